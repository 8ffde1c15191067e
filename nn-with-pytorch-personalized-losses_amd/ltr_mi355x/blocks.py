"""The building blocks of architeture/transformer.py and architeture/multiLayer.py called ON THEIR OWN (outside an LTRModel):
each is one torch.autograd.Function over the same C-ABI entry points the fused network uses (include/ltr_encoder.h), with
the same arithmetic -- bf16 operands on the matrix cores, fp32 accumulation / statistics / gradients -- and an analytic
backward that also returns the gradient w.r.t. the block's input.

    features(...)           FCModel.forward (multiLayer.py:36-46), EncoderLayer.forward (transformer.py:134-142),
                            Encoder.forward (:45-59), LTRModel.prepare_for_output (multiLayer.py:64-72)
    layer_norm(...)         LayerNorm.forward (transformer.py:78-88)
    linear(...)             OutputLayer.w_1 with d_output > 1 (multiLayer.py:113), any free-standing nn.Linear
    score_linear(...)       OutputLayer.w_1 with d_output = 1 (the fused network's scoring tail without its norm)
    multi_head_attention    MultiHeadedAttention.forward (transformer.py:187-212)
    attention_core(...)     attention() (:145-164), returns (output, p_attn)
    feed_forward(...)       PositionwiseFeedForward.forward (:231-237)

Device tensors only; there is no CPU fallback."""
import torch

from . import encoder as E
from ._lib import check, lib
from .functional import _ptr, _stream, require_device

_U16 = E._U16
_calls = [0]


def fresh_seed():
    """Seed of a standalone block's dropout streams: torch's global seed mixed with a per-call counter."""
    _calls[0] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _calls[0] * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D) & (2 ** 64 - 1)


def _tokens(x, width):
    if x.shape[-1] != width:
        raise ValueError(f"expected [..., {width}] features, got {tuple(x.shape)}")
    return x.detach().to(torch.float32).contiguous().view(-1, width)


def _f32(p):
    return p.detach().to(torch.float32).contiguous()


def _mult8(*dims):
    bad = [v for v in dims if v % 8]
    if bad:
        raise ValueError(f"the HIP encoder kernels need feature counts that are multiples of 8, got {bad}")


def slate_mask(mask, B, S, device):
    """The padding mask as the kernels take it: uint8 [B][S], 1 = padded.  Accepts what the reference's callers pass at the
    different levels -- [B, S] (Encoder), [B, 1, S] (EncoderLayer, transformer.py:55) or [B, 1, 1, S] (attention, :196)."""
    if mask is None:
        return None
    if mask.numel() != B * S:
        raise NotImplementedError(f"only per-document padding masks ([batch, slate], optionally with singleton head / query axes) "
                                  f"are built on the HIP path, got {tuple(mask.shape)} for batch {B}, slate {S}")
    return (mask.to(device) == 1).to(torch.uint8).contiguous().view(B, S)


# ----------------------------------------------------------------------------------------------------- FC / encoder bodies
class Features(torch.autograd.Function):
    """FC block and / or encoder blocks (+ the encoder's final norm when `final_norm`): x [B, S, F] -> [B, S, d_model] fp32.
    `params`: the body parameters in EncoderScores' order, then (a_2, b_2) of the final norm if `final_norm`."""

    @staticmethod
    def forward(ctx, spec, final_norm, x, mask, seed, training, *params):
        require_device(x, *params)
        B, S = x.shape[0], x.shape[1]
        ctx.spec, ctx.final_norm, ctx.seed = spec, final_norm, int(seed)
        ctx.meta = ([p.dtype for p in params], [tuple(p.shape) for p in params], x.dtype, tuple(x.shape))
        if B * S == 0:
            ctx.st = None
            return torch.empty((B, S, spec.d_model), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            st = E._run_forward(spec, x, mask, seed, training, params)
            out = st["final_x"]
            if final_norm:
                prm = st["prm"]
                _, out = E.layernorm_fwd(out, prm[-2], prm[-1], B * S, spec.d_model, E.LN_EPS, 0, want_f32=True)
            else:
                # the residual stream's last tensor stays in ctx.st for the backward (outside save_for_backward, so no version
                # counter guards it): hand out a copy, an in-place edit of the output must not reach it
                out = out.clone()
        ctx.st = st
        return out.view(B, S, spec.d_model)

    @staticmethod
    def backward(ctx, dout):
        dts, shapes, xdt, xshape = ctx.meta
        dev = dout.device
        if ctx.st is None:
            return (None, None, torch.zeros(xshape, dtype=xdt, device=dev), None, None, None,
                    *[torch.zeros(s, dtype=t, device=dev) for s, t in zip(shapes, dts)])
        spec, st = ctx.spec, ctx.st
        B, S, F = st["dims"]
        T, d, prm = B * S, spec.d_model, st["prm"]
        with torch.cuda.device(dev):
            g = dout.detach().to(torch.float32).contiguous().view(T, d)
            with E.deferred_reductions():
                tail = []
                if ctx.final_norm:
                    dx = torch.zeros((T, d), dtype=torch.float32, device=dev)
                    ga, gb = E.layernorm_bwd(st["final_x"], prm[-2], g, T, d, E.LN_EPS, 0, dx)
                    tail = [ga, gb]
                else:
                    dx = g.clone()                      # the body backward accumulates into it in place
                body, dxin = E._body_backward(spec, ctx.seed, st, dx, want_dx=ctx.needs_input_grad[2])
        grads = [None if gr is None else gr.to(t).reshape(s) for gr, t, s in zip(body + tail, dts, shapes)]
        return (None, None, None if dxin is None else dxin.view(xshape).to(xdt), None, None, None, *grads)


# ----------------------------------------------------------------------------------------------------- LayerNorm
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a, b, eps, standard):
        require_device(x, a, b)
        d = x.shape[-1]
        xf = _tokens(x, d)
        T = xf.shape[0]
        ctx.meta = (x.dtype, tuple(x.shape), a.dtype, b.dtype, float(eps), int(standard))
        if T == 0:
            ctx.xf = None
            return torch.empty(x.shape, dtype=torch.float32, device=x.device)
        af, bf = _f32(a), _f32(b)
        with torch.cuda.device(x.device):
            _, y = E.layernorm_fwd(xf, af, bf, T, d, eps, standard, want_f32=True)
        ctx.xf, ctx.af = xf, af
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        xdt, xshape, adt, bdt, eps, standard = ctx.meta
        d = xshape[-1]
        if ctx.xf is None:
            z = torch.zeros(d, device=dy.device)
            return torch.zeros(xshape, dtype=xdt, device=dy.device), z.to(adt), z.to(bdt), None, None
        T = ctx.xf.shape[0]
        with torch.cuda.device(dy.device):
            g = dy.detach().to(torch.float32).contiguous().view(T, d)
            dx = torch.zeros((T, d), dtype=torch.float32, device=dy.device)
            ga, gb = E.layernorm_bwd(ctx.xf, ctx.af, g, T, d, eps, standard, dx)
        return dx.view(xshape).to(xdt), ga.to(adt), gb.to(bdt), None, None


def layer_norm(x, a, b, eps, standard=False):
    return LayerNormFn.apply(x, a, b, float(eps), 1 if standard else 0)


# ----------------------------------------------------------------------------------------------------- Linear
class LinearFn(torch.autograd.Function):
    """y = x W^T + b for x [..., K], W [N, K]: bf16 operands, fp32 accumulate / output.  N is padded to a multiple of 8
    (zero rows) for the GEMM kernel; K must be one."""

    @staticmethod
    def forward(ctx, x, W, b):
        require_device(x, W, b)
        N, K = W.shape
        _mult8(K)
        xf = _tokens(x, K)
        T = xf.shape[0]
        Np = (N + 7) // 8 * 8
        ctx.meta = (x.dtype, tuple(x.shape), W.dtype, b.dtype, N, K, Np)
        if T == 0:
            ctx.x16 = None
            return torch.empty((*x.shape[:-1], N), dtype=torch.float32, device=x.device)
        dev = x.device
        with torch.cuda.device(dev):
            Wp, bp = _f32(W), _f32(b)
            if Np != N:
                Wp = torch.cat([Wp, torch.zeros((Np - N, K), dtype=torch.float32, device=dev)], 0)
                bp = torch.cat([bp, torch.zeros(Np - N, dtype=torch.float32, device=dev)], 0)
            x16, w16 = E.cast_bf16(xf), E.cast_bf16(Wp)
            y = torch.empty((T, Np), dtype=torch.float32, device=dev)
            E.gemm(x16, w16, T, Np, K, Cf=y, bias=bp)
        ctx.x16, ctx.w16 = x16, w16
        return (y if Np == N else y[:, :N].contiguous()).view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xdt, xshape, wdt, bdt, N, K, Np = ctx.meta
        dev = dy.device
        if ctx.x16 is None:
            return torch.zeros(xshape, dtype=xdt, device=dev), torch.zeros((N, K), dtype=wdt, device=dev), torch.zeros(N, dtype=bdt, device=dev)
        T = ctx.x16.shape[0]
        with torch.cuda.device(dev):
            g = dy.detach().to(torch.float32).contiguous().view(T, N)
            if Np != N:
                g = torch.cat([g, torch.zeros((T, Np - N), dtype=torch.float32, device=dev)], 1).contiguous()
            dy16, gb = E._drop_cast_colsum(g, T, Np, 0.0, 0, 0)
            gW = E._weight_grad(dy16, ctx.x16, T, Np, K)
            dx = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty((T, K), dtype=torch.float32, device=dev)
                E.gemm(dy16, ctx.w16, T, K, Np, b_kmajor=True, Cf=dx)
                dx = dx.view(xshape).to(xdt)
        return dx, gW[:N].to(wdt), gb[:N].to(bdt)


def linear(x, W, b):
    return LinearFn.apply(x, W, b)


class ScoreLinearFn(torch.autograd.Function):
    """scores[t] = w . x[t] + bias for w [1, d] (OutputLayer with d_output = 1): the fused network's scoring tail without
    its norm (ltr_enc_score_fwd / _bwd, fp32 throughout)."""

    @staticmethod
    def forward(ctx, x, W, b):
        require_device(x, W, b)
        d = W.shape[1]
        xf = _tokens(x, d)
        T = xf.shape[0]
        ctx.meta = (x.dtype, tuple(x.shape), W.dtype, b.dtype, d)
        out = torch.empty(T, dtype=torch.float32, device=x.device)
        ctx.xf = None
        if T:
            wf, bf = _f32(W), _f32(b)
            with torch.cuda.device(x.device):
                check(lib().ltr_enc_score_fwd(_ptr(xf), None, None, _ptr(wf), _ptr(bf), T, d, E.LN_EPS, 0, _ptr(out), _stream()),
                      "ltr_enc_score_fwd")
            ctx.xf, ctx.wf = xf, wf
        return out.view(*x.shape[:-1], 1)

    @staticmethod
    def backward(ctx, dscores):
        xdt, xshape, wdt, bdt, d = ctx.meta
        dev = dscores.device
        if ctx.xf is None:
            return torch.zeros(xshape, dtype=xdt, device=dev), torch.zeros((1, d), dtype=wdt, device=dev), torch.zeros(1, dtype=bdt, device=dev)
        T = ctx.xf.shape[0]
        with torch.cuda.device(dev):
            ds = dscores.detach().to(torch.float32).contiguous().view(T)
            nblk = max(1, min(E._NBLK, (T + 3) // 4))
            dx = torch.empty((T, d), dtype=torch.float32, device=dev)
            parts = torch.empty((nblk, 3 * d + 8), dtype=torch.float32, device=dev)
            check(lib().ltr_enc_score_bwd(_ptr(ctx.xf), None, None, _ptr(ctx.wf), _ptr(ds), T, d, E.LN_EPS, 0, _ptr(dx), _ptr(parts), nblk,
                                          _stream()), "ltr_enc_score_bwd")
            g = E.sum_partials(parts, nblk, 3 * d + 8)
        return dx.view(xshape).to(xdt), g[2 * d:3 * d].view(1, d).to(wdt), g[3 * d:3 * d + 1].to(bdt)


def score_linear(x, W, b):
    return ScoreLinearFn.apply(x, W, b)


# ----------------------------------------------------------------------------------------------------- attention
class MultiHeadFn(torch.autograd.Function):
    """MultiHeadedAttention.forward: the three input projections (one GEMM when query, key and value are the same tensor),
    the attention core over the slate, the output projection."""

    @staticmethod
    def forward(ctx, h, p, seed, same, query, key, value, mask_u8, Wq, bq, Wk, bk, Wv, bv, Wo, bo):
        prm = (Wq, bq, Wk, bk, Wv, bv, Wo, bo)
        require_device(query, key, value, *prm)
        B, S, d = query.shape
        if tuple(key.shape) != (B, S, d) or tuple(value.shape) != (B, S, d):
            raise NotImplementedError("key / value sets of a different shape than the query set are not built on the HIP path")
        _mult8(d)
        dk, T, dev = d // h, B * S, query.device
        ctx.meta = (h, float(p), int(seed), same, (B, S, d), [t.dtype for t in (query, key, value)], [t.dtype for t in prm])
        if T == 0:
            ctx.saved = None
            return torch.empty((B, S, d), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            srcs = [E.cast_bf16(_tokens(query, d))]
            if not same:
                srcs += [E.cast_bf16(_tokens(key, d)), E.cast_bf16(_tokens(value, d))]
            w16 = E.cast_bf16(torch.cat([_f32(Wq).reshape(-1), _f32(Wk).reshape(-1), _f32(Wv).reshape(-1), _f32(Wo).reshape(-1)]))
            wqkv, wo16 = w16[:3 * d * d].view(3 * d, d), w16[3 * d * d:].view(d, d)
            bqkv = torch.cat([_f32(bq), _f32(bk), _f32(bv)])
            qkv = torch.empty((T, 3 * d), dtype=_U16, device=dev)
            if same:
                E.gemm(srcs[0], wqkv, T, 3 * d, d, Cb=qkv, bias=bqkv)
            else:
                for j in range(3):
                    E.gemm(srcs[j], wqkv[j * d:(j + 1) * d], T, d, d, Cb=qkv[:, j * d:], ldc=3 * d, bias=bqkv[j * d:(j + 1) * d])
            ctxb, lse = E.attention_fwd(qkv, mask_u8, B, S, h, dk, p, seed, 0)
            out = torch.empty((T, d), dtype=torch.float32, device=dev)
            E.gemm(ctxb, wo16, T, d, d, Cf=out, bias=_f32(bo))
        ctx.saved = (srcs, wqkv, wo16, qkv, ctxb, mask_u8, lse)
        return out.view(B, S, d)

    @staticmethod
    def backward(ctx, dout):
        h, p, seed, same, (B, S, d), in_dts, p_dts = ctx.meta
        dev = dout.device
        if ctx.saved is None:
            z = lambda *s: torch.zeros(s, device=dev)                                     # noqa: E731
            gp = [z(d, d), z(d)] * 4
            return (None, None, None, None, z(B, S, d), None if same else z(B, S, d), None if same else z(B, S, d), None,
                    *[g.to(t) for g, t in zip(gp, p_dts)])
        srcs, wqkv, wo16, qkv, ctxb, mask_u8, lse = ctx.saved
        T, dk = B * S, d // h
        with torch.cuda.device(dev):
            g = dout.detach().to(torch.float32).contiguous().view(T, d)
            dy16, gbo = E._drop_cast_colsum(g, T, d, 0.0, 0, 0)
            gWo = E._weight_grad(dy16, ctxb, T, d, d)
            dctx = torch.empty((T, d), dtype=_U16, device=dev)
            E.gemm(dy16, wo16, T, d, d, b_kmajor=True, Cb=dctx)
            dqkv = E.attention_bwd(qkv, ctxb, dctx, lse, mask_u8, B, S, h, dk, p, seed, 0)
            gbqkv = E._colsum(dqkv, T, 3 * d)
            need = ctx.needs_input_grad[4:7]
            if same:
                gW = E._weight_grad(dqkv, srcs[0], T, 3 * d, d)
                gWs = [gW[j * d:(j + 1) * d] for j in range(3)]
                dins = [None, None, None]
                if need[0]:
                    dx = torch.empty((T, d), dtype=torch.float32, device=dev)
                    E.gemm(dqkv, wqkv, T, d, 3 * d, b_kmajor=True, Cf=dx)
                    dins[0] = dx.view(B, S, d).to(in_dts[0])
            else:
                gWs, dins = [], []
                for j in range(3):
                    sl = dqkv[:, j * d:]
                    gWs.append(E._weight_grad(sl, srcs[j], T, d, d, lda=3 * d))
                    dx = None
                    if need[j]:
                        dx = torch.empty((T, d), dtype=torch.float32, device=dev)
                        E.gemm(sl, wqkv[j * d:(j + 1) * d], T, d, d, lda=3 * d, b_kmajor=True, Cf=dx)
                        dx = dx.view(B, S, d).to(in_dts[j])
                    dins.append(dx)
        gp = [gWs[0], gbqkv[:d], gWs[1], gbqkv[d:2 * d], gWs[2], gbqkv[2 * d:], gWo, gbo]
        return (None, None, None, None, dins[0], dins[1], dins[2], None, *[g_.to(t) for g_, t in zip(gp, p_dts)])


class AttentionCoreFn(torch.autograd.Function):
    """attention(): query / key / value [B, h, S, dk] -> softmax(q k^T / sqrt(dk) masked) (dropout) v, [B, h, S, dk]."""

    @staticmethod
    def forward(ctx, query, key, value, mask_u8, p, seed):
        require_device(query, key, value)
        B, h, S, dk = query.shape
        if tuple(key.shape) != (B, h, S, dk) or tuple(value.shape) != (B, h, S, dk):
            raise NotImplementedError("key / value sets of a different shape than the query set are not built on the HIP path")
        d, T, dev = h * dk, B * S, query.device
        _mult8(d)
        ctx.meta = (B, h, S, dk, float(p), int(seed), [t.dtype for t in (query, key, value)])
        with torch.cuda.device(dev):
            packed = torch.cat([t.detach().to(torch.float32).transpose(1, 2).reshape(T, d) for t in (query, key, value)], 1).contiguous()
            qkv = E.cast_bf16(packed)
            ctxb, lse = E.attention_fwd(qkv, mask_u8, B, S, h, dk, p, seed, 0)
        ctx.saved = (qkv, ctxb, mask_u8, lse)
        return ctxb.view(torch.bfloat16).to(torch.float32).view(B, S, h, dk).transpose(1, 2).contiguous()

    @staticmethod
    def backward(ctx, dout):
        B, h, S, dk, p, seed, dts = ctx.meta
        qkv, ctxb, mask_u8, lse = ctx.saved
        d, T, dev = h * dk, B * S, dout.device
        with torch.cuda.device(dev):
            dctx = E.cast_bf16(dout.detach().to(torch.float32).transpose(1, 2).reshape(T, d))
            dqkv = E.attention_bwd(qkv, ctxb, dctx, lse, mask_u8, B, S, h, dk, p, seed, 0)
            g = dqkv.view(torch.bfloat16).to(torch.float32).view(B, S, 3, h, dk).permute(2, 0, 3, 1, 4)
        return g[0].to(dts[0]), g[1].to(dts[1]), g[2].to(dts[2]), None, None, None


def attention_probs(query, key, mask_u8, p, seed):
    """p_attn [B, h, S, S] fp32 (forward only; the same dropout stream as the core)."""
    B, h, S, dk = query.shape
    d, T, dev = h * dk, B * S, query.device
    with torch.cuda.device(dev), torch.no_grad():
        zeros = torch.zeros((T, d), dtype=torch.float32, device=dev)
        packed = torch.cat([query.detach().to(torch.float32).transpose(1, 2).reshape(T, d),
                            key.detach().to(torch.float32).transpose(1, 2).reshape(T, d), zeros], 1).contiguous()
        qkv = E.cast_bf16(packed)
        probs = torch.empty((B, h, S, S), dtype=torch.float32, device=dev)
        check(lib().ltr_enc_attention_probs(_ptr(qkv), _ptr(mask_u8), B, S, h, dk, float(p), int(seed), 0, _ptr(probs), _stream()),
              "ltr_enc_attention_probs")
    return probs


# ----------------------------------------------------------------------------------------------------- feed-forward
class FeedForwardFn(torch.autograd.Function):
    """w_2(dropout(relu(w_1 x))) (no residual, no output dropout: those belong to SublayerConnection)."""

    @staticmethod
    def forward(ctx, p, seed, x, W1, b1, W2, b2):
        require_device(x, W1, b1, W2, b2)
        dff, d = W1.shape
        _mult8(d, dff)
        xf = _tokens(x, d)
        T, dev = xf.shape[0], x.device
        ctx.meta = (float(p), int(seed), x.dtype, tuple(x.shape), [t.dtype for t in (W1, b1, W2, b2)], d, dff)
        if T == 0:
            ctx.saved = None
            return torch.empty(x.shape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            x16 = E.cast_bf16(xf)
            w116, w216 = E.cast_bf16(_f32(W1)), E.cast_bf16(_f32(W2))
            hid = torch.empty((T, dff), dtype=_U16, device=dev)
            E.gemm(x16, w116, T, dff, d, Cb=hid, bias=_f32(b1), relu=True, drop_p=p, seed=seed, drop_stream=2)
            out = torch.empty((T, d), dtype=torch.float32, device=dev)
            E.gemm(hid, w216, T, d, dff, Cf=out, bias=_f32(b2))
        ctx.saved = (x16, w116, w216, hid)
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, dout):
        p, seed, xdt, xshape, dts, d, dff = ctx.meta
        dev = dout.device
        if ctx.saved is None:
            z = lambda *s: torch.zeros(s, device=dev)                                     # noqa: E731
            return (None, None, torch.zeros(xshape, dtype=xdt, device=dev), z(dff, d).to(dts[0]), z(dff).to(dts[1]), z(d, dff).to(dts[2]),
                    z(d).to(dts[3]))
        x16, w116, w216, hid = ctx.saved
        T = x16.shape[0]
        with torch.cuda.device(dev):
            g = dout.detach().to(torch.float32).contiguous().view(T, d)
            dy16, gb2 = E._drop_cast_colsum(g, T, d, 0.0, 0, 0)
            gW2 = E._weight_grad(dy16, hid, T, d, dff)
            dz1 = torch.empty((T, dff), dtype=_U16, device=dev)
            E.gemm(dy16, w216, T, dff, d, b_kmajor=True, Cb=dz1, gate=hid, gate_scale=1.0 / (1.0 - p))
            gb1 = E._colsum(dz1, T, dff)
            gW1 = E._weight_grad(dz1, x16, T, dff, d)
            dx = None
            if ctx.needs_input_grad[2]:
                dx = torch.empty((T, d), dtype=torch.float32, device=dev)
                E.gemm(dz1, w116, T, d, dff, b_kmajor=True, Cf=dx)
                dx = dx.view(xshape).to(xdt)
        return None, None, dx, gW1.to(dts[0]), gb1.to(dts[1]), gW2.to(dts[2]), gb2.to(dts[3])
