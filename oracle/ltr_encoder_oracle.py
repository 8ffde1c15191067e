"""CPU ORACLE (test infrastructure, NOT product code) for the set-transformer scorer -- SURVEY.md row f-3.

Restates, in plain torch ops with autograd (any dtype; tests use fp64), what the reference computes in
    architeture/multiLayer.py:13-149     (FCModel :42-51, LTRModel :64-81, OutputLayer :107-113, make_model :127-149)
    architeture/transformer.py:29-257    (Encoder :44-59, LayerNorm :78-88, SublayerConnection :106-114,
                                          EncoderLayer :132-142, attention :145-164, MultiHeadedAttention :184-212,
                                          PositionwiseFeedForward :230-237)
as ONE function of a flat {state_dict key: tensor} mapping, with every dropout site taking an EXPLICIT keep mask
(the HIP path's counter-based streams are exported by ltr_enc_dropout_mask / ltr_enc_attn_dropout_mask), and with
an optional emulation of the HIP path's bf16 rounding points in the FORWARD (`bf16=True`; rounding is
straight-through for autograd).  Pinned against the imported reference by tests/golden/make_golden_r3.py (fixtures
tests/golden/encoder.npz).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.
"""
import math

import torch


class _RoundBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def _rb(x, on):
    return _RoundBf16.apply(x) if on else x


def _bf(x):
    return x.to(torch.bfloat16).to(x.dtype)


class _RoundGradBf16(torch.autograd.Function):
    """Identity whose BACKWARD rounds the gradient to bf16: placed where the HIP backward hands a gradient to the next
    kernel as bf16 (ltr_mi355x/encoder.py _body_backward: dy after the residual-branch dropout, dz1, dctx, dqkv)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


def _rg(x, on):
    return _RoundGradBf16.apply(x) if on else x


class _AttnCoreRounded(torch.autograd.Function):
    """softmax(q k^T / sqrt(dk) + mask) -> dropout -> @ v with the arithmetic of csrc/ltr_encoder.hip
    attention_fwd_kernel / attention_bwd_kernel: q, k, v, dO arrive as bf16 values; P is computed unrounded, Pd = drop(P)
    is rounded to bf16 before Pd @ v (and before dV = Pd^T dO); the backward uses D = dO . O with the ROUNDED saved
    output, dS = P ((keep ? dPd / (1-p) : 0) - D) / sqrt(dk) rounded to bf16 before dQ = dS k and dK = dS^T q."""

    @staticmethod
    def forward(ctx, q, k, v, pad, keep, p):
        dk = q.shape[-1]
        sc = (q @ k.transpose(-2, -1)) / math.sqrt(dk)
        sc = sc.masked_fill(pad, float("-inf"))
        P = torch.softmax(sc, dim=-1)
        P = torch.where(torch.isnan(P), torch.zeros_like(P), P)      # a slate without any unmasked key: zeros (kernel: l = 0)
        ks = 1.0 / (1.0 - p) if keep is not None and p > 0 else 1.0
        Pd = _bf(P * keep.to(P.dtype) * ks) if ks != 1.0 else _bf(P)
        O = _bf(Pd @ v)
        ctx.save_for_backward(q, k, v, P, Pd, O, keep if keep is not None else torch.ones((), dtype=torch.bool))
        ctx.ks, ctx.has_keep = ks, ks != 1.0
        return O

    @staticmethod
    def backward(ctx, dO):
        q, k, v, P, Pd, O, keep = ctx.saved_tensors
        dk = q.shape[-1]
        scale = 1.0 / math.sqrt(dk)
        dO = _bf(dO)
        D = (dO * O).sum(-1, keepdim=True)
        dPd = dO @ v.transpose(-2, -1)
        dP = dPd * keep.to(dPd.dtype) * ctx.ks if ctx.has_keep else dPd
        dS = _bf(P * (dP - D) * scale)
        dq = dS @ k
        dk_ = dS.transpose(-2, -1) @ q
        dv = Pd.transpose(-2, -1) @ dO
        return dq, dk_, dv, None, None, None


def layer_norm_annotated(x, a, b, eps=1e-6):
    """transformer.py:86-88: unbiased std, eps added to the std."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return a * (x - mean) / (std + eps) + b


def _drop(x, keep, p):
    """nn.Dropout with an explicit keep mask (None = eval mode / p = 0)."""
    if keep is None or p == 0.0:
        return x
    return x * keep.to(x.dtype) / (1.0 - p)


def encoder_scores(sd, x, mask, cfg, keep=None, bf16=False, round_bwd=False):
    """Scores [B, S] of a `make_model` network.
      sd    {state_dict key: tensor} (reference key names)
      x     [B, S, F];  mask [B, S] (1 / True = padded) or None when there is no encoder
      cfg   dict(n_fc, input_norm, fc_dropout, n_layers, heads, enc_dropout, has_encoder); optional keys for the blocks called
            on their own (tests/test_blocks_gpu.py): final_norm (default True: Encoder.norm, transformer.py:59; False = a bare
            EncoderLayer), output ("scores", default | "features": stop before the output layer = prepare_for_output)
      keep  None (eval) or {site: mask}: ("fc", i) -> [B*S, out_i]; ("attn", l) -> [B, h, S, S];
            ("attn_out", l), ("ffn_out", l) -> [B*S, d];  ("ffn_hidden", l) -> [B*S, d_ff]
      bf16       round where the HIP FORWARD rounds (operands of every GEMM, saved activations, P, ctx)
      round_bwd  (with bf16) also round where the HIP BACKWARD rounds: the gradient entering every weight-gradient /
                 input-gradient GEMM (dy behind the residual-branch dropout, dz1 behind the ReLU gate, dctx, dqkv) and the
                 attention core's dS, D = dO . O -- so that what is left between kernels and oracle is summation order
    """
    keep = keep or {}
    rbw = bool(bf16 and round_bwd)
    B, S, F = x.shape
    y = x
    if cfg.get("input_norm"):
        y = torch.nn.functional.layer_norm(y, (F,), sd["input_layer.input_norm.weight"], sd["input_layer.input_norm.bias"], 1e-5)
    for i in range(cfg["n_fc"]):                                      # multiLayer.py:48-50 (activation = Identity)
        W, b = sd[f"input_layer.layers.{i}.weight"], sd[f"input_layer.layers.{i}.bias"]
        y = _rg(_rb(y, bf16) @ _rb(W, bf16).t() + b, rbw)
        k = keep.get(("fc", i))
        y = _drop(y, None if k is None else k.view(B, S, -1), cfg["fc_dropout"])
    if cfg["has_encoder"]:
        h = cfg["heads"]
        d = y.shape[-1]
        dk = d // h
        p = cfg["enc_dropout"]
        pad = (mask == 1).view(B, 1, 1, S)                            # transformer.py:55, :196
        for l in range(cfg["n_layers"]):
            pre = f"encoder.layers.{l}."
            n1 = _rb(layer_norm_annotated(y, sd[pre + "sublayer.0.norm.a_2"], sd[pre + "sublayer.0.norm.b_2"]), bf16)
            q, k_, v = (_rb(_rg(n1 @ _rb(sd[pre + f"self_attn.linears.{j}.weight"], bf16).t() + sd[pre + f"self_attn.linears.{j}.bias"], rbw), bf16)
                        .view(B, S, h, dk).transpose(1, 2) for j in range(3))            # :199-201
            if rbw:
                ka = keep.get(("attn", l))
                ctx = _AttnCoreRounded.apply(q, k_, v, pad, ka, p if ka is not None else 0.0)
                ctx = _rg(ctx.transpose(1, 2).contiguous().view(B, S, d), True)
            else:
                sc = q @ k_.transpose(-2, -1) / math.sqrt(dk)              # :156
                sc = sc.masked_fill(pad, float("-inf"))                    # :158-159
                pa = torch.softmax(sc, dim=-1)                             # :161
                pa = _rb(_drop(pa, keep.get(("attn", l)), p), bf16)        # :162-163
                ctx = _rb((pa @ v).transpose(1, 2).contiguous().view(B, S, d), bf16)      # :207-209
            att = _rg(ctx @ _rb(sd[pre + "self_attn.linears.3.weight"], bf16).t() + sd[pre + "self_attn.linears.3.bias"], rbw)
            ko = keep.get(("attn_out", l))
            y = y + _drop(att, None if ko is None else ko.view(B, S, d), p)                # :113-114
            n2 = _rb(layer_norm_annotated(y, sd[pre + "sublayer.1.norm.a_2"], sd[pre + "sublayer.1.norm.b_2"]), bf16)
            hid = torch.relu(_rg(n2 @ _rb(sd[pre + "feed_forward.w_1.weight"], bf16).t() + sd[pre + "feed_forward.w_1.bias"], rbw))
            kh = keep.get(("ffn_hidden", l))
            hid = _rb(_drop(hid, None if kh is None else kh.view(B, S, -1), p), bf16)      # :237
            ff = _rg(hid @ _rb(sd[pre + "feed_forward.w_2.weight"], bf16).t() + sd[pre + "feed_forward.w_2.bias"], rbw)
            kf = keep.get(("ffn_out", l))
            y = y + _drop(ff, None if kf is None else kf.view(B, S, d), p)
        if cfg.get("final_norm", True):
            y = layer_norm_annotated(y, sd["encoder.norm.a_2"], sd["encoder.norm.b_2"])    # :59
    if cfg.get("output", "scores") == "features":
        return y
    W, b = sd["output_layer.w_1.weight"], sd["output_layer.w_1.bias"]
    if W.shape[0] > 1:          # d_output > 1 runs as a bf16-operand GEMM on the HIP path (d_output = 1: the fp32 scoring tail)
        out = _rg(_rb(y, bf16) @ _rb(W, bf16).t() + b, rbw)
    else:
        out = y @ W.t() + b                                                                # multiLayer.py:113
    return out.squeeze(dim=2)


def scores_and_grads(sd, x, mask, cfg, loss_fn, keep=None, bf16=False, dtype=torch.float64, round_bwd=False, want_dx=False):
    """(scores, loss, {key: grad}) in `dtype`; loss_fn maps scores [B, S] -> 0-dim.  want_dx: the gradient w.r.t. the input
    features is returned under the key "__x__"."""
    p = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    xin = x.detach().to(dtype).clone().requires_grad_(bool(want_dx))
    s = encoder_scores(p, xin, mask, cfg, keep, bf16, round_bwd)
    loss = loss_fn(s)
    leaves = list(p.values()) + ([xin] if want_dx else [])
    grads = torch.autograd.grad(loss, leaves, allow_unused=True)
    out = {k: (torch.zeros_like(p[k]) if g is None else g) for k, g in zip(p.keys(), grads)}
    if want_dx:
        out["__x__"] = grads[-1] if grads[-1] is not None else torch.zeros_like(xin)
    return s.detach(), loss.detach(), out


def config_of(model_kwargs, n_features):
    """cfg for encoder_scores from make_model's arguments (fc_model dict or None, transformer dict or None)."""
    fc, tr = model_kwargs.get("fc_model"), model_kwargs.get("transformer")
    return dict(n_fc=len(fc["sizes"]) if fc else 0, input_norm=bool(fc and fc.get("input_norm")),
                fc_dropout=float((fc or {}).get("dropout") or 0.0), has_encoder=bool(tr),
                n_layers=int(tr["N"]) if tr else 0, heads=int(tr["h"]) if tr else 1,
                enc_dropout=float(tr["dropout"]) if tr else 0.0)
