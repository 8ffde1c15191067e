"""Host side of the set-transformer scorer (BASELINE config 5): strings the kernels of
csrc/ltr_encoder.hip (C ABI: include/ltr_encoder.h) into the forward and backward of the reference's
`make_model` networks (architeture/multiLayer.py:13-149 + architeture/transformer.py:29-257):

    FCModel (Linear + dropout per layer, Identity activation)  ->  N x [LayerNorm -> MHA over the slate -> +res,
    LayerNorm -> FFN(ReLU) -> +res]  ->  LayerNorm  ->  OutputLayer.w_1 (d_model -> 1)  ->  scores [B, S]

ONE autograd node (`EncoderScores`) covers the whole network: the forward keeps exactly the tensors the analytic
backward needs, the backward walks the layers in reverse and hands every parameter gradient back in one go (no
gradient w.r.t. the input features is produced, like the FC scorers of this package).  bf16 operands, fp32
accumulation / residual stream / statistics / gradients.  Device tensors only; no CPU fallback."""
import ctypes

import torch

from ._lib import check, lib
from .functional import _ptr, _stream, require_device

_U16 = torch.int16      # raw bf16 bit patterns live in int16 tensors (torch.bfloat16 views are taken for tests only)


class GemmDesc(ctypes.Structure):
    """struct ltr_gemm_desc of include/ltr_encoder.h."""
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p),
                ("M", ctypes.c_int64), ("N", ctypes.c_int64), ("K", ctypes.c_int64),
                ("lda", ctypes.c_int64), ("ldb", ctypes.c_int64), ("ldc", ctypes.c_int64),
                ("a_kmajor", ctypes.c_int32), ("b_kmajor", ctypes.c_int32), ("splits", ctypes.c_int32),
                ("relu", ctypes.c_int32),
                ("Cf", ctypes.c_void_p), ("Cb", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("residual", ctypes.c_void_p), ("gate", ctypes.c_void_p),
                ("gate_scale", ctypes.c_float), ("drop_p", ctypes.c_float), ("seed", ctypes.c_uint64),
                ("drop_stream", ctypes.c_int32), ("reserved", ctypes.c_int32)]


def gemm(A, B, M, N, K, *, a_kmajor=False, b_kmajor=False, lda=None, ldb=None, ldc=None, Cf=None, Cb=None, bias=None, residual=None,
         gate=None, gate_scale=1.0, relu=False, drop_p=0.0, seed=0, drop_stream=0, splits=1):
    """C[m][n] = sum_k A(m,k) B(n,k) (+ epilogue) -> ltr_enc_gemm_bf16.  A, B: int16 tensors of bf16 bits."""
    d = GemmDesc()
    d.A, d.B = A.data_ptr(), B.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda = lda if lda is not None else (M if a_kmajor else K)
    d.ldb = ldb if ldb is not None else (N if b_kmajor else K)
    d.ldc = ldc if ldc is not None else N
    d.a_kmajor, d.b_kmajor, d.splits, d.relu = int(a_kmajor), int(b_kmajor), int(splits), int(relu)
    d.Cf, d.Cb, d.bias, d.residual, d.gate = _ptr(Cf), _ptr(Cb), _ptr(bias), _ptr(residual), _ptr(gate)
    d.gate_scale, d.drop_p, d.seed, d.drop_stream = float(gate_scale), float(drop_p), int(seed) & (2 ** 64 - 1), int(drop_stream)
    check(lib().ltr_enc_gemm_bf16(ctypes.byref(d), _stream()), "ltr_enc_gemm_bf16")


def cast_bf16(src):
    """fp32 tensor -> int16 tensor of bf16 bits, same shape."""
    src = src.contiguous()
    out = torch.empty(src.shape, dtype=_U16, device=src.device)
    check(lib().ltr_enc_cast_bf16(_ptr(src), _ptr(out), src.numel(), _stream()), "ltr_enc_cast_bf16")
    return out


class ReduceJob(ctypes.Structure):
    """struct ltr_reduce_job of include/ltr_encoder.h."""
    _fields_ = [("parts", ctypes.c_void_p), ("out", ctypes.c_void_p), ("n", ctypes.c_int64), ("stride", ctypes.c_int64),
                ("nsplit", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class deferred_reductions:
    """Within this context `sum_partials` only RECORDS its job and returns the (not yet written) output tensor; all jobs
    run at exit through ltr_enc_sum_partials_batch, 16 per launch.  The backward of one step ends in ~60 small
    reductions whose results nothing reads before the gradients are handed back."""
    current = None

    def __enter__(self):
        self.jobs, self.keep = [], []
        self.prev, deferred_reductions.current = deferred_reductions.current, self
        return self

    def __exit__(self, exc_type, exc, tb):
        deferred_reductions.current = self.prev
        if exc_type is None and self.jobs:
            arr = (ReduceJob * len(self.jobs))(*self.jobs)
            check(lib().ltr_enc_sum_partials_batch(arr, len(self.jobs), _stream()), "ltr_enc_sum_partials_batch")
        self.jobs, self.keep = [], []
        return False


def sum_partials(parts, nsplit, n, out=None, accumulate=False):
    q = deferred_reductions.current
    if q is not None and out is None and not accumulate:
        out = torch.empty(n, dtype=torch.float32, device=parts.device)
        q.jobs.append(ReduceJob(parts.data_ptr(), out.data_ptr(), n, 0, nsplit, 0))
        q.keep += [parts, out]
        return out
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=parts.device)
    check(lib().ltr_enc_sum_partials(_ptr(parts), nsplit, n, int(accumulate), _ptr(out), _stream()), "ltr_enc_sum_partials")
    return out


def seed_set(value):
    """Set the device-side dropout epoch (added to every launch's seed; include/ltr_encoder.h) on the current stream."""
    check(lib().ltr_enc_seed_set(int(value) & (2 ** 64 - 1), _stream()), "ltr_enc_seed_set")


def seed_advance(delta=1):
    """epoch += delta as a kernel on the current stream: the first node of a captured training step (graphs.GraphedTrainStep)."""
    check(lib().ltr_enc_seed_advance(int(delta) & (2 ** 64 - 1), _stream()), "ltr_enc_seed_advance")


def seed_get():
    v = ctypes.c_uint64(0)
    check(lib().ltr_enc_seed_get(ctypes.byref(v)), "ltr_enc_seed_get")
    return int(v.value)


def dropout_mask(seed, stream_id, n, p, device):
    out = torch.empty(n, dtype=torch.uint8, device=device)
    check(lib().ltr_enc_dropout_mask(int(seed), int(stream_id), n, float(p), _ptr(out), _stream()), "ltr_enc_dropout_mask")
    return out


def attn_dropout_mask(seed, stream_id, B, S, h, p, device):
    out = torch.empty((B, h, S, S), dtype=torch.uint8, device=device)
    check(lib().ltr_enc_attn_dropout_mask(int(seed), int(stream_id), B, S, h, float(p), _ptr(out), _stream()),
          "ltr_enc_attn_dropout_mask")
    return out


# dropout stream ids (one per dropout site; the reference has one nn.Dropout per site)
def stream_attn(layer): return 8 * layer
def stream_attn_out(layer): return 8 * layer + 1
def stream_ffn_hidden(layer): return 8 * layer + 2
def stream_ffn_out(layer): return 8 * layer + 3
def stream_fc(i): return 100000 + i


_NBLK = 512     # workgroups (= partial rows) of the column-sum style reductions


def _small_step_splits(M, N, K):
    """Split-K factor for an activation-shaped GEMM (M tokens x N) whose 128 x 128 output tiles do not fill the chip while its k loop is
    long (the GEMM-path FFN's dn2 = dz1 W1 at a few thousand tokens: 32 tiles x 32 k-steps): up to 4 slices, each at least 8 k-steps."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles >= 128 or K < 1024:
        return 1
    return max(1, min(4, 256 // tiles, K // 512))


def _dw_splits(M, N, K):
    """Split-K factor of a weight-gradient GEMM: enough (tile, slice) workgroups to fill the chip (~512), at least 256
    tokens (4 k-steps) per slice."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    s = max(1, min(256, 512 // tiles))
    return max(1, min(s, (K + 255) // 256))


def _weight_grad(dy, x, T, n_out, n_in, lda=None):
    """dW [n_out][n_in] = dy^T x over the T tokens (split-K GEMM + fixed-order reduce); `lda`: row stride of dy when it is a
    column slice of a wider tensor."""
    splits = _dw_splits(n_out, n_in, T)
    parts = torch.empty((splits, n_out, n_in), dtype=torch.float32, device=dy.device)
    gemm(dy, x, n_out, n_in, T, a_kmajor=True, b_kmajor=True, lda=lda, Cf=parts, splits=splits)
    return sum_partials(parts, splits, n_out * n_in).view(n_out, n_in) if splits > 1 else parts.view(n_out, n_in)


def _colsum(y, T, N):
    nblk = max(1, min(_NBLK, T))
    parts = torch.empty((nblk, N), dtype=torch.float32, device=y.device)
    check(lib().ltr_enc_colsum_bf16(_ptr(y), T, N, _ptr(parts), nblk, _stream()), "ltr_enc_colsum_bf16")
    return sum_partials(parts, nblk, N)


def _drop_cast_colsum(dx, T, N, p, seed, stream_id):
    nblk = max(1, min(_NBLK, T))
    out = torch.empty((T, N), dtype=_U16, device=dx.device)
    parts = torch.empty((nblk, N), dtype=torch.float32, device=dx.device)
    check(lib().ltr_enc_drop_cast_colsum(_ptr(dx), T, N, float(p), int(seed), int(stream_id), _ptr(out), _ptr(parts), nblk,
                                         _stream()), "ltr_enc_drop_cast_colsum")
    return out, sum_partials(parts, nblk, N)


def layernorm_fwd(x, a, b, T, d, eps, standard, want_f32=False):
    yb = torch.empty((T, d), dtype=_U16, device=x.device)
    yf = torch.empty((T, d), dtype=torch.float32, device=x.device) if want_f32 else None
    check(lib().ltr_enc_layernorm_fwd(_ptr(x), _ptr(a), _ptr(b), T, d, float(eps), int(standard), _ptr(yb), _ptr(yf), _stream()),
          "ltr_enc_layernorm_fwd")
    return (yb, yf) if want_f32 else yb


def layernorm_bwd(x, a, dy, T, d, eps, standard, dx):
    """dx += (through the norm); returns (d a, d b)."""
    nblk = max(1, min(4 * _NBLK, (T + 7) // 8))
    parts = torch.empty((nblk, 2 * d), dtype=torch.float32, device=x.device)
    check(lib().ltr_enc_layernorm_bwd(_ptr(x), _ptr(a), _ptr(dy), T, d, float(eps), int(standard), _ptr(dx), _ptr(parts), nblk,
                                      _stream()), "ltr_enc_layernorm_bwd")
    g = sum_partials(parts, nblk, 2 * d)
    return g[:d], g[d:]


def attention_fwd(qkv, mask_u8, B, S, h, dk, p, seed, stream_id):
    """ctx [T][d] bf16 and lse2 [B*h][S] fp32 (log2-sum-exp of the scaled scores; lets the backward evaluate every
    probability once, key-major: csrc/ltr_encoder.hip attention_bwd_km_kernel)."""
    T, d = B * S, h * dk
    ctxb = torch.empty((T, d), dtype=_U16, device=qkv.device)
    lse = torch.empty((B * h, S), dtype=torch.float32, device=qkv.device)
    check(lib().ltr_enc_attention_fwd_lse(_ptr(qkv), _ptr(mask_u8), B, S, h, dk, float(p), int(seed), int(stream_id), _ptr(ctxb),
                                          _ptr(lse), _stream()), "ltr_enc_attention_fwd_lse")
    return ctxb, lse


def attention_bwd(qkv, ctxb, dctx, lse, mask_u8, B, S, h, dk, p, seed, stream_id):
    dqkv = torch.empty((B * S, 3 * h * dk), dtype=_U16, device=qkv.device)
    check(lib().ltr_enc_attention_bwd_lse(_ptr(qkv), _ptr(ctxb), _ptr(dctx), _ptr(lse), _ptr(mask_u8), B, S, h, dk, float(p), int(seed),
                                          int(stream_id), _ptr(dqkv), _stream()), "ltr_enc_attention_bwd_lse")
    return dqkv


FUSED_FFN_MIN_TOKENS = 8193     # below this the GEMM path is faster (measured, profiles/r04_c5_graph_step.jsonl)


def fused_ffn_enabled(d, dff, T=None):
    """The fused FFN kernels (no [T, d_ff] tensor in HBM) cover d_model in {64, 128} with d_ff a multiple of 128;
    LTR_ENC_FUSED_FFN=0 forces the GEMM path, =1 the fused kernels (A/B measurements, tests).  Unset: fused from 8 193 tokens per
    step -- a fused-FFN workgroup walks all d_ff chunks of its 128 tokens serially, so 4 096 tokens are 32 workgroups on 256 CUs,
    while the GEMM path tiles the same work over 512 and its hidden tensor (16 MB there) is no traffic to speak of: the graphed
    config-5 step at 16 slates runs 2.05 ms on GEMMs against 2.35 fused, at 32 slates 2.45 / 2.54, at 64 slates 3.11 / 2.96."""
    import os
    env = os.environ.get("LTR_ENC_FUSED_FFN")
    if env == "0" or not lib().ltr_enc_ffn_supported(int(d), int(dff)):
        return False
    return env == "1" or T is None or int(T) >= FUSED_FFN_MIN_TOKENS


def ffn_fwd(n2, w1, b1, w2, b2, x1, T, d, dff, p, seed, s_hidden, s_out):
    x2 = torch.empty((T, d), dtype=torch.float32, device=n2.device)
    check(lib().ltr_enc_ffn_fwd(_ptr(n2), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(x1), T, d, dff, float(p), int(seed), s_hidden,
                                s_out, _ptr(x2), _stream()), "ltr_enc_ffn_fwd")
    return x2


def ffn_bwd(n2, w1, b1, w2, dy, T, d, dff, p, seed, s_hidden):
    """(dn2 [T, d] fp32, dW1 [dff, d], dW2 [d, dff], db1 [dff])."""
    dev = n2.device
    dn2 = torch.empty((T, d), dtype=torch.float32, device=dev)
    check(lib().ltr_enc_ffn_bwd_x(_ptr(n2), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(dy), T, d, dff, float(p), int(seed), s_hidden, _ptr(dn2),
                                  _stream()), "ltr_enc_ffn_bwd_x")
    nsplit = max(1, min(256 // (dff // 128), (T + 127) // 128))
    p1 = torch.empty((nsplit, dff, d), dtype=torch.float32, device=dev)
    p2 = torch.empty((nsplit, d, dff), dtype=torch.float32, device=dev)
    pb = torch.empty((nsplit, dff), dtype=torch.float32, device=dev)
    check(lib().ltr_enc_ffn_bwd_w(_ptr(n2), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(dy), T, d, dff, float(p), int(seed), s_hidden, nsplit,
                                  _ptr(p1), _ptr(p2), _ptr(pb), _stream()), "ltr_enc_ffn_bwd_w")
    return (dn2, sum_partials(p1, nsplit, dff * d).view(dff, d), sum_partials(p2, nsplit, d * dff).view(d, dff),
            sum_partials(pb, nsplit, dff))


class EncoderSpec:
    """Static shape of a `make_model` network (what the kernels need besides the parameter tensors)."""

    def __init__(self, n_features, fc_sizes, input_norm, fc_dropout, n_layers, heads, d_ff, enc_dropout, has_encoder):
        self.n_features = int(n_features)
        self.fc_sizes = [int(s) for s in fc_sizes]          # output sizes of the FC layers (may be empty)
        self.input_norm = bool(input_norm)
        self.fc_dropout = float(fc_dropout or 0.0)
        self.n_layers = int(n_layers) if has_encoder else 0
        self.heads = int(heads)
        self.d_ff = int(d_ff)
        self.enc_dropout = float(enc_dropout or 0.0)
        self.has_encoder = bool(has_encoder)
        self.d_model = self.fc_sizes[-1] if self.fc_sizes else self.n_features
        dims = [self.n_features] + self.fc_sizes + ([self.d_ff] if has_encoder else [])
        if self.input_norm and not self.fc_sizes:
            raise ValueError("input_norm without FC layers is not built on the HIP path")
        bad = [v for v in dims if v % 8]
        if bad:
            raise ValueError(f"the HIP encoder needs feature counts that are multiples of 8, got {bad}")
        if has_encoder:
            if self.d_model % self.heads:
                raise AssertionError("d_model % h == 0")          # transformer.py:179
            if self.d_model // self.heads > 32:
                raise ValueError("head dimension d_model / h must be <= 32")
        if max(dims) > 8192 or self.d_model > 512:
            raise ValueError("layer widths beyond what the kernels were sized for (d_model <= 512)")

    @property
    def dk(self):
        return self.d_model // self.heads

    def n_params(self):
        n = 2 * len(self.fc_sizes) + (2 if self.input_norm else 0)
        if self.has_encoder:
            n += self.n_layers * 16 + 2
        return n + 2


# Parameter order of EncoderScores.apply(..., *params):
#   [input_norm.weight, input_norm.bias]            if input_norm
#   fc[i].weight, fc[i].bias                         per FC layer
#   per encoder layer: ln1.a, ln1.b, Wq, bq, Wk, bk, Wv, bv, Wo, bo, ln2.a, ln2.b, W1, b1, W2, b2
#   final norm a, b                                  if encoder
#   out.weight [1, d], out.bias [1]
LN_EPS = 1e-6          # transformer.py:69
STD_LN_EPS = 1e-5      # nn.LayerNorm default (multiLayer.py:27)


def _run_forward(spec, x, mask, seed, training, params):
    """FCModel + encoder blocks.  Returns everything the scoring tail and the backward need."""
    require_device(x, *params)
    if x.dim() != 3 or x.shape[2] != spec.n_features:
        raise ValueError(f"input must be [batch, slate, {spec.n_features}], got {tuple(x.shape)}")
    B, S, F = x.shape
    T = B * S
    dev = x.device
    if spec.has_encoder and S > 512:
        raise ValueError("the attention kernels hold a whole slate: slate_length <= 512")
    if spec.has_encoder and mask is None:
        raise AttributeError("'NoneType' object has no attribute 'unsqueeze'")      # transformer.py:55
    p_fc = spec.fc_dropout if training else 0.0
    p_enc = spec.enc_dropout if training else 0.0
    st = {"dims": (B, S, F), "p": (p_fc, p_enc), "fc_in": [], "layers": [], "fc_w16": [], "enc_w16": [], "mask_u8": None}
    xin = x.detach().to(torch.float32).contiguous().view(T, F)
    prm = [p.detach().to(torch.float32).contiguous() for p in params]
    it = iter(prm)
    # ---- FCModel (multiLayer.py:42-51)
    act = None
    if spec.input_norm:
        ln_w, ln_b = next(it), next(it)
        act = layernorm_fwd(xin, ln_w, ln_b, T, F, STD_LN_EPS, 1)
    elif spec.fc_sizes:
        act = cast_bf16(xin)
    stream_x = xin
    n_in = F
    for i, n_out in enumerate(spec.fc_sizes):
        W, bvec = next(it), next(it)
        w16 = cast_bf16(W)
        st["fc_w16"].append(w16)
        last = i == len(spec.fc_sizes) - 1
        yf = torch.empty((T, n_out), dtype=torch.float32, device=dev) if last else None
        yb = torch.empty((T, n_out), dtype=_U16, device=dev) if not last else None
        gemm(act, w16, T, n_out, n_in, Cf=yf, Cb=yb, bias=bvec, drop_p=p_fc, seed=seed, drop_stream=stream_fc(i))
        st["fc_in"].append(act)
        act, n_in = yb, n_out
        if last:
            stream_x = yf
    d = spec.d_model
    # ---- Encoder blocks (transformer.py:44-59, 132-142)
    if spec.has_encoder:
        mask_u8 = (mask.to(dev) == 1).to(torch.uint8).contiguous().view(B, S)
        st["mask_u8"] = mask_u8
        h, dk, dff = spec.heads, spec.dk, spec.d_ff
        st["fused_ffn"] = fused_ffn_enabled(d, dff, T)
        # all encoder weights to bf16 in ONE cast: [Wq | Wk | Wv | Wo | W1 | W2] per block, concatenated (every piece is a
        # multiple of 8 elements, so the views stay 16-byte aligned)
        n_fc_prm = (2 if spec.input_norm else 0) + 2 * len(spec.fc_sizes)
        blocks = [prm[n_fc_prm + 16 * l:n_fc_prm + 16 * (l + 1)] for l in range(spec.n_layers)]
        w16_all = cast_bf16(torch.cat([blk[i].reshape(-1) for blk in blocks for i in (2, 4, 6, 8, 12, 14)]))
        per = 4 * d * d + 2 * d * dff
        for l in range(spec.n_layers):
            a1, b1n, Wq, bq, Wk, bk, Wv, bv, Wo, bo, a2, b2n, W1, b1, W2, b2 = (next(it) for _ in range(16))
            w16 = w16_all[l * per:(l + 1) * per]
            wqkv, wo16 = w16[:3 * d * d].view(3 * d, d), w16[3 * d * d:4 * d * d].view(d, d)
            w116, w216 = w16[4 * d * d:4 * d * d + d * dff].view(dff, d), w16[4 * d * d + d * dff:].view(d, dff)
            bqkv = torch.cat([bq, bk, bv], 0)
            st["enc_w16"].append((wqkv, wo16, w116, w216))
            x0 = stream_x
            n1 = layernorm_fwd(x0, a1, b1n, T, d, LN_EPS, 0)
            qkv = torch.empty((T, 3 * d), dtype=_U16, device=dev)
            gemm(n1, wqkv, T, 3 * d, d, Cb=qkv, bias=bqkv)
            ctxb, lse = attention_fwd(qkv, mask_u8, B, S, h, dk, p_enc, seed, stream_attn(l))
            x1 = torch.empty((T, d), dtype=torch.float32, device=dev)
            gemm(ctxb, wo16, T, d, d, Cf=x1, bias=bo, residual=x0, drop_p=p_enc, seed=seed, drop_stream=stream_attn_out(l))
            n2 = layernorm_fwd(x1, a2, b2n, T, d, LN_EPS, 0)
            if st["fused_ffn"]:
                hid = None
                x2 = ffn_fwd(n2, w116, b1, w216, b2, x1, T, d, dff, p_enc, seed, stream_ffn_hidden(l), stream_ffn_out(l))
            else:
                hid = torch.empty((T, dff), dtype=_U16, device=dev)
                gemm(n2, w116, T, dff, d, Cb=hid, bias=b1, relu=True, drop_p=p_enc, seed=seed, drop_stream=stream_ffn_hidden(l))
                x2 = torch.empty((T, d), dtype=torch.float32, device=dev)
                ks = _small_step_splits(T, d, dff)
                if ks > 1:          # 32 output tiles, 32 k-steps each: split-K partials, then bias + dropout + residual in the reduce
                    parts = torch.empty((ks, T, d), dtype=torch.float32, device=dev)
                    gemm(hid, w216, T, d, dff, Cf=parts, splits=ks)
                    check(lib().ltr_enc_splitk_epilogue(_ptr(parts), ks, T, d, _ptr(b2), float(p_enc), int(seed) & (2 ** 64 - 1),
                                                        stream_ffn_out(l), _ptr(x1), _ptr(x2), _stream()), "ltr_enc_splitk_epilogue")
                else:
                    gemm(hid, w216, T, d, dff, Cf=x2, bias=b2, residual=x1, drop_p=p_enc, seed=seed, drop_stream=stream_ffn_out(l))
            st["layers"].append((x0, n1, qkv, ctxb, x1, n2, hid, lse))
            stream_x = x2
    st["xin"], st["prm"], st["final_x"] = xin, prm, stream_x
    return st


class EncoderScores(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, x, mask, seed, training, *params):
        if len(params) != spec.n_params():
            raise ValueError(f"expected {spec.n_params()} parameter tensors, got {len(params)}")
        require_device(x, *params)
        ctx.param_dtypes = [p.dtype for p in params]
        if x.dim() == 3 and x.shape[0] * x.shape[1] == 0:          # empty batch: nothing to launch, zero gradients
            if spec.has_encoder and mask is None:
                raise AttributeError("'NoneType' object has no attribute 'unsqueeze'")
            ctx.spec, ctx.st = spec, None
            ctx.shapes = [tuple(p.shape) for p in params]
            return torch.empty(x.shape[:2], dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            st = _run_forward(spec, x, mask, seed, training, params)
            B, S, _ = st["dims"]
            prm = st["prm"]
            fa, fb = (prm[-4], prm[-3]) if spec.has_encoder else (None, None)
            ow, ob = prm[-2], prm[-1]
            if ow.shape[0] != 1:
                raise NotImplementedError("OutputLayer with d_output > 1 is not built on the HIP path")
            scores = torch.empty((B, S), dtype=torch.float32, device=x.device)
            check(lib().ltr_enc_score_fwd(_ptr(st["final_x"]), _ptr(fa), _ptr(fb), _ptr(ow), _ptr(ob), B * S, spec.d_model, LN_EPS,
                                          1 if spec.has_encoder else 0, _ptr(scores), _stream()), "ltr_enc_score_fwd")
        ctx.spec, ctx.seed, ctx.st = spec, int(seed), st
        return scores

    @staticmethod
    def backward(ctx, dscores):
        if ctx.st is None:
            zeros = [torch.zeros(sh, dtype=dt, device=dscores.device) for sh, dt in zip(ctx.shapes, ctx.param_dtypes)]
            return (None, None, None, None, None, *zeros)
        spec, seed, st = ctx.spec, ctx.seed, ctx.st
        B, S, F = st["dims"]
        T, d, prm = B * S, spec.d_model, st["prm"]
        dev = dscores.device
        with torch.cuda.device(dev):
            ds = dscores.detach().to(torch.float32).contiguous().view(T)
            # ---- output layer (+ final norm)
            fa, fb = (prm[-4], prm[-3]) if spec.has_encoder else (None, None)
            nblk = max(1, min(_NBLK, (T + 3) // 4))
            dx = torch.empty((T, d), dtype=torch.float32, device=dev)
            parts = torch.empty((nblk, 3 * d + 8), dtype=torch.float32, device=dev)
            check(lib().ltr_enc_score_bwd(_ptr(st["final_x"]), _ptr(fa), _ptr(fb), _ptr(prm[-2]), _ptr(ds), T, d, LN_EPS,
                                          1 if spec.has_encoder else 0, _ptr(dx), _ptr(parts), nblk, _stream()), "ltr_enc_score_bwd")
            with deferred_reductions():
                grads = _with_tail(spec, _body_backward(spec, seed, st, dx)[0], sum_partials(parts, nblk, 3 * d + 8))
        out = [g if g is None else g.to(dt).reshape(p.shape) for g, dt, p in zip(grads, ctx.param_dtypes, prm)]
        return (None, None, None, None, None, *out)


def _with_tail(spec, body, tail):
    """Body gradients + the reduced tail partials (d a_2 | d b_2 | d w | d bias) -> the full list in parameter order."""
    d = spec.d_model
    norm = [tail[:d], tail[d:2 * d]] if spec.has_encoder else []
    return body + norm + [tail[2 * d:3 * d].view(1, d), tail[3 * d:3 * d + 1]]


def _body_backward(spec, seed, st, dx, want_dx=False):
    """Everything below the final norm / scoring tail: `dx` = d loss / d (last residual-stream value) [T, d] (accumulated
    into in place).  Returns (gradient list of the FC + encoder-block parameters in parameter order, d loss / d input
    features [T, F] fp32 if want_dx else None)."""
    B, S, F = st["dims"]
    T = B * S
    p_fc, p_enc = st["p"]
    prm = st["prm"]
    fc_w16, enc_w16 = st["fc_w16"], st["enc_w16"]
    dev = dx.device
    d = spec.d_model
    n_fc0 = 2 if spec.input_norm else 0
    n_fc = n_fc0 + 2 * len(spec.fc_sizes)
    grads = [None] * (n_fc + 16 * spec.n_layers)
    if spec.has_encoder:
        h, dk, dff = spec.heads, spec.dk, spec.d_ff
        for l in reversed(range(spec.n_layers)):
            base = n_fc + 16 * l
            a1, _, Wq, _, _, _, _, _, Wo, _, a2, _, W1, _, W2, _ = prm[base:base + 16]
            wqkv, wo16, w116, w216 = enc_w16[l]
            x0, n1, qkv, ctxb, x1, n2, hid, lse = st["layers"][l]
            # FFN sublayer: x2 = x1 + drop(hid W2^T + b2)
            dy2, gb2 = _drop_cast_colsum(dx, T, d, p_enc, seed, stream_ffn_out(l))
            if st["fused_ffn"]:
                dn2, gW1, gW2, gb1 = ffn_bwd(n2, w116, prm[base + 13], w216, dy2, T, d, dff, p_enc, seed, stream_ffn_hidden(l))
            else:
                gW2 = _weight_grad(dy2, hid, T, d, dff)
                dz1 = torch.empty((T, dff), dtype=_U16, device=dev)
                gemm(dy2, w216, T, dff, d, b_kmajor=True, Cb=dz1, gate=hid, gate_scale=1.0 / (1.0 - p_enc))
                gb1 = _colsum(dz1, T, dff)
                gW1 = _weight_grad(dz1, n2, T, dff, d)
                dn2 = torch.empty((T, d), dtype=torch.float32, device=dev)
                ks = _small_step_splits(T, d, dff)
                if ks > 1:          # few output tiles and a long k loop: split-K partials + one reduce (32 workgroups -> 32 ks)
                    parts = torch.empty((ks, T, d), dtype=torch.float32, device=dev)
                    gemm(dz1, w116, T, d, dff, b_kmajor=True, Cf=parts, splits=ks)
                    sum_partials(parts, ks, T * d, out=dn2.view(-1))
                else:
                    gemm(dz1, w116, T, d, dff, b_kmajor=True, Cf=dn2)
            ga2, gb2n = layernorm_bwd(x1, a2, dn2, T, d, LN_EPS, 0, dx)
            # attention sublayer: x1 = x0 + drop(ctx Wo^T + bo)
            dyo, gbo = _drop_cast_colsum(dx, T, d, p_enc, seed, stream_attn_out(l))
            gWo = _weight_grad(dyo, ctxb, T, d, d)
            dctx = torch.empty((T, d), dtype=_U16, device=dev)
            gemm(dyo, wo16, T, d, d, b_kmajor=True, Cb=dctx)
            dqkv = attention_bwd(qkv, ctxb, dctx, lse, st["mask_u8"], B, S, h, dk, p_enc, seed, stream_attn(l))
            gbqkv = _colsum(dqkv, T, 3 * d)
            gWqkv = _weight_grad(dqkv, n1, T, 3 * d, d)
            dn1 = torch.empty((T, d), dtype=torch.float32, device=dev)
            gemm(dqkv, wqkv, T, d, 3 * d, b_kmajor=True, Cf=dn1)
            ga1, gb1n = layernorm_bwd(x0, a1, dn1, T, d, LN_EPS, 0, dx)
            grads[base:base + 16] = [ga1, gb1n, gWqkv[:d], gbqkv[:d], gWqkv[d:2 * d], gbqkv[d:2 * d], gWqkv[2 * d:],
                                     gbqkv[2 * d:], gWo, gbo, ga2, gb2n, gW1, gb1, gW2, gb2]
    # ---- FCModel backward
    sizes = [F] + spec.fc_sizes
    for i in reversed(range(len(spec.fc_sizes))):
        n_in, n_out = sizes[i], sizes[i + 1]
        dy, gb = _drop_cast_colsum(dx, T, n_out, p_fc, seed, stream_fc(i))
        grads[n_fc0 + 2 * i] = _weight_grad(dy, st["fc_in"][i], T, n_out, n_in)
        grads[n_fc0 + 2 * i + 1] = gb
        if i > 0 or spec.input_norm or want_dx:
            dx = torch.empty((T, n_in), dtype=torch.float32, device=dev)
            gemm(dy, fc_w16[i], T, n_in, n_out, b_kmajor=True, Cf=dx)
    if spec.input_norm:
        scratch = torch.zeros((T, F), dtype=torch.float32, device=dev)
        grads[0], grads[1] = layernorm_bwd(st["xin"], prm[0], dx, T, F, STD_LN_EPS, 1, scratch)
        dx = scratch
    return grads, (dx if want_dx else None)


def encoder_features(spec, x, mask, seed, training, params):
    """prepare_for_output (multiLayer.py:64-72): the encoder output [B, S, d_model] in fp32, with its analytic backward
    (parameters and input): ltr_mi355x.blocks.Features over the FC + encoder-block (+ final norm) parameters."""
    from .blocks import Features
    body = list(params[:-2])                       # everything but the output layer
    return Features.apply(spec, spec.has_encoder, x, mask, seed, training, *body)
