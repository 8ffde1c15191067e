"""GPU parity of the set-transformer scorer (BASELINE config 5): csrc/ltr_encoder.hip through the C ABI.

Tolerances (stated here, used below):
  * kernels in isolation, on bf16-representable inputs, vs torch fp64 on the SAME inputs:
        GEMM / column sums / LayerNorm / scoring tail: 2e-5 (fp32 accumulation order only; bf16 outputs: 2^-8 rounding)
        attention: 1e-2 of the tensor's max (P, dS and the outputs are rounded to bf16 inside the kernel)
  * whole network: the GATE is the fp64 oracle that rounds to bf16 wherever the kernels round, forward AND backward
    (oracle/ltr_encoder_oracle.py, bf16=True, round_bwd=True) -- what is left between the two is summation order and the
    rounding decisions that summation order flips.  That second part is MEASURED, not assumed: the same oracle is run with
    fp32 instead of fp64 accumulation and its deviation from itself (`self-noise`) is what a correct implementation with
    another summation order shows on that network (0.074 max-norm on one bias gradient of the three-block slate-256 golden,
    below 1e-2 on the others).  Bars per parameter tensor, on max(its own scale, 5 % of the case's largest gradient entry):
        max-norm  <= max(2e-2, 4 x self-noise)        L2 <= max(1.5e-2, 2 x self-noise (L2))
    (the max over up to 262 144 entries of two draws of the same noise differs by more than the L2 norm does: factor 4 vs 2;
    the floors are what a network WITHOUT any flipped rounding shows: 1.6e-2 / 1.0e-2 on the two-block slate-12 golden)
        cosine    >= 1 - max(1e-3, 2 x (1 - self cosine))   and   | |got| / |want| - 1 |  <=  max(2e-2, 2 x self)
    (cosine / norm ratio on tensors that are not noise: own max >= 5 % of the case's largest entry);  scores 1e-2.
    The comparison with the reference's fp32 goldens (tests/golden/encoder.npz, encoder_c5.npz) is recorded in the parity
    ledger and sanity-checked only: scores / loss 3e-2, cosine of the whole gradient > 0.99 -- the bf16-operand arithmetic
    BASELINE config 5 asks for, NOT the 1e-5 bar of the fp32 rows."""
import math

import numpy as np
import pytest
import torch

from conftest import golden, ledger_record, relerr, seeded_state_dict

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def enc():
    from ltr_mi355x import encoder
    return encoder


def bits(t):
    """fp32/fp64 tensor -> int16 tensor of bf16 bit patterns (values must already be what we want rounded)."""
    return t.to(torch.bfloat16).view(torch.int16)


def unbits(t):
    return t.view(torch.bfloat16).double()


def rnd(*shape, scale=1.0):
    """bf16-representable random values as fp64."""
    return (torch.randn(*shape, device=DEV) * scale).to(torch.bfloat16).double()


def err(a, b, floor=1e-30):
    a, b = a.double(), b.double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), floor)


# ------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 136), (1000, 384, 128), (77, 8, 8), (256, 2048, 128), (300, 128, 2048)])
def test_gemm_forward_form(enc, M, N, K):
    torch.manual_seed(M + N + K)
    A, W, bias = rnd(M, K), rnd(N, K, scale=0.2), torch.randn(N, device=DEV)
    Cf = torch.empty(M, N, device=DEV)
    Cb = torch.empty(M, N, dtype=torch.int16, device=DEV)
    enc.gemm(bits(A), bits(W), M, N, K, Cf=Cf, Cb=Cb, bias=bias)
    want = A @ W.t() + bias.double()
    assert err(Cf, want) < 2e-5
    assert err(unbits(Cb), want) < 5e-3


def test_gemm_epilogue_relu_dropout_residual_gate(enc):
    torch.manual_seed(7)
    M, N, K, p, seed, sid = 333, 264, 72, 0.25, 1234567, 5
    A, W, bias, res = rnd(M, K), rnd(N, K, scale=0.3), torch.randn(N, device=DEV), torch.randn(M, N, device=DEV)
    keep = enc.dropout_mask(seed, sid, M * N, p, DEV).view(M, N).double()
    assert abs(float(keep.mean()) - (1 - p)) < 0.01
    Cf = torch.empty(M, N, device=DEV)
    enc.gemm(bits(A), bits(W), M, N, K, Cf=Cf, bias=bias, relu=True, drop_p=p, seed=seed, drop_stream=sid, residual=res)
    want = torch.relu(A @ W.t() + bias.double()) * keep / (1 - p) + res.double()
    assert err(Cf, want) < 2e-5
    gate = rnd(M, N)
    Cb = torch.empty(M, N, dtype=torch.int16, device=DEV)
    enc.gemm(bits(A), bits(W), M, N, K, Cb=Cb, gate=bits(gate), gate_scale=1.0 / (1 - p))
    want = (A @ W.t()) * (gate > 0).double() / (1 - p)
    assert err(unbits(Cb), want) < 5e-3


@pytest.mark.parametrize("T,N,K", [(512, 128, 136), (1000, 136, 2048), (77, 264, 8)])
def test_gemm_input_gradient_form(enc, T, N, K):
    """dx = dy W: A = dy [T][N] k-contiguous, B = W [N][K] read k-major."""
    torch.manual_seed(T)
    dy, W = rnd(T, N), rnd(N, K, scale=0.2)
    Cf = torch.empty(T, K, device=DEV)
    enc.gemm(bits(dy), bits(W), T, K, N, b_kmajor=True, Cf=Cf)
    assert err(Cf, dy @ W) < 2e-5


@pytest.mark.parametrize("T,N,K,splits", [(512, 128, 136, 1), (5000, 2048, 128, 7), (1003, 136, 408, 3), (40, 8, 16, 4)])
def test_gemm_weight_gradient_form_split_k(enc, T, N, K, splits):
    """dW = dy^T x: both operands k-major, split over the tokens + fixed-order reduce."""
    torch.manual_seed(T)
    dy, x = rnd(T, N), rnd(T, K)
    parts = torch.empty(splits, N, K, device=DEV)
    enc.gemm(bits(dy), bits(x), N, K, T, a_kmajor=True, b_kmajor=True, Cf=parts, splits=splits)
    got = enc.sum_partials(parts, splits, N * K).view(N, K) if splits > 1 else parts[0]
    assert err(got, dy.t() @ x) < 2e-5
    assert err(enc._weight_grad(bits(dy), bits(x), T, N, K), dy.t() @ x) < 2e-5


def test_gemm_rejects_bad_shapes(enc):
    from ltr_mi355x._lib import LtrError
    a = torch.zeros(16, 12, dtype=torch.int16, device=DEV)
    with pytest.raises(LtrError):
        enc.gemm(a, a, 16, 16, 12, Cf=torch.empty(16, 16, device=DEV))      # K % 8 != 0


# ------------------------------------------------------------------------------------------------- reductions
def test_colsum_and_drop_cast_colsum(enc):
    torch.manual_seed(3)
    for T, N in ((1000, 128), (37, 2048), (5, 8), (4097, 136)):
        y = rnd(T, N)
        assert err(enc._colsum(bits(y), T, N), y.sum(0)) < 2e-5
        dx = torch.randn(T, N, device=DEV)
        p, seed, sid = 0.1, 99, 3
        out, cs = enc._drop_cast_colsum(dx, T, N, p, seed, sid)
        keep = enc.dropout_mask(seed, sid, T * N, p, DEV).view(T, N)
        want = (dx * keep / (1 - p)).to(torch.bfloat16)
        assert torch.equal(out.view(torch.bfloat16), want)
        assert err(cs, want.double().sum(0)) < 2e-5
        out0, _ = enc._drop_cast_colsum(dx, T, N, 0.0, seed, sid)
        assert torch.equal(out0.view(torch.bfloat16), dx.to(torch.bfloat16))


def test_batched_reductions_match_single_and_fp64(enc):
    """ltr_enc_sum_partials_batch (16 jobs per launch, deferred by the backward) == ltr_enc_sum_partials == fp64 sum."""
    torch.manual_seed(13)
    shapes = [(512, 1), (512, 128), (64, 2048), (7, 4096), (16, 262144), (3, 70000), (256, 16384)] * 3      # 21 jobs: 2 launches
    parts = [torch.randn(ns, n, device=DEV) for ns, n in shapes]
    single = [enc.sum_partials(p, p.shape[0], p.shape[1]) for p in parts]
    with enc.deferred_reductions():
        batched = [enc.sum_partials(p, p.shape[0], p.shape[1]) for p in parts]
    for p, a, b in zip(parts, single, batched):
        assert torch.equal(a, b)
        assert err(a, p.double().sum(0)) < 1e-6


# ------------------------------------------------------------------------------------------------- LayerNorm / scoring tail
def _ln_ref(x, a, b, eps, standard):
    if standard:
        return torch.nn.functional.layer_norm(x, (x.shape[-1],), a, b, eps)
    return a * (x - x.mean(-1, keepdim=True)) / (x.std(-1, keepdim=True) + eps) + b


@pytest.mark.parametrize("d", [16, 100, 128, 136, 200, 256, 264, 512])
@pytest.mark.parametrize("standard", [0, 1])
def test_layernorm_fwd_bwd(enc, d, standard):
    torch.manual_seed(d)
    T, eps = 301, 1e-5 if standard else 1e-6
    x = torch.randn(T, d, device=DEV) * 2 + 0.5
    a, b = torch.randn(d, device=DEV), torch.randn(d, device=DEV)
    yb, yf = enc.layernorm_fwd(x, a, b, T, d, eps, standard, want_f32=True)
    xr, ar, br = (t.double().requires_grad_(True) for t in (x, a, b))
    want = _ln_ref(xr, ar, br, eps, standard)
    assert err(yf, want) < 2e-5
    assert err(unbits(yb), want) < 5e-3
    dy = torch.randn(T, d, device=DEV)
    want.backward(dy.double())
    dx0 = torch.randn(T, d, device=DEV)
    dx = dx0.clone()
    ga, gb = enc.layernorm_bwd(x, a, dy, T, d, eps, standard, dx)
    assert err(dx - dx0, xr.grad) < 5e-5
    assert err(ga, ar.grad) < 2e-5 and err(gb, br.grad) < 2e-5


@pytest.mark.parametrize("norm", [0, 1])
def test_score_tail_fwd_bwd(enc, norm):
    from ltr_mi355x._lib import check, lib
    from ltr_mi355x.functional import _ptr, _stream
    torch.manual_seed(11)
    T, d = 515, 136
    x = torch.randn(T, d, device=DEV)
    a, b, w, bias = (torch.randn(n, device=DEV) for n in (d, d, d, 1))
    s = torch.empty(T, device=DEV)
    check(lib().ltr_enc_score_fwd(_ptr(x), _ptr(a), _ptr(b), _ptr(w), _ptr(bias), T, d, 1e-6, norm, _ptr(s), _stream()), "f")
    xr, ar, br, wr, cr = (t.double().requires_grad_(True) for t in (x, a, b, w, bias))
    y = _ln_ref(xr, ar, br, 1e-6, 0) if norm else xr
    want = y @ wr + cr
    assert err(s, want) < 2e-5
    ds = torch.randn(T, device=DEV)
    want.backward(ds.double())
    nblk = 64
    dx = torch.empty(T, d, device=DEV)
    parts = torch.empty(nblk, 3 * d + 8, device=DEV)
    check(lib().ltr_enc_score_bwd(_ptr(x), _ptr(a), _ptr(b), _ptr(w), _ptr(ds), T, d, 1e-6, norm, _ptr(dx), _ptr(parts), nblk,
                                  _stream()), "b")
    g = enc.sum_partials(parts, nblk, 3 * d + 8)
    assert err(dx, xr.grad) < 5e-5
    assert err(g[2 * d:3 * d], wr.grad) < 2e-5 and err(g[3 * d:3 * d + 1], cr.grad) < 2e-5
    if norm:
        assert err(g[:d], ar.grad) < 2e-5 and err(g[d:2 * d], br.grad) < 2e-5


# ------------------------------------------------------------------------------------------------- dropout stream
@pytest.mark.parametrize("p", [0.1, 0.5, 0.0037])
def test_dropout_stream_statistics(enc, p):
    """One 32-bit hash word serves four elements through overlapping 16-bit windows (csrc/ltr_encoder.hip, dropout stream):
    the keep rate must be exact to sampling error, and elements that SHARE a word (lags 1..3 inside a quad) as independent
    as elements that do not (lag 4), within 5 sigma of 4 M samples; two streams / two seeds are uncorrelated."""
    n = 1 << 22
    m = enc.dropout_mask(1234, 3, n, p, DEV).double()
    q = 1 - round(p * 65536) / 65536
    sig = math.sqrt(q * (1 - q) / n)
    assert abs(float(m.mean()) - q) < 5 * sig
    quad = m.view(-1, 4)
    for r in range(4):                                   # every position of the quad on its own
        assert abs(float(quad[:, r].mean()) - q) < 5 * 2 * sig
    var = q * (1 - q)
    for lag in (1, 2, 3, 4, 5, 64):
        c = float(((m[:-lag] - q) * (m[lag:] - q)).mean()) / var
        assert abs(c) < 5 / math.sqrt(n), f"lag {lag}: {c}"
    for a in range(4):                                   # pairs inside one quad (they share the hash word)
        for b in range(a + 1, 4):
            c = float(((quad[:, a] - q) * (quad[:, b] - q)).mean()) / var
            assert abs(c) < 5 / math.sqrt(n / 4), f"quad positions {a},{b}: {c}"
    for other in (enc.dropout_mask(1234, 4, n, p, DEV).double(), enc.dropout_mask(1235, 3, n, p, DEV).double()):
        c = float(((m - q) * (other - q)).mean()) / var
        assert abs(c) < 5 / math.sqrt(n)


# ------------------------------------------------------------------------------------------------- attention
def _attention_ref(q, k, v, pad, keep, p, dk):
    """transformer.py:145-164 on [B, h, S, dk] fp64 tensors; P rounded to bf16 after dropout like the kernel."""
    sc = q @ k.transpose(-2, -1) / math.sqrt(dk)
    sc = sc.masked_fill(pad, float("-inf"))
    pa = torch.softmax(sc, dim=-1)
    if keep is not None:
        pa = pa * keep / (1 - p)
    return pa @ v


@pytest.mark.parametrize("B,S,h,dk,p", [(2, 12, 4, 8, 0.0), (3, 33, 8, 17, 0.1), (2, 100, 8, 16, 0.1), (1, 256, 4, 32, 0.0),
                                        (2, 256, 8, 16, 0.2), (1, 300, 2, 24, 0.1), (1, 512, 1, 32, 0.0)])
def test_attention_fwd_bwd(enc, B, S, h, dk, p):
    from ltr_mi355x._lib import check, lib
    from ltr_mi355x.functional import _ptr, _stream
    torch.manual_seed(S * h + dk)
    d, T, seed, sid = h * dk, B * S, 4242, 16
    qkv = rnd(T, 3 * d, scale=1.5)
    mask = torch.zeros(B, S, dtype=torch.uint8, device=DEV)
    mask[0, S - S // 4:] = 1
    ctx = torch.empty(T, d, dtype=torch.int16, device=DEV)
    # (operand temporaries are held in names: a tensor created inside the argument list is freed -- and its block handed to
    #  the next allocation -- before the asynchronous launch has read it)
    qkv16 = bits(qkv)
    check(lib().ltr_enc_attention_fwd(_ptr(qkv16), _ptr(mask), B, S, h, dk, p, seed, sid, _ptr(ctx), _stream()), "fwd")
    keep = enc.attn_dropout_mask(seed, sid, B, S, h, p, DEV).double() if p else None
    if p:
        assert abs(float(keep.mean()) - (1 - p)) < 0.02
    qr = qkv.clone().requires_grad_(True)
    q, k, v = (qr[:, j * d:(j + 1) * d].view(B, S, h, dk).transpose(1, 2) for j in range(3))
    want = _attention_ref(q, k, v, (mask == 1).view(B, 1, 1, S), keep, p, dk).transpose(1, 2).reshape(T, d)
    assert err(unbits(ctx), want) < 1e-2
    dctx = rnd(T, d)
    want.backward(dctx)
    dqkv = torch.empty(T, 3 * d, dtype=torch.int16, device=DEV)
    dctx16 = bits(dctx)
    check(lib().ltr_enc_attention_bwd(_ptr(qkv16), _ptr(ctx), _ptr(dctx16), _ptr(mask), B, S, h, dk, p, seed, sid, _ptr(dqkv),
                                      _stream()), "bwd")
    got = unbits(dqkv)
    for j, name in enumerate("qkv"):
        e = err(got[:, j * d:(j + 1) * d], qr.grad[:, j * d:(j + 1) * d])
        assert e < 1.5e-2, f"d{name}: {e}"


@pytest.mark.parametrize("B,S,h,dk,p", [(2, 12, 4, 8, 0.0), (3, 40, 2, 16, 0.1), (2, 100, 8, 16, 0.1), (9, 128, 8, 16, 0.1), (2, 200, 4, 12, 0.0),
                                        (2, 256, 8, 16, 0.2), (10, 256, 8, 16, 0.1), (3, 250, 8, 16, 0.0),
                                        (2, 256, 8, 17, 0.1), (9, 256, 2, 32, 0.1), (3, 100, 4, 24, 0.1), (5, 128, 8, 17, 0.1), (2, 250, 2, 32, 0.0)])
def test_attention_key_major_backward(enc, B, S, h, dk, p):
    """ltr_enc_attention_fwd_lse / _bwd_lse (every probability evaluated once, key-major, for dk <= 16 and S <= 256; the dk 17 / 24 / 32
    cases take the two-phase kernel behind the same entry and exercise the funnel-shifted loads of unaligned head slices): against the
    fp64 reference at the bf16 bar of test_attention_fwd_bwd, against the two-phase kernel on the same inputs (same rounding
    points; the only difference is exp2(c2 s - lse2) for exp2(c2 s - max) / sum in the dQ contraction), bit-reproducible,
    and lse2 itself against log2-sum-exp of the reference scores."""
    from ltr_mi355x._lib import check, lib
    from ltr_mi355x.functional import _ptr, _stream
    torch.manual_seed(S * h + dk + B)
    d, T, seed, sid = h * dk, B * S, 99, 8
    qkv = rnd(T, 3 * d, scale=1.5)
    mask = torch.zeros(B, S, dtype=torch.uint8, device=DEV)
    mask[0, S - S // 4:] = 1
    if B > 2:
        mask[2] = 1                              # a slate without any unmasked document: zeros, lse2 = +inf
    qkv16, ctx, ctx0 = bits(qkv), torch.empty(T, d, dtype=torch.int16, device=DEV), torch.empty(T, d, dtype=torch.int16, device=DEV)
    lse = torch.full((B * h, S), float("nan"), device=DEV)
    check(lib().ltr_enc_attention_fwd_lse(_ptr(qkv16), _ptr(mask), B, S, h, dk, p, seed, sid, _ptr(ctx), _ptr(lse), _stream()), "fwd")
    check(lib().ltr_enc_attention_fwd(_ptr(qkv16), _ptr(mask), B, S, h, dk, p, seed, sid, _ptr(ctx0), _stream()), "fwd")
    assert torch.equal(ctx, ctx0)
    q, k, v = (unbits(qkv16)[:, j * d:(j + 1) * d].double().view(B, S, h, dk).transpose(1, 2) for j in range(3))
    sc = (q @ k.transpose(-2, -1) / math.sqrt(dk)).masked_fill((mask == 1).view(B, 1, 1, S), float("-inf"))
    want_lse = torch.logsumexp(sc, -1) / math.log(2.0)
    live = torch.isfinite(want_lse)
    got_lse = lse.view(B, h, S).double()
    assert torch.all(torch.isinf(got_lse[~live]) & (got_lse[~live] > 0))
    assert float((got_lse[live] - want_lse[live]).abs().max()) < 2e-5 * max(1.0, float(want_lse[live].abs().max()))
    dctx16 = bits(rnd(T, d))
    out = [torch.empty(T, 3 * d, dtype=torch.int16, device=DEV) for _ in range(3)]
    for o in out[:2]:
        check(lib().ltr_enc_attention_bwd_lse(_ptr(qkv16), _ptr(ctx), _ptr(dctx16), _ptr(lse), _ptr(mask), B, S, h, dk, p, seed, sid,
                                              _ptr(o), _stream()), "bwd_lse")
    check(lib().ltr_enc_attention_bwd(_ptr(qkv16), _ptr(ctx), _ptr(dctx16), _ptr(mask), B, S, h, dk, p, seed, sid, _ptr(out[2]),
                                      _stream()), "bwd")
    assert torch.equal(out[0], out[1])
    km, two = unbits(out[0]), unbits(out[2])
    for j, name in enumerate("qkv"):
        e = err(km[:, j * d:(j + 1) * d], two[:, j * d:(j + 1) * d])
        assert e < 8e-3, f"d{name} vs the two-phase kernel: {e}"           # one bf16 ulp of the largest entry = 3.9e-3
    # fp64 reference through autograd, dropout keep mask exported from the same stream
    keep = enc.attn_dropout_mask(seed, sid, B, S, h, p, DEV).double() if p else None
    qr = unbits(qkv16).double().clone().requires_grad_(True)
    q, k, v = (qr[:, j * d:(j + 1) * d].view(B, S, h, dk).transpose(1, 2) for j in range(3))
    okb = [b for b in range(B) if not bool((mask[b] == 1).all())]      # the reference yields NaN for an all-masked slate
    want = _attention_ref(q[okb], k[okb], v[okb], (mask[okb] == 1).view(len(okb), 1, 1, S), None if keep is None else keep[okb], p, dk)
    want = want.transpose(1, 2).reshape(len(okb) * S, d)
    rows = torch.cat([torch.arange(b * S, (b + 1) * S, device=DEV) for b in okb])
    want.backward(unbits(dctx16)[rows].double())
    for j, name in enumerate("qkv"):
        e = err(km[rows, j * d:(j + 1) * d], qr.grad[rows, j * d:(j + 1) * d])
        assert e < 1.5e-2, f"d{name}: {e}"
    dead = [b for b in range(B) if b not in okb]
    for b in dead:
        assert float(km[b * S:(b + 1) * S].abs().max()) == 0.0


def test_attention_without_mask_and_all_masked_slate(enc):
    from ltr_mi355x._lib import check, lib
    from ltr_mi355x.functional import _ptr, _stream
    B, S, h, dk = 2, 40, 2, 16
    d, T = h * dk, B * S
    qkv = rnd(T, 3 * d)
    ctx = torch.empty(T, d, dtype=torch.int16, device=DEV)
    qkv16 = bits(qkv)
    check(lib().ltr_enc_attention_fwd(_ptr(qkv16), None, B, S, h, dk, 0.0, 0, 0, _ptr(ctx), _stream()), "fwd")
    q, k, v = (qkv[:, j * d:(j + 1) * d].view(B, S, h, dk).transpose(1, 2) for j in range(3))
    want = _attention_ref(q, k, v, torch.zeros(B, 1, 1, S, dtype=torch.bool, device=DEV), None, 0.0, dk).transpose(1, 2).reshape(T, d)
    assert err(unbits(ctx), want) < 1e-2
    mask = torch.zeros(B, S, dtype=torch.uint8, device=DEV)
    mask[1] = 1                                  # the reference yields NaN for this slate; the kernel yields zeros (header)
    check(lib().ltr_enc_attention_fwd(_ptr(qkv16), _ptr(mask), B, S, h, dk, 0.0, 0, 0, _ptr(ctx), _stream()), "fwd")
    got = unbits(ctx).view(B, S, d)
    assert err(got[0], want.view(B, S, d)[0]) < 1e-2 and float(got[1].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------- fused FFN
@pytest.mark.parametrize("T,d,dff,p", [(700, 128, 384, 0.1), (256, 128, 128, 0.0), (1000, 64, 256, 0.2), (130, 64, 128, 0.0),
                                       (4096, 128, 2048, 0.1), (66000, 128, 256, 0.1), (65536, 64, 128, 0.0)])
def test_fused_ffn_matches_the_gemm_path(enc, T, d, dff, p):
    """ltr_enc_ffn_* (hidden activation in registers, recomputed in the backward) vs the same sublayer through
    ltr_enc_gemm_bf16: identical rounding points, so only the fp32 summation order differs.  T < 65 536 takes the
    128-tokens-per-workgroup variants of the forward / input-gradient kernels, T >= 65 536 the 256-token ones."""
    torch.manual_seed(T + d + dff)
    seed, sh, so = 987654321, 10, 11
    n2, dy = rnd(T, d), rnd(T, d, scale=0.3)
    w1, w2 = rnd(dff, d, scale=0.15), rnd(d, dff, scale=0.15)
    b1, b2, x1 = torch.randn(dff, device=DEV) * 0.2, torch.randn(d, device=DEV), torch.randn(T, d, device=DEV)
    n2b, dyb, w1b, w2b = bits(n2), bits(dy), bits(w1), bits(w2)
    hid = torch.empty(T, dff, dtype=torch.int16, device=DEV)
    enc.gemm(n2b, w1b, T, dff, d, Cb=hid, bias=b1, relu=True, drop_p=p, seed=seed, drop_stream=sh)
    x2 = torch.empty(T, d, device=DEV)
    enc.gemm(hid, w2b, T, d, dff, Cf=x2, bias=b2, residual=x1, drop_p=p, seed=seed, drop_stream=so)
    got = enc.ffn_fwd(n2b, w1b, b1, w2b, b2, x1, T, d, dff, p, seed, sh, so)
    assert err(got, x2) < 2e-5
    dz1 = torch.empty(T, dff, dtype=torch.int16, device=DEV)
    enc.gemm(dyb, w2b, T, dff, d, b_kmajor=True, Cb=dz1, gate=hid, gate_scale=1.0 / (1.0 - p))
    dn2 = torch.empty(T, d, device=DEV)
    enc.gemm(dz1, w1b, T, d, dff, b_kmajor=True, Cf=dn2)
    gW2, gW1, gb1 = enc._weight_grad(dyb, hid, T, d, dff), enc._weight_grad(dz1, n2b, T, dff, d), enc._colsum(dz1, T, dff)
    f_dn2, f_W1, f_W2, f_b1 = enc.ffn_bwd(n2b, w1b, b1, w2b, dyb, T, d, dff, p, seed, sh)
    # dz is rounded to bf16 from fp32 sums that differ in their last bits between the two paths: a few entries land on
    # the other side of a rounding boundary (2^-9 relative each)
    assert err(f_dn2, dn2) < 2e-3
    assert err(f_W1, gW1) < 1e-3 and err(f_W2, gW2) < 2e-5 and err(f_b1, gb1) < 1e-3


# ------------------------------------------------------------------------------------------------- whole network
def _build(case, g):
    from architeture.multiLayer import make_model
    import copy
    net = make_model(fc_model=copy.deepcopy(case["fc_model"]), transformer=copy.deepcopy(case["transformer"]),
                     post_model=dict(d_output=1, output_activation="Sigmoid"), n_features=case["n_features"])
    sd = {k: torch.from_numpy(g.arr(case, "w/" + k)) for k in case["keys"]}
    assert list(net.state_dict().keys()) == case["keys"]
    net.load_state_dict(sd)
    return net.to(DEV), sd


_NOISE_SCALED = {}


def _max_norm_factor_4():
    """Named single-entry outliers (tests/golden/encoder_noise_scaled.json): max-norm gated at 4 x the self-test's deviation."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "encoder_noise_scaled.json")
    if not os.path.exists(path):
        return set()
    with open(path) as f:
        return set(json.load(f).get("max_norm_factor_4", []))


def _noise_scaled(what, tensor, n_max, n_l2, e_max, e_l2):
    """A tensor gated at the noise-scaled bar: recorded (gpurun_out/encoder_noise_scaled.json) and checked against the committed
    list -- a NEW tensor drifting above the fixed bar fails here instead of quietly widening its own bar."""
    import json
    import os
    key = f"{what}::{tensor}"
    _NOISE_SCALED[key] = {"self_noise_max": n_max, "self_noise_l2": n_l2, "kernel_max": e_max, "kernel_l2": e_l2}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "encoder_noise_scaled.json"), "w") as f:
        json.dump(_NOISE_SCALED, f, indent=1)
    allowed = os.path.join(root, "tests", "golden", "encoder_noise_scaled.json")
    if os.path.exists(allowed):
        with open(allowed) as f:
            names = set(json.load(f)["tensors"])
        assert key in names, ("tensor above the fixed bf16 bar in the oracle's own self-test, but not on the committed list", key, n_max, n_l2)


def _oracle_gate(got, scores, sd, x, mask, cfg, y, keep=None, what="", loss_fn=None, want_dx=False, out_post=None):
    """Compare the output / parameter gradients of the HIP path with the rounding-faithful fp64 oracle at bars derived from
    the oracle's own fp32-vs-fp64 deviation (module docstring).  `loss_fn(out, dtype)`: the objective (default: approxNDCG
    against y).  Every tensor is measured first, then the bars are asserted (the message lists all offenders).  Returns the
    ledger numbers."""
    import ltr_encoder_oracle as EO
    import ltr_oracle as O
    res = {}
    for dt in (torch.float64, torch.float32):
        fn = (lambda s_, dt=dt: loss_fn(s_, dt)) if loss_fn is not None else (lambda s_, dt=dt: O.approx_ndcg(s_, y.cpu().to(dt)))
        res[dt] = EO.scores_and_grads(sd, x.cpu(), None if mask is None else mask.cpu(), cfg, fn, keep=keep, bf16=True, round_bwd=True,
                                      dtype=dt, want_dx=want_dx)
    s_o, _, g_o = res[torch.float64]
    g_o = {k: v.double() for k, v in g_o.items()}
    g_n = {k: v.double() for k, v in res[torch.float32][2].items()}
    if out_post is not None:
        s_o = out_post(s_o)
    e_s = relerr(scores.detach().cpu().numpy(), s_o.numpy())
    assert e_s < 1e-2, (what, "output", e_s)
    gmax = max(float(v.abs().max()) for v in g_o.values())
    out = dict(max=0.0, l2=0.0, noise_max=0.0, noise_l2=0.0, min_cos=1.0, scores=e_s)
    bad = []
    cos = lambda a_, b_: float(a_.flatten() @ b_.flatten() / max(float(a_.norm() * b_.norm()), 1e-300))      # noqa: E731
    for k, want in g_o.items():
        g, n = got[k].double().reshape(want.shape), g_n[k]
        e_max, n_max = _gerr(g, want, gmax), _gerr(n, want, gmax)
        e_l2, n_l2 = _l2err(g, want, gmax), _l2err(n, want, gmax)
        # Bars: max-norm max(2e-2, 2 x self-noise), L2 max(1.5e-2, 1.5 x self-noise) (round 3: 4 x / 2 x) -- the fixed part for every
        # tensor the oracle's own fp32-accumulation self-test leaves well inside it, the noise-scaled part only where that self-test
        # itself comes close to or passes the fixed bar (summation order flips bf16 roundings there for ANY implementation: on the
        # three-block slate-256 golden 24 tensors sit between 2e-2 and 9.2e-2 in the self-test alone).  Tensors the self-test puts
        # ABOVE the fixed bar are named and must be on the committed list (tests/golden/encoder_noise_scaled.json): a new tensor
        # drifting up there fails instead of widening its own bar.
        if n_max > 2e-2 or n_l2 > 1.5e-2:
            _noise_scaled(what, k, n_max, n_l2, e_max, e_l2)
        if e_max > max(2e-2, (4 if f"{what}::{k}" in _max_norm_factor_4() else 2) * n_max):
            bad.append((k, "max-norm", e_max, "self-noise", n_max))
        if e_l2 > max(1.5e-2, 1.5 * n_l2):
            bad.append((k, "L2", e_l2, "self-noise", n_l2))
        if float(want.abs().max()) >= 0.05 * gmax:
            c, cn = cos(g, want), cos(n, want)
            r, rn = float(g.norm() / want.norm()), float(n.norm() / want.norm())
            if c < 1 - max(1e-3, 2 * (1 - cn)):
                bad.append((k, "cosine", c, "self", cn))
            if abs(r - 1) > max(2e-2, 2 * abs(rn - 1)):
                bad.append((k, "norm ratio", r, "self", rn))
            out["min_cos"] = min(out["min_cos"], c)
        out.update(max=max(out["max"], e_max), l2=max(out["l2"], e_l2), noise_max=max(out["noise_max"], n_max),
                   noise_l2=max(out["noise_l2"], n_l2))
    assert not bad, (what, bad)
    return out


@pytest.mark.parametrize("case", golden("encoder").cases, ids=lambda c: c["id"])
def test_network_vs_reference_golden(case):
    import ltr_encoder_oracle as EO
    from losses.approxNDCG import approxNDCGLoss
    g = golden("encoder")
    net, sd = _build(case, g)
    net.eval()
    x = torch.from_numpy(g.arr(case, "x")).to(DEV)
    y = torch.from_numpy(g.arr(case, "y")).to(DEV)
    mask = torch.from_numpy(g.arr(case, "mask")).to(DEV) if case["has_mask"] else None
    scores = net(x, mask, None)
    assert scores.shape == (case["B"], case["S"])
    loss = approxNDCGLoss(scores, y)
    loss.backward()
    got = {k: p.grad.cpu().double() for k, p in net.named_parameters()}
    # (1) the gate: the rounding-faithful oracle at measured-noise bars
    cfg = EO.config_of(dict(fc_model=case["fc_model"], transformer=case["transformer"]), case["n_features"])
    gate = _oracle_gate(got, scores, sd, x, mask, cfg, y, what=case["id"])
    # (2) the reference's fp32 goldens: ledger + sanity
    want_s = g.arr(case, "scores")
    assert relerr(scores.detach().cpu().numpy(), want_s) < 3e-2
    assert abs(float(loss) - float(g.arr(case, "loss"))) < 3e-2 * abs(float(g.arr(case, "loss")))
    ref = {k: torch.from_numpy(g.arr(case, "g/" + k)).double() for k in case["keys"]}
    gmax = max(float(v.abs().max()) for v in ref.values())
    flat_got, flat_ref = torch.cat([got[k].flatten() for k in ref]), torch.cat([ref[k].flatten() for k in ref])
    cos = float(flat_got @ flat_ref / (flat_got.norm() * flat_ref.norm()))
    assert cos > 0.99, cos
    loose = max(_gerr(got[k], ref[k], gmax) for k in ref)
    note = "bf16-operand network (BASELINE config 5): bars are the bf16 ones stated in tests/test_encoder_gpu.py, not 1e-5"
    ledger_record("encoder worst param-grad vs rounding-faithful oracle (max-norm)", gate["max"], noise=gate["noise_max"],
                  tol=max(2e-2, 4 * gate["noise_max"]), note=note + f"; min cosine {gate['min_cos']:.6f}")
    ledger_record("encoder worst param-grad vs rounding-faithful oracle (L2)", gate["l2"], noise=gate["noise_l2"],
                  tol=max(1.5e-2, 2 * gate["noise_l2"]), note=note)
    ledger_record("encoder scores vs reference fp32 (ledger only)", relerr(scores.detach().cpu().numpy(), want_s), tol=3e-2, note=note)
    ledger_record("encoder worst param-grad vs reference fp32 (ledger only, not a gate)", loose, tol=1.0, asserted=False,
                  note=note + f"; whole-gradient cosine {cos:.5f}")


def test_benched_config5_network_vs_reference_golden():
    """The network bench.py / tools/bench_encoder.py time for BASELINE config 5 (FC 136 -> 128, 6 blocks, 8 heads, d_ff 2048,
    slate 256: fused-FFN kernels at 16 chunks, XCD-remapped attention), end to end against the reference."""
    import copy
    import ltr_encoder_oracle as EO
    from architeture.multiLayer import make_model
    from losses.approxNDCG import approxNDCGLoss
    g = golden("encoder_c5")
    case = g.cases[0]
    net = make_model(fc_model=copy.deepcopy(case["fc_model"]), transformer=copy.deepcopy(case["transformer"]),
                     post_model=dict(d_output=1, output_activation=None), n_features=case["n_features"])
    assert [[k, list(v.shape)] for k, v in net.state_dict().items()] == case["shapes"]
    sd = seeded_state_dict(case["shapes"], case["weight_seed"])
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    from ltr_mi355x import encoder as enc_mod
    assert enc_mod.fused_ffn_enabled(128, 2048)
    x, y = torch.from_numpy(g.arr(case, "x")).to(DEV), torch.from_numpy(g.arr(case, "y")).to(DEV)
    mask = torch.from_numpy(g.arr(case, "mask")).to(DEV)
    scores = net(x, mask, None)
    loss = approxNDCGLoss(scores, y)
    loss.backward()
    got = {k: p.grad.cpu().double() for k, p in net.named_parameters()}
    cfg = EO.config_of(dict(fc_model=case["fc_model"], transformer=case["transformer"]), case["n_features"])
    gate = _oracle_gate(got, scores, sd, x, mask, cfg, y, what=case["id"])
    # the reference's own numbers: scores, loss, every vector gradient in full, of every matrix gradient 4 rows + norm + sum
    e_s = relerr(scores.detach().cpu().numpy(), g.arr(case, "scores"))
    assert e_s < 3e-2
    assert abs(float(loss) - float(g.arr(case, "loss"))) < 3e-2 * abs(float(g.arr(case, "loss")))
    gmax = max([float(np.abs(g.arr(case, "g/" + k)).max()) for k, sh in case["shapes"] if len(sh) == 1] +
               [float(g.arr(case, "gstat/" + k)[2]) for k, sh in case["shapes"] if len(sh) == 2])
    worst, worst_norm = 0.0, 0.0
    for k, sh in case["shapes"]:
        if len(sh) == 1:
            ref = torch.from_numpy(g.arr(case, "g/" + k)).double()
            gk = got[k]
        else:
            ref = torch.from_numpy(g.arr(case, "grows/" + k)).double()
            gk = got[k][:4]
            nrm = float(g.arr(case, "gstat/" + k)[0])
            worst_norm = max(worst_norm, abs(float(got[k].norm()) - nrm) / max(nrm, 0.05 * gmax * math.sqrt(got[k].numel())))
        worst = max(worst, _gerr(gk, ref, gmax))
    note = "benched config-5 network (3.6 M parameters) vs the reference's fp32 CPU run; bf16 bars (tests/test_encoder_gpu.py)"
    assert worst_norm < 5e-2, worst_norm
    ledger_record("config-5 network scores vs reference fp32", e_s, tol=3e-2, note=note)
    ledger_record("config-5 network worst param-grad vs reference fp32 (stored rows; ledger only)", worst, tol=1.0, note=note, asserted=False)
    ledger_record("config-5 network worst matrix-gradient norm deviation vs reference fp32", worst_norm, tol=5e-2, note=note)
    ledger_record("config-5 network worst param-grad vs rounding-faithful oracle (max-norm)", gate["max"], noise=gate["noise_max"],
                  tol=max(2e-2, 4 * gate["noise_max"]), note=note + f"; min cosine {gate['min_cos']:.6f}")
    ledger_record("config-5 network worst param-grad vs rounding-faithful oracle (L2)", gate["l2"], noise=gate["noise_l2"],
                  tol=max(1.5e-2, 2 * gate["noise_l2"]), note=note)


def _l2err(a, b, gmax):
    """|a-b|_2 / |b|_2 with the same 5 % floor per entry."""
    return float((a - b).norm()) / max(float(b.norm()), 0.05 * gmax * math.sqrt(b.numel()))


def _gerr(a, b, gmax):
    """max|a-b| over max(|b|max, 5 % of the case's largest gradient entry)."""
    return float((a - b).abs().max()) / max(float(b.abs().max()), 0.05 * gmax)


def test_network_train_mode_dropout_matches_oracle_under_exported_masks(enc):
    import ltr_encoder_oracle as EO
    from architeture.multiLayer import make_model
    from losses.approxNDCG import approxNDCGLoss
    torch.manual_seed(5)
    F, B, S = 24, 3, 40
    fc = dict(sizes=[48, 32], input_norm=False, activation=None, dropout=0.2)
    tr = dict(N=2, d_ff=64, h=4, dropout=0.1, positional_encoding=None)
    import copy
    net = make_model(copy.deepcopy(fc), copy.deepcopy(tr), dict(d_output=1, output_activation=None), F).to(DEV)
    net.train()
    x = torch.randn(B, S, F, device=DEV)
    y = torch.randint(0, 5, (B, S), device=DEV).float()
    mask = torch.zeros(B, S, dtype=torch.bool, device=DEV)
    mask[1, 30:] = True
    y[mask] = -1
    net.ltr_seed = 77
    scores = net(x, mask, None)
    seed = (77 + 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
    approxNDCGLoss(scores, y).backward()
    T, d, dff, h = B * S, 32, 64, 4
    keep = {("fc", 0): enc.dropout_mask(seed, enc.stream_fc(0), T * 48, 0.2, DEV).view(T, 48).cpu(),
            ("fc", 1): enc.dropout_mask(seed, enc.stream_fc(1), T * 32, 0.2, DEV).view(T, 32).cpu()}
    for l in range(2):
        keep[("attn", l)] = enc.attn_dropout_mask(seed, enc.stream_attn(l), B, S, h, 0.1, DEV).cpu()
        keep[("attn_out", l)] = enc.dropout_mask(seed, enc.stream_attn_out(l), T * d, 0.1, DEV).view(T, d).cpu()
        keep[("ffn_hidden", l)] = enc.dropout_mask(seed, enc.stream_ffn_hidden(l), T * dff, 0.1, DEV).view(T, dff).cpu()
        keep[("ffn_out", l)] = enc.dropout_mask(seed, enc.stream_ffn_out(l), T * d, 0.1, DEV).view(T, d).cpu()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    cfg = EO.config_of(dict(fc_model=fc, transformer=tr), F)
    got = {k: p.grad.cpu().double() for k, p in net.named_parameters()}
    gate = _oracle_gate(got, scores, sd, x, mask, cfg, y, keep=keep, what="train-mode dropout")
    ledger_record("encoder train-mode worst param-grad vs rounding-faithful oracle under exported masks (max-norm)", gate["max"],
                  noise=gate["noise_max"], tol=max(2e-2, 4 * gate["noise_max"]), note="bf16 bars, tests/test_encoder_gpu.py")
    # a second training forward draws new masks; eval mode is deterministic and mask-free
    s2 = net(x, mask, None)
    assert not torch.equal(s2, scores)
    net.eval()
    assert torch.equal(net(x, mask, None), net(x, mask, None))


def test_small_steps_take_the_gemm_ffn_path_by_default(enc, monkeypatch):
    """Below 8 193 tokens per step the encoder's FFN runs as two GEMMs (faster there: a fused-FFN workgroup walks all d_ff chunks of
    its tokens serially), from there on fused; LTR_ENC_FUSED_FFN = 0 / 1 force either.  The two paths agree on a whole training step
    under the same dropout seed (bf16 operands on both: scores to 1e-2 of their range, every gradient to 5e-2 of its max-norm)."""
    from architeture.multiLayer import make_model
    from losses.approxNDCG import approxNDCGLoss
    from ltr_mi355x import blocks, encoder as E
    from architeture.multiLayer import LTRModel
    monkeypatch.delenv("LTR_ENC_FUSED_FFN", raising=False)
    assert not E.fused_ffn_enabled(128, 2048, 4096) and E.fused_ffn_enabled(128, 2048, 8193) and E.fused_ffn_enabled(128, 2048)
    assert not E.fused_ffn_enabled(136, 2048, 1 << 20)            # d_model outside {64, 128}: never
    monkeypatch.setenv("LTR_ENC_FUSED_FFN", "0")
    assert not E.fused_ffn_enabled(128, 2048, 1 << 20)
    monkeypatch.setenv("LTR_ENC_FUSED_FFN", "1")
    assert E.fused_ffn_enabled(128, 2048, 16)
    monkeypatch.setattr(LTRModel, "_ltr_next_seed", lambda self: 77)
    torch.manual_seed(5)
    assert E._small_step_splits(384, 128, 1024) == 2 and E._small_step_splits(4096, 128, 2048) == 4 and E._small_step_splits(65536, 128, 2048) == 1
    net = make_model(dict(sizes=[128], input_norm=False, activation=None, dropout=0.0), dict(N=2, d_ff=1024, h=8, dropout=0.1,
                     positional_encoding=None), dict(d_output=1, output_activation=None), 136).to(DEV).train()
    x = torch.randn(6, 64, 136, device=DEV)
    y = torch.randint(0, 5, (6, 64), device=DEV).float()
    mask = torch.zeros(6, 64, dtype=torch.bool, device=DEV)
    res = []
    for env in ("1", None):                                       # fused (forced) / default = GEMMs at 384 tokens
        if env is None:
            monkeypatch.delenv("LTR_ENC_FUSED_FFN", raising=False)
        else:
            monkeypatch.setenv("LTR_ENC_FUSED_FFN", env)
        net.zero_grad()
        s_ = net(x, mask, None)
        approxNDCGLoss(s_, y).backward()
        res.append((s_.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}))
    (sa, ga), (sb, gb) = res
    assert float((sa - sb).abs().max() / sa.abs().max()) < 1e-2
    top = max(float(v.abs().max()) for v in ga.values())
    for k in ga:                                                  # (the key bias has an exactly-zero gradient: noise against the top gradient)
        assert float((ga[k] - gb[k]).abs().max()) / max(float(ga[k].abs().max()), 1e-2 * top) < 5e-2, k


def test_network_api_errors_and_features(enc):
    from architeture.multiLayer import make_model
    from ltr_mi355x._lib import LtrDeviceError
    tr = dict(N=1, d_ff=32, h=2, dropout=0.0, positional_encoding=None)
    net = make_model(None, tr, dict(d_output=1), 16).to(DEV).eval()
    x = torch.randn(2, 10, 16, device=DEV)
    with pytest.raises(AttributeError):
        net(x, None, None)                      # transformer.py:55 dereferences the mask
    with pytest.raises(LtrDeviceError):
        net.cpu()(x.cpu(), torch.zeros(2, 10), None)
    net.to(DEV)
    mask = torch.zeros(2, 10, dtype=torch.bool, device=DEV)
    feats = net.prepare_for_output(x, mask, None)
    assert feats.shape == (2, 10, 16)
    w, b = net.output_layer.w_1.weight.detach(), net.output_layer.w_1.bias.detach()
    assert relerr((feats.detach() @ w.t() + b).squeeze(2).cpu().numpy(), net.score(x, mask, None).detach().cpu().numpy()) < 1e-5
    fc_only = make_model(dict(sizes=[32, 8], input_norm=False, activation=None, dropout=0.0), None, dict(d_output=1), 16).to(DEV)
    s = fc_only(x, None, None)                  # no encoder: mask may be None (main_batch_execution.py:124)
    assert s.shape == (2, 10)
    with pytest.raises(ValueError):
        make_model(dict(sizes=[30], input_norm=False, activation=None, dropout=0.0), None, dict(d_output=1), 16).to(DEV)(x, None, None)


def test_network_edge_shapes(enc):
    """Empty batch, one-document slates, tiny and ragged slate lengths, every mask dtype the callers use."""
    import ltr_encoder_oracle as EO
    from architeture.multiLayer import make_model
    torch.manual_seed(3)
    fc, tr = dict(sizes=[32], input_norm=False, activation=None, dropout=0.0), dict(N=1, d_ff=64, h=4, dropout=0.0, positional_encoding=None)
    import copy
    net = make_model(copy.deepcopy(fc), copy.deepcopy(tr), dict(d_output=1), 16).to(DEV).eval()
    sd = {k: v.detach().cpu().double() for k, v in net.state_dict().items()}
    cfg = EO.config_of(dict(fc_model=fc, transformer=tr), 16)
    out = net(torch.zeros(0, 7, 16, device=DEV), torch.zeros(0, 7, dtype=torch.bool, device=DEV), None)
    assert out.shape == (0, 7)
    out.sum().backward()
    assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for p in net.parameters())
    for B, S in ((3, 1), (2, 2), (1, 7), (5, 31), (2, 257)):
        x = torch.randn(B, S, 16, device=DEV)
        m = torch.zeros(B, S, dtype=torch.bool, device=DEV)
        if S > 2:
            m[0, S - 1] = True
        want = EO.encoder_scores(sd, x.cpu().double(), m.cpu(), cfg, bf16=True)
        for mask in (m, m.to(torch.uint8), m.to(torch.float32), m.to(torch.int64)):
            got = net(x, mask, None)
            assert got.shape == (B, S)
            assert relerr(got.detach().cpu().numpy(), want.numpy()) < 1e-2, (B, S, mask.dtype)


def test_network_trains(enc):
    """A few Adam steps on a fixed batch reduce the approxNDCG loss (the whole fwd/bwd chain has the right sign)."""
    from architeture.multiLayer import make_model
    from losses.approxNDCG import approxNDCGLoss
    torch.manual_seed(1)
    net = make_model(dict(sizes=[64], input_norm=False, activation=None, dropout=0.0),
                     dict(N=2, d_ff=128, h=4, dropout=0.0, positional_encoding=None), dict(d_output=1), 136).to(DEV)
    x = torch.randn(16, 64, 136, device=DEV)
    y = torch.randint(0, 5, (16, 64), device=DEV).float()
    mask = torch.zeros(16, 64, dtype=torch.bool, device=DEV)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = approxNDCGLoss(net(x, mask, None), y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0] - 0.05, losses
