#!/usr/bin/env python3
"""Per-kernel timings of csrc/ltr_encoder.hip at BASELINE config 5 shapes (T = B x 256 tokens, d_model 128, d_ff 2048,
8 heads): one JSON line per kernel with its algorithmic FLOP/s and bytes/s.  torch.cuda.Event on the current stream
(the launches go to torch's current stream)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"))

import torch  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--slate", type=int, default=256)
    ap.add_argument("--only", type=str, default="")
    a = ap.parse_args()
    from ltr_mi355x import encoder as E
    from ltr_mi355x._lib import check, lib
    from ltr_mi355x.functional import _ptr, _stream
    dev = "cuda:0"
    B, S, d, dff, h = a.batch, a.slate, 128, 2048, 8
    T, dk = B * S, d // h
    bf = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16).view(torch.int16)
    out = []

    def rec(name, sec, flops=0, bytes_=0):
        if a.only and a.only not in name:
            return
        r = {"kernel": name, "us": round(sec * 1e6, 1), "TFLOPs": round(flops / sec / 1e12, 1), "GBps": round(bytes_ / sec / 1e9, 1)}
        print(json.dumps(r), flush=True)

    x, w1, w2, wqkv, wo = bf(T, d), bf(dff, d), bf(d, dff), bf(3 * d, d), bf(d, d)
    hid, res = bf(T, dff), torch.randn(T, d, device=dev)
    b1, b2 = torch.randn(dff, device=dev), torch.randn(d, device=dev)
    Cb_h, Cf_d, Cb_q = torch.empty(T, dff, dtype=torch.int16, device=dev), torch.empty(T, d, device=dev), torch.empty(T, 3 * d, dtype=torch.int16, device=dev)
    rec("gemm fwd FFN w_1 (T x 2048 x 128, relu+dropout -> bf16)",
        timeit(lambda: E.gemm(x, w1, T, dff, d, Cb=Cb_h, bias=b1, relu=True, drop_p=0.1, seed=1, drop_stream=2)),
        2 * T * dff * d, T * d * 2 + T * dff * 2)
    rec("gemm fwd FFN w_2 (T x 128 x 2048, dropout+residual -> fp32)",
        timeit(lambda: E.gemm(hid, w2, T, d, dff, Cf=Cf_d, bias=b2, residual=res, drop_p=0.1, seed=1, drop_stream=3)),
        2 * T * dff * d, T * dff * 2 + T * d * 8)
    rec("gemm fwd QKV (T x 384 x 128 -> bf16)", timeit(lambda: E.gemm(x, wqkv, T, 3 * d, d, Cb=Cb_q, bias=torch.zeros(3 * d, device=dev))),
        2 * T * 3 * d * d, T * d * 2 + T * 3 * d * 2)
    rec("gemm fwd O (T x 128 x 128, dropout+residual -> fp32)",
        timeit(lambda: E.gemm(x, wo, T, d, d, Cf=Cf_d, bias=b2, residual=res, drop_p=0.1, seed=1, drop_stream=1)), 2 * T * d * d, T * d * 10)
    dy = bf(T, d)
    rec("gemm dx FFN: dz1 = dy W2 gated (T x 2048 x 128 -> bf16)",
        timeit(lambda: E.gemm(dy, w2, T, dff, d, b_kmajor=True, Cb=Cb_h, gate=hid, gate_scale=1.1)), 2 * T * dff * d, T * d * 2 + T * dff * 4)
    rec("gemm dx FFN: dn2 = dz1 W1 (T x 128 x 2048 -> fp32)",
        timeit(lambda: E.gemm(hid, w1, T, d, dff, b_kmajor=True, Cf=Cf_d)), 2 * T * dff * d, T * dff * 2 + T * d * 4)
    rec("dW2 = dy^T hid (128 x 2048 over T) + reduce", timeit(lambda: E._weight_grad(dy, hid, T, d, dff)), 2 * T * dff * d, T * dff * 2 + T * d * 2)
    rec("dW1 = dz1^T n2 (2048 x 128 over T) + reduce", timeit(lambda: E._weight_grad(hid, x, T, dff, d)), 2 * T * dff * d, T * dff * 2 + T * d * 2)
    rec("dWqkv (384 x 128 over T) + reduce", timeit(lambda: E._weight_grad(Cb_q, x, T, 3 * d, d)), 2 * T * 3 * d * d, T * 8 * d)
    rec("dWo (128 x 128 over T) + reduce", timeit(lambda: E._weight_grad(dy, x, T, d, d)), 2 * T * d * d, T * 4 * d)
    x1 = torch.randn(T, d, device=dev)
    for p in (0.0, 0.1):
        rec(f"fused FFN fwd p={p}", timeit(lambda: E.ffn_fwd(x, w1, b1, w2, b2, x1, T, d, dff, p, 1, 2, 3)), 4 * T * dff * d, T * d * 10)
        rec(f"fused FFN bwd (x + w + reduces) p={p}", timeit(lambda: E.ffn_bwd(x, w1, b1, w2, dy, T, d, dff, p, 1, 2)), 14 * T * dff * d, T * d * 8)
    qkv, ctx, dqkv = bf(T, 3 * d), torch.empty(T, d, dtype=torch.int16, device=dev), torch.empty(T, 3 * d, dtype=torch.int16, device=dev)
    mask = torch.zeros(B, S, dtype=torch.uint8, device=dev)
    att_f = 2 * 2 * B * h * S * S * dk
    for p in (0.0, 0.1):
        rec(f"attention fwd p={p}", timeit(lambda: check(lib().ltr_enc_attention_fwd(_ptr(qkv), _ptr(mask), B, S, h, dk, p, 1, 0, _ptr(ctx),
                                                                                       _stream()), "f")), att_f, T * d * 8)
        rec(f"attention bwd p={p}", timeit(lambda: check(lib().ltr_enc_attention_bwd(_ptr(qkv), _ptr(ctx), _ptr(dy), _ptr(mask), B, S, h, dk, p, 1, 0,
                                                                                       _ptr(dqkv), _stream()), "b")), att_f * 3.5, T * d * 16)
    lse = torch.empty(B * h, S, device=dev)
    for p in (0.0, 0.1):
        rec(f"attention fwd +lse p={p}", timeit(lambda: check(lib().ltr_enc_attention_fwd_lse(_ptr(qkv), _ptr(mask), B, S, h, dk, p, 1, 0, _ptr(ctx),
                                                                                                _ptr(lse), _stream()), "f")), att_f, T * d * 8)
        rec(f"attention bwd key-major (lse) p={p}", timeit(lambda: check(lib().ltr_enc_attention_bwd_lse(
            _ptr(qkv), _ptr(ctx), _ptr(dy), _ptr(lse), _ptr(mask), B, S, h, dk, p, 1, 0, _ptr(dqkv), _stream()), "b")), att_f * 2.5, T * d * 16)
    # the reference's default geometry: d_model 136 = 8 heads x 17
    d17, dk17 = 136, 17
    qkv17, ctx17, dq17, dy17 = bf(T, 3 * d17), torch.empty(T, d17, dtype=torch.int16, device=dev), torch.empty(T, 3 * d17, dtype=torch.int16, device=dev), bf(T, d17)
    rec("attention fwd +lse dk=17 p=0.1", timeit(lambda: check(lib().ltr_enc_attention_fwd_lse(_ptr(qkv17), _ptr(mask), B, S, h, dk17, 0.1, 1, 0, _ptr(ctx17),
                                                                                                  _ptr(lse), _stream()), "f")), 2 * 2 * B * h * S * S * dk17, T * d17 * 8)
    rec("attention bwd key-major (lse) dk=17 p=0.1", timeit(lambda: check(lib().ltr_enc_attention_bwd_lse(
        _ptr(qkv17), _ptr(ctx17), _ptr(dy17), _ptr(lse), _ptr(mask), B, S, h, dk17, 0.1, 1, 0, _ptr(dq17), _stream()), "b")), 5 * 2 * B * h * S * S * dk17, T * d17 * 16)
    rec("attention bwd two-phase dk=17 p=0.1", timeit(lambda: check(lib().ltr_enc_attention_bwd(
        _ptr(qkv17), _ptr(ctx17), _ptr(dy17), _ptr(mask), B, S, h, dk17, 0.1, 1, 0, _ptr(dq17), _stream()), "b")), 7 * 2 * B * h * S * S * dk17, T * d17 * 16)
    xf, a_, b_ = torch.randn(T, d, device=dev), torch.ones(d, device=dev), torch.zeros(d, device=dev)
    rec("layernorm fwd (-> bf16)", timeit(lambda: E.layernorm_fwd(xf, a_, b_, T, d, 1e-6, 0)), 0, T * d * 6)
    dxa = torch.zeros(T, d, device=dev)
    rec("layernorm bwd (+ partial reduce)", timeit(lambda: E.layernorm_bwd(xf, a_, res, T, d, 1e-6, 0, dxa)), 0, T * d * 16)
    rec("colsum bf16 [T, 2048] (+ reduce)", timeit(lambda: E._colsum(hid, T, dff)), 0, T * dff * 2)
    rec("colsum bf16 [T, 384] (+ reduce)", timeit(lambda: E._colsum(qkv, T, 3 * d)), 0, T * 3 * d * 2)
    rec("drop_cast_colsum [T, 128] (+ reduce)", timeit(lambda: E._drop_cast_colsum(xf, T, d, 0.1, 1, 1)), 0, T * d * 6)


if __name__ == "__main__":
    main()
