#!/usr/bin/env python3
"""Micro-benchmark of the standalone loss kernels (loss-only fwd+bwd, scores/labels resident in HBM).
These kernels are VALU/transcendental-bound (S^2 sigmoids vs 12*S bytes); reported separately from the
headline fused scorer+loss path (bench.py)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"))
from ltr_mi355x import lib  # noqa: E402
from ltr_mi355x.functional import _ptr, _stream, check  # noqa: E402


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
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    h = lib()
    out = []
    cases = (("approxndcg", 100_000, 32), ("approxndcg", 100_000, 128), ("approxndcg", 16_384, 512),
             ("lambda2pp", 100_000, 32), ("lambda2pp", 100_000, 128), ("lambda2pp", 8_192, 512),
             ("listnet", 100_000, 32), ("listnet", 100_000, 128), ("listnet", 16_384, 512))
    if "--only" in sys.argv:           # e.g. --only lambda512 (BASELINE config 3 shape, for profiling one kernel)
        want = sys.argv[sys.argv.index("--only") + 1]
        cases = tuple(c for c in cases if f"{c[0].replace('2pp', '')}{c[2]}" == want)
    for name, B, S in cases:
        s = torch.randn(B, S, device=dev)
        y = torch.randint(0, 5, (B, S), device=dev).float()
        sl = torch.empty(B, device=dev)
        cnt = torch.empty(B, device=dev)
        ds = torch.empty_like(s)
        if name == "approxndcg":
            fn = lambda: check(h.ltr_approxndcg_fwd_bwd(_ptr(s), _ptr(y), B, S, 1.0, 1e-10, -1.0, 1.0 / B, _ptr(sl),
                                                        _ptr(ds), _stream()), name)
        elif name == "lambda2pp":
            fn = lambda: check(h.ltr_lambda_fwd_bwd(_ptr(s), _ptr(y), B, S, 4, 0, 1.0, 10.0, 1e-10, -1.0, 0, 1.0,
                                                    _ptr(sl), _ptr(cnt), _ptr(ds), _stream()), name)
        else:
            fn = lambda: check(h.ltr_listnet_fwd_bwd(_ptr(y), _ptr(s), B, S, 0, 1.0, _ptr(sl), _ptr(ds), _stream()), name)
        ms = timeit(fn)
        rec = dict(kernel=name, B=B, S=S, ms=round(ms, 4), slates_per_s=round(B / ms * 1e3),
                   pairs_per_s=round(B * S * S / ms * 1e3), hbm_GBps=round(B * S * 12 / ms / 1e6, 1))
        out.append(rec)
        print(json.dumps(rec), flush=True)
    return out


if __name__ == "__main__":
    main()
