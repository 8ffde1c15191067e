#!/usr/bin/env python3
"""Where the slate-pipeline kernel's time goes: time MODE_FWD / MODE_BWD / MODE_FUSED launches.
With LTR_DEBUG_SKIP set in the environment phases are skipped (results wrong, timing only)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"))
from architeture.doubleLayer import DoubleLayerNet  # noqa: E402
from architeture.tripleLayer import TripleLayerNet  # noqa: E402
from ltr_mi355x import lib, scorer  # noqa: E402
from ltr_mi355x.functional import _ptr, _stream, check  # noqa: E402


def timeit(fn, iters=5, warm=2):
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
    B, S = 25_000, 128
    X = torch.randn(B, S, 136, device=dev)
    y = torch.randint(0, 5, (B, S), device=dev).float()
    h = lib()
    skip = os.environ.get("LTR_DEBUG_SKIP", "0")
    for name, cls in (("double", DoubleLayerNet), ("triple", TripleLayerNet)):
        net = cls(136).to(dev).eval()
        info = scorer.NetInfo.get(net._ltr_net)
        packed = scorer.pack_params(net._ltr_net, net._ltr_params())
        grid = scorer.cu_count(dev)
        n = B * S
        x2 = X.view(n, 136)
        sc = torch.empty(n, device=dev)
        part = torch.empty(grid * info.partial_floats, device=dev)
        sl = torch.empty(B, device=dev)
        ds = torch.randn(n, device=dev) * 1e-3
        fwd = lambda: check(h.ltr_mlp_forward(net._ltr_net, _ptr(x2), n, _ptr(packed), 0, 0, None, None, _ptr(sc), grid, _stream()), "f")
        bwd = lambda: check(h.ltr_mlp_backward(net._ltr_net, _ptr(x2), n, _ptr(packed), 0, 0, None, None, _ptr(ds), _ptr(part), grid, _stream()), "b")
        fus = lambda: check(h.ltr_fused_step(net._ltr_net, 0, _ptr(x2), _ptr(y), B, S, _ptr(packed), 0, 0, None, None, 1.0, 1e-10, -1.0, 0, 1.0 / B, _ptr(sl), _ptr(part), grid, _stream()), "u")
        rec = dict(net=name, skip=int(skip), fwd_ms=round(timeit(fwd), 3), bwd_ms=round(timeit(bwd), 3), fused_ms=round(timeit(fus), 3),
                   us_per_tile_fused=round(timeit(fus) * 1e3 / (B / grid), 2))
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
