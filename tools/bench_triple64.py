#!/usr/bin/env python3
"""TripleLayerNet on the 64-feature collection (TD2003 shape): fused approxNDCG step, folded (default) vs layer by layer (LTR_TRIPLE_FOLD=0)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"))
import torch
from architeture.tripleLayer import TripleLayerNet
from ltr_mi355x.scorer import FusedRanker
dev = "cuda:0"
B, S, F = 25_000, 128, 64
X = torch.randn(B, S, F, device=dev)
y = torch.randint(0, 5, (B, S), device=dev).float()
for fold in ("1", "0"):
    os.environ["LTR_TRIPLE_FOLD"] = fold
    net = TripleLayerNet(F).to(dev)
    r = FusedRanker(net, loss="approxNDCG")
    for _ in range(3):
        r.step(X, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        r.step(X, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(json.dumps({"workload": f"approxNDCG + TripleLayerNet(64) fused step, {B} x {S} x {F}", "folded": r.fold is not None, "ms_per_step": round(dt * 1e3, 4),
                      "slates_per_s": round(B / dt), "hbm_frac_of_8TBps": round(B * S * (F + 1) * 4 / dt / 8e12, 4)}), flush=True)
