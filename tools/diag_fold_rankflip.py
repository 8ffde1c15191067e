#!/usr/bin/env python3
"""Why the folded TripleLayerNet misses the 1e-5 bar on ONE lambdaLoss case (slate 512, seed 17): does the order of two nearly tied
scores differ from the fp64 oracle's?  (lambdaLoss weights pairs by the RANKS of the predicted scores: lambdaL.py:36-60.)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ltr_oracle as O
from architeture.tripleLayer import TripleLayerNet
dev = "cuda:0"
torch.manual_seed(17)
net = TripleLayerNet(136)
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
net = net.to(dev).eval()
gen = torch.Generator().manual_seed(512)
x = torch.randn(3, 512, 136, generator=gen)
p = {k: v.double() for k, v in sd.items()}
so = O.triple_layer_forward(x.double(), p).squeeze(-1).numpy()
for fold in ("1", "0"):
    os.environ["LTR_TRIPLE_FOLD"] = fold
    s = net(x.to(dev), None, None).squeeze(-1).detach().cpu().numpy().astype(np.float64)
    flips = 0
    for b in range(3):
        flips += int((np.argsort(-s[b], kind="stable") != np.argsort(-so[b], kind="stable")).sum())
    gaps = np.sort(np.abs(np.diff(np.sort(so, axis=1), axis=1)).ravel())[:3]
    print(f"fold={fold}: max|score - fp64| = {np.abs(s - so).max():.3e}, positions whose rank differs from the fp64 order: {flips}; smallest fp64 score gaps: {gaps}")
