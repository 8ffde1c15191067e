#!/usr/bin/env python3
"""Where does the bench-batch gradient error come from (fc2.weight 1.76e-3 of its own max against the fp64 oracle, the reference's
own fp32 path 8.0e-4; tests/test_fused_gaps_gpu.py::test_bench_batch_train_mode_vs_fp64_oracle)?  Same inputs and dropout masks;
the scorer's backward is driven once by the loss kernel's d loss / d scores and once by the fp64 oracle's.
Test infrastructure: imports oracle/.  Output: JSON lines."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ltr_oracle as O  # noqa: E402
from architeture.doubleLayer import DoubleLayerNet  # noqa: E402
from losses.approxNDCG import approxNDCGLoss  # noqa: E402
from ltr_mi355x import scorer  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(5)
net = DoubleLayerNet(136)
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
net = net.to(dev).train()
B, S = 2048, 128
gen = torch.Generator().manual_seed(77)
x = torch.randn(B, S, 136, generator=gen)
y = torch.multinomial(torch.tensor([0.52, 0.32, 0.13, 0.02, 0.01]), B * S, replacement=True, generator=gen).view(B, S).float()
seed = 0x0123456789ABCDEF
k1 = scorer.dropout_keep_mask(seed, 0, B * S, 136, dev).view(B, S, 136)
k2 = scorer.dropout_keep_mask(seed, 1, B * S, 136, dev).view(B, S, 136)


def oracle(dtype):
    p = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    s = O.double_layer_forward(x.to(dtype), p, k1.cpu().to(dtype), k2.cpu().to(dtype), 0.5).squeeze(-1)
    s.retain_grad()
    l = O.approx_ndcg(s, y.to(dtype))
    l.backward()
    return {k: v.grad.double() for k, v in p.items()}, s.detach().double(), s.grad.double()


g64, s64, ds64 = oracle(torch.float64)
g32, s32, ds32 = oracle(torch.float32)


def run(ds_from_oracle):
    net.zero_grad()
    s = net(x.to(dev), None, None, keep1=k1, keep2=k2).squeeze(-1)
    if ds_from_oracle:
        s.backward(ds64.float().to(dev))
        ds = None
    else:
        s.retain_grad()
        approxNDCGLoss(s, y.to(dev)).backward()
        ds = s.grad.detach().cpu().double()
    return {k: v.grad.detach().cpu().double() for k, v in net.named_parameters()}, s.detach().cpu().double(), ds


def err(a, b):
    return float((a - b).abs().max() / b.abs().max())


ga, sa, dsa = run(False)
gb, _, _ = run(True)
print(json.dumps({"scores: kernel vs fp64 oracle": err(sa, s64), "scores: fp32 oracle vs fp64": err(s32, s64),
                  "d loss / d scores: loss kernel vs fp64 oracle": err(dsa, ds64), "d loss / d scores: fp32 oracle vs fp64": err(ds32, ds64)}))
for k in g64:
    print(json.dumps({"tensor": k, "kernel backward driven by the LOSS KERNEL's ds": err(ga[k], g64[k]),
                      "kernel backward driven by the fp64 ORACLE's ds": err(gb[k], g64[k]), "fp32 oracle (the reference's own arithmetic)": err(g32[k], g64[k])}))

# ---- the structure of the deviation: ReLU gates that the fp32 forward decides differently from the fp64 forward.
# One flipped gate (document t, unit n) adds or removes that document's whole contribution ds_t w3_n keep2 2 * h1_t to row n of
# dW2 (and, through dh1, to dW1): a rank-one term the size of ONE document's contribution, where the full gradient is a sum over
# 262 144 documents that largely cancels.
with torch.no_grad():
    W1, b1, W2, b2 = (sd[k].double() for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"))
    xf = x.double().view(-1, 136)
    K1, K2 = k1.cpu().double().view(-1, 136), k2.cpu().double().view(-1, 136)
    z1 = xf @ W1.T + b1
    h1 = torch.relu(z1) * K1 * 2
    z2 = h1 @ W2.T + b2
    z1f = xf.float() @ W1.float().T + b1.float()
    h1f = torch.relu(z1f) * K1.float() * 2
    z2f = h1f @ W2.float().T + b2.float()
    flip1 = ((z1 > 0) != (z1f.double() > 0)) & (K1 > 0)
    flip2 = ((z2 > 0) != (z2f.double() > 0)) & (K2 > 0)
    d = (ga["fc2.weight"] - g64["fc2.weight"])
    u, sv, vt = torch.linalg.svd(d)
    e = (sv ** 2) / (sv ** 2).sum()
    n_top = int(torch.argmax(u[:, 0].abs()))
    cos = (h1 @ vt[0]) / (h1.norm(dim=1) * vt[0].norm()).clamp_min(1e-300)
    t_top = int(torch.argmax(cos.abs()))
    print(json.dumps({"ReLU gates decided differently by an fp32 forward (torch CPU) and the fp64 forward, kept units only": {
                          "layer 1": int(flip1.sum()), "layer 2": int(flip2.sum()), "of": int((K1 > 0).sum())},
                      "largest |z| among the flipped pre-activations": [float(z1[flip1].abs().max()) if flip1.any() else 0.0,
                                                                          float(z2[flip2].abs().max()) if flip2.any() else 0.0],
                      "fc2.weight deviation (kernel - fp64): share of its squared Frobenius norm in the top 1 / 2 / 4 singular directions":
                          [float(e[0]), float(e[:2].sum()), float(e[:4].sum())],
                      "top direction": {"unit (row of W2)": n_top, "document whose h1 it is parallel to": t_top, "cosine": float(cos[t_top].abs()),
                                        "that document's fp64 pre-activation z2[unit]": float(z2[t_top, n_top]),
                                        "typical |z2|": float(z2.abs().median())}}))

# ---- the deviation modulo gate flips: least-squares removal of the rank-one directions e_n (x) h1_t of every BORDERLINE kept
# layer-2 pre-activation (|z2| < 1e-6 in fp64) from fc2.weight's deviation; what is left is the arithmetic error proper.
with torch.no_grad():
    border = ((z2.abs() < 1e-6) & (K2 > 0)).nonzero()
    cols = []
    for t, n in border.tolist():
        m = torch.zeros(136, 136, dtype=torch.float64)
        m[n] = h1[t]
        cols.append(m.flatten())
    A = torch.stack(cols, 1)
    coef = torch.linalg.lstsq(A, d.flatten()[:, None]).solution
    res = d.flatten() - (A @ coef)[:, 0]
    top = float(g64["fc2.weight"].abs().max())
    print(json.dumps({"borderline kept layer-2 pre-activations (|z2| < 1e-6)": len(cols),
                      "fc2.weight deviation / max|tensor|": float(d.abs().max()) / top,
                      "... after removing the borderline documents' rank-one gate terms": float(res.abs().max()) / top}))
