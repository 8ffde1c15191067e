#!/usr/bin/env python3
"""CPU numerics study for the split-precision question (SURVEY.md section 7 hard part 1):
how far do the scorer GEMMs drift from the fp64 oracle when every fp32 operand is split into low-precision
pieces that the bf16 / f16 matrix cores can multiply (fp32 accumulate), compared with plain fp32?

Schemes (pieces per operand, products kept):
  fp32      : plain fp32 matmul (what v_mfma_f32_16x16x4_f32 computes)                      rate 1x
  bf16x3/6  : a = a1+a2+a3 (8 bits each), products with i+j <= 2  (error ~2^-24)            rate 16/6 = 2.7x
  bf16x2/3  : a = a1+a2, products with i+j <= 1                  (error ~2^-16)            rate 16/3 = 5.3x
  f16x2/3   : a = hi+lo in fp16 after an exact power-of-two scale to 2^14, i+j <= 1 (2^-22)  rate 5.3x
  f16x2/4   : same, all four products                                                        rate 4x

The emulation multiplies the exactly-representable pieces in fp32 (torch CPU matmul = fp32 accumulate), so it
carries the same accumulation noise as the hardware path.  The whole DoubleLayerNet forward + approxNDCG +
backward runs through the emulated GEMMs (fc1, fc2, dh1, dW2, dW1; fc3 is a VALU dot product in the kernel and
stays fp32).  Metric: max|delta| / max|ref| against the fp64 oracle (the repo's parity metric), printed next to
plain fp32's own deviation.  Usage: python tools/split_precision_study.py [--out profiles/r02_split_precision_study.json]
"""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ltr_oracle as O  # noqa: E402  (study tooling: checker only)


def pieces(a, scheme):
    """Exactly representable fp32 pieces of `a` and the power-of-two scale they were taken at."""
    if scheme == "fp32":
        return [a], 1.0
    if scheme.startswith("bf16"):
        n = 3 if scheme.startswith("bf16x3") else 2
        out, r = [], a.clone()
        for _ in range(n):
            p = r.to(torch.bfloat16).to(torch.float32)
            out.append(p)
            r = r - p
        return out, 1.0
    # f16x2: scale so that max|a| sits at 2^14 (exact), then hi/lo in fp16
    m = float(a.abs().max())
    s = 2.0 ** (14 - math.ceil(math.log2(m))) if m > 0 else 1.0
    x = a * s
    hi = x.to(torch.float16).to(torch.float32)
    lo = (x - hi).to(torch.float16).to(torch.float32)
    return [hi, lo], s


def kept(scheme, i, j):
    if scheme == "fp32":
        return True
    if scheme == "bf16x3/6":
        return i + j <= 2
    if scheme in ("bf16x2/3", "f16x2/3"):
        return i + j <= 1
    return True        # f16x2/4


def split_mm(a, b, scheme):
    """a [M,K] @ b [K,N] through the split scheme, fp32 accumulate."""
    pa, sa = pieces(a, scheme)
    pb, sb = pieces(b, scheme)
    acc = None
    # small products first (as a kernel would order them so the large term lands last)
    order = sorted(((i, j) for i in range(len(pa)) for j in range(len(pb)) if kept(scheme, i, j)), key=lambda t: -(t[0] + t[1]))
    for i, j in order:
        t = pa[i] @ pb[j]
        acc = t if acc is None else acc + t
    return acc / (sa * sb)


class SplitLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, scheme):
        ctx.save_for_backward(x, w)
        ctx.scheme = scheme
        return split_mm(x, w.t().contiguous(), scheme.split("+")[0]) + b

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        sch = ctx.scheme
        dx = split_mm(g, w, sch.split("+")[0])
        dw = split_mm(g.t().contiguous(), x, sch.split("+")[-1])
        return dx, dw, g.sum(0), None


def run(scheme, sd, X, y, k1, k2, loss_name):
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    B, S, F = X.shape
    x = X.reshape(B * S, F)
    h1 = torch.relu(SplitLinear.apply(x, p["fc1.weight"], p["fc1.bias"], scheme)) * (k1.reshape(B * S, -1) * 2.0)
    h2 = torch.relu(SplitLinear.apply(h1, p["fc2.weight"], p["fc2.bias"], scheme)) * (k2.reshape(B * S, -1) * 2.0)
    s = (h2 @ p["fc3.weight"].t() + p["fc3.bias"]).view(B, S)
    loss = O.approx_ndcg(s, y) if loss_name == "approxNDCG" else O.lambda_loss(s, y, weighing_scheme="ndcgLoss2PP_scheme")
    loss.backward()
    return float(loss), {k: v.grad.double() for k, v in p.items()}, s.detach().double()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_split_precision_study.json"))
    a = ap.parse_args()
    torch.manual_seed(2020)
    lin = [torch.nn.Linear(136, 136), torch.nn.Linear(136, 136), torch.nn.Linear(136, 1)]
    sd = {}
    for n, l in zip(("fc1", "fc2", "fc3"), lin):
        sd[f"{n}.weight"], sd[f"{n}.bias"] = l.weight.detach().clone(), l.bias.detach().clone()
    cases = [("approxNDCG", 37, 128, 1.0), ("approxNDCG", 512, 128, 1.0), ("lambdaLoss", 37, 128, 1.0),
             ("approxNDCG", 64, 128, 1000.0)]        # last: un-normalised features (x 1000) to stress the scaling
    rows = []
    for loss_name, B, S, xs in cases:
        gen = torch.Generator().manual_seed(B + S)
        X = torch.randn(B, S, 136, generator=gen) * xs
        y = torch.randint(0, 5, (B, S), generator=gen).float()
        k1 = (torch.rand(B, S, 136, generator=gen) < 0.5).float()
        k2 = (torch.rand(B, S, 136, generator=gen) < 0.5).float()
        # fp64 oracle
        p64 = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
        s64 = O.double_layer_forward(X.double(), p64, k1.double(), k2.double()).squeeze(-1)
        l64 = (O.approx_ndcg(s64, y.double()) if loss_name == "approxNDCG"
               else O.lambda_loss(s64, y.double(), weighing_scheme="ndcgLoss2PP_scheme"))
        l64.backward()
        ref = {k: v.grad for k, v in p64.items()}
        top = max(float(v.abs().max()) for v in ref.values())
        for scheme in ("fp32", "bf16x3/6", "f16x2/4", "f16x2/3", "bf16x2/3", "bf16x3/6+bf16x2/3", "bf16x3/6+fp32", "fp32+bf16x3/6"):
            l, g, s = run(scheme, sd, X, y, k1, k2, loss_name)
            row = {"case": f"{loss_name} B={B} S={S} x-scale={xs:g}", "scheme": scheme,
                   "loss_rel": abs(l - float(l64)) / abs(float(l64)),
                   "scores_rel": float((s - s64.detach()).abs().max() / s64.detach().abs().max()),
                   "grad_rel_whole": max(float((g[k] - ref[k]).abs().max()) for k in ref) / top,
                   "grad_rel_per_tensor_worst": max(float((g[k] - ref[k]).abs().max() / ref[k].abs().max())
                                                    for k in ref if float(ref[k].abs().max()) >= 1e-3 * top)}
            rows.append(row)
            print(json.dumps(row), flush=True)
    with open(a.out, "w") as f:
        json.dump({"what": __doc__.split("\n\n")[0], "metric": "max|delta|/max|ref| vs fp64 oracle", "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
