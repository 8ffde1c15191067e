#!/usr/bin/env python3
"""BASELINE config 5 micro-bench: transformer scorer + approxNDCG, slate 256, bf16 operands, one MI355X.
Prints one JSON line per configuration: slates/s for forward + loss + backward (+ Adam), algorithmic TFLOP/s and the
fraction of the dense bf16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"))

import torch  # noqa: E402


def flops_per_slate(S, F, fc, d, h, dff, N):
    """Multiply-adds x 2 of forward + backward (dX and dW of every Linear; attention QK^T / PV and their 5 backward GEMMs)."""
    fwd = 0
    n_in = F
    for n_out in fc:
        fwd += 2 * S * n_in * n_out
        n_in = n_out
    per_layer = 2 * S * d * 3 * d + 2 * S * d * d + 2 * 2 * S * d * dff      # QKV, O, FFN
    attn = 2 * 2 * S * S * d                                                   # QK^T + PV over all heads
    fwd += N * (per_layer + attn) + 2 * S * d
    return 3 * fwd + N * attn * 0.5          # backward = 2x forward for the Linears, 2.5x for attention


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--slate", type=int, default=256)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--dff", type=int, default=2048)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--fc", type=str, default="128")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--graph", action="store_true",
                    help="the whole step (seed advance + forward + loss + backward + Adam) as ONE hipGraph (graphs.GraphedTrainStep)")
    ap.add_argument("--adam", choices=["foreach", "fused"], default="fused",
                    help="torch.optim.Adam implementation: fused = one multi-tensor kernel; foreach (torch's default) is ~25 launches eager and, "
                         "with capturable=True, ~230 (202 of them 0-dim divisions) -- rounds 2-3 measured foreach")
    a = ap.parse_args()
    from architeture.multiLayer import make_model
    from losses.approxNDCG import approxNDCGLoss
    dev = "cuda:0"
    F = 136
    fc = [int(v) for v in a.fc.split(",")] if a.fc else []
    torch.manual_seed(0)
    net = make_model(dict(sizes=list(fc), input_norm=False, activation=None, dropout=0.0) if fc else None,
                     dict(N=a.layers, d_ff=a.dff, h=a.heads, dropout=a.dropout, positional_encoding=None),
                     dict(d_output=1, output_activation=None), F).to(dev)
    net.train()
    d = fc[-1] if fc else F
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=a.graph, fused=(a.adam == "fused"))
    B, S = a.batch, a.slate
    x = torch.randn(B, S, F, device=dev)
    y = torch.randint(0, 5, (B, S), device=dev).float()
    mask = torch.zeros(B, S, dtype=torch.bool, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = approxNDCGLoss(net(x, mask, None), y)
        loss.backward()
        opt.step()
        return loss

    if a.graph:
        from ltr_mi355x.graphs import GraphedTrainStep
        graphed = GraphedTrainStep(net, opt, lambda n, x, m, y: approxNDCGLoss(n(x, m, None), y), (x, mask, y))

        def step():                                  # noqa: F811
            return graphed(x, mask, y)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    fl = flops_per_slate(S, F, fc, d, a.heads, a.dff, a.layers)
    print(json.dumps({"workload": f"approxNDCG + make_model(fc={fc}, N={a.layers}, h={a.heads}, d_ff={a.dff}, dropout={a.dropout}) "
                                  f"train mode, {B} slates x {S} x {F} per step, fwd+loss+bwd+Adam({a.adam}), "
                                  + ("replayed as one hipGraph (device-side dropout epoch)" if a.graph else "eager launches"),
                      "graph": bool(a.graph),
                      "slates_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3), "flops_per_slate": fl,
                      "tflops": round(B * fl / dt / 1e12, 2), "frac_of_bf16_mfma_peak": round(B * fl / dt / 2.5e15, 4),
                      "final_loss": round(float(loss.detach()), 5), "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
