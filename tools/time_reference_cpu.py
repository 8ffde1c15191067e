#!/usr/bin/env python3
"""Time the IMPORTED reference (build container only: /root/reference, CPU, torch intra-op threads) on the shapes of
BASELINE.md section 3: one warm-up + median of >= 20 steps of  zero_grad -> net(x, None, None).squeeze(-1) -> loss ->
backward  (loss-only rows: the loss on random scores).  Prints a markdown table; the numbers in BASELINE.md come from
this script.  The reference source never leaves this container.    Usage: python tools/time_reference_cpu.py [iters]"""
import os
import statistics
import sys
import time

import torch

REF = os.environ.get("LTR_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
from architeture.doubleLayer import DoubleLayerNet      # noqa: E402  (reference)
from architeture.tripleLayer import TripleLayerNet      # noqa: E402
from losses.approxNDCG import approxNDCGLoss            # noqa: E402
from losses.lambdaL import lambdaLoss                   # noqa: E402
from losses.listnet import listnetLoss                  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(2020)


def median_ms(fn, iters):
    fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts), min(ts), max(ts)


def loss_only(name, B, S):
    s = torch.randn(B, S, requires_grad=True)
    y = torch.randint(0, 5, (B, S)).float()
    fn = {"approxNDCG": lambda: approxNDCGLoss(s, y), "ListNet": lambda: listnetLoss(y, s),
          "lambdaLoss ndcgLoss2PP": lambda: lambdaLoss(s, y, weighing_scheme="ndcgLoss2PP_scheme")}[name]

    def step():
        s.grad = None
        fn().backward()
    return step


def net_step(net, loss, B, S):
    x = torch.randn(B, S, 136)
    y = torch.randint(0, 5, (B, S)).float()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)

    def step():
        opt.zero_grad()
        out = net(x, None, None).squeeze(-1)
        l = approxNDCGLoss(out, y) if loss == "approxNDCG" else listnetLoss(y, out)
        l.backward()
        opt.step()
    return step


rows = []
for name in ("approxNDCG", "ListNet", "lambdaLoss ndcgLoss2PP"):
    for B, S in ((200, 32), (200, 128), (64, 512)):
        it = ITERS if S < 512 or name == "ListNet" else max(ITERS // 2, 5)
        med, lo, hi = median_ms(loss_only(name, B, S), it)
        rows.append((f"{name}, loss only", B, S, med, lo, hi, it))
for label, net, loss, B, S in (("DoubleLayerNet (136-136-136-1, dropout on) + approxNDCG + Adam", DoubleLayerNet(136), "approxNDCG", 200, 128),
                               ("DoubleLayerNet + ListNet + Adam (BASELINE config 1 shape)", DoubleLayerNet(136), "listnet", 200, 32),
                               ("TripleLayerNet + approxNDCG + Adam", TripleLayerNet(136), "approxNDCG", 200, 128)):
    net.train()
    med, lo, hi = median_ms(net_step(net, loss, B, S), ITERS)
    rows.append((label, B, S, med, lo, hi, ITERS))
# BASELINE config 5: the reference's make_model network (FC 136->128, 6 encoder blocks, 8 heads, d_ff 2048, dropout 0.1)
if "--no-config5" not in sys.argv:
    import attr
    from architeture.multiLayer import make_model           # noqa: E402  (reference)

    @attr.s(auto_attribs=True)
    class _Tr:
        N: int
        d_ff: int
        h: int
        dropout: float
        positional_encoding: object = None

    enc = make_model(fc_model=dict(sizes=[128], input_norm=False, activation=None, dropout=0.0), transformer=_Tr(6, 2048, 8, 0.1),
                     post_model=dict(d_output=1, output_activation=None), n_features=136)
    enc.train()
    B5, S5 = 8, 256
    x5, y5, m5 = torch.randn(B5, S5, 136), torch.randint(0, 5, (B5, S5)).float(), torch.zeros(B5, S5, dtype=torch.bool)
    opt5 = torch.optim.Adam(enc.parameters(), lr=1e-4)

    def step5():
        opt5.zero_grad()
        approxNDCGLoss(enc(x5, m5, None), y5).backward()
        opt5.step()
    it5 = max(ITERS // 4, 5)
    med, lo, hi = median_ms(step5, it5)
    rows.append(("make_model (FC 136-128, 6 blocks, 8 heads, d_ff 2048, dropout 0.1) + approxNDCG + Adam (BASELINE config 5, fp32)", B5, S5, med, lo, hi, it5))
print(f"threads={torch.get_num_threads()} nproc={os.cpu_count()} torch={torch.__version__} iters>={min(r[6] for r in rows)}")
print("| workload | B | S | median ms/step | min..max ms | slates/s (median) |")
print("|---|---|---|---|---|---|")
for label, B, S, med, lo, hi, it in rows:
    print(f"| {label} | {B} | {S} | {med:.2f} | {lo:.2f}..{hi:.2f} | {B / med * 1e3:,.0f} |")
