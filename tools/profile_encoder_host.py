#!/usr/bin/env python3
"""Host-side (Python) cost of one config-5 training step at a small batch, where the step is launch-bound:
cProfile of 20 steps, top functions by cumulative time."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"))
import torch  # noqa: E402
from architeture.multiLayer import make_model  # noqa: E402
from losses.approxNDCG import approxNDCGLoss  # noqa: E402

dev = "cuda:0"
net = make_model(dict(sizes=[128], input_norm=False, activation=None, dropout=0.0),
                 dict(N=6, d_ff=2048, h=8, dropout=0.1, positional_encoding=None), dict(d_output=1), 136).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 100
x, y, m = torch.randn(B, S, 136, device=dev), torch.randint(0, 5, (B, S), device=dev).float(), torch.zeros(B, S, dtype=torch.bool, device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    approxNDCGLoss(net(x, m, None), y).backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print(f"B={B} S={S}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
