#!/usr/bin/env python3
"""Phase stamps of the weights-stationary DoubleLayerNet kernel (csrc/ltr_wst.h; a -DLTR_STAMPS build via LTR_LIB):
cycles per phase of one mid-run tile, median over workgroups and the 4 waves."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"))
from architeture.doubleLayer import DoubleLayerNet  # noqa: E402
from ltr_mi355x import lib, scorer  # noqa: E402
from ltr_mi355x.functional import _ptr, _stream, check  # noqa: E402

NAMES = ["X wait + labels + barrier A", "fc1 (+ half tile) + act + h1 -> image", "barrier B + fc2 (+ half tile) + act", "score partials + barrier D",
         "loss (+ barrier E)", "dz2 (+dw3) + side image", "dW2 rows + barrier F1 + half-tile rows", "barrier F2 + dz2 -> image + barrier F3 + dh1 + dz1",
         "barrier G + next-X DMA issue + dW1"]
dev = torch.device("cuda:0")
B, S = 25_000, 128
X = torch.randn(B, S, 136, device=dev)
y = torch.randint(0, 5, (B, S), device=dev).float()
h = lib()
for train in (0, 1):
    net = DoubleLayerNet(136).to(dev)
    info = scorer.NetInfo.get(net._ltr_net)
    packed = scorer.pack_params(net._ltr_net, net._ltr_params())
    grid = scorer.cu_count(dev)
    part = torch.empty(grid * info.partial_floats, device=dev)
    sl = torch.empty(B, device=dev)
    stamps = torch.zeros((grid, 8, 16), dtype=torch.int64, device=dev)
    assert h.ltr_debug_set_stamps(stamps.data_ptr(), 40) == 1, "library not built with -DLTR_STAMPS"
    for _ in range(3):
        check(h.ltr_fused_step(info.net, 0, _ptr(X), _ptr(y), B, S, _ptr(packed), train, 7, None, None, 1.0, 1e-10, -1.0, 0, 1.0 / B, _ptr(sl),
                               _ptr(part), grid, _stream()), "fused")
    torch.cuda.synchronize()
    h.ltr_debug_set_stamps(None, 0)
    t = stamps[:, :4].cpu().double()
    d = (t[:, :, 1:10] - t[:, :, 0:9]).reshape(-1, 9).median(0).values
    tot = float((t[:, :, 9] - t[:, :, 0]).reshape(-1).median())
    lo = t[:, :, [4, 10, 12, 13, 14, 5]]
    ld = (lo[:, :, 1:] - lo[:, :, :-1]).reshape(-1, 5).median(0).values
    loss_names = ["prologue", "sweep 1 + row epilogue", "barrier + sums", "sweep 2", "exit barrier"]
    extra = float((t[:, :, 15] - t[:, :, 8]).reshape(-1).median())
    print(json.dumps({"dropout": bool(train), "barrier G + dW1 rows (stamp 8 -> 15)": extra, "total_cycles_per_tile": tot, "mfma_bound_cycles_per_simd": 3224 * 32,
                      "phases": {n: round(float(v)) for n, v in zip(NAMES, d)}, "loss_detail": {n: round(float(v)) for n, v in zip(loss_names, ld)}}), flush=True)
