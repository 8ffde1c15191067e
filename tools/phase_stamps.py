#!/usr/bin/env python3
"""In-kernel phase timing of the fused slate pipeline (diagnostic build with -DLTR_STAMPS, loaded via LTR_LIB):
per-phase shader cycles of one mid-run tile, median over workgroups and waves.  Shares, not absolute run time."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"))
from architeture.doubleLayer import DoubleLayerNet  # noqa: E402
from architeture.tripleLayer import TripleLayerNet  # noqa: E402
from ltr_mi355x import lib, scorer  # noqa: E402
from ltr_mi355x.extra_nets import TwoLayerNet  # noqa: E402
from ltr_mi355x.functional import _ptr, _stream, check  # noqa: E402

NAMES = ["barrier+X load->LDS", "fc1 (+act)", "fc2 (+act)", "fc3+scores", "loss", "L2 prefetch+dw3+dz2",
         "dW2 (2 chunks)", "dh1 (+mask)", "dW1 (2 chunks)"]
dev = torch.device("cuda:0")
B, S = 25_000, 128
X = torch.randn(B, S, 136, device=dev)
y = torch.randint(0, 5, (B, S), device=dev).float()
h = lib()
for name, cls in (("double", DoubleLayerNet), ("triple", TripleLayerNet)):      # the 136-64-1 net: tools/fcw_stamps.py
    net = cls(136).to(dev).eval()
    info = scorer.NetInfo.get(net._ltr_net)
    packed = scorer.pack_params(net._ltr_net, net._ltr_params())
    grid = scorer.cu_count(dev)
    part = torch.empty(grid * info.partial_floats, device=dev)
    sl = torch.empty(B, device=dev)
    stamps = torch.zeros((grid, 8, 16), dtype=torch.int64, device=dev)
    assert h.ltr_debug_set_stamps(stamps.data_ptr(), 40) == 1, "library not built with -DLTR_STAMPS"
    for _ in range(3):
        check(h.ltr_fused_step(net._ltr_net, 0, _ptr(X), _ptr(y), B, S, _ptr(packed), 0, 0, None, None, 1.0, 1e-10, -1.0, 0,
                               1.0 / B, _ptr(sl), _ptr(part), grid, _stream()), "fused")
    torch.cuda.synchronize()
    h.ltr_debug_set_stamps(None, 0)
    t = stamps.cpu().double()
    d = t[:, :, 1:10] - t[:, :, 0:9]
    med = d.reshape(-1, 9).median(0).values
    tot = float((t[:, :, 9] - t[:, :, 0]).reshape(-1).median())
    lo = t[:, :, [4, 10, 11, 12, 13, 14, 5]]
    ld = (lo[:, :, 1:] - lo[:, :, :-1]).reshape(-1, 6).median(0).values
    loss_names = ["entry barrier+range scan+minmax", "init u/mask + barrier", "rank+pos sweep", "sum2 (2 barriers)",
                  "grad sweep", "store + exit barrier"]
    ep = t[:, :, [1, 10, 11, 12, 13, 14, 15, 2]]
    epd = (ep[:, :, 1:] - ep[:, :, :-1]).reshape(-1, 7).median(0).values
    ep_names = ["fc1: start -> epoch1 dma waited", "epoch1 barrier", "epoch1 dma issue", "epoch1 lds reads + mfma", "epoch2 whole",
                "epochs 3..8", "activation + split"]
    rec = {"net": name, "split_fc1_detail(diag build only)": {n: round(float(v)) for n, v in zip(ep_names, epd)}, "loss_detail": {n: round(float(v)) for n, v in zip(loss_names, ld)}, "total_cycles": tot, "phases": {n: round(float(v)) for n, v in zip(NAMES, med)},
           "share_pct": {n: round(100 * float(v) / tot, 1) for n, v in zip(NAMES, med)}}
    print(json.dumps(rec), flush=True)
