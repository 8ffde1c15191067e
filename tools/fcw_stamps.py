#!/usr/bin/env python3
"""Phase stamps of the feature-partitioned two-layer kernel (csrc/ltr_fcw.h; a -DLTR_STAMPS build of either library via LTR_LIB):
cycles per phase of one mid-run tile, median over workgroups and the 4 waves, at 1 and at 2 workgroups per CU."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"))
from ltr_mi355x import lib, scorer  # noqa: E402
from ltr_mi355x.extra_nets import TwoLayerNet  # noqa: E402
from ltr_mi355x.functional import _ptr, _stream, check  # noqa: E402

NAMES = ["X loads -> registers (+max)", "barrier A + labels (+ barrier B)", "X -> LDS (convert) + barrier C", "fc1 (fp32: 288 MFMA x 32 cyc; f16x2: 160 x 16)",
         "activation + score partials", "barrier D + score sum + barrier", "loss", "dz1 (+dw3, scales)", "dW1 (fp32: 288 MFMA; f16x2: 144)"]
dev = torch.device("cuda:0")
B, S = 25_000, 128
X = torch.randn(B, S, 136, device=dev)
y = torch.randint(0, 5, (B, S), device=dev).float()
h = lib()
if len(sys.argv) > 1 and sys.argv[1] == "triplefold":       # the folded TripleLayerNet (two document-split copies of 32 sigmoid units)
    from architeture.tripleLayer import TripleLayerNet
    tnet = TripleLayerNet(136).to(dev).eval()
    pf = scorer._params_f32(tnet._ltr_params())
    info = scorer.NetInfo.get(scorer.NET_TRIPLE_FOLDED)
    packed = scorer.pack_params(scorer.NET_TRIPLE_FOLDED, scorer.triple_fold(pf, 2) + [pf[5]])
else:
    net = TwoLayerNet(136).to(dev).eval()
    info = scorer.NetInfo.get(net._ltr_net)
    packed = scorer.pack_params(net._ltr_net, net._ltr_params())
for per_cu in (1, 2):
    grid = scorer.cu_count(dev) * per_cu
    part = torch.empty(grid * info.partial_floats, device=dev)
    sl = torch.empty(B, device=dev)
    stamps = torch.zeros((grid, 8, 16), dtype=torch.int64, device=dev)
    assert h.ltr_debug_set_stamps(stamps.data_ptr(), 20 // per_cu) == 1, "library not built with -DLTR_STAMPS"
    for _ in range(3):
        check(h.ltr_fused_step(info.net, 0, _ptr(X), _ptr(y), B, S, _ptr(packed), 0, 0, None, None, 1.0, 1e-10, -1.0, 0, 1.0 / B, _ptr(sl),
                               _ptr(part), grid, _stream()), "fused")
    torch.cuda.synchronize()
    h.ltr_debug_set_stamps(None, 0)
    t = stamps[:, :4].cpu().double()
    d = (t[:, :, 1:10] - t[:, :, 0:9]).reshape(-1, 9).median(0).values
    tot = float((t[:, :, 9] - t[:, :, 0]).reshape(-1).median())
    lo = t[:, :, [6, 15, 11, 10, 12, 13, 14, 7]]
    ld = (lo[:, :, 1:] - lo[:, :, :-1]).reshape(-1, 7).median(0).values
    loss_names = ["score / label reads", "exponentials + u / mask writes", "path flags + label histogram + ideal DCG", "rank+pos sweep + row epilogue", "sums (barrier)",
                  "grad sweep", "exit barrier"]
    print(json.dumps({"loss_detail": {n: round(float(v)) for n, v in zip(loss_names, ld)}}))
    print(json.dumps({"workgroups_per_cu": per_cu, "total_cycles_per_tile_per_workgroup": tot, "phases": {n: round(float(v)) for n, v in zip(NAMES, d)}}), flush=True)
