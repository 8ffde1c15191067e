#!/usr/bin/env python3
"""Throughput of the rows around the hot path: risk-sensitive losses, eval metrics, data path.
One JSON line per measurement; the CPU side of each comparison is the oracle / the library the reference calls
(sklearn's svmlight parser), timed on this box's host cores.  Usage: python tools/bench_aux.py"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"), os.path.join(ROOT, "oracle")]


def gpu_ms(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def cpu_ms(fn, budget_s=6.0):
    fn()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        fn()
        n += 1
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    from losses.riskLosses import riskLosses as RL
    from ltr_mi355x import data
    from utils.metrics import getGeoRiskDefault, mNdcg_device
    import ltr_metrics_oracle as MO
    import ltr_risk_oracle as RO
    gen = torch.Generator().manual_seed(2020)

    # ---- f-1: a risk loss forward + backward (Lambda flavour = pair-matrix column sums + geoRisk), batch drivers' B = 100
    for B, S, nb in ((100, 128, 3), (2000, 128, 3)):
        yp, yt, yb = torch.randn(B, S, generator=gen), torch.randint(0, 5, (B, S), generator=gen).float(), torch.randn(B, S, nb, generator=gen)
        ypd, ytd, ybd = yp.to(dev).requires_grad_(True), yt.to(dev), yb.to(dev)

        def dev_step():
            ypd.grad = None
            RL.geoRiskLambdaLoss(ypd, ytd, ybd, listnet_transformation=2).sum().backward()
        rec = {"row": "f-1", "what": f"geoRiskLambdaLoss fwd+bwd, B={B}, S={S}, {nb} baselines (5 lambdaMask column sums + geoRisk)",
               "gpu_ms": round(gpu_ms(dev_step), 3)}
        from ltr_mi355x.graphs import GraphedLoss
        graphed = GraphedLoss(lambda a_, b_, c_: RL.geoRiskLambdaLoss(a_, b_, c_, listnet_transformation=2), (ypd.detach(), ytd, ybd))
        rec["gpu_ms_hipgraph"] = round(gpu_ms(lambda: graphed(ypd.detach(), ytd, ybd)), 3)
        if B <= 100:
            x = yp.clone().requires_grad_(True)

            def cpu_step():
                x.grad = None
                RO.geo_risk_lambda(x, yt, yb, lt=2).sum().backward()
            rec["cpu_oracle_ms"] = round(cpu_ms(cpu_step), 2)
            rec["cpu_threads"] = torch.get_num_threads()
        rec["slates_per_s_gpu"] = round(B / rec["gpu_ms"] * 1e3)
        print(json.dumps(rec), flush=True)

        # the Listnet flavour: the whole [queries, systems] matrix from ONE launch (ltr_risk_matrix_fwd) + flip + geoRisk
        def dev_step_ln():
            ypd.grad = None
            RL.geoRiskListnetLoss(ypd, ytd, ybd, listnet_transformation=1, add_ideal_ranking_to_mat=2).sum().backward()
        rec = {"row": "f-1", "what": f"geoRiskListnetLoss fwd+bwd, B={B}, S={S}, {nb} baselines + ideal (one matrix launch + flip + geoRisk)",
               "gpu_ms": round(gpu_ms(dev_step_ln), 3)}
        graphed = GraphedLoss(lambda a_, b_, c_: RL.geoRiskListnetLoss(a_, b_, c_, listnet_transformation=1, add_ideal_ranking_to_mat=2),
                              (ypd.detach(), ytd, ybd))
        rec["gpu_ms_hipgraph"] = round(gpu_ms(lambda: graphed(ypd.detach(), ytd, ybd)), 3)
        prof_launches = None
        try:
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CUDA]) as pr:
                dev_step_ln()
                torch.cuda.synchronize()
            prof_launches = sum(1 for e in pr.events() if e.device_type is not None and "cuda" in str(e.device_type).lower())
        except Exception:
            pass
        rec["device_kernels_per_step"] = prof_launches
        print(json.dumps(rec), flush=True)

        # tRisk (one baseline): Listnet flavour = matrix launch + tail launch; Lambda flavour = softmaxes + column sums + matrix + tail
        yb1 = ybd[:, :, 0].contiguous()
        for name, fn in (("tRiskListnetLoss", lambda a_, b_, c_: RL.tRiskListnetLoss(a_, b_, c_, listnet_transformation=1)),
                         ("tRiskLambdaLoss", lambda a_, b_, c_: RL.tRiskLambdaLoss(a_, b_, c_, listnet_transformation=1))):
            def dev_step_t():
                ypd.grad = None
                fn(ypd, ytd, yb1).sum().backward()
            rec = {"row": "f-1", "what": f"{name} fwd+bwd, B={B}, S={S}, 1 baseline, transformation 1 (flip + tRisk tail in one launch)",
                   "gpu_ms": round(gpu_ms(dev_step_t), 3)}
            graphed = GraphedLoss(fn, (ypd.detach(), ytd, yb1))
            rec["gpu_ms_hipgraph"] = round(gpu_ms(lambda: graphed(ypd.detach(), ytd, yb1)), 3)
            print(json.dumps(rec), flush=True)

    # ---- f-4: NDCG@10 per query (the reference loops over queries in Python after every epoch) and GeoRisk of 4 systems
    Q, S = 100_000, 128
    y = torch.randint(0, 5, (Q, S), device=dev).float()
    s = torch.randn(Q, S, device=dev)
    ms = gpu_ms(lambda: mNdcg_device(y, s, k=10))
    ycpu, scpu = y[:2000].cpu().numpy().astype(np.float64), s[:2000].cpu().numpy()
    cms = cpu_ms(lambda: MO.ndcg_per_query(ycpu, scpu, k=10), 4.0)
    print(json.dumps({"row": "f-4", "what": f"mNdcg k=10, {Q} queries x {S} docs", "gpu_ms": round(ms, 3),
                      "queries_per_s_gpu": round(Q / ms * 1e3), "cpu_numpy_oracle_queries_per_s": round(2000 / cms * 1e3)}), flush=True)
    mat = np.random.rand(Q, 4) * 0.9 + 0.05
    md = torch.as_tensor(mat, dtype=torch.float32, device=dev)
    ms = gpu_ms(lambda: getGeoRiskDefault(md, 5.0))
    print(json.dumps({"row": "f-4", "what": f"getGeoRiskDefault, {Q} queries x 4 systems (4 launches + host copy)", "gpu_ms": round(ms, 3),
                      "cpu_numpy_oracle_ms": round(cpu_ms(lambda: MO.geo_risk_all_systems(mat, 5.0), 3.0), 2)}), flush=True)

    # ---- f-2: LETOR parsing (host, native threads vs sklearn) and the per-epoch gather on the device
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "Norm.train.txt")
        rng = np.random.default_rng(0)
        Qf, Sf, F = 300, 128, 136
        with open(path, "w") as f:
            for q in range(Qf):
                X = rng.random((Sf, F)).round(6)
                lab = rng.integers(0, 5, Sf)
                f.write("".join(f"{lab[i]} qid:{q} " + " ".join(f"{j + 1}:{X[i, j]:.6f}" for j in range(F)) + "\n" for i in range(Sf)))
        size_mb = os.path.getsize(path) / 1e6
        t0 = time.perf_counter()
        Xn, yn, qn = data.load_svmlight(path)
        t_native = time.perf_counter() - t0
        from sklearn.datasets import load_svmlight_file
        t0 = time.perf_counter()
        ref = load_svmlight_file(path, query_id=True)
        dense = ref[0].toarray().astype(np.float32)
        t_sk = time.perf_counter() - t0
        assert np.array_equal(Xn, dense)
        print(json.dumps({"row": "f-2", "what": f"LETOR parse, {Qf * Sf} docs x {F} features ({size_mb:.0f} MB text)",
                          "native_MBps": round(size_mb / t_native, 1), "native_threads": os.cpu_count(),
                          "sklearn_parse_plus_densify_MBps": round(size_mb / t_sk, 1),
                          "note": "the reference additionally walks every document in Python (utils/dataset.py:54-64)"}), flush=True)
    Xd = torch.randn(25_000, 128, 136, device=dev)
    idx = torch.randperm(25_000, device=dev)
    out = torch.empty_like(Xd)
    ms = gpu_ms(lambda: data.gather_rows(Xd, idx, out=out))
    nbytes = Xd.numel() * 4
    print(json.dumps({"row": "f-2", "what": "epoch gather X[idx], 25 000 slates x 128 x 136 fp32 (1.74 GB)", "gpu_ms": round(ms, 3),
                      "GBps_read_plus_write": round(2 * nbytes / ms / 1e6, 1), "frac_of_8TBps": round(2 * nbytes / ms / 1e6 / 8000, 3),
                      "torch_index_ms": round(gpu_ms(lambda: Xd[idx]), 3)}), flush=True)


if __name__ == "__main__":
    main()
