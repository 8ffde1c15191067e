#!/usr/bin/env python3
"""Round-2 golden vectors (SURVEY.md rows f-1 risk losses, f-4 eval metrics) from the REAL reference.

Same rules as make_golden.py (which is untouched, so its 731 arrays stay bit-identical): runs ONLY in the
build container against the read-only reference checkout, calls the reference functions on seeded CPU inputs,
asserts that the oracle restatements (oracle/ltr_risk_oracle.py, oracle/ltr_metrics_oracle.py) reproduce
them -- this pins the oracles -- and stores inputs + expected outputs as plain arrays:

    tests/golden/risk.npz, tests/golden/metrics.npz, tests/golden/manifest_r2.json

Includes the reference's ONLY known-answer test: geoRisk(5x8 matrix, alpha=3) = 0.31438308416523303
(tests/georiskTorchTest.py:5-12); the matrix is data held by that test file.
Usage:  python tests/golden/make_golden_r2.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("LTR_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from losses.riskLosses import riskLosses as RL                 # noqa: E402  (reference)
from losses.riskLosses.riskFunctions import geoRisk, zRisk     # noqa: E402
from utils import metrics as RM                                # noqa: E402
import ltr_risk_oracle as RO                                   # noqa: E402
import ltr_metrics_oracle as MO                                # noqa: E402

torch.manual_seed(2020)
np.random.seed(2020)
torch.set_num_threads(4)

ARR, MANIFEST, WORST = {}, {}, {}


def relerr(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    if a.numel() == 0:
        return 0.0
    if not torch.isfinite(b).all():
        same = (torch.isfinite(a) == torch.isfinite(b)).all() and torch.allclose(
            torch.nan_to_num(a, 0, 0, 0), torch.nan_to_num(b, 0, 0, 0), rtol=1e-5, atol=1e-12)
        return 0.0 if same else float("inf")
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def note(kind, err, tol, what=""):
    assert err == err, f"{kind} {what}: NaN deviation"
    WORST[kind] = max(WORST.get(kind, 0.0), err)
    assert err <= tol, f"{kind} {what}: oracle deviates from reference by {err:.3e} > {tol:.1e}"


def put(group, case, **arrays):
    g = ARR.setdefault(group, {})
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        g[f"{case['id']}/{k}"] = np.asarray(v)
    MANIFEST.setdefault(group, []).append(case)


KAT = [[0.0500, 0.1500, 0.3000, 0.4500, 0.5500, 0.4000, 0.3500, 0.3000],
       [0.2500, 0.2000, 0.3000, 0.3000, 0.3000, 0.3000, 0.3000, 0.2500],
       [0.2500, 0.2500, 0.2500, 0.2500, 0.4000, 0.1500, 0.4000, 0.1500],
       [0.4000, 0.2000, 0.4500, 0.2000, 0.4500, 0.2000, 0.2542, 0.2629],
       [0.2802, 0.2975, 0.3061, 0.2918, 0.2994, 0.3147, 0.3301, 0.3378]]


# ------------------------------------------------------------------------------------------- risk functions
def gen_risk_functions():
    mats = [("kat", torch.tensor(KAT), 3.0)]
    for Q, n, alpha in ((5, 2, 5.0), (37, 3, 5.0), (100, 8, 1.0), (1000, 4, 2.0), (4096, 5, 5.0)):
        mats.append((f"Q{Q}_n{n}", torch.rand(Q, n) * 0.9 + 0.05, alpha))
    for name, m, alpha in mats:
        for fn_name, ref_fn, geo in (("zRisk", zRisk, False), ("geoRisk", geoRisk, True)):
            for i in (0, -1):
                cid = f"{fn_name}_{name}_i{i}"
                x = m.clone().requires_grad_(True)
                out = ref_fn(x, alpha, requires_grad=True, i=i)
                out.sum().backward()
                x64 = m.double().clone().requires_grad_(True)
                out64 = ref_fn(x64, alpha, requires_grad=True, i=i)
                out64.sum().backward()
                # pin the oracle: autograd restatement and closed form, fp32 and fp64
                xo = m.clone().requires_grad_(True)
                oo = (RO.geo_risk if geo else RO.z_risk)(xo, alpha, i)
                oo.sum().backward()
                note("risk/autograd32", max(relerr(oo.detach(), out.detach()), relerr(xo.grad, x.grad)), 2e-5)
                cv, cg = RO.risk_closed_form(m.double(), alpha, i, geo)
                note("risk/closed64", max(relerr(cv, out64.detach()), relerr(cg, x64.grad)), 2e-6)
                put("risk", dict(id=cid, kind="function", fn=fn_name, alpha=alpha, i=i, Q=m.shape[0], n=m.shape[1]),
                    mat=m, value=out.detach(), grad=x.grad, value64=out64.detach(), grad64=x64.grad)
    kat = geoRisk(torch.tensor(KAT), 3)
    assert abs(float(kat) - 0.31438308416523303) < 1e-6, float(kat)       # tests/georiskTorchTest.py:12


# ------------------------------------------------------------------------------------------------- risk losses
def gen_risk_losses():
    pairs = {
        "geoRiskListnetLoss": (RL.geoRiskListnetLoss, RO.geo_risk_listnet, (1, 2, 3)),
        "zRiskListnetLoss": (RL.zRiskListnetLoss, RO.z_risk_listnet, (1, 2, 3)),
        "geoRiskLambdaLoss": (RL.geoRiskLambdaLoss, RO.geo_risk_lambda, (1, 2)),
        "zRiskLambdaLoss": (RL.zRiskLambdaLoss, RO.z_risk_lambda, (1, 2)),
    }
    for name, (ref_fn, ora_fn, lts) in pairs.items():
        for B, S, nb in ((6, 8, 3), (16, 32, 2)):
            for lt in lts:
                for rs, ideal, with_base in ((1, 1, True), (2, 2, True), (3, 2, False), (1, 2, False), (2, 1, True)):
                    yp = torch.randn(B, S)
                    yt = torch.randint(0, 5, (B, S)).float()
                    yb = torch.randn(B, S, nb) if with_base else None
                    kw = dict(alpha=5, listnet_transformation=lt, return_strategy=rs, add_ideal_ranking_to_mat=ideal)
                    if "Lambda" in name:
                        kw["weighing_scheme"] = "ndcgLoss2PP_scheme" if (B + lt + rs) % 2 else "lamdbaRank_scheme"
                    cid = f"{name}_B{B}_S{S}_lt{lt}_rs{rs}_id{ideal}_{'b' if with_base else 'nb'}"
                    x = yp.clone().requires_grad_(True)
                    out = ref_fn(x, yt, yb, **kw)
                    out.sum().backward()
                    x64 = yp.double().clone().requires_grad_(True)
                    out64 = ref_fn(x64, yt.double(), None if yb is None else yb.double(), **kw)
                    out64.sum().backward()
                    okw = dict(alpha=5, lt=lt, rs=rs, add_ideal=ideal)
                    if "Lambda" in name:
                        okw["scheme"] = kw["weighing_scheme"]
                    xo = yp.double().clone().requires_grad_(True)
                    oo = ora_fn(xo, yt.double(), None if yb is None else yb.double(), **okw)
                    oo.sum().backward()
                    # geoRisk drops to fp32 inside the reference even for fp64 inputs (its Normal(tensor([0.]), tensor([1.]))
                    # is fp32 and a 0-dim fp64 value does not promote it, riskFunctions.py:31-32) and evaluates the normal
                    # cdf as 0.5 * (1 + erf(v / sqrt 2)) in fp32: for v = zRisk / Q below about -2.5 the 1 + erf cancels (at
                    # v = -4.6 it is off by 0.15 %, below -5.4 it is exactly 0).  Cases in that tail cannot pin anything
                    # -- one ulp of erf moves them by per cents -- so they are stored from the fp64 oracle and flagged.
                    geo = name.startswith("geo")
                    dv, dg = relerr(oo.detach(), out64.detach()), relerr(xo.grad, x64.grad)
                    pinned = True
                    if geo and max(dv, dg) > 1e-4:
                        pt, pp, pb = RO._softmaxes(yp.double(), yt.double(), None if yb is None else yb.double())
                        m = (RO.lambda_matrix(pt, pp, pb, lt, ideal, kw["weighing_scheme"], True) if "Lambda" in name
                             else RO.listnet_matrix(pt, pp, pb, lt, ideal))
                        v = min(float(RO.z_risk(m, 5, c)) / B for c in ((0,) if rs == 1 else (0, -1)))
                        assert v < -2.5, (cid, v, dv, dg)
                        pinned = False
                    else:
                        note("risklosses/value64" + ("/geo(fp32 cdf)" if geo else ""), dv, 1e-4 if geo else 1e-6, cid)
                        note("risklosses/grad64" + ("/geo(fp32 cdf)" if geo else ""), dg, 1e-4 if geo else 1e-6, cid)
                    noise = max(relerr(out.detach(), out64.detach()), relerr(x.grad, x64.grad))
                    case = dict(id=cid, kind="loss", fn=name, B=B, S=S, n_base=nb if with_base else 0, **kw,
                                ref_fp32_vs_fp64=noise, pinned=pinned)
                    if not pinned:
                        case["why"] = "reference's fp32 normal cdf cancels in the tail (zRisk/Q < -2.5); stored from the fp64 oracle"
                        arrays = dict(y_pred=yp, y_true=yt, value=oo.detach(), grad=xo.grad, value64=oo.detach(), grad64=xo.grad)
                    else:
                        arrays = dict(y_pred=yp, y_true=yt, value=out.detach(), grad=x.grad, value64=out64.detach(), grad64=x64.grad)
                    if yb is not None:
                        arrays["y_base"] = yb
                    put("risk", case, **arrays)
    for name, ref_fn, ora_fn in (("tRiskListnetLoss", RL.tRiskListnetLoss, RO.t_risk_listnet),
                                 ("tRiskLambdaLoss", RL.tRiskLambdaLoss, RO.t_risk_lambda)):
        for B, S in ((6, 8), (16, 32)):
            for lt in (1, 2, 3):
                if "Lambda" in name and lt == 3:
                    continue          # the reference sums a [B] vector over dim=1 there and raises
                yp, yt, yb = torch.randn(B, S), torch.randint(0, 5, (B, S)).float(), torch.randn(B, S, 1)
                kw = dict(alpha=5, listnet_transformation=lt)
                cid = f"{name}_B{B}_S{S}_lt{lt}"
                x = yp.clone().requires_grad_(True)
                out = ref_fn(x, yt, yb, **kw)
                out.sum().backward()
                x64 = yp.double().clone().requires_grad_(True)
                out64 = ref_fn(x64, yt.double(), yb.double(), **kw)
                out64.sum().backward()
                xo = yp.double().clone().requires_grad_(True)
                oo = ora_fn(xo, yt.double(), yb.double(), alpha=5, lt=lt)
                oo.sum().backward()
                note("risklosses/value64", relerr(oo.detach(), out64.detach()), 1e-6, cid)
                note("risklosses/grad64", relerr(xo.grad, x64.grad), 1e-6, cid)
                noise = max(relerr(out.detach(), out64.detach()), relerr(x.grad, x64.grad))
                put("risk", dict(id=cid, kind="loss", fn=name, B=B, S=S, n_base=1, ref_fp32_vs_fp64=noise, **kw),
                    y_pred=yp, y_true=yt, y_base=yb, value=out.detach(), grad=x.grad, value64=out64.detach(), grad64=x64.grad)


# ------------------------------------------------------------------------------------------------ eval metrics
def gen_metrics():
    """utils/metrics.py:48-104 (dcg / ndcg / mNdcg, numpy) and :8-45 (getGeoRiskDefault)."""
    for Q, S in ((7, 10), (40, 32), (25, 128)):
        for variant in ("plain", "ties", "no_relevant"):
            y = np.random.randint(0, 5, size=(Q, S)).astype(np.float64)
            s = np.random.randn(Q, S)
            if variant == "ties":
                s = np.round(s, 1)                       # many tied scores: exercises both tie-break modes
            if variant == "no_relevant":
                y[0] = 0.0
                y[Q // 2] = 0.0
            inputs = f"ndcg_inputs_Q{Q}_S{S}_{variant}"          # stored once, shared by the k / gains / mode cases below
            put("metrics", dict(id=inputs, kind="ndcg_inputs", Q=Q, S=S, variant=variant), y_true=y, y_score=s)
            for k in (5, 10, 100):
                for gains in ("exponential", "linear"):
                    for no_relevant in (False, True):
                        for use_numpy in (True, False):
                            if use_numpy and variant == "ties":
                                continue      # np.argsort's default kind is not stable: tie order there is unspecified
                            per_q = RM.mNdcg(y.tolist(), s.tolist(), k=k, gains=gains, no_relevant=no_relevant,
                                             use_numpy=use_numpy)
                            val = float(np.mean(per_q))
                            oq = MO.ndcg_per_query(y, s, k=k, gains=gains, no_relevant=no_relevant, stable=not use_numpy)
                            note("metrics/ndcg", relerr(oq, np.asarray(per_q)), 1e-12)
                            cid = f"ndcg_Q{Q}_S{S}_{variant}_k{k}_{gains}_{int(no_relevant)}_{'np' if use_numpy else 'py'}"
                            put("metrics", dict(id=cid, kind="ndcg", inputs=inputs, Q=Q, S=S, k=k, gains=gains,
                                                no_relevant=no_relevant, use_numpy=use_numpy, variant=variant),
                                mean=np.float64(val), per_query=np.asarray(per_q))
    mats = [("kat", np.asarray(KAT))] + [(f"Q{Q}_n{n}", np.random.rand(Q, n) * 0.9 + 0.05) for Q, n in ((50, 3), (500, 6))]
    for name, m in mats:
        for alpha in (1.0, 3.0, 5.0):
            ref = RM.getGeoRiskDefault(m, alpha)
            note("metrics/georisk", relerr(MO.geo_risk_all_systems(m, alpha), ref), 1e-12)
            put("metrics", dict(id=f"georisk_{name}_a{alpha:g}", kind="georisk", alpha=alpha), mat=m, value=np.asarray(ref))


if __name__ == "__main__":
    gen_risk_functions()
    gen_risk_losses()
    gen_metrics()
    total = 0
    for g, arrs in ARR.items():
        path = os.path.join(HERE, f"{g}.npz")
        np.savez_compressed(path, **arrs)
        total += os.path.getsize(path)
        print(f"{g}: {len(MANIFEST[g])} cases, {os.path.getsize(path) / 1024:.0f} KiB")
    MANIFEST["_oracle_vs_reference_worst_relerr"] = WORST
    MANIFEST["_generator"] = dict(torch=torch.__version__, seed=2020, reference="Haiga/nn-with-pytorch-personalized-losses @ v1")
    with open(os.path.join(HERE, "manifest_r2.json"), "w") as f:
        json.dump(MANIFEST, f, indent=1)
    print(f"total {total / 1024:.0f} KiB; worst oracle-vs-reference deviations:")
    for k, v in sorted(WORST.items()):
        print(f"  {k:24s} {v:.3e}")
