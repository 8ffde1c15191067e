"""CPU: the round-2 oracles (risk losses, eval metrics) reproduce every golden vector the real reference produced
(tests/golden/make_golden_r2.py) -- including the reference's only known-answer test,
geoRisk(5x8, alpha=3) = 0.31438308416523303 (tests/georiskTorchTest.py:5-12 of the reference)."""
import numpy as np
import pytest
import torch

import ltr_metrics_oracle as MO
import ltr_risk_oracle as RO
from conftest import golden, golden_cases, relerr

T = torch.from_numpy
RISK_FNS = {"geoRiskListnetLoss": RO.geo_risk_listnet, "zRiskListnetLoss": RO.z_risk_listnet,
            "geoRiskLambdaLoss": RO.geo_risk_lambda, "zRiskLambdaLoss": RO.z_risk_lambda,
            "tRiskListnetLoss": RO.t_risk_listnet, "tRiskLambdaLoss": RO.t_risk_lambda}


def oracle_risk_loss(case, yp, yt, yb):
    kw = dict(alpha=case["alpha"], lt=case["listnet_transformation"])
    if not case["fn"].startswith("tRisk"):
        kw.update(rs=case["return_strategy"], add_ideal=case["add_ideal_ranking_to_mat"])
    if "weighing_scheme" in case:
        kw["scheme"] = case["weighing_scheme"]
    return RISK_FNS[case["fn"]](yp, yt, yb, **kw)


def test_georisk_known_answer():
    g = golden("risk")
    case = next(c for c in g.cases if c["id"] == "geoRisk_kat_i0")
    mat = T(g.arr(case, "mat"))
    assert mat.shape == (5, 8) and case["alpha"] == 3.0
    assert abs(float(g.arr(case, "value")[0]) - 0.31438308416523303) < 1e-6       # the reference's own comment
    assert abs(float(RO.geo_risk(mat.double(), 3.0)) - 0.31438308416523303) < 1e-7
    v, _ = RO.risk_closed_form(mat.double(), 3.0, 0, True)
    assert abs(float(v) - 0.31438308416523303) < 1e-7
    assert abs(MO.geo_risk_all_systems(mat.numpy(), 3.0)[0] - 0.31438308416523303) < 1e-7


@pytest.mark.parametrize("case", [c for c in golden_cases("risk") if c["kind"] == "function"], ids=lambda c: c["id"])
def test_risk_functions(case):
    g = golden("risk")
    m = T(g.arr(case, "mat"))
    geo = case["fn"] == "geoRisk"
    x = m.double().clone().requires_grad_(True)
    out = (RO.geo_risk if geo else RO.z_risk)(x, case["alpha"], case["i"])
    out.sum().backward()
    assert relerr(out.detach().numpy(), g.arr(case, "value64")) < 2e-6
    assert relerr(x.grad.numpy(), g.arr(case, "grad64")) < 2e-6
    v, gr = RO.risk_closed_form(m.double(), case["alpha"], case["i"], geo)
    assert relerr(v.numpy(), g.arr(case, "value64")) < 2e-6 and relerr(gr.numpy(), g.arr(case, "grad64")) < 2e-6
    assert relerr(v.numpy(), g.arr(case, "value")) < 2e-5 and relerr(gr.numpy(), g.arr(case, "grad")) < 2e-5


@pytest.mark.parametrize("case", [c for c in golden_cases("risk") if c["kind"] == "loss"], ids=lambda c: c["id"])
def test_risk_losses(case):
    g = golden("risk")
    yp, yt = T(g.arr(case, "y_pred")).double(), T(g.arr(case, "y_true")).double()
    yb = T(g.arr(case, "y_base")).double() if g.has(case, "y_base") else None
    x = yp.clone().requires_grad_(True)
    out = oracle_risk_loss(case, x, yt, yb)
    out.sum().backward()
    tol = 1e-4 if case["fn"].startswith("geo") else 1e-6           # the reference's geoRisk is fp32 after its cdf
    assert out.shape == (1,)
    assert relerr(out.detach().numpy(), g.arr(case, "value64")) < tol
    assert relerr(x.grad.numpy(), g.arr(case, "grad64")) < tol


def _inputs(g, case):
    src = next(c for c in g.cases if c["id"] == case["inputs"])
    return g.arr(src, "y_true"), g.arr(src, "y_score")


@pytest.mark.parametrize("case", [c for c in golden_cases("metrics") if c["kind"] == "ndcg"], ids=lambda c: c["id"])
def test_ndcg(case):
    g = golden("metrics")
    y, s = _inputs(g, case)
    got = MO.ndcg_per_query(y, s, k=case["k"], gains=case["gains"], no_relevant=case["no_relevant"],
                            stable=not case["use_numpy"])
    assert relerr(got, g.arr(case, "per_query")) < 1e-12
    assert abs(got.mean() - float(g.arr(case, "mean"))) < 1e-12


@pytest.mark.parametrize("case", [c for c in golden_cases("metrics") if c["kind"] == "georisk"], ids=lambda c: c["id"])
def test_georisk_metric(case):
    g = golden("metrics")
    assert relerr(MO.geo_risk_all_systems(g.arr(case, "mat"), case["alpha"]), g.arr(case, "value")) < 1e-12
