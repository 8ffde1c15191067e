"""GPU parity tests for the risk-sensitive losses: zRisk / geoRisk kernels, the pair-matrix
column sums, the tRisk tail and the six losses through the reference's own import surface
(losses.riskLosses.riskFunctions / riskLosses) against the reference's golden vectors and the fp64 oracle.
Bar: max|delta| / max|ref| <= max(1e-5, 4 x the reference's own fp32-vs-fp64 deviation on that case)."""
import numpy as np
import pytest
import torch

import ltr_oracle as O
import ltr_risk_oracle as RO
from conftest import golden, golden_cases, ledger_record, relerr
from test_oracle_golden_r2 import oracle_risk_loss

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import ltr_mi355x
    ltr_mi355x.lib()
    return torch.device("cuda:0")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_georisk_known_answer(dev):
    """The reference's only KAT: tests/georiskTorchTest.py:5-12 prints geoRisk(5x8 matrix, 3) ~ 0.31438308416523303."""
    from losses.riskLosses.riskFunctions import geoRisk
    g = golden("risk")
    case = next(c for c in g.cases if c["id"] == "geoRisk_kat_i0")
    u = geoRisk(T(g.arr(case, "mat"), dev), 3)
    assert u.shape == (1,) and u.device.type == "cuda"
    assert abs(float(u) - 0.31438308416523303) < 1e-6


@pytest.mark.parametrize("case", [c for c in golden_cases("risk") if c["kind"] == "function"], ids=lambda c: c["id"])
def test_risk_functions_golden(case, dev):
    from losses.riskLosses.riskFunctions import geoRisk, zRisk
    g = golden("risk")
    fn = geoRisk if case["fn"] == "geoRisk" else zRisk
    m = T(g.arr(case, "mat"), dev).requires_grad_(True)
    out = fn(m, case["alpha"], requires_grad=True, i=case["i"])
    assert out.shape == ((1,) if case["fn"] == "geoRisk" else ())
    out.sum().backward()
    noise = max(relerr(g.arr(case, "value"), g.arr(case, "value64")), relerr(g.arr(case, "grad"), g.arr(case, "grad64")))
    ev, eg = relerr(out.detach().cpu().numpy(), g.arr(case, "value64")), relerr(m.grad.cpu().numpy(), g.arr(case, "grad64"))
    ledger_record(f"{case['fn']} value", ev, noise)
    ledger_record(f"{case['fn']} d/dmat", eg, noise)
    assert ev < max(TOL, 4 * noise) and eg < max(TOL, 4 * noise), (ev, eg, noise)


@pytest.mark.parametrize("Q,n", [(1, 3), (2, 2), (1000, 7), (100_000, 4)])
def test_risk_functions_vs_oracle(Q, n, dev):
    from losses.riskLosses.riskFunctions import geoRisk, zRisk
    gen = torch.Generator().manual_seed(Q + n)
    m = torch.rand(Q, n, generator=gen) * 0.9 + 0.05
    for geo, fn in ((False, zRisk), (True, geoRisk)):
        for i in (0, -1, 1):
            x = m.to(dev).requires_grad_(True)
            out = fn(x, 5, requires_grad=True, i=i)
            out.sum().backward()
            v, gr = RO.risk_closed_form(m.double(), 5, i, geo)
            if Q == 1:                # one query: every residual is exactly 0 up to rounding -- compare absolutely
                assert abs(float(out.detach().sum()) - float(v.sum())) < 1e-5
                continue
            assert relerr(out.detach().cpu().numpy(), v.numpy()) < TOL, (geo, i)
            assert relerr(x.grad.cpu().numpy(), gr.numpy()) < TOL, (geo, i)


@pytest.mark.parametrize("S", [8, 32, 128, 200])
@pytest.mark.parametrize("scheme", ["ndcgLoss2PP_scheme", "lamdbaRank_scheme", "ndcgLoss1_scheme", None])
def test_pair_colsum_vs_oracle(S, scheme, dev):
    """Column sums of the pair matrix without the [B,S,S] tensor == sum(dim=1) of the oracle's full matrix; also equal
    to summing this package's own lambdaMask(return_losses=True); and the backward."""
    from losses.lambdaL import lambdaMask
    from ltr_mi355x.risk import lambda_colsum
    B = 5
    gen = torch.Generator().manual_seed(S)
    p = torch.softmax(torch.randn(B, S, generator=gen), dim=1)
    pt = torch.softmax(torch.randint(0, 5, (B, S), generator=gen).float(), dim=1)
    x = p.double().clone().requires_grad_(True)
    ref = RO.pair_colsum(x, pt.double(), scheme)
    gup = torch.randn(B, S, generator=gen)
    ref.backward(gup.double())
    xd = p.to(dev).requires_grad_(True)
    got = lambda_colsum(xd, pt.to(dev), scheme)
    got.backward(gup.to(dev))
    assert relerr(got.detach().cpu().numpy(), ref.detach().numpy()) < TOL
    assert relerr(xd.grad.cpu().numpy(), x.grad.numpy()) < TOL
    full = lambdaMask(p.to(dev), pt.to(dev), weighing_scheme=scheme, return_losses=True)
    assert relerr(got.detach().cpu().numpy(), torch.sum(full, dim=1).cpu().numpy()) < 1e-6


@pytest.mark.parametrize("case", [c for c in golden_cases("risk") if c["kind"] == "loss"], ids=lambda c: c["id"])
def test_risk_losses_golden(case, dev):
    from losses.riskLosses import riskLosses as RL
    g = golden("risk")
    yp, yt = T(g.arr(case, "y_pred"), dev), T(g.arr(case, "y_true"), dev)
    yb = T(g.arr(case, "y_base"), dev) if g.has(case, "y_base") else None
    kw = {k: case[k] for k in ("alpha", "listnet_transformation", "return_strategy", "add_ideal_ranking_to_mat", "weighing_scheme")
          if k in case}
    x = yp.clone().requires_grad_(True)
    out = getattr(RL, case["fn"])(x, yt, yb, **kw)
    assert out.shape == (1,) and out.device.type == "cuda"
    out.sum().backward()
    # exact-arithmetic target: the fp64 oracle on the same inputs (pinned to the reference by make_golden_r2.py)
    xo = T(g.arr(case, "y_pred"), "cpu").double().requires_grad_(True)
    oo = oracle_risk_loss(case, xo, T(g.arr(case, "y_true"), "cpu").double(),
                          None if yb is None else T(g.arr(case, "y_base"), "cpu").double())
    oo.sum().backward()
    noise = case["ref_fp32_vs_fp64"]
    ev, eg = relerr(out.detach().cpu().numpy(), oo.detach().numpy()), relerr(x.grad.cpu().numpy(), xo.grad.numpy())
    ledger_record(f"{case['fn']} value", ev, noise)
    ledger_record(f"{case['fn']} d/dy_pred", eg, noise)
    assert ev < max(TOL, 4 * noise) and eg < max(TOL, 4 * noise), (ev, eg, noise)
    if case.get("pinned", True):
        # and the reference's own fp32 output, within the reference's own fp32 noise.  Strategies 2 / 3 of the geoRisk
        # losses subtract two risks of ~0.1-0.6 that the reference rounds to fp32 (its normal cdf is fp32 even for
        # fp64 inputs, so `noise` above cannot see it): 1 ulp of each is 1e-5 of a difference of 2e-3.
        bar = max(TOL, 4 * noise, 1e-4 if case["fn"].startswith("geo") and case.get("return_strategy", 1) > 1 else 0.0)
        assert relerr(out.detach().cpu().numpy(), g.arr(case, "value")) < bar
        assert relerr(x.grad.cpu().numpy(), g.arr(case, "grad")) < bar


@pytest.mark.parametrize("B,S,nb", [(2, 8, 2), (7, 33, 3), (100, 128, 3), (5, 600, 2), (4, 40, 0)])
@pytest.mark.parametrize("lt", [1, 2, 3])
@pytest.mark.parametrize("ideal", [1, 2])
def test_fused_risk_matrix_matches_the_tensor_algebra_path(B, S, nb, lt, ideal, dev):
    """ltr_risk_matrix_fwd (one launch: softmaxes, transformation, every system, d mat[:, 0] / d y_pred) against the [B, S]-sized
    tensor algebra it replaces, in fp64 (`_listnet_mat` on fp64 inputs does not take the fused path), value and gradient; and the
    Lambda-type effectiveness (mode 1) against `_effectiveness`."""
    from losses.riskLosses import riskLosses as RL
    from ltr_mi355x import risk as R
    gen = torch.Generator().manual_seed(B * 1000 + S + lt)
    yp, yt = torch.randn(B, S, generator=gen), torch.randint(0, 5, (B, S), generator=gen).float()
    yb = torch.randn(B, S, nb, generator=gen) if nb else None
    x = yp.to(dev).requires_grad_(True)
    mat, flip = RL._listnet_mat_fused(x, yt.to(dev), None if yb is None else yb.to(dev), lt, ideal)
    assert mat.shape == (B, 1 + nb + (ideal == 2)) and flip == (lt in (1, 3))
    mat = RL._flip(mat, lt)
    w = torch.randn(mat.shape, generator=gen).to(dev)
    (mat * w).sum().backward()
    x64 = yp.double().to(dev).requires_grad_(True)
    pt, pp, pb = RL._probs(x64, yt.double().to(dev), None if yb is None else yb.double().to(dev))
    ref = RL._listnet_mat(pt, pp, pb, lt, ideal)
    (ref * w.double()).sum().backward()
    assert relerr(mat.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 2e-6
    assert relerr(x.grad.cpu().numpy(), x64.grad.cpu().numpy()) < 1e-5
    if lt < 3:                                                   # mode 1: vectors taken as they are (the Lambda losses have no lt 3)
        tt, rest = torch.rand(B, S, generator=gen).to(dev), (torch.rand(nb, B, S, generator=gen).to(dev) if nb else None)
        c = torch.rand(B, S, generator=gen).to(dev).requires_grad_(True)
        m1 = R.risk_matrix(tt, c, rest, 1, lt, ideal == 2)
        (m1 * w).sum().backward()
        c64 = c.detach().double().requires_grad_(True)
        parts = [c64.unsqueeze(0)] + ([rest.double()] if rest is not None else []) + ([tt.double().unsqueeze(0)] if ideal == 2 else [])
        r1 = RL._effectiveness(tt.double(), torch.cat(parts, 0), lt).t()
        (r1 * w.double()).sum().backward()
        assert relerr(m1.detach().cpu().numpy(), r1.detach().cpu().numpy()) < 2e-6
        assert relerr(c.grad.cpu().numpy(), c64.grad.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("ideal", [False, True])
def test_fused_risk_matrix_cosine_of_a_zero_vector(ideal, dev):
    """Cosine effectiveness (mode 1: Lambda-type column sums) when one query's model vector is all zeros: F.cosine_similarity of the
    installed torch clamps EACH norm at eps = 1e-8 (x / max(|x|, eps)), not the product -- the value is 0 and the gradient w.r.t.
    the zero vector is ref / (|ref| eps).  A near-zero vector (norm 3e-9 < eps) shows how ATen differentiates the clamp: the norms
    are clamped in place under no-grad, so the value uses eps while the gradient still runs through d|v|/dv = v/|v|."""
    from losses.riskLosses import riskLosses as RL
    from ltr_mi355x import risk as R
    gen = torch.Generator().manual_seed(11)
    B, S, nb = 5, 24, 2
    tt, rest = torch.rand(B, S, generator=gen).to(dev), torch.rand(nb, B, S, generator=gen).to(dev)
    c0 = torch.rand(B, S, generator=gen)
    c0[1] = 0.0
    c0[3] = 0.0
    c0[3, 5] = 3e-9
    rest[1, 2] = 0.0                                   # a baseline with a zero vector too (no gradient flows there)
    c = c0.to(dev).requires_grad_(True)
    w = torch.randn(B, 1 + nb + int(ideal), generator=gen).to(dev)
    m1 = R.risk_matrix(tt, c, rest, 1, 2, ideal)
    (m1 * w).sum().backward()
    c64 = c0.double().to(dev).requires_grad_(True)
    parts = [c64.unsqueeze(0), rest.double()] + ([tt.double().unsqueeze(0)] if ideal else [])
    r1 = RL._effectiveness(tt.double(), torch.cat(parts, 0), 2).t()
    (r1 * w.double()).sum().backward()
    assert float(m1[1, 0]) == 0.0 and float(m1[2, 2]) == 0.0
    assert relerr(m1.detach().cpu().numpy(), r1.detach().cpu().numpy()) < 2e-6
    for b in range(B):                                 # per query: the clamped rows carry gradients of 1e8, the others of 1
        assert relerr(c.grad[b].cpu().numpy(), c64.grad[b].cpu().numpy()) < 1e-5, b


@pytest.mark.parametrize("Q,n", [(2, 2), (100, 5), (37, 1), (2000, 4)])
@pytest.mark.parametrize("kind", ["geo", "z"])
@pytest.mark.parametrize("strategy", [1, 2, 3])
@pytest.mark.parametrize("flip", [False, True])
def test_fused_risk_tail_matches_the_tensor_algebra_path(Q, n, kind, strategy, flip, dev):
    """ltr_risk_tail_fwd_bwd (flip + risk of column 0 and of the last column + return strategy + `negative` in one launch) against
    the same tail through torch ops and the per-column risk kernels (an fp64 matrix takes that path), value and gradient, incl. the
    reference's precedence quirk of zRiskListnetLoss and the max's gradient under the flip."""
    from losses.riskLosses import riskLosses as RL
    from ltr_mi355x import risk as R
    gen = torch.Generator().manual_seed(Q * 10 + n + strategy)
    base = torch.rand(Q, n, generator=gen) * 0.8 + 0.1
    k = R.RISK_GEO if kind == "geo" else R.RISK_Z
    for negative, zq in ((1, False), (-1, False), (-1, kind == "z")):
        m32 = base.to(dev).requires_grad_(True)
        out = RL._tail(k, m32, flip, 5, strategy, negative, zquirk=zq)
        out.sum().backward()
        m64 = base.double().to(dev).requires_grad_(True)
        ref = RL._tail(k, m64, flip, 5, strategy, negative, zquirk=zq)
        ref.sum().backward()
        assert out.shape == (1,)
        if bool(torch.isnan(ref).any()):                         # one system, flipped: the maximal entry becomes e = 0 -> 0 / 0, both ways
            assert bool(torch.isnan(out).all())
            continue
        # strategies 2 / 3 subtract two risks that both paths round to fp32 first: compare on the scale of the risks themselves
        scale = max(float(ref.abs().max()), 1e-2)                 # (a single system has zRisk = 0 identically)
        assert float((out.double() - ref.double()).abs().max()) / scale < 5e-5
        gs = max(float(m64.grad.abs().max()), 1e-6)
        assert float((m32.grad.double() - m64.grad).abs().max()) / gs < 2e-4, (kind, strategy, flip, negative)


@pytest.mark.parametrize("B,S", [(2, 8), (7, 33), (100, 128), (5, 600)])
@pytest.mark.parametrize("lt", [1, 2, 3])
@pytest.mark.parametrize("fn", ["tRiskListnetLoss", "tRiskLambdaLoss"])
def test_trisk_pair_through_the_fused_matrix(B, S, lt, fn, dev):
    """The tRisk pair takes its [queries, 2] matrix from ltr_risk_matrix_fwd (mode 2: the Listnet flavour's cosine is taken between
    the PRODUCTS t^2 and t p; mode 1 for the column sums of the Lambda flavour): fp32 inputs (fused) against fp64 inputs (the
    tensor-algebra path), value and gradient."""
    from losses.riskLosses import riskLosses as RL
    gen = torch.Generator().manual_seed(B * 100 + S + lt)
    yp, yt, yb = torch.randn(B, S, generator=gen), torch.randint(0, 5, (B, S), generator=gen).float(), torch.randn(B, S, generator=gen)
    x = yp.to(dev).requires_grad_(True)
    out = getattr(RL, fn)(x, yt.to(dev), yb.to(dev), alpha=5, listnet_transformation=lt)
    out.sum().backward()
    x64 = yp.double().to(dev).requires_grad_(True)
    ref = getattr(RL, fn)(x64, yt.double().to(dev), yb.double().to(dev), alpha=5, listnet_transformation=lt)
    ref.sum().backward()
    # mean / std of per-query deltas, a ratio of small differences, from fp32 column sums against fp64 ones: 1e-3 (the goldens
    # pin the same functions at max(1e-5, 4 x the reference's own fp32 noise) in test_risk_losses_golden)
    assert relerr(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 1e-3
    assert relerr(x.grad.cpu().numpy(), x64.grad.cpu().numpy()) < 1e-3


def test_risk_loss_errors_and_larger_batch(dev):
    from losses.riskLosses import riskLosses as RL
    gen = torch.Generator().manual_seed(5)
    B, S, nb = 200, 128, 3
    yp, yt, yb = torch.randn(B, S, generator=gen), torch.randint(0, 5, (B, S), generator=gen).float(), torch.randn(B, S, nb, generator=gen)
    with pytest.raises(UnboundLocalError):                       # the reference has no transformation 3 for the Lambda losses
        RL.geoRiskLambdaLoss(yp.to(dev), yt.to(dev), yb.to(dev), listnet_transformation=3)
    from ltr_mi355x import LtrDeviceError
    with pytest.raises(LtrDeviceError):
        RL.geoRiskListnetLoss(yp, yt, yb)
    assert RL.zRiskListnetLoss(yp.to(dev), yt.to(dev), yb.to(dev), return_strategy=7) is None
    for fn, ora in ((RL.zRiskLambdaLoss, RO.z_risk_lambda), (RL.geoRiskListnetLoss, RO.geo_risk_listnet)):
        x = yp.to(dev).requires_grad_(True)
        out = fn(x, yt.to(dev), yb.to(dev), listnet_transformation=2, return_strategy=1, add_ideal_ranking_to_mat=2)
        out.sum().backward()
        xo = yp.double().requires_grad_(True)
        oo = ora(xo, yt.double(), yb.double(), lt=2, rs=1, add_ideal=2)
        oo.sum().backward()
        assert relerr(out.detach().cpu().numpy(), oo.detach().numpy()) < 5e-5
        assert relerr(x.grad.cpu().numpy(), xo.grad.numpy()) < 5e-5


def test_graphed_risk_loss_replays_the_eager_step_bit_for_bit():
    """ltr_mi355x.graphs.GraphedLoss: forward + backward of a risk loss captured once into a hipGraph (ctypes launches of this
    package + ATen glue), replayed on new values: identical to the eager call on those values."""
    from losses.riskLosses import riskLosses as RL
    from ltr_mi355x.graphs import GraphedLoss
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(99)
    B, S, nb = 12, 40, 3
    mk = lambda: (torch.randn(B, S, generator=gen).to(dev), torch.randint(0, 5, (B, S), generator=gen).float().to(dev),
                  torch.randn(B, S, nb, generator=gen).to(dev))                                      # noqa: E731
    for fn in (lambda yp, yt, yb: RL.geoRiskLambdaLoss(yp, yt, yb, listnet_transformation=2),
               lambda yp, yt, yb: RL.zRiskListnetLoss(yp, yt, yb, listnet_transformation=1, return_strategy=2, add_ideal_ranking_to_mat=2)):
        step = GraphedLoss(fn, mk())
        for _ in range(3):
            yp, yt, yb = mk()
            loss, (g,) = step(yp, yt, yb)
            ype = yp.clone().requires_grad_(True)
            want = fn(ype, yt, yb)
            (gw,) = torch.autograd.grad(want.sum(), [ype])
            assert torch.equal(loss.detach(), want.detach()) and torch.equal(g, gw)
    with pytest.raises(ValueError):
        step(yp[:5], yt[:5], yb[:5])
