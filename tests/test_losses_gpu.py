"""GPU parity tests proper: the HIP loss kernels (through the C ABI and the drop-in `losses` modules)
against (1) the golden vectors produced by the real reference and (2) the oracle on fresh seeded inputs,
plus size-independent properties at BASELINE.json's full sizes.

Parity bar (BASELINE.md / north_star): max|delta| / max|ref| <= 1e-5, fp32.
"""
import numpy as np
import pytest
import torch

import ltr_oracle as O
from conftest import golden, golden_cases, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import ltr_mi355x
    ltr_mi355x.lib()           # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def run(fn, s_dev):
    s = s_dev.clone().requires_grad_(True)
    out = fn(s)
    out.backward()
    return out.detach().cpu().numpy(), s.grad.cpu().numpy()


# ----------------------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("case", golden_cases("approx"), ids=lambda c: c["id"])
def test_approx_golden(case, dev):
    from losses.approxNDCG import approxNDCGLoss
    g = golden("approx")
    s, y = T(g.arr(case, "y_pred"), dev), T(g.arr(case, "y_true"), dev)
    s0, y0 = s.clone(), y.clone()
    loss, grad = run(lambda p: approxNDCGLoss(p, y, alpha=case["alpha"]), s)
    assert relerr(loss, g.arr(case, "loss64")) < TOL
    assert relerr(grad, g.arr(case, "grad64")) < TOL
    assert torch.equal(s, s0) and torch.equal(y, y0)      # inputs never mutated


@pytest.mark.parametrize("case", golden_cases("listnet"), ids=lambda c: c["id"])
def test_listnet_golden(case, dev):
    from losses.listnet import listnetLoss
    g = golden("listnet")
    s, y = T(g.arr(case, "y_pred"), dev), T(g.arr(case, "y_true"), dev)
    loss, grad = run(lambda p: listnetLoss(y, p, apply_sigmoid=case["apply_sigmoid"]), s)
    assert relerr(loss, g.arr(case, "loss64")) < TOL
    assert relerr(grad, g.arr(case, "grad64")) < TOL


def _lkw(case):
    return dict(weighing_scheme=case["scheme"], k=case["k"], sigma=case["sigma"], mu=case["mu"],
                reduction=case["reduction"], reduction_log=case["reduction_log"])


@pytest.mark.parametrize("case", golden_cases("lambda"), ids=lambda c: c["id"])
def test_lambda_golden(case, dev):
    from losses.lambdaL import lambdaLoss, lambdaMask
    g = golden("lambda")
    s, y = T(g.arr(case, "y_pred"), dev), T(g.arr(case, "y_true"), dev)
    loss, grad = run(lambda p: lambdaLoss(p, y, **_lkw(case)), s)
    assert relerr(loss, g.arr(case, "loss64")) < TOL
    assert relerr(grad, g.arr(case, "grad64")) < TOL
    if case["has_full"]:
        sr = s.clone().requires_grad_(True)
        full = lambdaMask(sr, y, return_losses=True, **_lkw(case))
        assert relerr(full.detach().cpu().numpy(), g.arr(case, "full")) < TOL
        full.backward(T(g.arr(case, "full_gup"), dev))
        assert relerr(sr.grad.cpu().numpy(), g.arr(case, "full_grad")) < 2e-5
        masked = lambdaMask(s, y, **_lkw(case))
        assert masked.shape == g.arr(case, "masked").shape
        assert relerr(masked.cpu().numpy(), g.arr(case, "masked")) < TOL


@pytest.mark.parametrize("case", golden_cases("ordinal"), ids=lambda c: c["id"])
def test_ordinal_golden(case, dev):
    from losses.ordinal import ordinalLoss, with_ordinals
    g = golden("ordinal")
    p, y = T(g.arr(case, "y_pred"), dev), T(g.arr(case, "y_true"), dev)
    assert np.array_equal(with_ordinals(y, case["n"]).cpu().numpy(), g.arr(case, "ordinals"))
    loss, grad = run(lambda q: ordinalLoss(q, y, case["n"]), p)
    assert relerr(loss, g.arr(case, "loss")) < TOL
    assert relerr(grad, g.arr(case, "grad")) < TOL


# ----------------------------------------------------------------------------- oracle, fresh inputs
SHAPES = [(1, 1), (3, 2), (5, 7), (9, 32), (6, 100), (2, 128), (3, 250), (2, 1000), (1, 2048)]


def _inputs(B, S, seed, padded):
    gen = torch.Generator().manual_seed(seed)
    s = torch.randn(B, S, generator=gen) * 2.0
    y = torch.randint(0, 5, (B, S), generator=gen).float()
    if padded and S > 2:
        for b in range(B):
            c = int(torch.randint(0, S - 1, (1,), generator=gen))
            y[b, S - c:] = -1.0
    return s, y


@pytest.mark.parametrize("B,S", SHAPES)
@pytest.mark.parametrize("padded", [False, True])
def test_approx_oracle(B, S, padded, dev):
    from losses.approxNDCG import approxNDCGLoss
    s, y = _inputs(B, S, 100 + S, padded)
    for alpha in (1.0, 3.0):
        loss, grad = run(lambda p: approxNDCGLoss(p, y.to(dev), alpha=alpha), s.to(dev))
        rl, rg, _ = O.approx_ndcg_closed_form(s.double(), y.double(), alpha=alpha)
        assert relerr(loss, rl.numpy()) < TOL and relerr(grad, rg.numpy()) < TOL


@pytest.mark.parametrize("B,S", SHAPES)
def test_listnet_oracle(B, S, dev):
    from losses.listnet import listnetLoss
    s, y = _inputs(B, S, 200 + S, False)
    for sig in (False, True):
        loss, grad = run(lambda p: listnetLoss(y.to(dev), p, apply_sigmoid=sig), s.to(dev))
        rl, rg = O.listnet_closed_form(y.double(), s.double(), sig)
        assert relerr(loss, rl.numpy()) < TOL and relerr(grad, rg.numpy()) < TOL
    # [B,S,1] inputs are squeezed like the reference does
    l3 = listnetLoss(y.to(dev)[:, :, None], s.to(dev)[:, :, None])
    assert relerr(l3.cpu().numpy(), O.listnet_closed_form(y.double(), s.double())[0].numpy()) < TOL


@pytest.mark.parametrize("B,S", [(1, 1), (3, 2), (5, 7), (9, 32), (6, 100), (2, 128), (2, 512), (1, 1000)])
@pytest.mark.parametrize("scheme", list(O.SCHEMES))
def test_lambda_oracle(B, S, scheme, dev):
    from losses.lambdaL import lambdaLoss
    s, y = _inputs(B, S, 300 + S, padded=(S % 2 == 0))
    for k, red, lg, sigma in ((None, "sum", "binary", 1.0), (5, "mean", "natural", 2.0)):
        kw = dict(weighing_scheme=scheme, k=k, sigma=sigma, mu=10.0, reduction=red, reduction_log=lg)
        rl, rg, n = O.lambda_loss_closed_form(s.double(), y.double(), **kw)
        loss, grad = run(lambda p: lambdaLoss(p, y.to(dev), **kw), s.to(dev))
        if int(n) == 0 and red == "mean":
            assert np.isnan(loss)
            continue
        assert relerr(loss, rl.numpy()) < TOL, (k, red, lg)
        assert relerr(grad, rg.numpy()) < TOL, (k, red, lg)


def test_ordinal_oracle(dev):
    from losses.ordinal import ordinalLoss
    gen = torch.Generator().manual_seed(5)
    for B, S, n in ((1, 1, 1), (3, 50, 4), (7, 333, 5)):
        p = torch.rand(B, S, n, generator=gen) * 0.98 + 0.01
        y = torch.randint(-1, 5, (B, S), generator=gen).float()
        loss, grad = run(lambda q: ordinalLoss(q, y.to(dev), n), p.to(dev))
        rl, rg = O.ordinal_closed_form(p, y, n)
        if not np.isfinite(rl.numpy()):
            continue
        assert relerr(loss, rl.numpy()) < TOL and relerr(grad, rg.numpy()) < TOL


def test_dtypes_and_contract(dev):
    from losses.approxNDCG import approxNDCGLoss
    from losses.lambdaL import lambdaLoss
    s, y = _inputs(4, 32, 1, True)
    # fp64 labels (what the reference's loader produces, utils/dataset.py:62) with fp32 scores
    l = approxNDCGLoss(s.to(dev), y.double().to(dev))
    assert l.dtype == torch.float64 and l.dim() == 0
    rl, _, _ = O.approx_ndcg_closed_form(s.double(), y.double())
    assert relerr(l.cpu().numpy(), rl.numpy()) < TOL
    with pytest.raises(ValueError, match="Reduction logarithm base can be either natural or binary"):
        lambdaLoss(s.to(dev), y.to(dev), reduction_log="decimal")
    with pytest.raises(ValueError, match="Reduction method can be either sum or mean"):
        lambdaLoss(s.to(dev), y.to(dev), reduction="max")
    with pytest.raises(KeyError):
        lambdaLoss(s.to(dev), y.to(dev), weighing_scheme="nope_scheme")
    # upstream gradient scaling
    sd = s.to(dev).requires_grad_(True)
    (3.0 * approxNDCGLoss(sd, y.to(dev))).backward()
    _, rg, _ = O.approx_ndcg_closed_form(s.double(), y.double())
    assert relerr(sd.grad.cpu().numpy(), 3.0 * rg.numpy()) < TOL


# ----------------------------------------------------------------------------- full-size properties
def test_approx_full_size_properties(dev):
    """BASELINE config 2 loss shape: 100k slates x 128.  Properties that need no oracle:
    shift invariance => per-slate gradient sums to 0; loss in [-1, 0]; permutation equivariance;
    bit-reproducibility; and a 64-slate sample against the oracle."""
    from losses.approxNDCG import approxNDCGLoss
    B, S = 100_000, 128
    gen = torch.Generator(device=dev).manual_seed(2020)
    s = torch.randn(B, S, device=dev, generator=gen)
    y = torch.randint(0, 5, (B, S), device=dev, generator=gen).float()
    sr = s.clone().requires_grad_(True)
    loss = approxNDCGLoss(sr, y)
    loss.backward()
    g1 = sr.grad.clone()
    assert -1.0 <= float(loss) <= 0.0
    assert float(g1.sum(1).abs().max()) < 1e-5 * float(g1.abs().max()) * S
    sr2 = s.clone().requires_grad_(True)
    loss2 = approxNDCGLoss(sr2, y)
    loss2.backward()
    assert torch.equal(loss, loss2) and torch.equal(g1, sr2.grad)
    perm = torch.randperm(S, device=dev)
    sp = s[:, perm].clone().requires_grad_(True)
    lp = approxNDCGLoss(sp, y[:, perm])
    lp.backward()
    assert abs(float(lp) - float(loss)) < 1e-5 * abs(float(loss))
    assert float((sp.grad - g1[:, perm]).abs().max()) < 1e-5 * float(g1.abs().max())
    idx = torch.arange(0, B, B // 64, device=dev)[:64]
    _, rg, per = O.approx_ndcg_closed_form(s[idx].cpu().double(), y[idx].cpu().double())
    assert relerr((g1[idx] * (B / 64.0)).cpu().numpy(), rg.numpy()) < TOL


def test_lambda_full_size_properties(dev):
    """BASELINE config 3 shape: S=512 (8192 slates)."""
    from losses.lambdaL import lambdaLoss
    B, S = 8192, 512
    gen = torch.Generator(device=dev).manual_seed(2020)
    s = torch.randn(B, S, device=dev, generator=gen)
    y = torch.randint(0, 5, (B, S), device=dev, generator=gen).float()
    sr = s.clone().requires_grad_(True)
    loss = lambdaLoss(sr, y, weighing_scheme="ndcgLoss2PP_scheme")
    loss.backward()
    g1 = sr.grad
    assert float(loss) > 0
    assert float(g1.sum(1).abs().max()) < 2e-5 * float(g1.abs().max()) * S
    idx = torch.arange(0, B, B // 8, device=dev)[:8]
    rl, rg, _ = O.lambda_loss_closed_form(s[idx].cpu().double(), y[idx].cpu().double(),
                                          weighing_scheme="ndcgLoss2PP_scheme")
    assert relerr(g1[idx].cpu().numpy(), rg.numpy()) < TOL
    l8 = lambdaLoss(s[idx], y[idx], weighing_scheme="ndcgLoss2PP_scheme")
    assert relerr(l8.cpu().numpy(), rl.numpy()) < TOL


def test_listnet_full_size_properties(dev):
    from losses.listnet import listnetLoss
    B, S = 100_000, 32
    gen = torch.Generator(device=dev).manual_seed(2020)
    s = torch.randn(B, S, device=dev, generator=gen).requires_grad_(True)
    y = torch.randint(0, 5, (B, S), device=dev, generator=gen).float()
    loss = listnetLoss(y, s)
    loss.backward()
    assert float(s.grad.sum(1).abs().max()) < 1e-5
    rl, _ = O.listnet_closed_form(y[:256].cpu().double(), s.detach()[:256].cpu().double())
    l256 = listnetLoss(y[:256], s.detach()[:256])
    assert relerr(l256.cpu().numpy(), rl.numpy()) < TOL
