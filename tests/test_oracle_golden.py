"""CPU: the oracle restatement reproduces every golden vector the real reference produced
(tests/golden/make_golden.py).  This is what keeps the oracle pinned on boxes without /root/reference."""
import numpy as np
import pytest
import torch

import ltr_oracle as O
from conftest import golden, golden_cases, relerr

T = torch.from_numpy


def _grad(fn, s):
    s = s.clone().requires_grad_(True)
    out = fn(s)
    out.backward()
    return out.detach().numpy(), s.grad.numpy()


@pytest.mark.parametrize("case", golden_cases("approx"), ids=lambda c: c["id"])
def test_approx(case):
    g = golden("approx")
    s, y = T(g.arr(case, "y_pred")), T(g.arr(case, "y_true"))
    loss, grad = _grad(lambda p: O.approx_ndcg(p, y, case["eps"], case["pad"], case["alpha"]), s)
    assert relerr(loss, g.arr(case, "loss")) < 5e-6 and relerr(grad, g.arr(case, "grad")) < 5e-6
    cl, cg, _ = O.approx_ndcg_closed_form(s, y, case["eps"], case["pad"], case["alpha"])
    assert relerr(cl, g.arr(case, "loss")) < 1e-5 and relerr(cg, g.arr(case, "grad")) < 1e-5
    cl, cg, _ = O.approx_ndcg_closed_form(s.double(), y.double(), case["eps"], case["pad"], case["alpha"])
    assert relerr(cl, g.arr(case, "loss64")) < 1e-9 and relerr(cg, g.arr(case, "grad64")) < 1e-9


@pytest.mark.parametrize("case", golden_cases("listnet"), ids=lambda c: c["id"])
def test_listnet(case):
    g = golden("listnet")
    s, y = T(g.arr(case, "y_pred")), T(g.arr(case, "y_true"))
    loss, grad = _grad(lambda p: O.listnet(y, p, case["apply_sigmoid"]), s)
    assert relerr(loss, g.arr(case, "loss")) < 2e-6 and relerr(grad, g.arr(case, "grad")) < 2e-6
    cl, cg = O.listnet_closed_form(y.double(), s.double(), case["apply_sigmoid"])
    assert relerr(cl, g.arr(case, "loss64")) < 1e-9 and relerr(cg, g.arr(case, "grad64")) < 1e-9


def _lkw(case):
    return dict(weighing_scheme=case["scheme"], k=case["k"], sigma=case["sigma"], mu=case["mu"],
                reduction=case["reduction"], reduction_log=case["reduction_log"])


@pytest.mark.parametrize("case", golden_cases("lambda"), ids=lambda c: c["id"])
def test_lambda(case):
    g = golden("lambda")
    s, y = T(g.arr(case, "y_pred")), T(g.arr(case, "y_true"))
    loss, grad = _grad(lambda p: O.lambda_loss(p, y, **_lkw(case)), s)
    assert relerr(loss, g.arr(case, "loss")) < 1e-5 and relerr(grad, g.arr(case, "grad")) < 1e-5
    cl, cg, n = O.lambda_loss_closed_form(s.double(), y.double(), **_lkw(case))
    assert relerr(cl, g.arr(case, "loss64")) < 1e-7 and relerr(cg, g.arr(case, "grad64")) < 1e-7
    assert int(n) == int(g.arr(case, "n_kept"))
    if case["has_full"]:
        kw = _lkw(case)
        kw.pop("reduction")
        full, keep = O.lambda_pairs(s, y, **kw)
        assert relerr(full.numpy(), g.arr(case, "full")) < 1e-5
        assert np.array_equal(keep.numpy().astype(np.uint8), g.arr(case, "keep"))
        assert relerr(full[keep].numpy(), g.arr(case, "masked")) < 1e-5


@pytest.mark.parametrize("case", golden_cases("ordinal"), ids=lambda c: c["id"])
def test_ordinal(case):
    g = golden("ordinal")
    p, y = T(g.arr(case, "y_pred")), T(g.arr(case, "y_true"))
    assert np.array_equal(O.with_ordinals(y, case["n"]).numpy(), g.arr(case, "ordinals"))
    loss, grad = _grad(lambda q: O.ordinal(q, y, case["n"]), p)
    assert relerr(loss, g.arr(case, "loss")) < 2e-6 and relerr(grad, g.arr(case, "grad")) < 2e-6
    cl, cg = O.ordinal_closed_form(p, y, case["n"])
    assert relerr(cl, g.arr(case, "loss")) < 1e-5 and relerr(cg, g.arr(case, "grad")) < 1e-5


def test_scorers():
    g = golden("scorers")
    tri, dbl = g.cases
    sd = {k: T(g.arr(tri, f"sd.{k}")) for k in tri["keys"]}
    assert relerr(O.triple_layer_forward(T(g.arr(tri, "x")), sd).numpy(), g.arr(tri, "out")) < 2e-6
    sd = {k: T(g.arr(dbl, f"sd.{k}")) for k in dbl["keys"]}
    x = T(g.arr(dbl, "x"))
    assert relerr(O.double_layer_forward(x, sd).numpy(), g.arr(dbl, "out_eval")) < 2e-6
    k1, k2 = T(g.arr(dbl, "keep1")).float(), T(g.arr(dbl, "keep2")).float()
    assert relerr(O.double_layer_forward(x, sd, k1, k2).numpy(), g.arr(dbl, "out_train")) < 2e-6
    # backward of the oracle scorer (autograd over the restatement) == the reference's param grads
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    O.double_layer_forward(x, sdr, k1, k2).backward(T(g.arr(dbl, "gs")))
    for k in dbl["keys"]:
        assert relerr(sdr[k].grad.numpy(), g.arr(dbl, f"gtrain.{k}")) < 5e-6, k


def test_error_contract():
    s, y = torch.randn(2, 8), torch.randint(0, 5, (2, 8)).float()
    with pytest.raises(ValueError, match="Reduction logarithm base"):
        O.lambda_loss(s, y, reduction_log="decimal")
    with pytest.raises(ValueError, match="Reduction method"):
        O.lambda_loss(s, y, reduction="max")
    with pytest.raises(KeyError):
        O.lambda_loss(s, y, weighing_scheme="nope_scheme")
