"""GPU: TripleLayerNet's one-launch step in its FOLDED form.

tripleLayer.py:14-16 applies no activation between l1 and l2 -- l3(sigmoid(l2(l1 x))) = w3 . sigmoid((W2 W1) x + (W2 b1 + b2)) + b3 --
so the fused step runs the 136 -> 32 -> 1 network (csrc/ltr_scorer.hip: TripleFolded on the two-layer kernel, its 64 hidden rows two
document-split copies of the 32 units) and maps the gradient back per step:
  * ltr_triple_fold           against W2 @ W1, W2 @ b1 + b2 in fp64 (rows u and 32 + u equal);
  * ltr_triple_unfold_grads   against the chain rule written out in fp64 on random folded gradients;
  * FusedRanker(TripleLayerNet) folded (default) vs layer by layer (LTR_TRIPLE_FOLD=0) vs the fp64 oracle of the reference's
    layer-by-layer network: loss 1e-5, gradients at the fused tests' bar, for the three losses, ragged batches and padded slates.
(tests/test_scorer_gpu.py runs every TripleLayerNet fused case -- goldens included -- through the folded form by default.)"""
import numpy as np
import pytest
import torch

import ltr_oracle as O
from conftest import relerr
from test_scorer_gpu import _grads, _make, _oracle_step, assert_grads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import ltr_mi355x
    ltr_mi355x.lib()
    return torch.device("cuda:0")


def _weights(dev, seed, F=136):
    g = torch.Generator().manual_seed(seed)
    W1, b1 = torch.randn(64, F, generator=g) * 0.2, torch.randn(64, generator=g) * 0.1
    W2, b2 = torch.randn(32, 64, generator=g) * 0.3, torch.randn(32, generator=g) * 0.1
    w3 = torch.randn(1, 32, generator=g)
    return [t.to(dev).contiguous() for t in (W1, b1, W2, b2, w3)]


@pytest.mark.parametrize("copies,F", [(1, 136), (2, 136), (1, 64)])
def test_fold_kernel(dev, copies, F):
    from ltr_mi355x.scorer import triple_fold
    W1, b1, W2, b2, w3 = _weights(dev, 1, F)
    W1e, b1e, w3e = triple_fold([W1, b1, W2, b2, w3], copies)
    assert tuple(W1e.shape) == (32 * copies, F) and tuple(b1e.shape) == (32 * copies,) and tuple(w3e.shape) == (1, 32 * copies)
    ref_W = (W2.double() @ W1.double())
    ref_b = W2.double() @ b1.double() + b2.double()
    for c in range(copies):
        assert relerr(W1e[32 * c:32 * c + 32].cpu().numpy(), ref_W.cpu().numpy()) < 1e-7
        assert relerr(b1e[32 * c:32 * c + 32].cpu().numpy(), ref_b.cpu().numpy()) < 1e-7
        assert torch.equal(w3e[0, 32 * c:32 * c + 32], w3[0])
    if copies == 2:
        assert torch.equal(W1e[:32], W1e[32:]) and torch.equal(b1e[:32], b1e[32:])
    from ltr_mi355x import lib
    from ltr_mi355x.functional import _ptr, _stream
    assert lib().ltr_triple_fold(_ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(w3), F, 3, _ptr(W1e), _ptr(b1e), _ptr(w3e), _stream()) != 0


@pytest.mark.parametrize("copies,F", [(1, 136), (2, 136), (1, 64)])
def test_unfold_kernel(dev, copies, F):
    from ltr_mi355x.scorer import triple_unfold
    W1, b1, W2, b2, w3 = _weights(dev, 2, F)
    g = torch.Generator().manual_seed(3)
    R = 32 * copies
    g2 = torch.randn(R * F + R + R + 1, generator=g).to(dev)
    flat = torch.empty(64 * F + 64 + 32 * 64 + 32 + 32 + 1, device=dev)
    triple_unfold(g2, copies, [W1, b1, W2], flat)
    d = g2.double()
    GW = d[:R * F].view(copies, 32, F)
    G = GW.sum(0)
    gb = d[R * F:R * F + R].view(copies, 32).sum(0)
    gw3 = d[R * F + R:R * F + 2 * R].view(copies, 32).sum(0)
    ref = torch.cat([(W2.double().t() @ G).reshape(-1), W2.double().t() @ gb, (G @ W1.double().t() + gb[:, None] * b1.double()[None, :]).reshape(-1),
                     gb, gw3, d[-1:]])
    assert relerr(flat.cpu().numpy(), ref.cpu().numpy()) < 1e-7
    # piecewise too: a small tensor must not hide behind the largest one
    off = 0
    for n in (64 * F, 64, 32 * 64, 32, 32, 1):
        assert relerr(flat[off:off + n].cpu().numpy(), ref[off:off + n].cpu().numpy()) < 1e-6
        off += n


@pytest.mark.parametrize("loss", ["approxNDCG", "listnet", "lambdaLoss"])
@pytest.mark.parametrize("S,B", [(128, 37), (64, 5), (32, 130), (128, 1)])
def test_folded_step_equals_layerwise_step_and_oracle(loss, S, B, dev, monkeypatch):
    from ltr_mi355x.scorer import FusedRanker
    net, sd = _make("triple", dev, 11)
    gen = torch.Generator().manual_seed(100 * S + B)
    x = torch.randn(B, S, 136, generator=gen)
    y = torch.randint(0, 5, (B, S), generator=gen).float()
    if S >= 64:                                   # padded tails
        y[0, S - 7:] = -1.0
        if B > 2:
            y[2, 3:] = -1.0
    xd, yd = x.to(dev), y.to(dev)
    kw = dict(weighing_scheme="ndcgLoss2PP_scheme") if loss == "lambdaLoss" else {}
    folded = FusedRanker(net, loss=loss, **kw)
    assert folded.fold is not None
    lf = float(folded.step(xd, yd))
    gf = {k: v.copy() for k, v in _grads(net).items()}
    monkeypatch.setenv("LTR_TRIPLE_FOLD", "0")
    plain = FusedRanker(net, loss=loss, **kw)
    assert plain.fold is None
    lp = float(plain.step(xd, yd))
    gp = _grads(net)
    rl, rg, _ = _oracle_step("triple", sd, x, y, loss)
    _, rg32, _ = _oracle_step("triple", sd, x, y, loss, dtype=torch.float32)
    rl = float(rl)
    assert abs(lf - rl) <= 1e-5 * max(1.0, abs(rl)) and abs(lp - rl) <= 1e-5 * max(1.0, abs(rl))
    assert_grads(gf, rg, ref32=rg32)
    assert_grads(gp, rg, ref32=rg32)
    top = max(float(np.abs(v).max()) for v in rg.values())
    for k in gf:                                  # the two kernels against each other: fp32 noise of two different summation orders
        scale = float(np.abs(rg[k]).max())        # (a tensor whose exact gradient is ~0 -- the last bias -- against the top gradient)
        assert float(np.abs(gf[k] - gp[k]).max()) / (scale if scale >= 1e-3 * top else top) < 2e-5, k


def test_folded_full_size_linearity(dev):
    """BASELINE batch size (25 000 x 128): the gradient of a batch is the sum of its halves' (sum-type loss), folded form."""
    from ltr_mi355x.scorer import FusedRanker
    net, _ = _make("triple", dev, 5)
    gen = torch.Generator(device=dev).manual_seed(9)
    X = torch.randn(25_000, 128, 136, device=dev, generator=gen)
    y = torch.randint(0, 5, (25_000, 128), device=dev, generator=gen).float()
    r = FusedRanker(net, loss="listnet")
    assert r.fold is not None
    l_all = float(r.step(X, y))
    g_all = r.flat_grad.double().clone()
    l_a = float(r.step(X[:12_345], y[:12_345]))
    g_a = r.flat_grad.double().clone()
    l_b = float(r.step(X[12_345:], y[12_345:]))
    g_b = r.flat_grad.double().clone()
    assert abs(l_a + l_b - l_all) / abs(l_all) < 1e-5
    assert float((g_a + g_b - g_all).abs().max() / g_all.abs().max()) < 1e-4


@pytest.mark.parametrize("n_docs", [1, 100, 128 * 37 + 5])
def test_module_path_folded_equals_layerwise_and_oracle(n_docs, dev, monkeypatch):
    """`net(x, None, None)` + autograd (main_batch_execution.py:128-170): forward and backward launches of the folded network."""
    net, sd = _make("triple", dev, 21)
    gen = torch.Generator().manual_seed(n_docs)
    x = torch.randn(n_docs, 136, generator=gen)
    w = torch.randn(n_docs, 1, generator=gen)
    xd, wd = x.to(dev), w.to(dev)
    outs = []
    for fold in ("1", "0"):
        monkeypatch.setenv("LTR_TRIPLE_FOLD", fold)
        net.zero_grad()
        s_ = net(xd, None, None)
        (s_ * wd).sum().backward()
        outs.append((s_.detach().cpu().numpy(), _grads(net)))
    p = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    so = O.triple_layer_forward(x.double(), p)
    (so * w.double()).sum().backward()
    rg = {k: v.grad.numpy() for k, v in p.items()}
    p32 = {k: v.float().clone().requires_grad_(True) for k, v in sd.items()}
    (O.triple_layer_forward(x, p32) * w).sum().backward()
    rg32 = {k: v.grad.numpy() for k, v in p32.items()}
    for s_, g_ in outs:
        assert relerr(s_, so.detach().numpy()) < 1e-5
        assert_grads(g_, rg, ref32=rg32)
    assert relerr(outs[0][0], outs[1][0]) < 2e-6


@pytest.mark.parametrize("loss,S", [("lambdaLoss", 512), ("approxNDCG", 50), ("listnet", 200)])
def test_three_launch_path_folded(loss, S, dev):
    """Slates other than 32 / 64 / 128: forward(+save) + loss kernel + backward(saved) of the folded 136 -> 32 -> 1 network."""
    from ltr_mi355x.scorer import FusedRanker
    net, sd = _make("triple", dev, 31)
    B = 9
    gen = torch.Generator().manual_seed(S)
    x = torch.randn(B, S, 136, generator=gen)
    y = torch.randint(0, 5, (B, S), generator=gen).float()
    y[1, S - 5:] = -1.0
    kw = dict(weighing_scheme="ndcgLoss2PP_scheme") if loss == "lambdaLoss" else {}
    r = FusedRanker(net, loss=loss, **kw)
    assert r.fold32 is not None
    l = float(r.step(x.to(dev), y.to(dev)))
    rl, rg, _ = _oracle_step("triple", sd, x, y, loss)
    _, rg32, _ = _oracle_step("triple", sd, x, y, loss, dtype=torch.float32)
    assert abs(l - float(rl)) <= 1e-5 * max(1.0, abs(float(rl)))
    assert_grads(_grads(net), rg, ref32=rg32)


@pytest.mark.parametrize("loss,S,B", [("approxNDCG", 128, 9), ("listnet", 64, 20), ("lambdaLoss", 32, 33), ("approxNDCG", 50, 6)])
def test_64_feature_network_folded(loss, S, B, dev, monkeypatch):
    """TripleLayerNet(64) (TD2003): one-launch and three-launch steps and the module path run the folded 64 -> 32 -> 1 network on
    the generic pipeline; against the layer-by-layer kernels and the fp64 oracle."""
    from ltr_mi355x.scorer import FusedRanker
    net, sd = _make("triple", dev, 41, F=64)
    gen = torch.Generator().manual_seed(7 * S + B)
    x = torch.randn(B, S, 64, generator=gen)
    y = torch.randint(0, 5, (B, S), generator=gen).float()
    y[0, S - 3:] = -1.0
    xd, yd = x.to(dev), y.to(dev)
    kw = dict(weighing_scheme="ndcgLoss2PP_scheme") if loss == "lambdaLoss" else {}
    folded = FusedRanker(net, loss=loss, **kw)
    assert folded.fold is not None and folded.fold.F == 64 and folded.fold.H1 == 32
    lf = float(folded.step(xd, yd))
    gf = {k: v.copy() for k, v in _grads(net).items()}
    rl, rg, rs = _oracle_step("triple", sd, x, y, loss)
    _, rg32, _ = _oracle_step("triple", sd, x, y, loss, dtype=torch.float32)
    assert abs(lf - float(rl)) <= 1e-5 * max(1.0, abs(float(rl)))
    assert_grads(gf, rg, ref32=rg32)
    sm = net(xd, None, None).squeeze(-1).detach().cpu().numpy()                 # module path, folded
    assert relerr(sm, rs) < 1e-5
    monkeypatch.setenv("LTR_TRIPLE_FOLD", "0")
    plain = FusedRanker(net, loss=loss, **kw)
    assert plain.fold is None
    lp = float(plain.step(xd, yd))
    assert abs(lp - float(rl)) <= 1e-5 * max(1.0, abs(float(rl)))
    assert_grads(_grads(net), rg, ref32=rg32)
