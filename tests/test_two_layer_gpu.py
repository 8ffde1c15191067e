"""GPU: the bench-only 136-64-1 two-layer scorer (LTR_NET_TWO_LAYER_64H; the commented-out DoubleLayerNet variant of
architeture/doubleLayer.py:38-51 that BASELINE.json configs[0] names) -- module path and fused pass vs the fp64 oracle."""
import numpy as np
import pytest
import torch

import ltr_oracle as O
from conftest import relerr
from test_scorer_gpu import _grads, assert_grads

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import ltr_mi355x
    ltr_mi355x.lib()
    return torch.device("cuda:0")


def _oracle(sd, x, y, loss, dtype):
    p = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    s = O.two_layer_forward(x.to(dtype), p).squeeze(-1)
    l = O.approx_ndcg(s, y.to(dtype)) if loss == "approxNDCG" else O.listnet(y.to(dtype), s)
    l.backward()
    return l.detach().numpy(), {k: v.grad.numpy() for k, v in p.items()}, s.detach().numpy()


@pytest.mark.parametrize("S,B", [(128, 9), (32, 37), (64, 5), (100, 4)])
@pytest.mark.parametrize("loss", ["approxNDCG", "listnet"])
def test_two_layer_vs_oracle(S, B, loss, dev):
    from ltr_mi355x.extra_nets import TwoLayerNet
    from ltr_mi355x.scorer import FusedRanker
    torch.manual_seed(3)
    net = TwoLayerNet(136)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    assert list(sd) == ["fc1.weight", "fc1.bias", "fc4.weight", "fc4.bias"]
    net = net.to(dev)
    gen = torch.Generator().manual_seed(S + B)
    x = torch.randn(B, S, 136, generator=gen)
    y = torch.randint(0, 5, (B, S), generator=gen).float()
    rl, rg, rs = _oracle(sd, x, y, loss, torch.float64)
    _, rg32, _ = _oracle(sd, x, y, loss, torch.float32)
    out = net(x.to(dev), None, None)
    assert out.shape == (B, S, 1) and relerr(out.detach().cpu().numpy().squeeze(-1), rs) < TOL
    ranker = FusedRanker(net, loss="approxNDCG" if loss == "approxNDCG" else "listnet")
    l = ranker.step(x.to(dev), y.to(dev))
    assert relerr(l.cpu().numpy(), rl) < TOL
    assert_grads(_grads(net), rg, ref32=rg32)
    assert ranker.flat_grad.numel() == 64 * 136 + 64 + 64 + 1


def test_two_layer_rejected_by_other_sizes(dev):
    from ltr_mi355x.extra_nets import TwoLayerNet
    with pytest.raises(NotImplementedError):
        TwoLayerNet(64)
