"""GPU: BASELINE config 4's per-GPU shard shape -- tensors past 2^31 elements (a 1.25 M x 128 x 136 shard is 21.8 G floats).

The kernels index documents with 64-bit integers (csrc/ltr_scorer.hip, ltr_fcw.h, ltr_data.hip, ltr_losses.hip); nothing had executed
them there.  These tests allocate a 130 000 x 128 x 136 fp32 tensor (2.26 G floats = 9.05 GB; the card holds 288 GB) and check
  * FusedRanker.step on a 256-slate SLICE that starts beyond element 2^31 against the fp64 oracle (1e-5, SURVEY 8c metric);
  * ONE launch over the whole tensor (n_docs = 16.64 M, X offsets past 2^31 inside the kernel): the per-slate losses of the last 256
    slates equal those of a launch on just that slice, bit for bit, and the gradient matches a chunked accumulation;
  * ltr_gather_rows_f32 moving more than 2^31 floats in one launch (whole-query rows and 136-float document rows), bit-exact;
  * the standalone approxNDCG / lambdaLoss kernels on [17 M, 128] scores (2.18 G elements): the tail's gradients equal a small launch's.
Reference: main_batch_execution.py:112-171 (epoch gather, minibatch loop), losses/approxNDCG.py:7-53, losses/lambdaL.py:7-93."""
import numpy as np
import pytest
import torch

import ltr_oracle as O
from conftest import relerr
from test_scorer_gpu import _grads, _make, _oracle_step, assert_grads

pytestmark = pytest.mark.gpu
Q, S, F = 130_000, 128, 136
TWO31 = 2 ** 31


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import ltr_mi355x
    ltr_mi355x.lib()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def big(dev):
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 60 * 2 ** 30:
        pytest.skip("needs ~40 GB of free device memory")
    gen = torch.Generator(device=dev).manual_seed(4)
    X = torch.empty((Q, S, F), dtype=torch.float32, device=dev)
    for i in range(0, Q, 8192):
        X[i:i + 8192].normal_(generator=gen)
    y = torch.randint(0, 5, (Q, S), generator=gen, device=dev).float()
    assert X.numel() > TWO31
    yield X, y
    del X, y
    torch.cuda.empty_cache()


@pytest.mark.parametrize("kind", ["double", "triple", "two64"])
def test_fused_step_on_a_slice_beyond_two_to_the_31_elements(kind, big, dev):
    from ltr_mi355x.scorer import FusedRanker
    X, y = big
    lo = 124_000
    assert lo * S * F > TWO31
    if kind == "two64":
        from ltr_mi355x.extra_nets import TwoLayerNet
        torch.manual_seed(9)
        net = TwoLayerNet(F)
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        net = net.to(dev).eval()
    else:
        net, sd = _make(kind, dev, 9)
        net.eval()
    xs, ys = X[lo:lo + 256], y[lo:lo + 256]
    ranker = FusedRanker(net, loss="approxNDCG")
    out = ranker.step(xs, ys)

    def oracle(dtype):
        p = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
        xd = xs.cpu().to(dtype)
        s = (O.triple_layer_forward(xd, p) if kind == "triple" else O.two_layer_forward(xd, p) if kind == "two64"
             else O.double_layer_forward(xd, p, None, None)).squeeze(-1)
        l = O.approx_ndcg(s, ys.cpu().to(dtype))
        l.backward()
        return l.detach().numpy(), {k: v.grad.numpy() for k, v in p.items()}
    rl, rg = oracle(torch.float64)
    _, rg32 = oracle(torch.float32)
    assert relerr(out.cpu().numpy(), rl) < 1e-5
    assert_grads(_grads(net), rg, ref32=rg32)


@pytest.mark.parametrize("kind", ["double", "two64"])
def test_one_launch_over_more_than_two_to_the_31_floats(kind, big, dev):
    """n_docs = 16.64 M in ONE launch: the persistent workgroups walk X offsets past 2^31 floats."""
    from ltr_mi355x.scorer import FusedRanker
    X, y = big
    if kind == "two64":
        from ltr_mi355x.extra_nets import TwoLayerNet
        torch.manual_seed(2)
        net = TwoLayerNet(F).to(dev).eval()
    else:
        net, _ = _make(kind, dev, 2)
        net.eval()
    ranker = FusedRanker(net, loss="approxNDCG")
    loss_all = ranker.step(X, y).clone()
    slate_all = ranker._slate[:Q].clone()
    g_all = ranker.flat_grad.double().clone()
    assert torch.isfinite(slate_all).all() and torch.isfinite(g_all).all()
    # the tail of the launch (slates whose X lies beyond 2^31 floats) against a launch on just that slice: per-slate losses bit for bit
    ranker.step(X[-256:], y[-256:])
    assert torch.equal(ranker._slate[:256], slate_all[-256:])
    # the whole-launch loss / gradient against an accumulation over 13 chunks of 10 000 slates (each chunk: its own launch)
    acc_g = torch.zeros_like(g_all)
    acc_l = 0.0
    for i in range(0, Q, 10_000):
        l = ranker.step(X[i:i + 10_000], y[i:i + 10_000])
        n = min(10_000, Q - i)
        acc_l += float(l) * n / Q
        acc_g += ranker.flat_grad.double() * (n / Q)
    assert abs(acc_l - float(loss_all)) / abs(acc_l) < 1e-5
    assert float((acc_g - g_all).abs().max() / acc_g.abs().max()) < 1e-4      # fp32 partials of 16.6 M documents vs 13 x 1.28 M


def test_gather_of_more_than_two_to_the_31_floats_in_one_launch(big, dev):
    from ltr_mi355x.data import gather_rows
    X, _ = big
    gen = torch.Generator(device=dev).manual_seed(1)
    # whole-query rows (17 408 floats): the per-epoch shuffle of main_batch_execution.py:112-117
    idx = torch.randperm(Q, generator=gen, device=dev)
    out = gather_rows(X, idx)
    assert out.shape == X.shape and out.numel() > TWO31
    for k in (0, 1, Q // 2, Q - 2, Q - 1):               # destination rows at both ends (the last ones beyond 2^31 floats)
        assert torch.equal(out[k], X[idx[k]])
    sample = torch.randint(0, Q, (512,), generator=gen, device=dev)
    assert torch.equal(out[sample], X[idx[sample]])
    del out
    # document rows (136 floats): 16.64 M rows, sources and destinations on both sides of 2^31
    docs = X.view(Q * S, F)
    idx2 = torch.randperm(Q * S, generator=gen, device=dev)
    out2 = gather_rows(docs, idx2)
    for k in (0, Q * S // 2, Q * S - 1):
        assert torch.equal(out2[k], docs[idx2[k]])
    sample = torch.randint(0, Q * S, (4096,), generator=gen, device=dev)
    assert torch.equal(out2[sample], docs[idx2[sample]])
    far = torch.nonzero(idx2[-4096:] * F > TWO31).flatten()          # far destination rows that read far source rows
    assert far.numel() > 0
    k = Q * S - 4096 + int(far[0])
    assert torch.equal(out2[k], docs[idx2[k]])


@pytest.mark.parametrize("loss", ["approxNDCG", "lambdaLoss"])
def test_loss_kernels_on_more_than_two_to_the_31_scores(loss, dev):
    from losses.approxNDCG import approxNDCGLoss
    from losses.lambdaL import lambdaLoss
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 60 * 2 ** 30:
        pytest.skip("needs ~40 GB of free device memory")
    B = 17_000_000
    assert B * S > TWO31
    gen = torch.Generator(device=dev).manual_seed(8)
    s = torch.empty((B, S), device=dev)
    for i in range(0, B, 1_000_000):
        s[i:i + 1_000_000].normal_(generator=gen)
    y = torch.randint(0, 5, (B, S), generator=gen, device=dev).float()
    s.requires_grad_(True)
    fn = (lambda a, b: approxNDCGLoss(a, b)) if loss == "approxNDCG" else (lambda a, b: lambdaLoss(a, b, weighing_scheme="ndcgLoss2PP_scheme"))
    l = fn(s, y)
    l.backward()
    g_tail = s.grad[-64:].clone()
    assert bool(torch.isfinite(l)) and bool(torch.isfinite(g_tail).all())
    st = s.detach()[-64:].clone().requires_grad_(True)
    lt = fn(st, y[-64:])
    lt.backward()
    scale = (64.0 / B) if loss == "approxNDCG" else 1.0                 # approxNDCG is a batch mean, lambdaLoss("sum") a sum
    assert relerr(g_tail.cpu().numpy(), (st.grad * scale).cpu().numpy()) < 1e-6
    del s, y
    torch.cuda.empty_cache()
