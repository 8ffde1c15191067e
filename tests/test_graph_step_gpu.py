"""GPU: device-side dropout epoch + a whole training step of the transformer scorer captured into ONE hipGraph.

The dropout seeds of csrc/ltr_encoder.hip are kernel arguments: a recorded launch sequence would replay one mask.  Every kernel
therefore adds the device word EPOCH to its seed (include/ltr_encoder.h: ltr_enc_seed_set / _advance / _get), and
ltr_mi355x.graphs.GraphedTrainStep records `advance; forward; loss; backward; optimizer` once.  Checked here:
  * the stream of (seed, epoch e) is the stream of (seed + e, epoch 0) -- for the plain and the attention mask exporters;
  * replay k of the graph leaves the parameters BIT-IDENTICAL to k eager steps run with the epochs the graph went through
    (same kernels, same order, same masks), and consecutive replays draw different masks (losses on the same batch differ);
  * without dropout, graph and eager steps agree bit for bit as well (FC-only and FC + encoder).
Reference: the per-minibatch step of main_batch_execution.py:119-171 on architeture/multiLayer.py:40-66 (nn.Dropout draws a new
mask every forward: transformer.py:30,52,161)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def enc():
    assert torch.cuda.is_available()
    import ltr_mi355x
    ltr_mi355x.lib()
    from ltr_mi355x import encoder
    yield encoder
    encoder.seed_set(0)
    torch.cuda.synchronize()


def test_epoch_is_added_to_the_seed(enc):
    n = 1 << 16
    enc.seed_set(0)
    assert enc.seed_get() == 0
    ref = enc.dropout_mask(1000 + 7, 3, n, 0.1, DEV).clone()
    ref_attn = enc.attn_dropout_mask(1000 + 7, 2, 3, 40, 4, 0.25, DEV).clone()
    enc.seed_set(5)
    enc.seed_advance(2)
    assert enc.seed_get() == 7
    assert torch.equal(enc.dropout_mask(1000, 3, n, 0.1, DEV), ref)
    assert torch.equal(enc.attn_dropout_mask(1000, 2, 3, 40, 4, 0.25, DEV), ref_attn)
    assert not torch.equal(enc.dropout_mask(1001, 3, n, 0.1, DEV), ref)
    # the carry into the high word reaches the second key
    enc.seed_set(2 ** 32 - 1)
    enc.seed_advance(1)
    hi = enc.dropout_mask(0, 3, n, 0.1, DEV).clone()
    enc.seed_set(0)
    assert torch.equal(enc.dropout_mask(2 ** 32, 3, n, 0.1, DEV), hi)
    assert enc.seed_get() == 0


def _net(encoder_cfg, dropout):
    from architeture.multiLayer import make_model
    torch.manual_seed(3)
    fc = dict(sizes=[128], input_norm=False, activation=None, dropout=dropout)
    tr = dict(N=2, d_ff=256, h=4, dropout=dropout, positional_encoding=None) if encoder_cfg else None
    return make_model(fc, tr, dict(d_output=1, output_activation=None), 136).to(DEV).train()


@pytest.mark.parametrize("encoder_cfg,dropout", [(True, 0.1), (True, 0.0), (False, 0.2)])
def test_graphed_training_step_equals_eager_steps(enc, monkeypatch, encoder_cfg, dropout):
    from losses.approxNDCG import approxNDCGLoss
    from ltr_mi355x import blocks
    from ltr_mi355x.graphs import GraphedTrainStep
    SEED = 0x1234567890ABCDEF
    from architeture.multiLayer import LTRModel
    monkeypatch.setattr(blocks, "fresh_seed", lambda: SEED)        # the host seed every launch is recorded with
    monkeypatch.setattr(LTRModel, "_ltr_next_seed", lambda self: SEED)
    B, S = 24, 64
    gen = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B, S, 136, device=DEV, generator=gen)
    y = torch.randint(0, 5, (B, S), device=DEV, generator=gen).float()
    mask = torch.zeros(B, S, dtype=torch.bool, device=DEV)
    mask[3, 50:] = True
    y[3, 50:] = -1.0

    def loss_fn(net, x, mask, y):
        return approxNDCGLoss(net(x, mask, None), y)

    net_g = _net(encoder_cfg, dropout)
    net_e = copy.deepcopy(net_g)
    opt_g = torch.optim.Adam(net_g.parameters(), lr=1e-3, capturable=True)
    opt_e = torch.optim.Adam(net_e.parameters(), lr=1e-3, capturable=True)
    E0 = 1000
    enc.seed_set(E0)
    step = GraphedTrainStep(net_g, opt_g, loss_fn, (x, mask, y), warmup=2)      # two real steps (epochs E0+1, E0+2); the capture runs nothing
    losses_g = [float(step(x, mask, y).detach()) for _ in range(3)]              # epochs E0+3 .. E0+5
    assert enc.seed_get() == E0 + 5
    losses_e = []
    for k in range(1, 6):
        enc.seed_set(E0 + k)
        opt_e.zero_grad(set_to_none=True)
        l = loss_fn(net_e, x, mask, y)
        l.backward()
        opt_e.step()
        losses_e.append(float(l.detach()))
    assert losses_g == losses_e[2:]
    for (n, a), b in zip(net_g.named_parameters(), net_e.parameters()):
        assert torch.equal(a, b), n
    if dropout > 0:
        # same batch, consecutive replays: different masks (a frozen mask with lr 1e-3 moves the loss by ~1e-3, a new one by more;
        # the direct check is the equality with the eager epochs above -- here only that the masks are not all one)
        enc.seed_set(E0 + 5)
        m5 = enc.dropout_mask(SEED, 0, 4096, dropout, DEV).clone()
        enc.seed_set(E0 + 4)
        assert not torch.equal(enc.dropout_mask(SEED, 0, 4096, dropout, DEV), m5)


def test_graphed_step_rejects_a_host_side_optimizer(enc):
    from ltr_mi355x.graphs import GraphedTrainStep
    net = _net(False, 0.0)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x = torch.randn(4, 8, 136, device=DEV)
    with pytest.raises(ValueError, match="capturable"):
        GraphedTrainStep(net, opt, lambda net, x: net(x, None, None).sum(), (x,))
    with pytest.raises(ValueError, match="device tensors"):
        GraphedTrainStep(net, torch.optim.Adam(net.parameters(), capturable=True), lambda net, x: net(x, None, None).sum(), (x.cpu(),))
