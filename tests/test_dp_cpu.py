"""CPU (gloo, world_size 2): the query-sharded trainer's only collective -- one all-reduce(SUM) of the flat
[grads | loss] buffer -- reproduces the single-process full-batch step.  The per-rank compute is a stand-in
built on the oracle (the HIP path needs a GPU; the -m gpu tests cover it)."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["l1.weight", "l1.bias", "l2.weight", "l2.bias", "l3.weight", "l3.bias"]
SHAPES = [(64, 136), (64,), (32, 64), (32,), (1, 32), (1,)]


def _init_params():
    g = torch.Generator().manual_seed(2020)
    return [torch.nn.Parameter(torch.randn(s, generator=g, dtype=torch.float64) * 0.1) for s in SHAPES]


class OracleLocalStep:
    """Same contract as ltr_mi355x.scorer.FusedRanker: .flat = [grads | loss], p.grad aliases it,
    .step(X, y, world_batch) leaves this rank's contribution scaled for the global batch; with deferred=True also the
    deferred-normalisation protocol (.flat_ext = [grads | loss | normaliser], step(defer_norm=True), finish_norm())."""

    def __init__(self, params, loss="approxNDCG", deferred=True):
        import ltr_oracle as O
        self.O = O
        self.params = params
        self.loss = loss
        n = sum(p.numel() for p in params)
        ext = torch.zeros(n + 2, dtype=torch.float64)
        self.flat = ext[:n + 1]
        if deferred:
            self.flat_ext = ext
            self.finish_norm = self._finish_norm
        self._norm = ext[n + 1:]
        self.seed_salt = -1        # QueryShardedTrainer must set the per-rank dropout salt
        self.calls = []
        off = 0
        for p in params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    @property
    def mean_kind(self):
        return "batch" if self.loss == "approxNDCG" else "pairs"

    def _finish_norm(self):
        self.flat.div_(self._norm)

    def step(self, X, y, world_batch=None, defer_norm=False):
        self.calls.append("defer" if defer_norm else "prescaled")
        B = X.shape[0]
        gb = world_batch or B
        sd = dict(zip(KEYS, self.params))
        s = self.O.triple_layer_forward(X, sd).squeeze(-1)
        if self.loss == "approxNDCG":
            loss = self.O.approx_ndcg(s, y) * (B if defer_norm else B / gb)      # sum over slates | mean over the GLOBAL batch
            self._norm.fill_(float(B))
        else:                                                                     # lambdaLoss reduction="mean" (lambdaL.py:88-89)
            assert defer_norm or gb == B
            losses, keep = self.O.lambda_pairs(s, y, weighing_scheme="ndcgLoss2PP_scheme")
            pairs = losses[keep]
            loss = -pairs.sum() if defer_norm else -pairs.mean()
            self._norm.fill_(float(pairs.numel()))
        grads = torch.autograd.grad(loss, self.params)
        off = 0
        for g in grads:
            self.flat[off:off + g.numel()] = g.reshape(-1)
            off += g.numel()
        self.flat[-1] = loss.detach()
        return self.flat[-1]


def _data():
    g = torch.Generator().manual_seed(7)
    X = torch.randn(12, 16, 136, generator=g, dtype=torch.float64)
    y = torch.randint(0, 5, (12, 16), generator=g).double()
    return X, y


def _worker(rank, world, port, out_dir, loss_name, deferred):
    sys.path[:0] = [os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"), os.path.join(ROOT, "oracle")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ltr_mi355x.dp import QueryShardedTrainer, shard_range, sync_parameters
    torch.manual_seed(100 + rank)                       # ranks start from DIFFERENT weights ...
    module = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(s, dtype=torch.float64)) for s in SHAPES])
    if rank == 0:
        with torch.no_grad():
            for p, q in zip(module, _init_params()):
                p.copy_(q)
    sync_parameters(module)                             # ... and are synchronised from rank 0
    params = list(module)
    local = OracleLocalStep(params, loss=loss_name, deferred=deferred)
    opt = torch.optim.Adam(params, lr=1e-2)
    tr = QueryShardedTrainer(local, opt)
    X, y = _data()
    lo, hi = shard_range(X.shape[0], rank, world)
    losses = []
    for _ in range(3):
        losses.append(float(tr.step(X[lo:hi], y[lo:hi], global_batch=X.shape[0])))
    assert local.seed_salt == rank
    # ragged shards (11 queries over 2 ranks: 6 + 5), no global batch given: the normaliser rides in the flat buffer
    # (deferred protocol: ONE collective, no host read) -- or, for local steps without it, an all-reduce of the sizes
    lo, hi = shard_range(11, rank, world)
    losses.append(float(tr.step(X[lo:hi], y[lo:hi])))
    if deferred:
        assert local.calls == (["prescaled"] * 3 if loss_name == "approxNDCG" else ["defer"] * 3) + ["defer"]
    else:
        assert local.calls == ["prescaled"] * 4
    torch.save({"losses": losses, "flat": local.flat.clone(), "params": [p.detach().clone() for p in params]},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("loss_name,deferred", [("approxNDCG", True), ("approxNDCG", False), ("lambdaLoss_mean", True)])
def test_two_rank_allreduce_equals_single_process(loss_name, deferred):
    """approxNDCG (batch mean) with and without the deferred-normalisation protocol, and lambdaLoss reduction="mean"
    (lambdaL.py:88-89: mean over the GLOBAL kept pairs -- the count is all-reduced inside the same flat buffer)."""
    import ltr_oracle  # noqa: F401  (on sys.path via conftest)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d, loss_name, deferred), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "rank0.pt"), weights_only=True)
        r1 = torch.load(os.path.join(d, "rank1.pt"), weights_only=True)
    # single process, full batch
    params = _init_params()
    local = OracleLocalStep(params, loss=loss_name)
    opt = torch.optim.Adam(params, lr=1e-2)
    X, y = _data()
    ref_losses = []
    for _ in range(3):
        local.step(X, y)
        ref_losses.append(float(local.flat[-1]))
        opt.step()
    local.step(X[:11], y[:11])
    ref_losses.append(float(local.flat[-1]))
    opt.step()
    assert r0["losses"] == r1["losses"]
    assert torch.allclose(torch.tensor(r0["losses"]), torch.tensor(ref_losses), rtol=1e-12, atol=0)
    assert torch.equal(r0["flat"], r1["flat"])
    assert torch.allclose(r0["flat"], local.flat, rtol=1e-10, atol=1e-14)
    for a, b, c in zip(r0["params"], r1["params"], params):
        assert torch.equal(a, b) and torch.allclose(a, c.detach(), rtol=1e-10, atol=1e-14)


def test_shard_range_partitions_queries():
    from ltr_mi355x.dp import shard_range
    for n in (0, 1, 7, 8, 100_000, 10_000_001):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


# ------------------------------------------------------------------------------------------------- ModuleShardedTrainer
def _encoder_stand_in(params, X, mask, y):
    """Per-rank compute of the make_model path, on the CPU: the encoder ORACLE (the HIP path needs a GPU)."""
    import ltr_encoder_oracle as EO
    import ltr_oracle as O
    cfg = dict(n_fc=1, input_norm=False, fc_dropout=0.0, has_encoder=True, n_layers=1, heads=2, enc_dropout=0.0)
    return O.approx_ndcg(EO.encoder_scores(params, X, mask, cfg), y)


def _enc_params():
    sys.path[:0] = [os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd")]
    from architeture.multiLayer import make_model
    torch.manual_seed(3)
    net = make_model(dict(sizes=[16], input_norm=False, activation=None, dropout=0.0),
                     dict(N=1, d_ff=32, h=2, dropout=0.0, positional_encoding=None), dict(d_output=1), 8).double()
    return net


def _enc_data():
    g = torch.Generator().manual_seed(9)
    X = torch.randn(7, 10, 8, generator=g, dtype=torch.float64)           # 7 slates: ragged shards (4 + 3)
    y = torch.randint(0, 5, (7, 10), generator=g).double()
    return X, y, torch.zeros(7, 10, dtype=torch.bool)


def _module_worker(rank, world, port, out_dir):
    sys.path[:0] = [os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"), os.path.join(ROOT, "oracle")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ltr_mi355x.dp import ModuleShardedTrainer, shard_range, sync_parameters
    net = _enc_params()
    if rank == 1:
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.5)
    sync_parameters(net)
    seed0 = net.ltr_seed
    tr = ModuleShardedTrainer(net, torch.optim.SGD(net.parameters(), lr=0.1), reduction="mean")
    assert (net.ltr_seed != seed0) == (rank != 0)                          # per-rank dropout streams
    X, y, m = _enc_data()
    lo, hi = shard_range(X.shape[0], rank, world)
    sd = dict(net.named_parameters())
    losses = [float(tr.step(lambda: _encoder_stand_in(sd, X[lo:hi], m[lo:hi], y[lo:hi]), hi - lo)) for _ in range(3)]
    torch.save({"losses": losses, "params": [p.detach().clone() for p in net.parameters()]}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_module_trainer_two_ranks_ragged_shards_equal_full_batch():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_module_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "rank0.pt"), weights_only=True)
        r1 = torch.load(os.path.join(d, "rank1.pt"), weights_only=True)
    sys.path[:0] = [os.path.join(ROOT, "oracle")]
    net = _enc_params()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    X, y, m = _enc_data()
    sd = dict(net.named_parameters())
    ref = []
    for _ in range(3):
        opt.zero_grad()
        loss = _encoder_stand_in(sd, X, m, y)
        loss.backward()
        opt.step()
        ref.append(float(loss))
    assert r0["losses"] == r1["losses"]
    assert max(abs(a - b) / abs(b) for a, b in zip(r0["losses"], ref)) < 1e-9
    for a, b, c in zip(r0["params"], r1["params"], net.parameters()):
        assert torch.equal(a, b)
        assert float((a - c.detach()).abs().max()) < 1e-9
