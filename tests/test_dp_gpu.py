"""GPU: the data-parallel step with the REAL fused kernel.  Two ranks share the one card of the test box, so the
collective runs on gloo (RCCL refuses two ranks on one device; on a multi-GPU node the same code uses
backend "nccl" = RCCL): query-sharded FusedRanker steps + one all-reduce of the flat [grads | loss] buffer must
reproduce the single-process full-batch step."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data():
    g = torch.Generator().manual_seed(11)
    X = torch.randn(24, 128, 136, generator=g)
    y = torch.randint(0, 5, (24, 128), generator=g).float()
    return X, y


def _net(dev):
    sys.path.insert(0, os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"))
    from architeture.tripleLayer import TripleLayerNet
    torch.manual_seed(2020)
    return TripleLayerNet(136).to(dev)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    net = _net(dev)
    from ltr_mi355x.dp import QueryShardedTrainer, shard_range, sync_parameters
    from ltr_mi355x.scorer import FusedRanker
    if rank == 1:                                   # rank 1 starts from different weights ...
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    sync_parameters(net)                            # ... and receives rank 0's
    ranker = FusedRanker(net, loss="approxNDCG")
    tr = QueryShardedTrainer(ranker, torch.optim.SGD(net.parameters(), lr=0.5))
    X, y = _data()
    lo, hi = shard_range(X.shape[0], rank, world)
    losses = [float(tr.step(X[lo:hi].to(dev), y[lo:hi].to(dev))) for _ in range(3)]
    torch.save({"losses": losses, "flat": ranker.flat.cpu(), "params": [p.detach().cpu() for p in net.parameters()]},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_equals_single_process():
    assert torch.cuda.is_available()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "rank0.pt"), weights_only=True)
        r1 = torch.load(os.path.join(d, "rank1.pt"), weights_only=True)
    dev = torch.device("cuda:0")
    net = _net(dev)
    from ltr_mi355x.scorer import FusedRanker
    ranker = FusedRanker(net, loss="approxNDCG")
    opt = torch.optim.SGD(net.parameters(), lr=0.5)
    X, y = _data()
    ref = []
    for _ in range(3):
        ref.append(float(ranker.step(X.to(dev), y.to(dev))))
        opt.step()
    assert r0["losses"] == r1["losses"]                       # both ranks see the same global loss
    assert torch.equal(r0["flat"], r1["flat"])
    assert max(abs(a - b) / abs(b) for a, b in zip(r0["losses"], ref)) < 1e-5
    top = float(ranker.flat_grad.abs().max())
    assert float((r0["flat"][:-1] - ranker.flat_grad.cpu()).abs().max()) / top < 1e-5
    for a, b, c in zip(r0["params"], r1["params"], net.parameters()):
        assert torch.equal(a, b)
        assert float((a - c.detach().cpu()).abs().max()) / max(float(c.abs().max()), 1e-30) < 1e-5
