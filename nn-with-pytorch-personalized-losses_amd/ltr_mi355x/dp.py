"""Query-sharded data parallelism: one process per GPU, one all-reduce per step.

Slates are independent, so the path shards by query with no data-path collective; the only exchange is a
single all-reduce(SUM) of ONE flat fp32 buffer [all parameter gradients | loss] (37 402 floats = 150 KB for
DoubleLayerNet) over RCCL/xGMI (`torch.distributed` backend "nccl" on ROCm).  At this size the collective is
latency-bound, so it is issued once per step, not bucketed.  Reduction rules:
approxNDCG is a MEAN over the global batch -> each rank pre-scales by 1/B_global; ListNet is a SUM -> no
scaling.  Works unchanged on the gloo backend (CPU tests, world_size 2).
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_queries, rank, world_size):
    """Contiguous query shard [lo, hi) of rank; sizes differ by at most one."""
    base, rem = divmod(n_queries, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sync_parameters(module, src=0, group=None):
    """Every rank starts from rank `src`'s weights (one broadcast of the flattened parameters)."""
    _, ws = world()
    if ws == 1:
        return
    flat = torch.cat([p.detach().reshape(-1) for p in module.parameters()])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()


class QueryShardedTrainer:
    """local_step: object with `.flat` ([grads | loss] fp32 tensor whose slices alias every param.grad) and
    `.step(X, y, world_batch=...)` (ltr_mi355x.scorer.FusedRanker on the GPU; any stand-in in tests)."""

    def __init__(self, local_step, optimizer, group=None):
        self.local = local_step
        self.opt = optimizer
        self.group = group
        self.rank, self.world_size = world()
        # every rank draws its OWN dropout stream: the keep bits are keyed on (seed, local document index), so
        # without a per-rank salt document i of every shard would share one mask (ltr_scorer.hip keep_word)
        if hasattr(self.local, "seed_salt"):
            self.local.seed_salt = self.rank

    def global_batch_of(self, b_local, device):
        """Sum of the ranks' local batch sizes (one 8-byte all-reduce + host read).  `shard_range` shards differ by
        one slate, so B_local * world_size is wrong for ragged shards; callers with equal shards pass
        `global_batch` to `step` and skip this collective (bench.py does)."""
        if self.world_size == 1:
            return b_local
        t = torch.tensor([b_local], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return int(t.item())

    def step(self, X, y, global_batch=None):
        """One optimizer step on this rank's slates.  Returns the GLOBAL loss (0-dim tensor, no host sync).
        global_batch: total slates over all ranks this step; default = all-reduced sum of the local sizes
        (every rank must then take the default, or every rank pass the value)."""
        gb = int(global_batch) if global_batch else self.global_batch_of(int(X.shape[0]), X.device)
        self.local.step(X, y, world_batch=gb)
        if self.world_size > 1:
            dist.all_reduce(self.local.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.opt.step()
        return self.local.flat[-1]
