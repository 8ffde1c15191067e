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


class ModuleShardedTrainer:
    """Query-sharded data parallelism for scorers whose backward fills `param.grad` (the `make_model` networks of
    architeture/multiLayer.py): every rank runs forward + loss + backward on its slates, then ONE all-reduce(SUM) of the
    flattened gradients (+ the loss) and the optimizer step.  `reduction`: "mean" (approxNDCG: each rank's gradient is
    weighted by B_local / B_global first) or "sum" (ListNet, lambdaLoss with reduction="sum").  Every rank gets its own
    dropout streams (module.ltr_seed is offset by the rank)."""

    def __init__(self, module, optimizer, reduction="mean", group=None):
        if reduction not in ("mean", "sum"):
            raise ValueError("reduction must be 'mean' or 'sum'")
        self.module, self.opt, self.reduction, self.group = module, optimizer, reduction, group
        self.rank, self.world_size = world()
        if hasattr(module, "ltr_seed"):
            module.ltr_seed = (int(module.ltr_seed) + 0xA24BAED4963EE407 * self.rank) & ((1 << 64) - 1)
        self.params = [p for p in module.parameters() if p.requires_grad]

    def step(self, loss_closure, b_local, global_batch=None):
        """loss_closure() -> this rank's loss (0-dim, attached to the graph).  Returns the GLOBAL loss (0-dim)."""
        self.opt.zero_grad(set_to_none=True)
        loss = loss_closure()
        loss.backward()
        dev = loss.device
        if self.world_size > 1:
            if global_batch:
                gb = int(global_batch)
            else:
                t = torch.tensor([int(b_local)], dtype=torch.int64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                gb = int(t.item())
            w = float(b_local) / gb if self.reduction == "mean" else 1.0
            dt = torch.float64 if any(p.dtype == torch.float64 for p in self.params) else torch.float32
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dt) for p in self.params]
                             + [loss.detach().reshape(1).to(dt)]) * w
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            off = 0
            for p in self.params:
                g = flat[off:off + p.numel()].view_as(p).to(p.dtype)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += p.numel()
            loss = flat[-1]
        self.opt.step()
        return loss.detach()
