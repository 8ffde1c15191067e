"""Query-sharded data parallelism: one process per GPU, one all-reduce per step.

Slates are independent, so the path shards by query with no data-path collective; the only exchange is a
single all-reduce(SUM) of ONE flat fp32 buffer [all parameter gradients | loss] (37 402 floats = 150 KB for
DoubleLayerNet) over RCCL/xGMI (`torch.distributed` backend "nccl" on ROCm).  At this size the collective is
latency-bound, so it is issued once per step, not bucketed.  Reduction rules:
approxNDCG is a MEAN over the global batch -> each rank pre-scales by 1/B_global when the caller knows it, else the
local batch size rides along in the buffer and the division follows the all-reduce; ListNet / lambdaLoss("sum") are
SUMs -> no scaling; lambdaLoss("mean") divides by the GLOBAL kept-pair count, all-reduced in the same buffer.
Works unchanged on the gloo backend (CPU tests, world_size 2).
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
    `.step(X, y, world_batch=...)` (ltr_mi355x.scorer.FusedRanker on the GPU; any stand-in in tests).

    Deferred normalisation (local steps that have `.flat_ext`, `.finish_norm()` and `step(defer_norm=True)`): the rank
    leaves SUM-form contributions in [grads | loss] and appends its normaliser -- its batch size for the batch-mean loss
    (approxNDCG.py:53), its kept-pair count for lambdaLoss reduction="mean" (lambdaL.py:88-89) -- as one more float of
    the SAME buffer; after the single all-reduce every rank divides by the global normaliser on the device.  Ragged
    shards and data-dependent normalisers therefore cost no second collective and no host read."""

    def __init__(self, local_step, optimizer, group=None, always_collective=False):
        self.local = local_step
        self.opt = optimizer
        self.group = group
        self.rank, self.world_size = world()
        # always_collective: issue the all-reduce even at world_size 1 (exercises the backend on a one-GPU box)
        self.collective = self.world_size > 1 or (always_collective and dist.is_available() and dist.is_initialized())
        self.deferred = all(hasattr(local_step, a) for a in ("flat_ext", "finish_norm"))
        self.comm_events = None
        # every rank draws its OWN dropout stream: the keep bits are keyed on (seed, local document index), so
        # without a per-rank salt document i of every shard would share one mask (ltr_scorer.hip keep_word)
        if hasattr(self.local, "seed_salt"):
            self.local.seed_salt = self.rank

    def global_batch_of(self, b_local, device):
        """Sum of the ranks' local batch sizes (one 8-byte all-reduce + host read): only for local steps WITHOUT the
        deferred-normalisation protocol."""
        if self.world_size == 1:
            return b_local
        t = torch.tensor([b_local], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return int(t.item())

    def step(self, X, y, global_batch=None):
        """One optimizer step on this rank's slates.  Returns the GLOBAL loss (0-dim tensor, no host sync).
        global_batch: total slates over all ranks this step when the caller knows it (equal shards: bench.py) -- the
        batch-mean loss is then pre-scaled inside the launch.  Default: deferred normalisation (see the class docstring);
        every rank must make the same choice."""
        data_dependent = getattr(self.local, "mean_kind", None) == "pairs"
        ev = self.comm_events                   # optional (start, end) torch.cuda.Event pair around the collective (bench.py)
        if self.deferred and (global_batch is None or data_dependent):
            self.local.step(X, y, defer_norm=True)
            if self.collective:
                if ev is not None:
                    ev[0].record()
                dist.all_reduce(self.local.flat_ext, op=dist.ReduceOp.SUM, group=self.group)
                if ev is not None:
                    ev[1].record()
            self.local.finish_norm()
        else:
            gb = int(global_batch) if global_batch else self.global_batch_of(int(X.shape[0]), X.device)
            self.local.step(X, y, world_batch=gb)
            if self.collective:
                if ev is not None:
                    ev[0].record()
                dist.all_reduce(self.local.flat, op=dist.ReduceOp.SUM, group=self.group)
                if ev is not None:
                    ev[1].record()
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

    def step(self, loss_closure, b_local, global_batch=None, weight=None):
        """loss_closure() -> this rank's loss (0-dim, attached to the graph).  Returns the GLOBAL loss (0-dim).
        reduction "mean": every rank's gradient and loss are weighted by `weight` (default: its batch size b_local; a
        0-dim device tensor for data-dependent normalisers such as a kept-pair count) and the weight rides along as the
        last float of the one all-reduced buffer -- no size collective, no host read."""
        self.opt.zero_grad(set_to_none=True)
        loss = loss_closure()
        loss.backward()
        if self.world_size > 1:
            dt = torch.float64 if any(p.dtype == torch.float64 for p in self.params) else torch.float32
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dt) for p in self.params]
                             + [loss.detach().reshape(1).to(dt)])
            if self.reduction == "mean":
                if weight is None and global_batch:
                    flat = flat * (float(b_local) / int(global_batch))
                    w = None
                else:
                    w = (weight.detach().reshape(1).to(dt) if torch.is_tensor(weight) else
                         torch.full((1,), float(b_local if weight is None else weight), dtype=dt, device=flat.device))
                    flat = torch.cat([flat * w, w])
            else:
                w = None
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if w is not None:
                flat = flat[:-1] / flat[-1]
            off = 0
            for p in self.params:
                g = flat[off:off + p.numel()].view_as(p).to(p.dtype)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += p.numel()
            loss = flat[off]
        self.opt.step()
        return loss.detach()
