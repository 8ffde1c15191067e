"""Host side of the slate pipeline (csrc/ltr_scorer.hip): FC scorers and the fused training pass.

Two ways in, same kernels:
  * `mlp_scores(module, x, ...)` -- the nn.Module path (`net(x, None, None)` like the reference,
    main_batch_execution.py:128): forward launch now, backward launch when autograd asks for it
    (the forward is recomputed from X inside the backward launch; activations are never stored in HBM).
  * `FusedRanker.step(...)`      -- scorer forward + listwise loss + backward + weight gradients in ONE
    launch over a batch of slates; gradients land in one flat buffer that aliases every `param.grad`
    (the buffer the data-parallel all-reduce runs on).
"""
import ctypes

import torch

from ._lib import LtrError, check, lib
from .functional import _ptr, _stream, require_device

NET_DOUBLE, NET_TRIPLE, NET_DOUBLE_64, NET_TRIPLE_64, NET_TWO_LAYER_64H, NET_TRIPLE_FOLDED, NET_TRIPLE_FOLDED_32 = 0, 1, 2, 3, 4, 5, 6
NET_TRIPLE_FOLDED_32_64 = 7
NET_WIDE = -1       # more than 136 input features (neither of the reference's collections): see wide_forward
COMPILED_FEATURES = {136: (NET_DOUBLE, NET_TRIPLE), 64: (NET_DOUBLE_64, NET_TRIPLE_64)}


def net_id(kind, n_features):
    """kind 'double' | 'triple' -> network handle for this input size: the compiled id for 136 / 64 features (MSLR-WEB /
    TD2003, the reference's collections), otherwise `compiled id | n_features << 8` -- the network runs zero-padded on the next
    compiled geometry (ltr_mlp_pack_sub / ltr_mlp_reduce_grads_sub; X rows are padded by `_docs`)."""
    n_features = int(n_features)
    which = 0 if kind == "double" else 1
    if n_features in COMPILED_FEATURES:
        return COMPILED_FEATURES[n_features][which]
    if n_features > 136:
        return NET_WIDE            # wider than the fused kernels' LDS tile: the modules run their layers as library GEMMs (wide_forward)
    if n_features < 1:
        raise ValueError(f"input_size must be positive, got {n_features}")
    return COMPILED_FEATURES[64 if n_features <= 64 else 136][which] | (n_features << 8)
LOSS_APPROXNDCG, LOSS_LISTNET, LOSS_LAMBDA = 0, 1, 2
_MASK64 = (1 << 64) - 1


class NetInfo:
    """Static geometry of a compiled network (ltr_net_info)."""
    _cache = {}

    def __init__(self, handle):
        buf = (ctypes.c_int32 * 8)()
        self.net = handle & 0xFF                     # the compiled network the kernels run
        check(lib().ltr_net_info(self.net, buf), "ltr_net_info")
        (self.F, self.H1, self.H2, self.n_params, self.packed_floats, self.partial_floats, self.tile_docs,
         self.lds_bytes) = list(buf)
        self.handle = handle
        self.cF, self.cH1, self.cH2 = self.F, self.H1, self.H2          # compiled widths (X row length the kernels read)
        # two-Linear-layer nets (no fc2) carry [W1, b1, w3, b3] only
        self.two_layer = self.n_params == self.H1 * self.F + self.H1 + self.H2 + 1
        if handle >> 8:                              # a narrower network, zero-padded onto the compiled geometry
            f = handle >> 8
            self.F = f
            if self.net in (NET_DOUBLE, NET_DOUBLE_64):
                self.H1 = self.H2 = f                # input_size -> input_size -> input_size -> 1 (doubleLayer.py:55-60)
            self.n_params = self.H1 * self.F + self.H1 + self.H2 * self.H1 + self.H2 + self.H2 + 1
        self.padded = self.F != self.cF
        self.shapes = ([(self.H1, self.F), (self.H1,), (1, self.H2), (1,)] if self.two_layer else
                       [(self.H1, self.F), (self.H1,), (self.H2, self.H1), (self.H2,), (1, self.H2), (1,)])

    @classmethod
    def get(cls, net):
        if net not in cls._cache:
            cls._cache[net] = cls(net)
        return cls._cache[net]


def cu_count(device):
    return torch.cuda.get_device_properties(device).multi_processor_count


def default_grid(device, n_docs, tile_docs=128):
    """Persistent workgroups: one per CU, never more than there are 128-document tiles."""
    return max(1, min(cu_count(device), (n_docs + tile_docs - 1) // tile_docs))


def next_seed(counter):
    """Dropout stream seed: torch's global seed mixed with a per-call counter (no device sync)."""
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + counter * 0xD1B54A32D192ED03 + 0x8CB92BA72F3D8DD7) & _MASK64


def _params_f32(params):
    out = []
    for p in params:
        t = p.detach()
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(torch.float32).contiguous()
        out.append(t)
    return out


def pack_params(net, params, out=None):
    """nn.Linear weights/biases [W1,b1,W2,b2,w3,b3] -> lane-ordered MFMA fragments (ltr_mlp_pack)."""
    info = NetInfo.get(net)
    ps = _params_f32(params)
    if len(ps) != len(info.shapes):
        raise ValueError(f"expected {len(info.shapes)} parameter tensors, got {len(ps)}")
    for t, shape in zip(ps, info.shapes):
        if tuple(t.shape) != shape:
            raise ValueError(f"parameter shape {tuple(t.shape)} != expected {shape}")
    dev = ps[0].device
    if out is None:
        out = torch.empty(info.packed_floats, dtype=torch.float32, device=dev)
    ptrs = [_ptr(t) for t in ps]
    if info.two_layer:
        ptrs = ptrs[:2] + [None, None] + ptrs[2:]          # no fc2: W2 / b2 are NULL in the C ABI
    check(lib().ltr_mlp_pack_sub(info.net, info.F, info.H1, info.H2, *ptrs, _ptr(out), _stream()), "ltr_mlp_pack_sub")
    return out


def triple_folds(handle):
    """TripleLayerNet on the 136-feature collection runs FOLDED: tripleLayer.py:14-16 has no activation between l1 and l2, so
    l2 . l1 is ONE 136 -> 32 layer (csrc/ltr_scorer.hip TripleFolded / TripleFolded32: 2.46 x fewer multiply-adds per document).
    ltr_triple_fold makes the folded weights from the six tensors every step, ltr_triple_unfold_grads turns the folded
    gradient into theirs.  LTR_TRIPLE_FOLD=0 keeps the layer-by-layer kernels (A/B, tests)."""
    import os
    return handle in (NET_TRIPLE, NET_TRIPLE_64) and os.environ.get("LTR_TRIPLE_FOLD", "1") != "0"


def folded_net32(handle):
    """The plain folded network (32 units, copies = 1) of a compiled TripleLayerNet: 136 features -> NET_TRIPLE_FOLDED_32 (the
    one-launch step of that collection runs NET_TRIPLE_FOLDED on ltr_fcw.h instead), 64 features -> NET_TRIPLE_FOLDED_32_64 (every path)."""
    return NET_TRIPLE_FOLDED_32 if handle == NET_TRIPLE else NET_TRIPLE_FOLDED_32_64


def triple_fold(pf, copies, out=None):
    """pf = [W1, b1, W2, b2, w3, b3] fp32 -> [W1e [32 copies, F], b1e [32 copies], w3e [1, 32 copies]] (+ b3 unchanged)."""
    dev = pf[0].device
    R, F = 32 * copies, int(pf[0].shape[1])
    if out is None:
        out = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in ((R, F), (R,), (1, R))]
    check(lib().ltr_triple_fold(*[_ptr(t) for t in pf[:5]], F, int(copies), *[_ptr(t) for t in out], _stream()), "ltr_triple_fold")
    return out


def triple_unfold(g2, copies, pf, flat):
    check(lib().ltr_triple_unfold_grads(_ptr(g2), int(pf[0].shape[1]), int(copies), _ptr(pf[0]), _ptr(pf[1]), _ptr(pf[2]), _ptr(flat),
                                        _stream()), "ltr_triple_unfold_grads")


def reduce_grads(info, partials, grid, flat):
    check(lib().ltr_mlp_reduce_grads_sub(info.net, info.F, info.H1, info.H2, _ptr(partials), grid, _ptr(flat), _stream()),
          "ltr_mlp_reduce_grads_sub")


def _docs(x, info):
    if x.dim() < 2 or x.shape[-1] != info.F:
        raise ValueError(f"expected [..., {info.F}] features, got {tuple(x.shape)}")
    if x.dtype != torch.float32:
        raise TypeError(f"scorer kernels take fp32 features, got {x.dtype}")
    x2 = x.detach().reshape(-1, info.F)
    if info.padded:                                  # zero-pad the rows to the compiled feature count (one extra pass over X)
        return torch.nn.functional.pad(x2, (0, info.cF - info.F))
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    if x2.data_ptr() % 16:
        x2 = x2.clone()
    return x2


def _mask(m, n_docs, H, cH=None):
    """Explicit dropout keep mask [n_docs, H] -> bytes; padded with ones to the compiled width cH."""
    if m is None:
        return None
    m = m.detach().reshape(n_docs, H).to(torch.uint8)
    if cH is not None and cH != H:
        m = torch.nn.functional.pad(m, (0, cH - H), value=1)
    return m.contiguous()


class _MLPScores(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, dropout, seed, keep1, keep2, *params):
        info = NetInfo.get(net)
        dev = x.device
        ctx.fold_pf = None
        with torch.cuda.device(dev):
            x2 = _docs(x, info)
            n = x2.shape[0]
            if triple_folds(net) and keep1 is None and keep2 is None:
                # forward and backward launches of the folded 136 -> 32 -> 1 network (NET_TRIPLE_FOLDED_32); the six tensors'
                # gradients come back through ltr_triple_unfold_grads
                pf = _params_f32(params)
                info = NetInfo.get(folded_net32(net))
                packed = pack_params(info.handle, triple_fold(pf, 1) + [pf[5]])
                ctx.fold_pf = pf[:3]
            else:
                packed = pack_params(net, params)
            k1, k2 = _mask(keep1, n, info.H1, info.cH1), _mask(keep2, n, info.H2, info.cH2)
            scores = torch.empty(n, dtype=torch.float32, device=dev)
            grid = default_grid(dev, n, info.tile_docs)
            acts = None
            if int(dropout) > 1 and k1 is None and any(p.requires_grad for p in params):
                # a dropout probability other than 0.5: only the forward kernels carry its stream, so the forward keeps the
                # hidden activations and the backward reads them (ltr_mlp_forward_save / ltr_mlp_backward_saved)
                acts = torch.empty(int(lib().ltr_mlp_acts_floats(info.net, n)), dtype=torch.float32, device=dev)
                check(lib().ltr_mlp_forward_save(info.net, _ptr(x2), n, _ptr(packed), int(dropout), seed, _ptr(k1), _ptr(k2),
                                                 _ptr(scores), _ptr(acts), grid, _stream()), "ltr_mlp_forward_save")
            else:
                check(lib().ltr_mlp_forward(info.net, _ptr(x2), n, _ptr(packed), int(dropout), seed, _ptr(k1), _ptr(k2),
                                            _ptr(scores), grid, _stream()), "ltr_mlp_forward")
        ctx.save_for_backward(x2, packed, k1, k2)
        ctx.acts = acts
        ctx.meta = (net, int(dropout), seed, grid, [p.dtype for p in params])
        return scores.view(*x.shape[:-1], 1)

    @staticmethod
    def backward(ctx, g):
        x2, packed, k1, k2 = ctx.saved_tensors
        net, dropout, seed, grid, dtypes = ctx.meta
        info = NetInfo.get(folded_net32(net) if ctx.fold_pf is not None else net)
        dev = x2.device
        n = x2.shape[0]
        with torch.cuda.device(dev):
            gs = g.detach().reshape(-1).to(torch.float32).contiguous()
            partials = torch.empty(grid * info.partial_floats, dtype=torch.float32, device=dev)
            flat = torch.empty(info.n_params, dtype=torch.float32, device=dev)
            if ctx.acts is not None:
                check(lib().ltr_mlp_backward_saved(info.net, _ptr(x2), n, _ptr(packed), dropout, _ptr(ctx.acts), _ptr(gs), _ptr(partials),
                                                   grid, _stream()), "ltr_mlp_backward_saved")
            else:
                check(lib().ltr_mlp_backward(info.net, _ptr(x2), n, _ptr(packed), dropout, seed, _ptr(k1), _ptr(k2), _ptr(gs),
                                             _ptr(partials), grid, _stream()), "ltr_mlp_backward")
            reduce_grads(info, partials, grid, flat)
            if ctx.fold_pf is not None:
                info = NetInfo.get(net)
                flat6 = torch.empty(info.n_params, dtype=torch.float32, device=dev)
                triple_unfold(flat, 1, ctx.fold_pf, flat6)
                flat = flat6
        grads, off = [], 0
        for shape, dt in zip(info.shapes, dtypes):
            cnt = 1
            for s in shape:
                cnt *= s
            grads.append(flat[off:off + cnt].view(shape).to(dt))
            off += cnt
        return (None, None, None, None, None, None) + tuple(grads)


def wide_forward(kind, params, x, train=False, p=0.5, keep1=None, keep2=None):
    """FC scorers with MORE than 136 input features (doubleLayer.py:55-60 / tripleLayer.py:6-10 accept any width; the reference's
    collections have 136 and 64): the fused slate kernels hold a 128-document tile of at most 136 features in LDS, so these widths run
    layer by layer on the device as plain library GEMMs (rocBLAS through torch.nn.functional.linear -- the one place the hand-written
    kernels are not used) with autograd's own backward.  DoubleLayerNet: fc1, ReLU, dropout, fc2, ReLU, dropout, fc3 -- torch's
    dropout in training mode (exactly the reference's stream), or the explicit keep masks.  TripleLayerNet: l2 . l1 folded into one
    layer (no activation between them, tripleLayer.py:14-15), sigmoid, l3.  The listwise losses on the resulting scores are the HIP
    kernels as always; FusedRanker does not take these networks."""
    import torch.nn.functional as Fn
    require_device(x, *params)
    if kind == "triple":
        W1, b1, W2, b2, w3, b3 = params
        return Fn.linear(torch.sigmoid(Fn.linear(x, W2 @ W1, W2 @ b1 + b2)), w3, b3)
    W1, b1, W2, b2, w3, b3 = params

    def drop(h, keep):
        if keep is not None:
            return h * keep.to(h.dtype).reshape(h.shape) * (1.0 / (1.0 - p))
        return Fn.dropout(h, p, training=train)
    h = drop(torch.relu(Fn.linear(x, W1, b1)), keep1)
    h = drop(torch.relu(Fn.linear(h, W2, b2)), keep2)
    return Fn.linear(h, w3, b3)


def mlp_scores(net, params, x, dropout=False, seed=0, keep1=None, keep2=None):
    """scores[..., 1] = net(x) on the device.  x: [..., F] fp32 device tensor.  No gradient w.r.t. x."""
    require_device(x, *params)
    return _MLPScores.apply(x, net, int(dropout), int(seed), keep1, keep2, *params)


def drop_code(train, p=0.5):
    """The `dropout` argument of the C launchers (include/ltr_mi355x.h): 0 = off, 1 = on with the reference's p = 0.5
    (doubleLayer.py:60), otherwise 1 | (the bits of float32(p) with the lowest one cleared)."""
    p = float(p)
    if not train or p <= 0.0:
        return 0
    if not p < 1.0:
        raise ValueError(f"dropout probability has to be between 0 and 1, but got {p}")
    if p == 0.5:
        return 1
    import struct
    return 1 | (struct.unpack("<I", struct.pack("<f", p))[0] & ~1)


def dropout_keep_mask(seed, layer, n_docs, H, device, p=0.5):
    if p != 0.5:
        out = torch.empty((n_docs, H), dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            check(lib().ltr_dropout_keep_mask_p(int(seed) & _MASK64, layer, n_docs, H, float(p), _ptr(out), _stream()),
                  "ltr_dropout_keep_mask_p")
        return out
    return _dropout_keep_mask_half(seed, layer, n_docs, H, device)


def _dropout_keep_mask_half(seed, layer, n_docs, H, device):
    out = torch.empty((n_docs, H), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        check(lib().ltr_dropout_keep_mask(int(seed) & _MASK64, layer, n_docs, H, _ptr(out), _stream()),
              "ltr_dropout_keep_mask")
    return out


class FusedRanker:
    """One-launch training pass for a DoubleLayerNet / TripleLayerNet with a listwise loss.

        ranker = FusedRanker(net, loss="approxNDCG")
        loss = ranker.step(X, y)          # X [B,S,F] fp32 device, y [B,S]; fills p.grad for every parameter
        opt.step()

    Equivalent to `loss = lossfn(net(X, None, None).squeeze(-1), y); loss.backward()` of the reference loop
    (main_batch_execution.py:128-170), with X read from HBM once and nothing but the loss written back.
    `flat_grad` is a single fp32 buffer that every `param.grad` aliases: one all-reduce for data parallel.
    `grad_div` rescales the mean-type losses for a global batch (B_global = B * world_size).
    """

    LOSSES = {"approxNDCG": LOSS_APPROXNDCG, "listnet": LOSS_LISTNET, "lambdaLoss": LOSS_LAMBDA}

    def __init__(self, module, loss="approxNDCG", alpha=1.0, eps=1e-10, padded_value_indicator=-1,
                 apply_sigmoid=False, grid=None, weighing_scheme=None, k=None, sigma=1.0, mu=10.0,
                 reduction="sum", reduction_log="binary"):
        if loss not in self.LOSSES:
            raise KeyError(f"fused loss must be one of {sorted(self.LOSSES)}, got {loss!r}")
        self.module = module
        if module._ltr_net == NET_WIDE:
            raise NotImplementedError("the fused step is built for scorers of up to 136 input features; wider networks run as "
                                      "net(x, None, None) + the loss + backward() (ltr_mi355x.scorer.wide_forward)")
        self.info = NetInfo.get(module._ltr_net)
        self.net = self.info.net           # the compiled network id the kernels run
        self.loss = loss
        self.loss_kind = self.LOSSES[loss]
        self.alpha, self.eps, self.pad = float(alpha), float(eps), float(padded_value_indicator)
        self.apply_sigmoid = bool(apply_sigmoid)
        if self.loss_kind == LOSS_LAMBDA:
            from .functional import _lambda_args
            if reduction not in ("sum", "mean"):
                raise ValueError("Reduction method can be either sum or mean")
            self.lambda_args = _lambda_args(eps, padded_value_indicator, weighing_scheme, k, sigma, mu, reduction_log)
        self.reduction = reduction
        self.params = module._ltr_params()
        require_device(*self.params)
        dev = self.params[0].device
        self.device = dev
        self.grid = int(grid) if grid else int(lib().ltr_fused_grid(self.net, cu_count(dev)))     # persistent workgroups (1 or 2 per CU)
        # one flat fp32 buffer [all parameter gradients | loss | normaliser]: the ONLY thing data parallel all-reduces.
        # `flat` = [grads | loss] (what the optimizer and callers read); `flat_ext` adds the normaliser slot of the
        # deferred-normalisation protocol (step(defer_norm=True) -> all-reduce(flat_ext) -> finish_norm()): the local
        # batch size for the batch-mean loss (approxNDCG.py:53), the local kept-pair count for lambdaLoss
        # reduction="mean" (lambdaL.py:88-89) -- so the GLOBAL mean needs neither a second collective nor a host sync.
        self.flat_ext = torch.zeros(self.info.n_params + 2, dtype=torch.float32, device=dev)
        self.flat = self.flat_ext[:self.info.n_params + 1]
        self.flat_grad = self.flat[:self.info.n_params]
        self._norm = self.flat_ext[self.info.n_params + 1:]          # 1-element view
        self._grad_views = []
        off = 0
        for p in self.params:           # every p.grad is a view into the flat buffer
            self._grad_views.append(self.flat_grad[off:off + p.numel()].view_as(p))
            off += p.numel()
        self._bind_grads()
        self.packed = torch.empty(self.info.packed_floats, dtype=torch.float32, device=dev)
        self.partials = torch.empty(self.grid * self.info.partial_floats, dtype=torch.float32, device=dev)
        # TripleLayerNet on the 136-feature collection runs FOLDED (triple_folds above): the one-launch step on the two-layer
        # kernel with two document-split copies of the 32 units (NET_TRIPLE_FOLDED), the three-launch path on the plain
        # 136 -> 32 -> 1 network (NET_TRIPLE_FOLDED_32)
        self.fold = self.fold32 = None
        if triple_folds(self.info.handle):
            fi32 = NetInfo.get(folded_net32(self.info.handle))
            fi = NetInfo.get(NET_TRIPLE_FOLDED) if self.info.handle == NET_TRIPLE else fi32      # 64 features: the plain form everywhere
            self.fold, self.fold32 = fi, fi32
            self.fold_grid = int(grid) if grid else int(lib().ltr_fused_grid(fi.net, cu_count(dev)))
            self.fold_w = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in ((64, self.info.F), (64,), (1, 64))]
            self.fold_packed = torch.empty(max(fi.packed_floats, fi32.packed_floats), dtype=torch.float32, device=dev)
            self.fold_partials = torch.empty(max(self.fold_grid * fi.partial_floats, self.grid * fi32.partial_floats), dtype=torch.float32,
                                             device=dev)
            self.fold_flat = torch.empty(fi.n_params, dtype=torch.float32, device=dev)
        self._loss_out = self.flat[self.info.n_params]
        self._slate = None
        self._acts = None              # saved hidden activations of the three-launch path (grown on demand)
        self._calls = 0
        self.seed_salt = 0             # per-rank salt of the dropout stream (data parallel)
        self.kernel_events = None      # optional (start, end) torch.cuda.Event pair bracketing the pipeline kernel

    def _bind_grads(self):
        """Make every `param.grad` the view of the flat buffer again.  `opt.zero_grad()` / `net.zero_grad()` default
        to set_to_none=True (they drop the aliasing) and the reference loop calls zero_grad between the loss and
        backward (main_batch_execution.py:167): re-binding at the end of every step keeps `opt.step()` correct
        whichever side of `ranker.step` the caller zeroes on.  (zero_grad(set_to_none=False) AFTER step() would
        wipe the gradients just computed -- like zeroing after backward() in any torch loop.)"""
        for p, v in zip(self.params, self._grad_views):
            if p.grad is not v:
                p.grad = v

    @property
    def mean_kind(self):
        """How the loss normalises: "batch" (approxNDCG: mean over slates), "pairs" (lambdaLoss reduction="mean": mean
        over kept pairs -- data dependent), None (plain sums: ListNet, lambdaLoss "sum")."""
        if self.loss_kind == LOSS_APPROXNDCG:
            return "batch"
        if self.loss_kind == LOSS_LAMBDA and self.reduction == "mean":
            return "pairs"
        return None

    def finish_norm(self):
        """Second half of the deferred-normalisation protocol: divide [grads | loss] by the (all-reduced) normaliser.
        Device ops only.  0 / 0 = nan is the reference's "mean of nothing"."""
        if self.mean_kind is not None:
            self._divide_by_norm()
        self._bind_grads()
        return self._loss_out

    def _divide_by_norm(self):
        """[grads | loss] /= normaliser, as device ops (no host read).  An EMPTY mean -- no kept pair anywhere (every slate with
        uniform labels, all ranks empty under data parallel) -- leaves a nan LOSS and ZERO gradients, like the reference
        (lambdaL.py:88-89: torch.mean of an empty selection is nan, its autograd gradient is zeros): the loss slot is divided by
        the raw count (0 / 0 = nan), the gradient slice by the count with 0 replaced by 1 (the sum-form gradients are exactly
        zero then) -- never 0 / 0 into a tensor the optimizer steps on."""
        n = self._norm
        self.flat_grad.div_(torch.where(n > 0, n, torch.ones_like(n)))
        self.flat[self.info.n_params:].div_(n)

    def step(self, X, y, world_batch=None, keep1=None, keep2=None, seed=None, train=None, defer_norm=False):
        """Run the fused pass on this rank's slates.  Returns the 0-dim LOCAL loss contribution, already
        scaled for the global batch (sum over ranks == the reference's loss on the global batch).
        defer_norm=True: leave SUM-form contributions in `flat` and this rank's normaliser in `flat_ext[-1]`; the caller
        all-reduces `flat_ext` and calls `finish_norm()` (ltr_mi355x.dp.QueryShardedTrainer)."""
        info = self.info
        require_device(X, y)
        if X.dim() != 3 or X.shape[2] != info.F or tuple(y.shape[:2]) != tuple(X.shape[:2]):
            raise ValueError(f"expected X [B,S,{info.F}] and y [B,S], got {tuple(X.shape)} / {tuple(y.shape)}")
        B, S = int(X.shape[0]), int(X.shape[1])
        if S < 1 or S > 2048:
            raise ValueError(f"slate_length {S} outside the supported range 1..2048")
        lambda_mean = self.loss_kind == LOSS_LAMBDA and self.reduction == "mean"
        if lambda_mean and not defer_norm and world_batch not in (None, B):
            raise ValueError('lambdaLoss reduction="mean" divides by the GLOBAL kept-pair count, which no rank knows before '
                             "the all-reduce: under data parallel call step(defer_norm=True) (QueryShardedTrainer does)")
        # one launch when the slate tiles a 128-document super-tile; otherwise forward launch + loss kernel +
        # backward launch -- same flat gradient buffer either way
        one_launch = S in (32, 64, 128)       # (and the reference's dropout probability: see below)
        if B == 0:
            # no slates on this rank: zero gradient contribution; the loss of an empty batch is what the
            # reference's reduction gives (mean of nothing = nan, sum of nothing = 0) unless a global batch is set
            self.flat_ext.zero_()
            if self.loss_kind == LOSS_APPROXNDCG and not world_batch and not defer_norm:
                self.flat[self.info.n_params] = float("nan")
            self._bind_grads()
            return self._loss_out
        if self.loss_kind == LOSS_LAMBDA and self.lambda_args[1] < 0:
            # k = 0: `ndcg_at_k_mask[:0, :0]` keeps no pair (lambdaL.py:29-30) -> loss 0 ("sum") / nan ("mean" of
            # nothing), zero gradient, for every slate length
            self.flat_ext.zero_()
            if lambda_mean and not defer_norm:
                self.flat[self.info.n_params] = float("nan")
            self._bind_grads()
            return self._loss_out
        gb = int(world_batch) if world_batch else B
        # mean over the batch (approxNDCG.py:53) vs sum (listnet.py:16); deferred: sums now, one division after the all-reduce
        scale = 1.0 / gb if (self.loss_kind == LOSS_APPROXNDCG and not defer_norm) else 1.0
        if defer_norm and self.loss_kind == LOSS_APPROXNDCG:
            self._norm.fill_(float(B))
        count = None
        train = self.module.training if train is None else train
        dropout = drop_code(bool(train and self.module._ltr_dropout), getattr(getattr(self.module, "dropout", None), "p", 0.5))
        if seed is None:
            seed = next_seed(self._calls) ^ ((self.seed_salt * 0xA24BAED4963EE407) & _MASK64)
        self._calls += 1
        with torch.cuda.device(self.device):
            x2 = _docs(X, info)
            yy = y.detach().reshape(B, S).to(torch.float32).contiguous()
            k1, k2 = _mask(keep1, B * S, info.H1, info.cH1), _mask(keep2, B * S, info.H2, info.cH2)
            if self._slate is None or self._slate.numel() < B:
                self._slate = torch.empty(B, dtype=torch.float32, device=self.device)
            h = lib()
            three = not one_launch or (dropout > 1 and k1 is None)     # p != 0.5: only the forward kernels carry that stream
            fold = (self.fold32 if three else self.fold) if (k1 is None and k2 is None) else None
            net, packed, partials, grid = self.net, self.packed, self.partials, self.grid
            pf = None
            if fold is None:
                pack_params(self.info.handle, self.params, out=self.packed)
            else:
                pf = _params_f32(self.params)                       # W1, b1, W2, b2, w3, b3
                copies = fold.H1 // 32
                fw = triple_fold(pf, copies, [self.fold_w[0][:fold.H1], self.fold_w[1][:fold.H1], self.fold_w[2][:, :fold.H1]])
                packed = self.fold_packed[:fold.packed_floats]
                pack_params(fold.handle, fw + [pf[5]], out=packed)
                net, partials, grid = fold.net, self.fold_partials, (self.grid if three else self.fold_grid)
            if three:
                out = self._step_three_launches(h, x2, yy, B, S, dropout, int(seed) & _MASK64, k1, k2, scale, lambda_mean,
                                                defer_norm, fold, pf, net, packed, partials)
                self._bind_grads()
                return out
            if self.kernel_events is not None:
                self.kernel_events[0].record()
            if self.loss_kind == LOSS_LAMBDA:
                sid, kk, sigma, mu, eps, pad, lb = self.lambda_args
                if lambda_mean:
                    count = torch.empty(B, dtype=torch.float32, device=self.device)
                check(h.ltr_fused_step_lambda(net, _ptr(x2), _ptr(yy), B, S, _ptr(packed), int(dropout),
                                              int(seed) & _MASK64, _ptr(k1), _ptr(k2), sid, kk, sigma, mu, eps, pad,
                                              lb, 1.0, _ptr(self._slate), _ptr(count), _ptr(partials), grid,
                                              _stream()), "ltr_fused_step_lambda")
            else:
                check(h.ltr_fused_step(net, self.loss_kind, _ptr(x2), _ptr(yy), B, S, _ptr(packed),
                                       int(dropout), int(seed) & _MASK64, _ptr(k1), _ptr(k2), self.alpha, self.eps,
                                       self.pad, int(self.apply_sigmoid), scale, _ptr(self._slate), _ptr(partials),
                                       grid, _stream()), "ltr_fused_step")
            if self.kernel_events is not None:
                self.kernel_events[1].record()
            self._reduce(fold, pf, partials, grid)
            check(h.ltr_reduce_sum_f32(_ptr(self._slate), B, scale, self.flat.data_ptr() + 4 * self.info.n_params,
                                       _stream()), "ltr_reduce_sum_f32")
            if lambda_mean:
                # the pair-mean is linear in d loss / d scores: the launch ran in sum form, one division of
                # [grads | loss] by the kept-pair count follows (here, or after the all-reduce when deferred)
                torch.sum(count, dim=0, keepdim=True, out=self._norm)
                if not defer_norm:
                    self._divide_by_norm()
        self._bind_grads()
        return self._loss_out

    def _reduce(self, fold, pf, partials, grid):
        """Per-workgroup partials -> the flat gradient of the module's own tensors (through the unfold for a folded TripleLayerNet)."""
        if fold is None:
            reduce_grads(self.info, partials, grid, self.flat_grad)
        else:
            g2 = self.fold_flat[:fold.n_params]
            reduce_grads(fold, partials, grid, g2)
            triple_unfold(g2, fold.H1 // 32, pf, self.flat_grad)

    def _step_three_launches(self, h, x2, yy, B, S, dropout, seed, k1, k2, scale, lambda_mean, defer_norm=False, fold=None, pf=None,
                             net=None, packed=None, partials=None):
        """Any slate length (and lambdaLoss "mean"): scorer forward launch (writes the scores AND the post-activation
        hidden layers, 1 152 B per document for the 136-wide net) -> loss kernel (forward + dL/dscores) -> scorer backward
        launch that reads h1 / h2 back instead of recomputing fc1 / fc2: ONE forward, like the reference's autograd
        (main_batch_execution.py:128-170).  The backward kernel is bound by the fp32 matrix pipe; the activation round trip
        rides on HBM bandwidth it leaves idle."""
        n = B * S
        dev = self.device
        if net is None:
            net, packed, partials = self.net, self.packed, self.partials
        scores = torch.empty(n, dtype=torch.float32, device=dev)
        ds = torch.empty(n, dtype=torch.float32, device=dev)
        n_acts = int(h.ltr_mlp_acts_floats(net, n))
        if self._acts is None or self._acts.numel() < n_acts:
            self._acts = torch.empty(n_acts, dtype=torch.float32, device=dev)
        check(h.ltr_mlp_forward_save(net, _ptr(x2), n, _ptr(packed), int(dropout), seed, _ptr(k1), _ptr(k2),
                                     _ptr(scores), _ptr(self._acts), self.grid, _stream()), "ltr_mlp_forward_save")
        if self.loss_kind == LOSS_APPROXNDCG:
            check(h.ltr_approxndcg_fwd_bwd(_ptr(scores), _ptr(yy), B, S, self.alpha, self.eps, self.pad, scale,
                                           _ptr(self._slate), _ptr(ds), _stream()), "ltr_approxndcg_fwd_bwd")
        elif self.loss_kind == LOSS_LISTNET:
            check(h.ltr_listnet_fwd_bwd(_ptr(yy), _ptr(scores), B, S, int(self.apply_sigmoid), 1.0, _ptr(self._slate),
                                        _ptr(ds), _stream()), "ltr_listnet_fwd_bwd")
        else:
            sid, kk, sigma, mu, eps, pad, lb = self.lambda_args
            count = torch.empty(B, dtype=torch.float32, device=dev)
            check(h.ltr_lambda_fwd_bwd(_ptr(scores), _ptr(yy), B, S, sid, kk, sigma, mu, eps, pad, lb, 1.0,
                                       _ptr(self._slate), _ptr(count), _ptr(ds), _stream()), "ltr_lambda_fwd_bwd")
        if self.kernel_events is not None:
            self.kernel_events[0].record()
        check(h.ltr_mlp_backward_saved(net, _ptr(x2), n, _ptr(packed), int(dropout), _ptr(self._acts), _ptr(ds),
                                       _ptr(partials), self.grid, _stream()), "ltr_mlp_backward_saved")
        if self.kernel_events is not None:
            self.kernel_events[1].record()
        self._reduce(fold, pf, partials, self.grid)
        check(h.ltr_reduce_sum_f32(_ptr(self._slate), B, scale, self.flat.data_ptr() + 4 * self.info.n_params,
                                   _stream()), "ltr_reduce_sum_f32")
        if lambda_mean:
            torch.sum(count, dim=0, keepdim=True, out=self._norm)
            if not defer_norm:
                self._divide_by_norm()
        return self._loss_out
