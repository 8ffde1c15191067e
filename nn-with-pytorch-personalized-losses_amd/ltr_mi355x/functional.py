"""torch.autograd.Function wrappers around the HIP loss kernels (C ABI: include/ltr_mi355x.h).

Every Function runs forward AND the analytic backward in its forward launch (the kernels compute
dL/dscores in the same pass that computes the loss), saves the gradient, and scales it by the incoming
grad_output in backward.  Inputs must be ROCm device tensors; there is no CPU fallback.
"""
import math

import torch

from ._lib import LtrDeviceError, check, lib

SCHEME_IDS = {
    None: 0,
    "ndcgLoss1_scheme": 1,
    "ndcgLoss2_scheme": 2,
    "lamdbaRank_scheme": 3,
    "ndcgLoss2PP_scheme": 4,
    "rankNet_scheme": 5,
    "rankNetWeightedByGTDiff_scheme": 6,
    "rankNetWeightedByGTDiffPowed_scheme": 7,
}
MAX_SLATE = 2048


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def require_device(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise LtrDeviceError(
                "ltr_mi355x ops run on the MI355X only (HIP kernels); got a CPU tensor and there is no CPU "
                "fallback.  Move inputs to the device first.")
    dev = tensors[0].device
    for t in tensors[1:]:
        if t.device != dev:
            raise LtrDeviceError(f"tensors on different devices: {dev} vs {t.device}")


def slate_2d(t, name):
    """[B,S] or [B,S,1] -> [B,S] (the reference callers squeeze, main_batch_execution.py:130-132)."""
    if t.dim() == 3 and t.shape[2] == 1:
        t = t[:, :, 0]
    if t.dim() != 2:
        raise ValueError(f"{name} must have shape [batch_size, slate_length], got {tuple(t.shape)}")
    return t


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _check_slates(y_pred, y_true):
    if y_pred.shape != y_true.shape:
        raise ValueError(f"y_pred {tuple(y_pred.shape)} and y_true {tuple(y_true.shape)} differ in shape")
    B, S = y_pred.shape
    if S < 1 or S > MAX_SLATE:
        raise ValueError(f"slate_length {S} outside the supported range 1..{MAX_SLATE}")
    return B, S


def _reduce(vec, scale):
    out = torch.empty((), dtype=torch.float32, device=vec.device)
    check(lib().ltr_reduce_sum_f32(_ptr(vec), vec.numel(), float(scale), _ptr(out), _stream()), "ltr_reduce_sum_f32")
    return out


class ApproxNDCG(torch.autograd.Function):
    """losses/approxNDCG.py:7-53 -> ltr_approxndcg_fwd_bwd + ltr_reduce_sum_f32 (mean over slates)."""

    @staticmethod
    def forward(ctx, y_pred, y_true, eps, pad, alpha):
        B, S = _check_slates(y_pred, y_true)
        out_dtype = torch.result_type(y_pred, y_true)
        ctx.in_dtype = y_pred.dtype
        if B == 0:
            ctx.save_for_backward(torch.zeros_like(y_pred))
            return torch.full((), float("nan"), dtype=out_dtype, device=y_pred.device)
        with torch.cuda.device(y_pred.device):
            s, y = _f32(y_pred), _f32(y_true)
            slate = torch.empty(B, dtype=torch.float32, device=s.device)
            ds = torch.empty_like(s) if ctx.needs_input_grad[0] else None
            check(lib().ltr_approxndcg_fwd_bwd(_ptr(s), _ptr(y), B, S, float(alpha), float(eps), float(pad), 1.0 / B,
                                               _ptr(slate), _ptr(ds), _stream()), "ltr_approxndcg_fwd_bwd")
            loss = _reduce(slate, 1.0 / B)
        ctx.save_for_backward(ds)
        return loss.to(out_dtype)

    @staticmethod
    def backward(ctx, go):
        (ds,) = ctx.saved_tensors
        return (ds * go.to(torch.float32)).to(ctx.in_dtype), None, None, None, None


class ListNet(torch.autograd.Function):
    """losses/listnet.py:5-16 -> ltr_listnet_fwd_bwd; loss is the SUM over batch and slate."""

    @staticmethod
    def forward(ctx, y_true, y_pred, apply_sigmoid):
        B, S = _check_slates(y_pred, y_true)
        out_dtype = torch.result_type(y_pred, y_true)
        ctx.in_dtype = y_pred.dtype
        if B == 0:
            ctx.save_for_backward(torch.zeros_like(y_pred))
            return torch.zeros((), dtype=out_dtype, device=y_pred.device)
        with torch.cuda.device(y_pred.device):
            s, y = _f32(y_pred), _f32(y_true)
            slate = torch.empty(B, dtype=torch.float32, device=s.device)
            ds = torch.empty_like(s) if ctx.needs_input_grad[1] else None
            check(lib().ltr_listnet_fwd_bwd(_ptr(y), _ptr(s), B, S, int(bool(apply_sigmoid)), 1.0, _ptr(slate),
                                            _ptr(ds), _stream()), "ltr_listnet_fwd_bwd")
            loss = _reduce(slate, 1.0)
        ctx.save_for_backward(ds)
        return loss.to(out_dtype)

    @staticmethod
    def backward(ctx, go):
        (ds,) = ctx.saved_tensors
        return None, (ds * go.to(torch.float32)).to(ctx.in_dtype), None


def _lambda_args(eps, pad, scheme, k, sigma, mu, reduction_log):
    if reduction_log == "natural":
        lb = 1
    elif reduction_log == "binary":
        lb = 0
    else:
        raise ValueError("Reduction logarithm base can be either natural or binary")   # lambdaL.py:57
    sid = SCHEME_IDS[scheme]        # KeyError for unknown names, like globals()[...] in lambdaL.py:46
    kk = 0 if k is None else int(k)
    if k is not None and kk <= 0:
        kk = -1                     # k=0 keeps nothing in the reference; encoded below as "empty"
    return sid, kk, float(sigma), float(mu), float(eps), float(pad), lb


class LambdaLoss(torch.autograd.Function):
    """losses/lambdaL.py:67-93 -> ltr_lambda_fwd_bwd.  reduction in {"sum", "mean"}."""

    @staticmethod
    def forward(ctx, y_pred, y_true, eps, pad, scheme, k, sigma, mu, reduction, reduction_log):
        sid, kk, sigma, mu, eps, pad, lb = _lambda_args(eps, pad, scheme, k, sigma, mu, reduction_log)
        if reduction not in ("sum", "mean"):
            raise ValueError("Reduction method can be either sum or mean")             # lambdaL.py:91
        B, S = _check_slates(y_pred, y_true)
        out_dtype = torch.result_type(y_pred, y_true)
        ctx.in_dtype = y_pred.dtype
        dev = y_pred.device
        if B == 0 or kk < 0:
            ctx.save_for_backward(torch.zeros_like(y_pred, dtype=torch.float32))
            v = 0.0 if reduction == "sum" else float("nan")
            return torch.full((), v, dtype=out_dtype, device=dev)
        with torch.cuda.device(dev):
            s, y = _f32(y_pred), _f32(y_true)
            slate = torch.empty(B, dtype=torch.float32, device=dev)
            count = torch.empty(B, dtype=torch.float32, device=dev)
            ds = torch.empty_like(s) if ctx.needs_input_grad[0] else None
            check(lib().ltr_lambda_fwd_bwd(_ptr(s), _ptr(y), B, S, sid, kk, sigma, mu, eps, pad, lb, 1.0, _ptr(slate),
                                           _ptr(count), _ptr(ds), _stream()), "ltr_lambda_fwd_bwd")
            loss = _reduce(slate, 1.0)
            if reduction == "mean":
                n = _reduce(count, 1.0)
                loss = loss / n
                if ds is not None:
                    ds = ds / n
        ctx.save_for_backward(ds)
        return loss.to(out_dtype)

    @staticmethod
    def backward(ctx, go):
        (ds,) = ctx.saved_tensors
        return ((ds * go.to(torch.float32)).to(ctx.in_dtype),) + (None,) * 9


class LambdaPairs(torch.autograd.Function):
    """lambdaMask(return_losses=True), losses/lambdaL.py:7-60 -> ltr_lambda_pairs_{fwd,bwd}.
    Returns (losses[B,S,S] in predicted-rank order, keep[B,S,S] uint8); keep is non-differentiable."""

    @staticmethod
    def forward(ctx, y_pred, y_true, eps, pad, scheme, k, sigma, mu, reduction_log):
        args = _lambda_args(eps, pad, scheme, k, sigma, mu, reduction_log)
        sid, kk = args[0], args[1]
        B, S = _check_slates(y_pred, y_true)
        out_dtype = torch.result_type(y_pred, y_true)
        ctx.in_dtype = y_pred.dtype
        dev = y_pred.device
        with torch.cuda.device(dev):
            s, y = _f32(y_pred), _f32(y_true)
            losses = torch.empty((B, S, S), dtype=torch.float32, device=dev)
            keep = torch.empty((B, S, S), dtype=torch.uint8, device=dev)
            if B > 0:
                # k=0 (nothing kept) is served by k=None losses with an all-zero mask
                check(lib().ltr_lambda_pairs_fwd(_ptr(s), _ptr(y), B, S, sid, max(kk, 0), *args[2:], _ptr(losses),
                                                 _ptr(keep), None, _stream()), "ltr_lambda_pairs_fwd")
                if kk < 0:
                    keep.zero_()
        ctx.save_for_backward(s, y)
        ctx.largs = (sid, max(kk, 0)) + args[2:]
        ctx.mark_non_differentiable(keep)
        return losses.to(out_dtype), keep

    @staticmethod
    def backward(ctx, g_losses, _g_keep):
        s, y = ctx.saved_tensors
        B, S = s.shape
        ds = torch.zeros_like(s)
        if B > 0:
            with torch.cuda.device(s.device):
                g = g_losses.to(torch.float32).contiguous()
                check(lib().ltr_lambda_pairs_bwd(_ptr(s), _ptr(y), B, S, *ctx.largs, _ptr(g), _ptr(ds), _stream()),
                      "ltr_lambda_pairs_bwd")
        return (ds.to(ctx.in_dtype),) + (None,) * 8


class Ordinal(torch.autograd.Function):
    """losses/ordinal.py:27-53 -> ltr_ordinal_fwd_bwd."""

    @staticmethod
    def forward(ctx, y_pred, y_true, n, pad):
        if y_pred.dim() != 3 or y_pred.shape[2] != n or tuple(y_pred.shape[:2]) != tuple(y_true.shape):
            raise ValueError(f"ordinalLoss expects y_pred [B,S,{n}] and y_true [B,S], got "
                             f"{tuple(y_pred.shape)} / {tuple(y_true.shape)}")
        ctx.in_dtype = y_pred.dtype
        dev = y_pred.device
        n_docs = y_true.numel()
        with torch.cuda.device(dev):
            p, y = _f32(y_pred), _f32(y_true)
            nb = int(lib().ltr_ordinal_num_blocks(n_docs))
            partials = torch.empty(max(2 * nb, 2), dtype=torch.float32, device=dev)
            sums = torch.empty(2, dtype=torch.float32, device=dev)
            dp = torch.empty_like(p) if ctx.needs_input_grad[0] else None
            check(lib().ltr_ordinal_fwd_bwd(_ptr(p), _ptr(y), n_docs, int(n), float(pad), _ptr(partials), _ptr(sums),
                                            _ptr(dp), _stream()), "ltr_ordinal_fwd_bwd")
            loss = sums[0] / sums[1]
            if dp is not None:
                dp = dp / sums[1]
        ctx.save_for_backward(dp)
        return loss.to(y_pred.dtype)

    @staticmethod
    def backward(ctx, go):
        (dp,) = ctx.saved_tensors
        return (dp * go.to(torch.float32)).to(ctx.in_dtype), None, None, None


LOG2_E = 1.0 / math.log(2.0)
