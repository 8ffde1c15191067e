"""Host side of the risk-sensitive losses: autograd Functions over the HIP kernels of
csrc/ltr_risk.hip (zRisk / geoRisk / tRisk reductions on a [queries x systems] matrix) and the pair-matrix
column sums of csrc/ltr_losses.hip (ltr_lambda_colsum_*), used by losses/riskLosses/*.py.
Device tensors only; no CPU fallback."""
import torch

from ._lib import check, lib
from .functional import _f32, _lambda_args, _ptr, _stream, require_device

RISK_Z, RISK_GEO = 0, 1


class RiskEval(torch.autograd.Function):
    """zRisk / geoRisk of column `col` of mat [Q, n_systems] (riskFunctions.py:4-33) -> ltr_risk_fwd_bwd.
    Forward value and d value / d mat come out of one launch."""

    @staticmethod
    def forward(ctx, mat, alpha, col, kind):
        if mat.dim() != 2:
            raise ValueError(f"risk functions take a [queries, systems] matrix, got {tuple(mat.shape)}")
        Q, n = mat.shape
        ctx.in_dtype = mat.dtype
        with torch.cuda.device(mat.device):
            m = _f32(mat)
            out = torch.empty(1, dtype=torch.float32, device=mat.device)
            dmat = torch.empty_like(m) if ctx.needs_input_grad[0] else None
            check(lib().ltr_risk_fwd_bwd(_ptr(m), Q, n, int(col), float(alpha), int(kind), _ptr(out), _ptr(dmat), _stream()),
                  "ltr_risk_fwd_bwd")
        ctx.save_for_backward(dmat)
        return out

    @staticmethod
    def backward(ctx, go):
        (dmat,) = ctx.saved_tensors
        return (dmat * go.to(torch.float32)).to(ctx.in_dtype), None, None, None


class TRisk(torch.autograd.Function):
    """mean / std of the alpha-weighted per-query deltas (riskLosses.py:278-291) -> ltr_trisk_fwd_bwd."""

    @staticmethod
    def forward(ctx, model, baseline, alpha):
        if model.dim() != 1 or model.shape != baseline.shape:
            raise ValueError(f"tRisk takes two [queries] vectors, got {tuple(model.shape)} / {tuple(baseline.shape)}")
        ctx.dtypes = (model.dtype, baseline.dtype)
        with torch.cuda.device(model.device):
            a, b = _f32(model), _f32(baseline)
            out = torch.empty(1, dtype=torch.float32, device=a.device)
            da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
            db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
            check(lib().ltr_trisk_fwd_bwd(_ptr(a), _ptr(b), a.numel(), float(alpha), _ptr(out), _ptr(da), _ptr(db), _stream()),
                  "ltr_trisk_fwd_bwd")
        ctx.save_for_backward(da, db)
        return out

    @staticmethod
    def backward(ctx, go):
        da, db = ctx.saved_tensors
        g = go.to(torch.float32)
        return (None if da is None else (da * g).to(ctx.dtypes[0]), None if db is None else (db * g).to(ctx.dtypes[1]), None)


class LambdaColsum(torch.autograd.Function):
    """torch.sum(lambdaMask(y_pred, y_true, ..., return_losses=True), dim=1) without the [B,S,S] tensor
    (riskLosses.py:72-83) -> ltr_lambda_colsum_{fwd,bwd}.  Result [B, S] indexed by predicted rank."""

    @staticmethod
    def forward(ctx, y_pred, y_true, eps, pad, scheme, k, sigma, mu, reduction_log):
        args = _lambda_args(eps, pad, scheme, k, sigma, mu, reduction_log)
        if y_pred.dim() != 2 or y_pred.shape != y_true.shape:
            raise ValueError(f"expected y_pred / y_true [batch_size, slate_length], got {tuple(y_pred.shape)} / {tuple(y_true.shape)}")
        B, S = y_pred.shape
        ctx.in_dtype = y_pred.dtype
        largs = (args[0], max(args[1], 0)) + args[2:]
        with torch.cuda.device(y_pred.device):
            s, y = _f32(y_pred), _f32(y_true)
            out = torch.empty((B, S), dtype=torch.float32, device=s.device)
            if B > 0:
                check(lib().ltr_lambda_colsum_fwd(_ptr(s), _ptr(y), B, S, *largs, _ptr(out), _stream()), "ltr_lambda_colsum_fwd")
        ctx.save_for_backward(s, y)
        ctx.largs = largs
        return out.to(torch.result_type(y_pred, y_true))

    @staticmethod
    def backward(ctx, g):
        s, y = ctx.saved_tensors
        B, S = s.shape
        ds = torch.zeros_like(s)
        if B > 0:
            with torch.cuda.device(s.device):
                gg = g.to(torch.float32).contiguous()
                check(lib().ltr_lambda_colsum_bwd(_ptr(s), _ptr(y), B, S, *ctx.largs, _ptr(gg), _ptr(ds), _stream()),
                      "ltr_lambda_colsum_bwd")
        return (ds.to(ctx.in_dtype),) + (None,) * 8


class LambdaColsumSys(torch.autograd.Function):
    """The lambdaMask column sums of EVERY system of a Lambda-type risk loss in ONE launch (riskLosses.py:63-83, :183-203):
    [model, baselines..., ideal ranking] x [B, S], with the slate softmaxes of :65-70 inside the kernel -> ltr_lambda_colsum_sys_fwd.
    Inputs are the RAW y_predicted / y_true [B, S] and y_baselines [B, S, n] (or None); only y_predicted carries a gradient
    (ltr_lambda_colsum_sys_bwd: pair backward + softmax Jacobian in one launch)."""

    @staticmethod
    def forward(ctx, y_pred, y_true, y_base, eps, pad, scheme, k, sigma, mu, reduction_log):
        args = _lambda_args(eps, pad, scheme, k, sigma, mu, reduction_log)
        B, S = y_pred.shape
        nb = 0 if y_base is None else int(y_base.shape[2])
        largs = (args[0], max(args[1], 0)) + args[2:]
        with torch.cuda.device(y_pred.device):
            s, y = _f32(y_pred), _f32(y_true)
            b = None if y_base is None else _f32(y_base)
            out = torch.empty((nb + 2, B, S), dtype=torch.float32, device=s.device)
            check(lib().ltr_lambda_colsum_sys_fwd(_ptr(s), _ptr(y), _ptr(b), B, S, nb, *largs, _ptr(out), _stream()),
                  "ltr_lambda_colsum_sys_fwd")
        ctx.save_for_backward(s, y)
        ctx.largs = largs
        ctx.in_dtype = y_pred.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        s, y = ctx.saved_tensors
        B, S = s.shape
        ds = torch.empty_like(s)
        with torch.cuda.device(s.device):
            g0 = g[0].to(torch.float32).contiguous()
            check(lib().ltr_lambda_colsum_sys_bwd(_ptr(s), _ptr(y), B, S, *ctx.largs, _ptr(g0), _ptr(ds), _stream()),
                  "ltr_lambda_colsum_sys_bwd")
        return (ds.to(ctx.in_dtype),) + (None,) * 9


class LambdaRiskLoss(torch.autograd.Function):
    """A whole Lambda-type geoRisk / zRisk loss as ONE autograd node (riskLosses.py:63-125, :183-244): forward = three launches
    (every system's column sums with the slate softmaxes inside -> every system's effectiveness + d mat[:, 0] / d colsum ->
    flip + risks + return strategy + d value / d mat), backward = one multiply and ONE launch (pair backward + softmax Jacobian).
    The chain of separate nodes it replaces cost ~13 launches and as many Python-level autograd hops per step."""

    @staticmethod
    def forward(ctx, y_pred, y_true, y_base, scheme, lt, ideal, ones_col, kind, alpha, strategy, flip, factor):
        args = _lambda_args(1e-10, -1, scheme, None, 1., 10., "binary")
        largs = (args[0], max(args[1], 0)) + args[2:]
        B, S = y_pred.shape
        nb = 0 if y_base is None else int(y_base.shape[2])
        dev = y_pred.device
        h = lib()
        with torch.cuda.device(dev):
            s, y = _f32(y_pred), _f32(y_true)
            b = None if y_base is None else _f32(y_base)
            cs = torch.empty((nb + 2, B, S), dtype=torch.float32, device=dev)
            check(h.ltr_lambda_colsum_sys_fwd(_ptr(s), _ptr(y), _ptr(b), B, S, nb, *largs, _ptr(cs), _stream()), "ltr_lambda_colsum_sys_fwd")
            nsys = 1 + nb + (1 if ideal else 0)
            mat = torch.empty((B, nsys + (1 if ones_col else 0)), dtype=torch.float32, device=dev)
            m0 = mat if not ones_col else torch.empty((B, nsys), dtype=torch.float32, device=dev)
            jac = torch.empty((B, S), dtype=torch.float32, device=dev)
            check(h.ltr_risk_matrix_fwd(_ptr(cs[nb + 1]), _ptr(cs[0]), _ptr(cs[1:nb + 1]) if nb else None, B, S, nb, 1, int(lt), int(bool(ideal)),
                                        _ptr(m0), _ptr(jac), _stream()), "ltr_risk_matrix_fwd")
            if ones_col:
                mat[:, :nsys] = m0
                mat[:, nsys] = 1.0                                   # the ideal ranking's cosine with itself (:106)
            out = torch.empty(1, dtype=torch.float32, device=dev)
            dmat = torch.empty_like(mat)
            check(h.ltr_risk_tail_fwd_bwd(_ptr(mat), B, mat.shape[1], float(alpha), int(kind), int(strategy), int(bool(flip)), float(factor), 0,
                                          _ptr(out), _ptr(dmat), _stream()), "ltr_risk_tail_fwd_bwd")
            g0 = jac.mul_(dmat[:, 0:1])                              # d value / d colsum[model]
        ctx.save_for_backward(s, y, g0)
        ctx.largs = largs
        ctx.in_dtype = y_pred.dtype
        return out

    @staticmethod
    def backward(ctx, go):
        s, y, g0 = ctx.saved_tensors
        B, S = s.shape
        ds = torch.empty_like(s)
        with torch.cuda.device(s.device):
            g = g0 * go.to(torch.float32)
            check(lib().ltr_lambda_colsum_sys_bwd(_ptr(s), _ptr(y), B, S, *ctx.largs, _ptr(g), _ptr(ds), _stream()), "ltr_lambda_colsum_sys_bwd")
        return (ds.to(ctx.in_dtype),) + (None,) * 11


def lambda_risk_loss(y_pred, y_true, y_base, scheme, lt, ideal, ones_col, kind, alpha, strategy, flip, factor):
    require_device(y_pred, y_true)
    return LambdaRiskLoss.apply(y_pred, y_true, y_base, scheme, lt, ideal, ones_col, kind, alpha, strategy, flip, factor)


def lambda_colsum_systems(y_pred, y_true, y_base, weighing_scheme, eps=1e-10, padded_value_indicator=-1, k=None, sigma=1., mu=10.,
                          reduction_log="binary"):
    require_device(y_pred, y_true)
    return LambdaColsumSys.apply(y_pred, y_true, y_base, eps, padded_value_indicator, weighing_scheme, k, sigma, mu, reduction_log)


class RiskMatrix(torch.autograd.Function):
    """The [queries, systems] effectiveness matrix of the risk-sensitive losses in ONE launch (riskLosses.py:8-49, :63-117) ->
    ltr_risk_matrix_fwd: column 0 the model (`x0`, the only input that carries a gradient), then the baselines (`rest`), optionally
    the ideal ranking.  mode 0: raw labels / scores / baselines [B, S, n], soft-maxed inside; mode 1: lambdaMask column sums, rest
    [n, B, S].  The forward also writes d mat[:, 0] / d x0, so the backward is one multiply."""

    @staticmethod
    def forward(ctx, ref, x0, rest, mode, lt, ideal):
        B, S = x0.shape
        nr = 0 if rest is None else (rest.shape[0] if mode == 1 else rest.shape[2])
        with torch.cuda.device(x0.device):
            r, x = _f32(ref), _f32(x0)
            rs = None if rest is None else _f32(rest)
            mat = torch.empty((B, 1 + nr + (1 if ideal else 0)), dtype=torch.float32, device=x0.device)
            jac = torch.empty((B, S), dtype=torch.float32, device=x0.device) if ctx.needs_input_grad[1] else None
            check(lib().ltr_risk_matrix_fwd(_ptr(r), _ptr(x), _ptr(rs), B, S, nr, int(mode), int(lt), int(bool(ideal)), _ptr(mat), _ptr(jac),
                                            _stream()), "ltr_risk_matrix_fwd")
        ctx.save_for_backward(jac)
        ctx.in_dtype = x0.dtype
        return mat

    @staticmethod
    def backward(ctx, g):
        (jac,) = ctx.saved_tensors
        return None, (jac * g[:, 0:1].to(torch.float32)).to(ctx.in_dtype), None, None, None, None


class RiskTail(torch.autograd.Function):
    """Flip + risk of column 0 (and of the last column) + return strategy + `negative` factor of a geoRisk / zRisk loss in one
    launch (riskLosses.py:47-60, :118-125) -> ltr_risk_tail_fwd_bwd; value [1] and d value / d mat from the same launch."""

    @staticmethod
    def forward(ctx, mat, alpha, kind, strategy, flip, factor, zquirk):
        Q, n = mat.shape
        with torch.cuda.device(mat.device):
            m = _f32(mat)
            out = torch.empty(1, dtype=torch.float32, device=mat.device)
            dmat = torch.empty_like(m) if ctx.needs_input_grad[0] else None
            check(lib().ltr_risk_tail_fwd_bwd(_ptr(m), Q, n, float(alpha), int(kind), int(strategy), int(bool(flip)), float(factor),
                                              int(bool(zquirk)), _ptr(out), _ptr(dmat), _stream()), "ltr_risk_tail_fwd_bwd")
        ctx.save_for_backward(dmat)
        return out

    @staticmethod
    def backward(ctx, go):
        (dmat,) = ctx.saved_tensors
        return (dmat * go.to(torch.float32),) + (None,) * 6


class TRiskTail(torch.autograd.Function):
    """Flip + alpha-weighted deltas + mean / std + `negative` factor of a tRisk loss on its [queries, 2] matrix in one launch
    (riskLosses.py:269-291, :332-345) -> ltr_trisk_tail_fwd_bwd."""

    @staticmethod
    def forward(ctx, mat, alpha, flip, factor):
        with torch.cuda.device(mat.device):
            m = _f32(mat)
            out = torch.empty(1, dtype=torch.float32, device=mat.device)
            dmat = torch.empty_like(m) if ctx.needs_input_grad[0] else None
            check(lib().ltr_trisk_tail_fwd_bwd(_ptr(m), m.shape[0], float(alpha), int(bool(flip)), float(factor), _ptr(out), _ptr(dmat),
                                               _stream()), "ltr_trisk_tail_fwd_bwd")
        ctx.save_for_backward(dmat)
        return out

    @staticmethod
    def backward(ctx, go):
        (dmat,) = ctx.saved_tensors
        return (dmat * go.to(torch.float32),) + (None,) * 3


def trisk_tail(mat, alpha, flip, factor):
    require_device(mat)
    return TRiskTail.apply(mat, alpha, flip, factor)


def risk_tail(mat, alpha, kind, strategy, flip, factor, zquirk=False):
    require_device(mat)
    return RiskTail.apply(mat, alpha, kind, strategy, flip, factor, zquirk)


def risk_matrix(ref, x0, rest, mode, lt, ideal):
    require_device(ref, x0)
    return RiskMatrix.apply(ref, x0, rest, mode, lt, ideal)


def z_risk(mat, alpha, i=0):
    require_device(mat)
    return RiskEval.apply(mat, alpha, i, RISK_Z)


def geo_risk(mat, alpha, i=0):
    require_device(mat)
    return RiskEval.apply(mat, alpha, i, RISK_GEO)


def lambda_colsum(y_pred, y_true, weighing_scheme, eps=1e-10, padded_value_indicator=-1, k=None, sigma=1., mu=10.,
                  reduction_log="binary"):
    require_device(y_pred, y_true)
    return LambdaColsum.apply(y_pred, y_true, eps, padded_value_indicator, weighing_scheme, k, sigma, mu, reduction_log)
