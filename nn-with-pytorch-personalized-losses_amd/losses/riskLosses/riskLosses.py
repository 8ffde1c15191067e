"""MI355X drop-in for losses/riskLosses/riskLosses.py of the reference: the six risk-sensitive listwise losses
(geoRisk / zRisk / tRisk  x  Listnet / Lambda, :8, :63, :128, :183, :247, :294) with the reference's signatures,
defaults, shapes and quirks.

Each loss builds a [queries, systems] effectiveness matrix -- column 0 the model, then the baseline rankers,
optionally the ideal ranking -- from row-softmaxed labels / predictions / baseline scores, and hands it to a risk
function.  On the MI355X: the pair-matrix column sums of the Lambda variants come from `ltr_lambda_colsum_*`
(no [B,S,S] tensor is ever written), the risk reductions from `ltr_risk_fwd_bwd` / `ltr_trisk_fwd_bwd`
(forward + analytic gradient per launch), and -- for the regular shapes of the geoRisk / zRisk losses -- the whole
[queries, systems] matrix from ONE launch (`ltr_risk_matrix_fwd`: softmaxes, transformation, every system, and the
gradient of the model's column); the tRisk pair takes its [queries, 2] matrix from the same kernel.  Irregular shapes (a
squeezed single baseline, B = 1, fp64 inputs, ...) walk the tensor-algebra path, which reproduces the reference's squeeze
quirks literally.  Device tensors only.
"""
import torch
import torch.nn.functional as F

from losses.riskLosses.riskFunctions import geoRisk, zRisk
from ltr_mi355x import risk as _risk
from ltr_mi355x.functional import require_device


def _probs(y_predicted, y_true, y_baselines):
    require_device(y_predicted, y_true)
    pb = None if y_baselines is None else torch.squeeze(F.softmax(y_baselines, dim=1))     # :10-11
    return torch.squeeze(F.softmax(y_true, dim=1)), torch.squeeze(F.softmax(y_predicted, dim=1)), pb


def _cos(a, b):
    return torch.nn.CosineSimilarity(dim=1)(a, b)


def _unbound():
    # the reference leaves `mat` unassigned for transformations it does not implement (e.g. :66-86 has no
    # listnet_transformation == 3 branch) and dies on its first use; same exception type and message
    raise UnboundLocalError("local variable 'mat' referenced before assignment")


def _systems(p_pred, p_base, p_ideal):
    """[n_systems, B, S]: the model, the baseline rankers, optionally the ideal ranking -- ONE tensor, so that the effectiveness
    transformation below is one set of launches for all systems instead of one set per system."""
    parts = [p_pred.unsqueeze(0)]
    if p_base is not None:
        p_base.shape[2]                         # a single baseline was squeezed away by _probs: IndexError, like the reference (:26)
        parts.append(p_base.permute(2, 0, 1))
    if p_ideal is not None:
        parts.append(p_ideal.unsqueeze(0))
    return parts[0] if len(parts) == 1 else torch.cat(parts, 0)


def _effectiveness(ref, sys3, lt):
    """Per (system, query) effectiveness against `ref` [B, S] (:16-49, :71-117): 1 = squared differences summed over the slate,
    2 = cosine similarity, 3 = squared difference of the dot products."""
    if lt == 1:
        return torch.sum((sys3 - ref) ** 2, dim=2)
    if lt == 2:
        return F.cosine_similarity(ref.unsqueeze(0).expand_as(sys3), sys3, dim=2)
    return (torch.sum(sys3, dim=2) - torch.sum(ref, dim=1)) ** 2


_FUSED_MAX_SLATE = 2048


def _regular(y_predicted, y_true, y_baselines):
    """The shapes the one-launch matrix kernel takes (anything else walks the tensor-algebra path below, which reproduces the
    reference's squeeze quirks literally): fp32 [B, S] or [B, S, 1] scores / labels with B, S > 1, baselines None or [B, S, n >= 2]."""
    def two_d(t):
        if t.dim() == 3 and t.shape[2] == 1:
            t = t[:, :, 0]
        ok = t.dim() == 2 and t.shape[0] > 1 and 1 < t.shape[1] <= _FUSED_MAX_SLATE and t.dtype == torch.float32
        return t if ok else None
    yp, yt = two_d(y_predicted), two_d(y_true)
    if yp is None or yt is None or yp.shape != yt.shape:
        return None
    yb = y_baselines
    if yb is not None and (yb.dim() != 3 or tuple(yb.shape[:2]) != tuple(yp.shape) or yb.shape[2] < 2 or yb.shape[2] > 64
                           or yb.dtype != torch.float32):
        return None
    return yp, yt, yb


def _flip(mat, lt):
    # larger distance = worse: flip so that larger = better (:47-49)
    return -mat + torch.max(mat) if lt in (1, 3) else mat


def _listnet_mat_fused(y_predicted, y_true, y_baselines, lt, add_ideal):
    """Queries x systems matrix of the Listnet-type losses in one launch (softmaxes, transformation, every system), BEFORE the
    "larger = better" flip of :47-49, and whether that flip applies (the risk tail does it); None when the shapes are not the
    regular ones."""
    require_device(y_predicted, y_true)
    reg = _regular(y_predicted, y_true, y_baselines)
    if reg is None:
        return None
    if lt not in (1, 2, 3):
        _unbound()
    yp, yt, yb = reg
    return _risk.risk_matrix(yt, yp, yb, 0, lt, add_ideal == 2), lt in (1, 3)       # (matrix BEFORE the flip, whether it is flipped)


def _listnet_mat(p_true, p_pred, p_base, lt, add_ideal):
    """Queries x systems matrix of the Listnet-type losses (:16-49, :136-169)."""
    if lt not in (1, 2, 3):
        _unbound()
    sys3 = _systems(p_pred, p_base, p_true if add_ideal == 2 else None)
    if lt == 2:
        mat = _effectiveness(p_true, sys3, 2)
    else:                        # 1: (p_true p - p_true^2)^2 summed over the slate; 3: (sum p_true p - sum p_true^2)^2
        mat = _effectiveness(p_true * p_true, p_true * sys3, lt)
    mat = mat.t()
    if lt == 1 or lt == 3:       # larger distance = worse: flip so that larger = better (:47-49)
        mat = -mat + torch.max(mat)
    return mat


def _lambda_mat(p_true, p_pred, p_base, lt, add_ideal, scheme, ideal_is_ones):
    """Queries x systems matrix of the Lambda-type losses (:71-117, :191-236): effectiveness from the column sums of
    lambdaMask(p, p_true, weighing_scheme, return_losses=True).  Three colsum launches whatever the number of systems: the ideal
    ranking, the model (the one that carries a gradient), and ALL baselines as one [n_base * B, S] batch."""
    if lt not in (1, 2):
        _unbound()
    B, S = p_true.shape
    with torch.no_grad():
        tt = _risk.lambda_colsum(p_true, p_true, scheme)
        cb = None
        if p_base is not None:
            nb = p_base.shape[2]                # (IndexError for a single, squeezed baseline: like the reference, :79)
            stacked = p_base.permute(2, 0, 1).reshape(nb * B, S).contiguous()
            cb = _risk.lambda_colsum(stacked, p_true.repeat(nb, 1), scheme).view(nb, B, S)
    cp = _risk.lambda_colsum(p_pred, p_true, scheme)
    ones_col = add_ideal == 2 and lt == 2 and ideal_is_ones
    if tt.dtype == torch.float32 and cp.dtype == torch.float32 and 1 < S <= _FUSED_MAX_SLATE and (cb is None or cb.shape[0] <= 64):
        # one launch for every system's effectiveness (+ d mat[:, 0] / d cp for the backward)
        mat = _risk.risk_matrix(tt, cp, None if cb is None else cb.contiguous(), 1, lt, add_ideal == 2 and not ones_col)
        if ones_col:
            mat = torch.cat([mat, torch.ones((B, 1), dtype=mat.dtype, device=mat.device)], 1)        # :106
        return mat, lt == 1                      # (matrix BEFORE the flip, whether it is flipped: the risk tail does it)
    parts = [cp.unsqueeze(0)] + ([cb] if cb is not None else [])
    if add_ideal == 2 and not ones_col:
        parts.append(tt.unsqueeze(0))
    mat = _effectiveness(tt, parts[0] if len(parts) == 1 else torch.cat(parts, 0), lt)
    if ones_col:
        mat = torch.cat([mat, torch.ones((1, B), dtype=mat.dtype, device=mat.device)], 0)            # :106
    mat = mat.t()
    if lt == 1:
        mat = -mat + torch.max(mat)
    return mat, False


def _lambda_mat_fused(y_predicted, y_true, y_baselines, lt, add_ideal, scheme, ideal_is_ones):
    """The Lambda-type matrix for the regular shapes in TWO launches: every system's lambdaMask column sums (slate softmaxes
    inside) -> ltr_lambda_colsum_sys_fwd, then every system's effectiveness -> ltr_risk_matrix_fwd (mode 1).  None for the shapes
    the tensor-algebra path has to reproduce literally."""
    reg = _regular(y_predicted, y_true, y_baselines)
    if reg is None or lt not in (1, 2):
        return None
    require_device(y_predicted, y_true)
    yp, yt, yb = reg
    cs = _risk.lambda_colsum_systems(yp, yt, yb, scheme)          # [model, baselines..., ideal] x [B, S]
    nb = cs.shape[0] - 2
    ones_col = add_ideal == 2 and lt == 2 and ideal_is_ones
    mat = _risk.risk_matrix(cs[nb + 1], cs[0], cs[1:nb + 1] if nb else None, 1, lt, add_ideal == 2 and not ones_col)
    if ones_col:
        mat = torch.cat([mat, torch.ones((mat.shape[0], 1), dtype=mat.dtype, device=mat.device)], 1)        # :106
    return mat, lt == 1                          # (matrix BEFORE the flip, whether it is flipped: the risk tail does it)


def _lambda_loss_fused(kind, y_predicted, y_true, y_baselines, alpha, lt, return_strategy, negative, add_ideal, scheme, ideal_is_ones):
    """The whole Lambda-type geoRisk / zRisk loss as one autograd node (three launches forward, two backward) for the regular shapes
    and a plain-number `negative`; None otherwise."""
    if return_strategy not in (1, 2, 3) or lt not in (1, 2) or isinstance(negative, torch.Tensor):
        return None
    reg = _regular(y_predicted, y_true, y_baselines)
    if reg is None:
        return None
    yp, yt, yb = reg
    ones_col = add_ideal == 2 and lt == 2 and ideal_is_ones
    return _risk.lambda_risk_loss(yp, yt, yb, scheme, lt, add_ideal == 2 and not ones_col, ones_col, kind, alpha, return_strategy, lt == 1,
                                  float(negative))


def _lambda_matrix(y_predicted, y_true, y_baselines, lt, add_ideal, scheme, ideal_is_ones):
    fm = _lambda_mat_fused(y_predicted, y_true, y_baselines, lt, add_ideal, scheme, ideal_is_ones)
    if fm is not None:
        return fm
    p_true, p_pred, p_base = _probs(y_predicted, y_true, y_baselines)
    return _lambda_mat(p_true, p_pred, p_base, lt, add_ideal, scheme, ideal_is_ones)


def _tail(kind, mat, flip, alpha, return_strategy, negative, zquirk=False):
    """Flip, risk of the model column (and of the last one), return strategy, `negative`: one launch for a regular fp32 matrix."""
    if return_strategy not in (1, 2, 3):
        return None
    if mat.dim() == 2 and mat.dtype == torch.float32 and mat.shape[0] > 0 and not isinstance(negative, torch.Tensor):
        return _risk.risk_tail(mat, alpha, kind, return_strategy, flip, float(negative), zquirk)
    if flip:
        mat = -mat + torch.max(mat)
    risk = geoRisk if kind == _risk.RISK_GEO else zRisk
    factor = _factor(negative, mat)
    if zquirk:
        # the reference's operator precedence (:176-178): `factor` multiplies only the first term of strategy 2 and the squared
        # difference of strategy 3
        if return_strategy == 1:
            return factor * risk(mat, alpha, requires_grad=True)
        if return_strategy == 2:
            return factor * risk(mat, alpha, requires_grad=True, i=-1) - risk(mat, alpha, requires_grad=True)
        return factor * (risk(mat, alpha, requires_grad=True, i=-1) - risk(mat, alpha, requires_grad=True)) ** 2
    return _by_strategy(risk, mat, alpha, return_strategy, factor)


def _factor(negative, like):
    # (a fill kernel, not a host-to-device copy: the whole loss can then be captured in a hipGraph -- ltr_mi355x.graphs)
    return torch.full((1,), float(negative), dtype=torch.float, device=like.device, requires_grad=True)


def _by_strategy(risk, mat, alpha, return_strategy, factor):
    if return_strategy == 1:
        return factor * risk(mat, alpha, requires_grad=True)
    elif return_strategy == 2:
        return factor * (risk(mat, alpha, requires_grad=True, i=-1) - risk(mat, alpha, requires_grad=True))
    elif return_strategy == 3:
        return factor * ((risk(mat, alpha, requires_grad=True, i=-1) - risk(mat, alpha, requires_grad=True)) ** 2)
    return None


def _listnet_matrix(y_predicted, y_true, y_baselines, lt, add_ideal):
    """(matrix, flip still to be applied?) of the Listnet-type losses: one launch for the regular shapes, else the tensor algebra."""
    fm = _listnet_mat_fused(y_predicted, y_true, y_baselines, lt, add_ideal)
    if fm is not None:
        return fm
    p_true, p_pred, p_base = _probs(y_predicted, y_true, y_baselines)
    return _listnet_mat(p_true, p_pred, p_base, lt, add_ideal), False


def geoRiskListnetLoss(y_predicted, y_true, y_baselines=None, alpha=5, listnet_transformation=1, return_strategy=1,
                       negative=1, add_ideal_ranking_to_mat=1):
    mat, flip = _listnet_matrix(y_predicted, y_true, y_baselines, listnet_transformation, add_ideal_ranking_to_mat)
    return _tail(_risk.RISK_GEO, mat, flip, alpha, return_strategy, negative)


def geoRiskLambdaLoss(y_predicted, y_true, y_baselines=None, alpha=5, listnet_transformation=1, return_strategy=1,
                      negative=1, add_ideal_ranking_to_mat=1, weighing_scheme="ndcgLoss2PP_scheme"):
    fused = _lambda_loss_fused(_risk.RISK_GEO, y_predicted, y_true, y_baselines, alpha, listnet_transformation, return_strategy, negative,
                               add_ideal_ranking_to_mat, weighing_scheme, True)
    if fused is not None:
        return fused
    mat, flip = _lambda_matrix(y_predicted, y_true, y_baselines, listnet_transformation, add_ideal_ranking_to_mat, weighing_scheme, True)
    return _tail(_risk.RISK_GEO, mat, flip, alpha, return_strategy, negative)


def zRiskListnetLoss(y_predicted, y_true, y_baselines=None, alpha=5, listnet_transformation=1, return_strategy=1,
                     negative=1, add_ideal_ranking_to_mat=1):
    mat, flip = _listnet_matrix(y_predicted, y_true, y_baselines, listnet_transformation, add_ideal_ranking_to_mat)
    # (zquirk: the reference's operator precedence, :176-178 -- `factor` multiplies only the first term of strategy 2)
    return _tail(_risk.RISK_Z, mat, flip, alpha, return_strategy, negative, zquirk=True)


def zRiskLambdaLoss(y_predicted, y_true, y_baselines=None, alpha=5, listnet_transformation=1, return_strategy=1,
                    negative=1, add_ideal_ranking_to_mat=1, weighing_scheme="ndcgLoss2PP_scheme"):
    fused = _lambda_loss_fused(_risk.RISK_Z, y_predicted, y_true, y_baselines, alpha, listnet_transformation, return_strategy, negative,
                               add_ideal_ranking_to_mat, weighing_scheme, False)
    if fused is not None:
        return fused
    mat, flip = _lambda_matrix(y_predicted, y_true, y_baselines, listnet_transformation, add_ideal_ranking_to_mat, weighing_scheme, False)
    return _tail(_risk.RISK_Z, mat, flip, alpha, return_strategy, negative)


def _trisk_tail(mat, lt, alpha, negative):
    """:269-291: flip (transformation 1 only), alpha-weight the per-query deltas against the baseline, mean / std."""
    if lt == 1:
        mat = torch.sum(mat, dim=2)
        mat = -mat + torch.max(mat)
    return _factor(negative, mat) * _risk.TRisk.apply(mat[0], mat[1], alpha)


def _regular_pair(y_predicted, y_true, y_baseline):
    """tRisk: ONE baseline [B, S] (or [B, S, 1]) next to regular scores / labels."""
    if y_baseline is None or not isinstance(y_baseline, torch.Tensor):
        return None
    yb = y_baseline[:, :, 0] if y_baseline.dim() == 3 and y_baseline.shape[2] == 1 else y_baseline
    reg = _regular(y_predicted, y_true, None)
    if reg is None or yb.dim() != 2 or tuple(yb.shape) != tuple(reg[0].shape) or yb.dtype != torch.float32:
        return None
    return reg[0], reg[1], yb


def _trisk_pair_tail(mat, lt, alpha, negative):
    """[B, 2] (model, baseline) -> flip (transformation 1 only, :269-271) -> tRisk: one launch for a regular fp32 matrix."""
    if mat.dim() == 2 and mat.shape[1] == 2 and mat.shape[0] > 1 and mat.dtype == torch.float32 and not isinstance(negative, torch.Tensor):
        return _risk.trisk_tail(mat, alpha, lt == 1, float(negative))
    if lt == 1:
        mat = -mat + torch.max(mat)
    return _factor(negative, mat) * _risk.TRisk.apply(mat[:, 0], mat[:, 1], alpha)


def tRiskListnetLoss(y_predicted, y_true, y_baselines, alpha=5, listnet_transformation=1, negative=1):
    reg = _regular_pair(y_predicted, y_true, y_baselines)
    if reg is not None and listnet_transformation in (1, 2, 3):
        require_device(y_predicted, y_true)
        yp, yt, yb = reg                          # one launch: softmaxes, transformation, model and baseline columns (+ gradient)
        return _trisk_pair_tail(_risk.risk_matrix(yt, yp, yb.unsqueeze(2).contiguous(), 2, listnet_transformation, False),
                                listnet_transformation, alpha, negative)
    p_true, p_pred, p_base = _probs(y_predicted, y_true, y_baselines)
    q_true, q_pred, q_base = p_true * p_true, p_true * p_pred, p_true * p_base
    if listnet_transformation == 1:
        mat = [(q_pred - q_true) ** 2, (q_base - q_true) ** 2]
    elif listnet_transformation == 2:
        mat = [_cos(q_true, q_pred), _cos(q_true, q_base)]
    elif listnet_transformation == 3:
        t2 = torch.sum(q_true, dim=1)
        mat = [(torch.sum(q_pred, dim=1) - t2) ** 2, (torch.sum(q_base, dim=1) - t2) ** 2]
    else:
        mat = []                       # torch.stack([]) raises, as in the reference (:269)
    return _trisk_tail(torch.stack(mat), listnet_transformation, alpha, negative)


def tRiskLambdaLoss(y_predicted, y_true, y_baselines, alpha=5, listnet_transformation=1, negative=1,
                    weighing_scheme="ndcgLoss2PP_scheme"):
    reg = _regular_pair(y_predicted, y_true, y_baselines)
    if reg is not None and listnet_transformation in (1, 2, 3) and 1 < reg[0].shape[1] <= _FUSED_MAX_SLATE:
        # three launches: column sums of model / baseline / ideal with the slate softmaxes inside, the [B, 2] matrix, flip + tRisk
        require_device(y_predicted, y_true)
        yp, yt, yb = reg
        cs3 = _risk.lambda_colsum_systems(yp, yt, yb.unsqueeze(2).contiguous(), weighing_scheme)
        mat = _risk.risk_matrix(cs3[2], cs3[0], cs3[1:2], 1, listnet_transformation, False)
        return _trisk_pair_tail(mat, listnet_transformation, alpha, negative)
    p_true, p_pred, p_base = _probs(y_predicted, y_true, y_baselines)
    cs = lambda p: _risk.lambda_colsum(p, p_true, weighing_scheme)                          # noqa: E731
    q_true, q_pred, q_base = cs(p_true), cs(p_pred), cs(p_base)
    if (listnet_transformation in (1, 2, 3) and q_pred.dim() == 2 and q_pred.dtype == torch.float32 and q_base.shape == q_pred.shape
            and 1 < q_pred.shape[1] <= _FUSED_MAX_SLATE):
        mat = _risk.risk_matrix(q_true, q_pred, q_base.unsqueeze(0).contiguous(), 1, listnet_transformation, False)
        return _trisk_pair_tail(mat, listnet_transformation, alpha, negative)
    if listnet_transformation == 1:
        mat = [(q_pred - q_true) ** 2, (q_base - q_true) ** 2]
    elif listnet_transformation == 2:
        mat = [_cos(q_true, q_pred), _cos(q_true, q_base)]
    elif listnet_transformation == 3:
        t2 = torch.sum(q_true, dim=1)
        mat = [(torch.sum(q_pred, dim=1) - t2) ** 2, (torch.sum(q_base, dim=1) - t2) ** 2]
    else:
        mat = []
    return _trisk_tail(torch.stack(mat), listnet_transformation, alpha, negative)
