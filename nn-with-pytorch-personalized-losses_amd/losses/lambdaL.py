"""MI355X drop-in for losses/lambdaL.py of the reference (lambdaMask :7-64, lambdaLoss :67-93 and the
seven *_scheme names of :96-127, which the reference dispatches by string, :46)."""
import torch

from ltr_mi355x.functional import SCHEME_IDS, LambdaLoss, LambdaPairs, require_device, slate_2d


def lambdaMask(y_pred, y_true, eps=1e-10, padded_value_indicator=-1, weighing_scheme=None, k=None, sigma=1., mu=10.,
               reduction="sum", reduction_log="binary", return_losses=False):
    """Pairwise LambdaLoss terms log_b(clamp(clamp(sigmoid(sigma*(s_i-s_j)), eps) ** w_ij, eps)).

    return_losses=True : the full [B,S,S] matrix indexed by PREDICTED RANK (row = rank of the first doc).
    return_losses=False: the 1-D tensor of the kept pairs only (both docs real, y_i > y_j unless
    weighing_scheme == "ndcgLoss1_scheme", both ranks < k), in row-major (b, rank_i, rank_j) order.
    """
    y_pred, y_true = slate_2d(y_pred, "y_pred"), slate_2d(y_true, "y_true")
    require_device(y_pred, y_true)
    losses, keep = LambdaPairs.apply(y_pred, y_true, eps, padded_value_indicator, weighing_scheme, k, sigma, mu,
                                     reduction_log)
    if return_losses:
        return losses
    return losses[keep.bool()]


def lambdaLoss(y_pred, y_true, eps=1e-10, padded_value_indicator=-1, weighing_scheme=None, k=None, sigma=1., mu=10.,
               reduction="sum", reduction_log="binary"):
    """LambdaLoss framework ("The LambdaLoss Framework for Ranking Metric Optimization").

    :param weighing_scheme: None or one of the *_scheme names below (KeyError otherwise)
    :param k: rank truncation (None = whole slate)
    :param sigma: score-difference weight inside the sigmoid;  :param mu: NDCGLoss2++ mixing weight
    :param reduction: "sum" | "mean" (ValueError otherwise); :param reduction_log: "binary" | "natural"
    :return: 0-dim loss tensor, differentiable w.r.t. y_pred.  One fused HIP launch (no [B,S,S] tensors).
    """
    y_pred, y_true = slate_2d(y_pred, "y_pred"), slate_2d(y_true, "y_true")
    require_device(y_pred, y_true)
    return LambdaLoss.apply(y_pred, y_true, eps, padded_value_indicator, weighing_scheme, k, sigma, mu, reduction,
                            reduction_log)


# The scheme functions are part of the module surface (string-dispatched in the reference).  They take the
# rank-ordered G [B,S], D [1,S] and return pair weights, as device tensor expressions; the fused kernels do
# not call them (the weights are computed in registers), they exist for callers that import them.
def ndcgLoss1_scheme(G, D, *args):
    return (G / D)[:, :, None]


def ndcgLoss2_scheme(G, D, *args):
    S = G.shape[1]
    r = torch.arange(S, device=G.device)
    m = (r[:, None] - r[None, :]).abs()
    inv = 1.0 / D[0]
    delta = (inv[(m - 1).clamp(min=0)] - inv[m]).abs() * (m > 0)
    return delta[None] * (G[:, :, None] - G[:, None, :]).abs()


def lamdbaRank_scheme(G, D, *args):
    inv = 1.0 / D
    return (inv[:, :, None] - inv[:, None, :]).abs() * (G[:, :, None] - G[:, None, :]).abs()


def ndcgLoss2PP_scheme(G, D, *args):
    return args[0] * ndcgLoss2_scheme(G, D) + lamdbaRank_scheme(G, D)


def rankNet_scheme(G, D, *args):
    return 1.


def rankNetWeightedByGTDiff_scheme(G, D, *args):
    return (args[1][:, :, None] - args[1][:, None, :]).abs()


def rankNetWeightedByGTDiffPowed_scheme(G, D, *args):
    return (args[1][:, :, None] ** 2 - args[1][:, None, :] ** 2).abs()


assert all(name is None or name in globals() for name in SCHEME_IDS)
