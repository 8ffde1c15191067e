"""MI355X drop-in for losses/approxNDCG.py of the reference (approxNDCGLoss, :7-53)."""
from ltr_mi355x.functional import ApproxNDCG, require_device, slate_2d


def approxNDCGLoss(y_pred, y_true, eps=1e-10, padded_value_indicator=-1, alpha=1.):
    """ApproxNDCG listwise loss ("A General Approximation Framework for Direct Optimization of Information
    Retrieval Measures"), no truncation.  Same signature and semantics as the reference:

    :param y_pred: model scores, [batch_size, slate_length] (a trailing singleton dim is squeezed)
    :param y_true: relevance labels, same shape; `padded_value_indicator` marks padded documents
    :param eps: clamp for the pairwise sigmoids and the ideal DCG
    :param alpha: sigmoid temperature on score differences
    :return: 0-dim loss tensor (-mean over slates of approximate NDCG), differentiable w.r.t. y_pred

    Inputs are never modified.  Runs one HIP workgroup-slice per slate (scores/labels in LDS); forward
    and the analytic gradient are produced by one launch.  Device tensors only.
    """
    y_pred, y_true = slate_2d(y_pred, "y_pred"), slate_2d(y_true, "y_true")
    require_device(y_pred, y_true)
    return ApproxNDCG.apply(y_pred, y_true, eps, padded_value_indicator, alpha)
