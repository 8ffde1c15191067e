"""MI355X drop-in for losses/listnet.py of the reference (listnetLoss, :5-16)."""
from ltr_mi355x.functional import ListNet, require_device, slate_2d


def listnetLoss(y_true, y_predicted, apply_sigmoid=False):
    """ListNet top-one loss: -sum_{b,i} softmax(y_true)_i * log softmax(y_predicted)_i, softmax over dim 1,
    SUMMED over batch and slate (no mean, no padding handling, like the reference).  NOTE the argument
    order (y_true first).  apply_sigmoid=True gives the reference's -sum sigmoid(p * log q) variant.
    Device tensors only; [B,S] or [B,S,1]."""
    y_true, y_predicted = slate_2d(y_true, "y_true"), slate_2d(y_predicted, "y_predicted")
    require_device(y_predicted, y_true)
    return ListNet.apply(y_true, y_predicted, apply_sigmoid)
