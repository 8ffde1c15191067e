"""MI355X drop-in for losses/ordinal.py of the reference (with_ordinals :9-24, ordinalLoss :27-53)."""
import torch

from ltr_mi355x.functional import Ordinal, require_device

PADDED_Y_VALUE = -1


def with_ordinals(y, n, padded_value_indicator=PADDED_Y_VALUE):
    """Labels [batch_size, slate_length] -> cumulative ordinal targets [batch_size, slate_length, n]:
    target_k = 1[y >= k], k = 1..n; documents whose label equals the indicator get the indicator in all n
    slots.  Stays on y's device (the reference builds `one_to_n` on the CPU only, ordinal.py:19)."""
    one_to_n = torch.arange(start=1, end=n + 1, dtype=torch.float, device=y.device)
    rep = y.unsqueeze(2).repeat(1, 1, n)
    ordinals = (rep >= one_to_n).type(torch.float)
    ordinals[rep == padded_value_indicator] = padded_value_indicator
    return ordinals


def ordinalLoss(y_pred, y_true, n, padded_value_indicator=PADDED_Y_VALUE):
    """Ordinal regression loss: BCE (each log clamped at -100 as torch's BCELoss does) between
    y_pred[B,S,n] probabilities and with_ordinals(y_true, n), masked entries zeroed, summed and divided by
    the number of documents with at least one unmasked target.  As in the reference, the targets are
    always built with the default indicator (-1) and `padded_value_indicator` only selects the mask.
    One elementwise HIP launch + a fixed-order reduction.  Device tensors only."""
    require_device(y_pred, y_true)
    return Ordinal.apply(y_pred, y_true, n, padded_value_indicator)
