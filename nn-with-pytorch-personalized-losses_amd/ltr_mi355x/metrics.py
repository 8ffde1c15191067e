"""Host side of the on-device evaluation metrics: NDCG@k per query (csrc/ltr_metrics.hip) and
GeoRisk of every system of a [queries x systems] matrix (csrc/ltr_risk.hip with the numpy metric's zero guard).
Inputs may be device tensors (no copy) or host arrays / lists (moved to the current device first); results stay on
the device unless the reference-shaped wrappers in utils/metrics.py convert them."""
import numpy as np
import torch

from ._lib import LtrDeviceError, check, lib
from .functional import _ptr, _stream

GAINS = {"linear": 0, "exponential": 1}
RISK_GEO, RISK_ZERO_GUARD = 1, 4


def _device():
    if not torch.cuda.is_available():
        raise LtrDeviceError("ltr_mi355x metrics run on the MI355X only (HIP kernels); no ROCm device is visible and "
                             "there is no CPU fallback.")
    return torch.device("cuda", torch.cuda.current_device())


def to_device_f32(a, like=None):
    """Device fp32 contiguous view/copy of a tensor, ndarray or (nested) list."""
    if torch.is_tensor(a):
        if not a.is_cuda:
            a = a.to(like.device if like is not None and like.is_cuda else _device())
        return a.detach().to(torch.float32).contiguous()
    return torch.as_tensor(np.asarray(a, dtype=np.float32), device=like.device if like is not None and like.is_cuda else _device())


def ndcg_at_k(y_true, y_score, k=5, no_relevant=True, gains="linear", reverse_ties=False, want="ndcg"):
    """Per-query NDCG@k (want="ndcg") or un-normalised DCG@k (want="dcg"), [Q] fp64 device tensor
    (utils/metrics.py:48-80).  y_true, y_score: [Q, S].  Labels and scores are RANKED IN FP32 (the reference's numpy path ranks
    whatever dtype it is given, float64 included): float64 scores that differ below fp32 resolution become ties here and are then
    ordered by index.  The accumulation and the result are fp64.  Parity unpinned for such inputs: the reference's fixtures and
    this repo's goldens only hold fp32-representable values."""
    if gains not in GAINS:
        raise ValueError("Invalid gains option.")                                  # metrics.py:62
    s = to_device_f32(y_score)
    y = to_device_f32(y_true, like=s)
    if y.dim() == 1:
        y, s = y[None, :], s.reshape(1, -1)
    if s.dim() == 3 and s.shape[2] == 1:
        s = s[:, :, 0].contiguous()
    if y.shape != s.shape or y.dim() != 2:
        raise ValueError(f"expected y_true / y_score [queries, docs], got {tuple(y.shape)} / {tuple(s.shape)}")
    Q, S = y.shape
    out = torch.empty(Q, dtype=torch.float64, device=s.device)
    with torch.cuda.device(s.device):
        check(lib().ltr_ndcg_at_k(_ptr(y), _ptr(s), Q, S, int(k) if k is not None else S, GAINS[gains], int(bool(no_relevant)),
                                  int(bool(reverse_ties)), _ptr(out) if want == "ndcg" else None,
                                  _ptr(out) if want == "dcg" else None, _stream()), "ltr_ndcg_at_k")
    return out


def geo_risk_all_systems(mat, alpha):
    """GeoRisk of EVERY column of mat [Q, n] (utils/metrics.py:8-45), [n] fp32 device tensor: one launch per system."""
    m = to_device_f32(mat)
    if m.dim() != 2:
        raise ValueError(f"expected a [queries, systems] matrix, got {tuple(m.shape)}")
    Q, n = m.shape
    out = torch.empty(n, dtype=torch.float32, device=m.device)
    with torch.cuda.device(m.device):
        for i in range(n):
            check(lib().ltr_risk_fwd_bwd(_ptr(m), Q, n, i, float(alpha), RISK_GEO | RISK_ZERO_GUARD,
                                         out.data_ptr() + 4 * i, None, _stream()), "ltr_risk_fwd_bwd")
    return out
