"""MI355X drop-in for utils/metrics.py of the reference: NDCG@k (dcg / ndcg / mNdcg :48-80, torchNdcg :83-104) and
GeoRisk (getGeoRiskDefault :8-45), same names, signatures, defaults and return types -- computed by HIP kernels
(one workgroup per query, rank-by-counting in LDS) instead of per-query Python loops.

Inputs may be what the reference's drivers pass (numpy arrays / nested lists: copied to the device) or device tensors
(no copy).  `mNdcg` returns a list of per-query floats like the reference; `mNdcg_device` returns the [Q] fp64 device
tensor without a host sync.  Ties in the scores: lower index first (the reference's default `use_numpy=False` path,
Python's stable sort); `use_numpy=True` = higher index first, which is what `np.argsort(...)[::-1]` gives under a stable
argsort (numpy's default sort kind does not promise one: the reference is unspecified there).
(`utils` is a namespace package here, as in the reference: modules this directory does not provide --
computeMetrics, runSklearn -- still resolve in the caller's tree.)"""
import numpy as np
import torch

from ltr_mi355x import metrics as _m


def getGeoRiskDefault(mat, alpha):
    """GeoRisk of every system (column) of the [queries x systems] effectiveness matrix; numpy array [systems]."""
    return _m.geo_risk_all_systems(mat, alpha).double().cpu().numpy()


def mNdcg_device(true_relevance, pred_relevance, k=5, no_relevant=True, gains='linear', use_numpy=False):
    return _m.ndcg_at_k(true_relevance, pred_relevance, k=k, no_relevant=no_relevant, gains=gains, reverse_ties=use_numpy)


def mNdcg(true_relevance, pred_relevance, k=5, no_relevant=True, gains='linear', use_numpy=False):
    """One NDCG@k per query, as a list of floats (np.mean(...) of it is what the drivers log)."""
    return mNdcg_device(true_relevance, pred_relevance, k, no_relevant, gains, use_numpy).cpu().tolist()


def ndcg(true_relevance, pred_relevance, k=5, no_relevant=True, gains='linear', use_numpy=False):
    """NDCG@k of ONE query (1-D inputs)."""
    return mNdcg([np.asarray(true_relevance, dtype=np.float64)], [np.asarray(pred_relevance, dtype=np.float64)], k,
                 no_relevant, gains, use_numpy)[0]


def dcg(true_relevance, pred_relevance, k=5, gains='linear', use_numpy=False):
    """DCG@k of ONE query (1-D inputs)."""
    return float(_m.ndcg_at_k([np.asarray(true_relevance, dtype=np.float64)], [np.asarray(pred_relevance, dtype=np.float64)],
                              k=k, gains=gains, reverse_ties=use_numpy, want="dcg")[0])


def torchNdcg(ys_true, ys_pred, k=None, return_type='list'):
    """NDCG with exponential gains per query (k=None: whole slate).  'tensor': NaN (no relevant document) -> 0 like the
    reference (:99-102); 'list': a Python list in which such queries are NaN (the reference's 0/0)."""
    S = ys_true.shape[1]
    r = _m.ndcg_at_k(ys_true, torch.squeeze(ys_pred, -1) if torch.is_tensor(ys_pred) and ys_pred.dim() == 3 else ys_pred,
                     k=S if k is None else k, no_relevant=False, gains="exponential")
    if return_type == 'tensor':
        return r.to(torch.float32)
    ideal_zero = (_m.to_device_f32(ys_true, like=r) <= 0).all(dim=1)
    return torch.where(ideal_zero, torch.full_like(r, float("nan")), r).cpu().tolist()
