"""CPU ORACLE (test infrastructure, NOT product code) for the evaluation metrics -- SURVEY.md row f-4.

Vectorised numpy (fp64) restatement of what the reference computes with per-query Python loops in
    utils/metrics.py:48-80   dcg / ndcg / mNdcg      (NDCG@k, linear | exponential gains, no_relevant rule)
    utils/metrics.py:8-45    getGeoRiskDefault       (GeoRisk of every system of a [queries x systems] matrix)
Sort-free like the kernels: a document's rank is the number of documents that beat it.  Pinned against the
imported reference by tests/golden/make_golden_r2.py (fixtures tests/golden/metrics.npz).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np


def ranks_desc(v, stable=True):
    """0-based rank of every entry of each row under a descending sort.
    stable=True : ties keep index order -- Python's sorted(..., reverse=True) (metrics.py:51, the default path).
    stable=False: ties in REVERSE index order -- np.argsort(v)[::-1] when argsort is stable (metrics.py:53); numpy's
                  default sort kind is not guaranteed stable, so that path is only pinned on tie-free data."""
    v = np.asarray(v, dtype=np.float64)
    gt = v[:, None, :] > v[:, :, None]                       # [q, i, j]: j beats i
    idx = np.arange(v.shape[1])
    tie = (v[:, None, :] == v[:, :, None]) & ((idx[None, None, :] < idx[None, :, None]) if stable
                                              else (idx[None, None, :] > idx[None, :, None]))
    return (gt | tie).sum(axis=2)


def dcg_at_k(y_true, order_key, k, gains, stable=True):
    y = np.asarray(y_true, dtype=np.float64)
    r = ranks_desc(order_key, stable)
    k = min(k, y.shape[1])                                   # metrics.py:54-55
    if gains == "exponential":
        g = 2.0 ** y - 1.0
    elif gains == "linear":
        g = y
    else:
        raise ValueError("Invalid gains option.")
    return np.where(r < k, g / np.log2(r + 2.0), 0.0).sum(axis=1)


def ndcg_per_query(y_true, y_score, k=5, no_relevant=True, gains="linear", stable=True):
    """metrics.py:67-78: per-query NDCG@k; a query with ideal DCG 0 scores 1.0 (no_relevant) or 0.0."""
    d = dcg_at_k(y_true, y_score, k, gains, stable)
    ideal = dcg_at_k(y_true, y_true, k, gains, stable)
    out = np.where(ideal == 0.0, 1.0 if no_relevant else 0.0, d / np.where(ideal == 0.0, 1.0, ideal))
    return out


def geo_risk_all_systems(mat, alpha):
    """metrics.py:8-45: GeoRisk of every column.  e == 0 contributes 0; negative residuals weigh (1 + alpha)."""
    m = np.asarray(mat, dtype=np.float64)
    Q = m.shape[0]
    S, T = m.sum(axis=0), m.sum(axis=1)
    e = S[None, :] * (T[:, None] / T.sum())
    x = m - e
    z = np.where(e != 0.0, x / np.sqrt(np.where(e != 0.0, e, 1.0)), 0.0)
    z = np.where(x < 0.0, (1.0 + alpha) * z, z).sum(axis=0)
    ncd = np.array([0.5 * math.erfc(-(v / Q) / math.sqrt(2.0)) for v in z])
    return np.sqrt((S / Q) * ncd)
