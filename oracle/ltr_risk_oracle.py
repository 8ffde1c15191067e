"""CPU ORACLE (test infrastructure, NOT product code) for the risk-sensitive losses -- SURVEY.md row f-1.

Restates, in plain torch CPU ops with autograd, what the reference computes in
    losses/riskLosses/riskFunctions.py:4-33   (zRisk, geoRisk)
    losses/riskLosses/riskLosses.py:8-345     (geoRisk/zRisk/tRisk x Listnet/Lambda losses)
plus closed-form gradients of the two risk functions (what csrc/ltr_risk.hip implements).  Pinned against the
imported reference by tests/golden/make_golden_r2.py (fixtures tests/golden/risk.npz), including the
reference's only known-answer test, geoRisk(5x8 matrix, alpha=3) = 0.31438308416523303
(tests/georiskTorchTest.py:5-12).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.
"""
import math

import torch

import ltr_oracle as O


# ------------------------------------------------------------------------------------------ risk functions
def z_risk(mat, alpha, i=0):
    """riskFunctions.py:4-22.  mat [Q, n]; column i under test.  0-dim."""
    col = mat[:, i]
    s_i = col.sum()
    t = mat.sum(dim=1)
    e = s_i * (t / t.sum())                       # expected effectiveness of system i on each query (:10)
    d = (col - e) / torch.sqrt(e)
    w = 1.0 + float(alpha) * (d < 0).to(mat.dtype)   # the boolean mask carries no gradient (:15-18)
    return (d * w.detach()).sum()


def geo_risk(mat, alpha, i=0):
    """riskFunctions.py:25-33.  Shape [1] (the reference broadcasts against Normal(tensor([0.]), tensor([1.])))."""
    Q = mat.shape[0]
    v = z_risk(mat, alpha, i) / Q
    phi_cdf = 0.5 * (1.0 + torch.erf(v / math.sqrt(2.0)))
    return torch.sqrt((mat[:, i].sum() / Q) * phi_cdf).reshape(1)


def risk_closed_form(mat, alpha, i, geo):
    """Value and d value / d mat without autograd (the formulas of csrc/ltr_risk.hip), in mat's dtype."""
    Q, n = mat.shape
    if i < 0:
        i += n
    x = mat[:, i]
    s_i, t = x.sum(), mat.sum(dim=1)
    N = t.sum()
    e = s_i * (t / N)
    d = (x - e) / torch.sqrt(e)
    c = 1.0 + float(alpha) * (d < 0).to(mat.dtype)
    Z = (c * d).sum()
    A = c * (-0.5 * (x + e) / (e * torch.sqrt(e)))
    T1 = (A * t).sum() / N
    gz = (A * s_i / N - s_i * T1 / N)[:, None].expand(Q, n).clone()
    gz[:, i] += c / torch.sqrt(e) + T1
    if not geo:
        return Z, gz
    v = Z / Q
    Phi = 0.5 * (1.0 + torch.erf(v / math.sqrt(2.0)))
    phi = torch.exp(-0.5 * v * v) / math.sqrt(2.0 * math.pi)
    M = s_i / Q
    val = torch.sqrt(M * Phi)
    g = (0.5 / val * M * phi / Q) * gz
    g[:, i] += 0.5 / val * Phi / Q
    return val.reshape(1), g


def t_risk_tail(model, baseline, alpha):
    """riskLosses.py:278-289: alpha-weighted per-query deltas, mean over unbiased std."""
    delta = model - baseline
    delta = delta * (1.0 + float(alpha) * (model < baseline).to(model.dtype))
    return delta.mean() / delta.std()


# --------------------------------------------------------------------------------------- effectiveness matrices
def _softmaxes(y_pred, y_true, y_base):
    sq = torch.squeeze
    return (sq(torch.softmax(y_true, dim=1)), sq(torch.softmax(y_pred, dim=1)),
            None if y_base is None else sq(torch.softmax(y_base, dim=1)))


def _cos(a, b):
    return torch.nn.functional.cosine_similarity(a, b, dim=1)


def pair_colsum(p, p_true, scheme):
    """torch.sum(lambdaMask(p, p_true, weighing_scheme=scheme, return_losses=True), dim=1): [B,S] by predicted rank."""
    full, _ = O.lambda_pairs(p, p_true, weighing_scheme=scheme)
    return full.sum(dim=1)


def listnet_matrix(p_true, p_pred, p_base, lt, add_ideal):
    """riskLosses.py:16-49 (= :136-169).  Systems: model, baselines..., [ideal]."""
    systems = [p_pred] + ([] if p_base is None else [p_base[:, :, j] for j in range(p_base.shape[2])])
    if add_ideal == 2:
        systems.append(p_true)
    if lt == 1:
        cols = [((p_true * p - p_true * p_true) ** 2).sum(dim=1) for p in systems]
    elif lt == 2:
        cols = [_cos(p_true, p) for p in systems]
    else:
        ref = (p_true * p_true).sum(dim=1)
        cols = [((p_true * p).sum(dim=1) - ref) ** 2 for p in systems]
    mat = torch.stack(cols, dim=1)
    return mat.max() - mat if lt in (1, 3) else mat


def lambda_matrix(p_true, p_pred, p_base, lt, add_ideal, scheme, ideal_ones):
    """riskLosses.py:71-117 (geo) / :191-236 (z)."""
    tt = pair_colsum(p_true, p_true, scheme)
    systems = [pair_colsum(p_pred, p_true, scheme)]
    if p_base is not None:
        systems += [pair_colsum(p_base[:, :, j], p_true, scheme) for j in range(p_base.shape[2])]
    if lt == 1:
        cols = [((c - tt) ** 2).sum(dim=1) for c in systems]
        if add_ideal == 2:
            cols.append(torch.zeros_like(cols[0]))
    else:
        cols = [_cos(tt, c) for c in systems]
        if add_ideal == 2:
            cols.append(torch.ones(tt.shape[0], dtype=torch.float) if ideal_ones else _cos(tt, tt))
    mat = torch.stack(cols, dim=1)
    return mat.max() - mat if lt == 1 else mat


def _strategy(fn, mat, alpha, rs):
    if rs == 1:
        return fn(mat, alpha)
    gap = fn(mat, alpha, -1) - fn(mat, alpha)
    return gap if rs == 2 else gap ** 2


def geo_risk_listnet(y_pred, y_true, y_base=None, alpha=5, lt=1, rs=1, negative=1, add_ideal=1):
    pt, pp, pb = _softmaxes(y_pred, y_true, y_base)
    return negative * _strategy(geo_risk, listnet_matrix(pt, pp, pb, lt, add_ideal), alpha, rs).reshape(1)


def geo_risk_lambda(y_pred, y_true, y_base=None, alpha=5, lt=1, rs=1, negative=1, add_ideal=1, scheme="ndcgLoss2PP_scheme"):
    pt, pp, pb = _softmaxes(y_pred, y_true, y_base)
    return negative * _strategy(geo_risk, lambda_matrix(pt, pp, pb, lt, add_ideal, scheme, True), alpha, rs).reshape(1)


def z_risk_listnet(y_pred, y_true, y_base=None, alpha=5, lt=1, rs=1, negative=1, add_ideal=1):
    pt, pp, pb = _softmaxes(y_pred, y_true, y_base)
    mat = listnet_matrix(pt, pp, pb, lt, add_ideal)
    if rs == 2:          # the reference's precedence: only the first term is multiplied by `negative` (:176)
        return (negative * z_risk(mat, alpha, -1) - z_risk(mat, alpha)).reshape(1)
    return (negative * _strategy(z_risk, mat, alpha, rs)).reshape(1)


def z_risk_lambda(y_pred, y_true, y_base=None, alpha=5, lt=1, rs=1, negative=1, add_ideal=1, scheme="ndcgLoss2PP_scheme"):
    pt, pp, pb = _softmaxes(y_pred, y_true, y_base)
    return (negative * _strategy(z_risk, lambda_matrix(pt, pp, pb, lt, add_ideal, scheme, False), alpha, rs)).reshape(1)


def _t_cols(q_true, q_pred, q_base, lt):
    if lt == 1:
        m = torch.stack([((q_pred - q_true) ** 2).sum(dim=1), ((q_base - q_true) ** 2).sum(dim=1)])
        return m.max() - m
    if lt == 2:
        return torch.stack([_cos(q_true, q_pred), _cos(q_true, q_base)])
    ref = q_true.sum(dim=1)
    return torch.stack([(q_pred.sum(dim=1) - ref) ** 2, (q_base.sum(dim=1) - ref) ** 2])


def t_risk_listnet(y_pred, y_true, y_base, alpha=5, lt=1, negative=1):
    pt, pp, pb = _softmaxes(y_pred, y_true, y_base)
    m = _t_cols(pt * pt, pt * pp, pt * pb, lt)
    return (negative * t_risk_tail(m[0], m[1], alpha)).reshape(1)


def t_risk_lambda(y_pred, y_true, y_base, alpha=5, lt=1, negative=1, scheme="ndcgLoss2PP_scheme"):
    pt, pp, pb = _softmaxes(y_pred, y_true, y_base)
    m = _t_cols(pair_colsum(pt, pt, scheme), pair_colsum(pp, pt, scheme), pair_colsum(pb, pt, scheme), lt)
    return (negative * t_risk_tail(m[0], m[1], alpha)).reshape(1)
