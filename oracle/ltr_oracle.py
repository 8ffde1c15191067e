"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.

CPU restatement (PyTorch CPU tensor ops, any float dtype) of the listwise-LTR hot path of
Haiga/nn-with-pytorch-personalized-losses, written from the maths in SURVEY.md section 8(a):

    approx_ndcg            <-> losses/approxNDCG.py:7-53      (approxNDCGLoss)
    listnet                <-> losses/listnet.py:5-16          (listnetLoss)
    lambda_pairs/_loss     <-> losses/lambdaL.py:7-64, 67-93   (lambdaMask / lambdaLoss)
    scheme weights         <-> losses/lambdaL.py:96-127
    with_ordinals/ordinal  <-> losses/ordinal.py:9-24, 27-53
    double_layer_forward   <-> architeture/doubleLayer.py:54-73
    triple_layer_forward   <-> architeture/tripleLayer.py:5-17

Two flavours of every loss live here:
  * an autograd flavour (plain differentiable tensor ops, [B,S,S] intermediates like the reference),
  * a closed-form flavour (`*_closed_form`) returning (loss, dL/dscores) with the analytic gradient the
    HIP kernels implement (rank-by-counting, no sort, no autograd).

Pinned (parity NOT unpinned): `tests/golden/make_golden.py` imports the real reference in the build
container, asserts both flavours equal it on every golden case, and commits the vectors under
`tests/golden/`; `tests/test_oracle_golden.py` re-checks the oracle against those vectors everywhere.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module,
and only as the checker / the timed CPU baseline -- never as a fallback for the HIP path.
"""
import math

import torch

LN2 = math.log(2.0)

SCHEMES = (
    None,
    "ndcgLoss1_scheme",
    "ndcgLoss2_scheme",
    "lamdbaRank_scheme",
    "ndcgLoss2PP_scheme",
    "rankNet_scheme",
    "rankNetWeightedByGTDiff_scheme",
    "rankNetWeightedByGTDiffPowed_scheme",
)


# --------------------------------------------------------------------------------------- helpers
def _discounts(S, device):
    """log2(1 + r), r = 1..S, ALWAYS fp32 (approxNDCG.py:41-42, lambdaL.py:37-38: `pos_idxs.float()`)."""
    return torch.log2(1.0 + torch.arange(1, S + 1, device=device).float())


def rank_desc(v):
    """0-based descending rank by counting, ties broken by index (lower index first).

    rank_i = #{j : v_j > v_i} + #{j < i : v_j == v_i}.  Equals the position of i in a stable
    descending sort; the reference uses torch.sort (approxNDCG.py:27-28, lambdaL.py:18-19) whose
    tie order is unspecified, so ties are outside the parity contract.
    """
    S = v.shape[-1]
    vj = v[:, None, :]
    vi = v[:, :, None]
    before = torch.tril(torch.ones(S, S, dtype=torch.bool, device=v.device), -1)  # [i, j] : j < i
    return ((vj > vi) | ((vj == vi) & before)).sum(-1)


def _gains(y_true, pad):
    """(pad mask, clamped labels, 2^y - 1).  Padded docs get label 0 -> gain 0 (approxNDCG.py:37-38)."""
    padm = y_true == pad
    yc = torch.where(padm, torch.zeros_like(y_true), y_true.clamp(min=0.0))
    return padm, yc, torch.pow(2.0, yc) - 1.0


def _ideal_dcg(y_true, padm, gain, D, eps, k=None):
    """maxDCG: gains in label-sorted order over discounts, first k ranks (lambdaL.py:39), clamp eps."""
    neg_inf = torch.full_like(y_true, float("-inf"))
    r_lab = rank_desc(torch.where(padm, neg_inf, y_true))
    contrib = gain / D[r_lab]
    if k is not None:
        contrib = torch.where(r_lab < k, contrib, torch.zeros_like(contrib))
    return contrib.sum(-1).clamp(min=eps)


# ------------------------------------------------------------------------------------ approxNDCG
def approx_ndcg(y_pred, y_true, eps=1e-10, pad=-1, alpha=1.0):
    """Autograd flavour of approxNDCGLoss (losses/approxNDCG.py:7-53).  Sort-free: the soft-rank sum
    is permutation invariant, only the ideal DCG needs an order (by rank counting)."""
    B, S = y_pred.shape
    dev = y_pred.device
    padm, _, gain = _gains(y_true, pad)
    D = _discounts(S, dev)
    G = gain / _ideal_dcg(y_true, padm, gain, D, eps)[:, None]
    valid = ~padm
    pair = valid[:, :, None] & valid[:, None, :] & ~torch.eye(S, dtype=torch.bool, device=dev)
    diff = y_pred[:, :, None] - y_pred[:, None, :]
    diff = torch.where(pair, diff, torch.zeros_like(diff))
    c = torch.sigmoid(-alpha * diff).clamp(min=eps)
    pos = 1.0 + (pair.float() * c).sum(-1)
    return -(G / torch.log2(1.0 + pos)).sum(-1).mean()


def approx_ndcg_closed_form(y_pred, y_true, eps=1e-10, pad=-1, alpha=1.0):
    """(loss, dL/dy_pred, per-slate -NDCG) with the analytic gradient:
         g_i  = G_i / (log2(1+pos_i)^2 (1+pos_i) ln2) / B
         t_ij = c_ij (1 - c_ij) [c_ij >= eps],   dL/ds_k = alpha * sum_j (g_j t_jk - g_k t_kj)."""
    with torch.no_grad():
        B, S = y_pred.shape
        dev = y_pred.device
        padm, _, gain = _gains(y_true, pad)
        D = _discounts(S, dev)
        G = gain / _ideal_dcg(y_true, padm, gain, D, eps)[:, None]
        valid = ~padm
        pair = valid[:, :, None] & valid[:, None, :] & ~torch.eye(S, dtype=torch.bool, device=dev)
        diff = torch.where(pair, y_pred[:, :, None] - y_pred[:, None, :], torch.zeros(1, dtype=y_pred.dtype))
        raw = torch.sigmoid(-alpha * diff)
        c = raw.clamp(min=eps)
        pm = pair.to(c.dtype)
        pos = 1.0 + (pm * c).sum(-1)
        L = torch.log2(1.0 + pos)
        per_slate = -(G / L).sum(-1)
        g = G / (L * L * (1.0 + pos) * LN2) / B
        t = pm * c * (1.0 - c) * (raw >= eps).to(c.dtype)          # t[k, j]
        grad = alpha * ((t * g[:, :, None]).sum(1) - g * t.sum(2))  # sum_j g_j t_jk - g_k sum_j t_kj
        return per_slate.mean(), grad.to(y_pred.dtype), per_slate


# --------------------------------------------------------------------------------------- ListNet
def listnet(y_true, y_pred, apply_sigmoid=False):
    """Autograd flavour of listnetLoss (losses/listnet.py:5-16).  NOTE argument order (true, pred).
    log(softmax) is taken literally (not log_softmax), summed over batch AND slate."""
    p = torch.softmax(y_true, dim=1)
    q = torch.softmax(y_pred, dim=1)
    if apply_sigmoid:
        return -torch.sigmoid(p * torch.log(q)).sum()
    return -(p * torch.log(q)).sum()


def listnet_closed_form(y_true, y_pred, apply_sigmoid=False):
    """(loss, dL/dy_pred).  Plain: grad = q*sum(p) - p.  Sigmoid variant: w = r(1-r)p, grad = q*sum(w) - w."""
    with torch.no_grad():
        p = torch.softmax(y_true, dim=1)
        q = torch.softmax(y_pred, dim=1)
        lq = torch.log(q)
        if apply_sigmoid:
            r = torch.sigmoid(p * lq)
            w = r * (1.0 - r) * p
            return -r.sum(), q * w.sum(1, keepdim=True) - w
        return -(p * lq).sum(), q * p.sum(1, keepdim=True) - p


# ------------------------------------------------------------------------------------ LambdaLoss
def scheme_weights(scheme, G, D, mu, yc):
    """Pair weights [B,S,S] (or broadcastable) in PRED-RANK order (lambdaL.py:96-127).
    G, yc: [B,S] gathered by predicted rank; D: [S] fp32 discounts log2(2+r)."""
    S = G.shape[1]
    inv = 1.0 / D
    dG = (G[:, :, None] - G[:, None, :]).abs()
    if scheme is None or scheme == "rankNet_scheme":
        return torch.ones((), dtype=G.dtype)
    if scheme == "ndcgLoss1_scheme":
        return (G / D)[:, :, None]

    def ndcg2():
        r = torch.arange(S)
        m = (r[:, None] - r[None, :]).abs()                   # |rank distance|
        # delta_m = |1/D[m-1] - 1/D[m]|, delta_0 := 0 (the reference's D[-1] wrap is zeroed, :104)
        delta = (inv[(m - 1).clamp(min=0)] - inv[m]).abs()
        delta = torch.where(m == 0, torch.zeros_like(delta), delta)
        return delta[None] * dG

    def lrank():
        return (inv[:, None] - inv[None, :]).abs()[None] * dG

    if scheme == "ndcgLoss2_scheme":
        return ndcg2()
    if scheme == "lamdbaRank_scheme":
        return lrank()
    if scheme == "ndcgLoss2PP_scheme":
        return mu * ndcg2() + lrank()
    if scheme == "rankNetWeightedByGTDiff_scheme":
        return (yc[:, :, None] - yc[:, None, :]).abs()
    if scheme == "rankNetWeightedByGTDiffPowed_scheme":
        return (yc[:, :, None] ** 2 - yc[:, None, :] ** 2).abs()
    raise KeyError(scheme)


def lambda_pairs(y_pred, y_true, eps=1e-10, pad=-1, weighing_scheme=None, k=None, sigma=1.0, mu=10.0,
                 reduction_log="binary"):
    """Full pair-loss matrix and keep-mask, both [B,S,S] indexed by PREDICTED RANK (what
    lambdaMask(return_losses=True) returns, lambdaL.py:59-60, plus the mask of :62)."""
    if reduction_log not in ("binary", "natural"):
        raise ValueError("Reduction logarithm base can be either natural or binary")
    B, S = y_pred.shape
    dev = y_pred.device
    padm = y_true == pad
    ninf_p = torch.full_like(y_pred, float("-inf"))
    ninf_t = torch.full_like(y_true, float("-inf"))
    s_m = torch.where(padm, ninf_p, y_pred)
    y_m = torch.where(padm, ninf_t, y_true)
    r_pred = rank_desc(s_m)
    perm = torch.empty_like(r_pred)
    perm.scatter_(1, r_pred, torch.arange(S, device=dev).expand(B, S))    # perm[b, rank] = doc
    ss = torch.gather(s_m, 1, perm)
    yb = torch.gather(y_m, 1, perm)
    tdiff = yb[:, :, None] - yb[:, None, :]
    keep = torch.isfinite(tdiff)
    if weighing_scheme != "ndcgLoss1_scheme":
        keep = keep & (tdiff > 0)
    if k is not None:
        topk = torch.arange(S, device=dev) < k
        keep = keep & topk[None, :, None] & topk[None, None, :]
    ybc = yb.clamp(min=0.0)
    gain = torch.pow(2.0, ybc) - 1.0
    D = _discounts(S, dev)
    _, _, gain_doc = _gains(y_true, pad)
    ideal = _ideal_dcg(y_true, padm, gain_doc, D, eps, k)
    G = gain / ideal[:, None]
    w = scheme_weights(weighing_scheme, G, D, mu, ybc)
    d = (ss[:, :, None] - ss[:, None, :]).clamp(min=-1e8, max=1e8)
    d = torch.where(torch.isnan(d), torch.zeros_like(d), d)
    P = (torch.sigmoid(sigma * d).clamp(min=eps) ** w).clamp(min=eps)
    losses = torch.log2(P) if reduction_log == "binary" else torch.log(P)
    return losses, keep


def lambda_loss(y_pred, y_true, eps=1e-10, pad=-1, weighing_scheme=None, k=None, sigma=1.0, mu=10.0,
                reduction="sum", reduction_log="binary"):
    """Autograd flavour of lambdaLoss (losses/lambdaL.py:67-93)."""
    losses, keep = lambda_pairs(y_pred, y_true, eps, pad, weighing_scheme, k, sigma, mu, reduction_log)
    kept = losses[keep]
    if reduction == "sum":
        return -kept.sum()
    if reduction == "mean":
        return -kept.mean()
    raise ValueError("Reduction method can be either sum or mean")


def lambda_loss_closed_form(y_pred, y_true, eps=1e-10, pad=-1, weighing_scheme=None, k=None, sigma=1.0,
                            mu=10.0, reduction="sum", reduction_log="binary"):
    """(loss, dL/dy_pred, n_kept) in DOC space the way the HIP kernel computes it:
    for a kept pair (i, j):  l = max(w*log(max(u,eps)), log(eps)),  u = sigmoid(sigma (s_i - s_j)),
    dl/dD = w sigma (1 - u) / ln(base) unless a clamp is active; loss = -sum l (or / n_kept)."""
    with torch.no_grad():
        B, S = y_pred.shape
        dev = y_pred.device
        padm = y_true == pad
        valid = ~padm
        s_m = torch.where(padm, torch.full_like(y_pred, float("-inf")), y_pred)
        r = rank_desc(s_m)                                            # 0-based predicted rank per doc
        _, yc, gain = _gains(y_true, pad)
        D = _discounts(S, dev)
        G = gain / _ideal_dcg(y_true, padm, gain, D, eps, k)[:, None]
        keep = valid[:, :, None] & valid[:, None, :]
        if weighing_scheme != "ndcgLoss1_scheme":
            keep = keep & (y_true[:, :, None] > y_true[:, None, :])
        if k is not None:
            keep = keep & (r < k)[:, :, None] & (r < k)[:, None, :]
        inv = (1.0 / D)
        dG = (G[:, :, None] - G[:, None, :]).abs()
        m = (r[:, :, None] - r[:, None, :]).abs()
        delta = (inv[(m - 1).clamp(min=0)] - inv[m]).abs() * (m > 0)
        lr = (inv[r][:, :, None] - inv[r][:, None, :]).abs()
        one = torch.ones((B, S, S), dtype=y_pred.dtype)
        w = {
            None: one, "rankNet_scheme": one,
            "ndcgLoss1_scheme": ((G / D[r])[:, :, None]).expand(B, S, S),
            "ndcgLoss2_scheme": delta * dG,
            "lamdbaRank_scheme": lr * dG,
            "ndcgLoss2PP_scheme": mu * (delta * dG) + lr * dG,
            "rankNetWeightedByGTDiff_scheme": (yc[:, :, None] - yc[:, None, :]).abs(),
            "rankNetWeightedByGTDiffPowed_scheme": (yc[:, :, None] ** 2 - yc[:, None, :] ** 2).abs(),
        }[weighing_scheme].to(y_pred.dtype)
        d = torch.where(keep, y_pred[:, :, None] - y_pred[:, None, :], torch.zeros(1, dtype=y_pred.dtype))
        d = d.clamp(min=-1e8, max=1e8)
        u_raw = torch.sigmoid(sigma * d)
        u = u_raw.clamp(min=eps)
        lb = 1.0 / LN2 if reduction_log == "binary" else 1.0
        if reduction_log not in ("binary", "natural"):
            raise ValueError("Reduction logarithm base can be either natural or binary")
        wl = w * torch.log(u) * lb
        floor = math.log(eps) * lb
        ell = torch.maximum(wl, torch.full_like(wl, floor))
        km = keep.to(y_pred.dtype)
        n = keep.sum()
        live = km * (u_raw >= eps).to(km.dtype) * (wl >= floor).to(km.dtype)
        lam = live * w * sigma * (1.0 - u) * lb                       # d ell / d (s_i - s_j)
        if reduction == "sum":
            scale = 1.0
        elif reduction == "mean":
            scale = 1.0 / float(n)  if int(n) > 0 else float("nan")
        else:
            raise ValueError("Reduction method can be either sum or mean")
        loss = -(km * ell).sum() * scale if int(n) > 0 or reduction == "sum" else torch.tensor(float("nan"))
        grad = -(lam.sum(2) - lam.sum(1)) * scale                     # -(sum_j lam_ij - sum_j lam_ji)
        return loss, grad.to(y_pred.dtype), n


# --------------------------------------------------------------------------------------- ordinal
def with_ordinals(y, n, pad=-1):
    """[B,S] labels -> [B,S,n] cumulative targets 1[y >= k], k = 1..n; padded docs -> pad (ordinal.py:9-24)."""
    ks = torch.arange(1, n + 1, dtype=torch.float32)
    yy = y.unsqueeze(2).expand(-1, -1, n)
    o = (yy >= ks).float()
    return torch.where(yy == pad, torch.full_like(o, float(pad)), o)


def ordinal(y_pred, y_true, n, pad=-1):
    """Autograd flavour of ordinalLoss (ordinal.py:27-53).  The reference builds the targets with the
    DEFAULT indicator (-1) whatever `pad` is (:39) and then masks target == pad (:41-42)."""
    t = with_ordinals(y_true, n)                     # default -1 on purpose, see docstring
    masked = t == pad
    # ATen's BCE (the op the reference calls, ordinal.py:44): logs clamped at -100, analytic backward.
    # Masked targets are swapped for 0 first: torch >= 1.5 rejects targets outside [0,1].
    ls = torch.nn.functional.binary_cross_entropy(y_pred, torch.where(masked, torch.zeros_like(t), t),
                                                  reduction="none")
    ls = torch.where(masked, torch.zeros_like(ls), ls)
    n_docs = ((~masked).sum(2) > 0).sum()
    return ls.sum() / n_docs


def ordinal_closed_form(y_pred, y_true, n, pad=-1):
    """(loss, dL/dy_pred): BCE backward as ATen does it, (p - t) / max((1-p) p, 1e-12), / #valid docs."""
    with torch.no_grad():
        t = with_ordinals(y_true, n)
        masked = t == pad
        logp = torch.log(y_pred).clamp(min=-100.0)
        log1p = torch.log(1.0 - y_pred).clamp(min=-100.0)
        ls = torch.where(masked, torch.zeros_like(y_pred), -(t * logp + (1.0 - t) * log1p))
        n_docs = ((~masked).sum(2) > 0).sum().to(y_pred.dtype)
        g = (y_pred - t) / ((1.0 - y_pred) * y_pred).clamp(min=1e-12)
        g = torch.where(masked, torch.zeros_like(g), g) / n_docs
        return ls.sum() / n_docs, g


# --------------------------------------------------------------------------------------- scorers
def double_layer_forward(x, p, keep1=None, keep2=None, drop_p=0.5):
    """DoubleLayerNet (architeture/doubleLayer.py:54-73): fc3(drop(relu(fc2(drop(relu(fc1 x)))))).
    keep1/keep2: optional {0,1} keep masks [.., hidden]; dropout scales kept units by 1 / (1 - drop_p) (the reference's
    nn.Dropout(p=0.5), :60: by 2).  None -> predict()/eval() path (no dropout).  `p` maps state_dict keys to tensors."""
    scale = 1.0 / (1.0 - drop_p)
    h = torch.relu(x @ p["fc1.weight"].T + p["fc1.bias"])
    if keep1 is not None:
        h = h * keep1 * scale
    h = torch.relu(h @ p["fc2.weight"].T + p["fc2.bias"])
    if keep2 is not None:
        h = h * keep2 * scale
    return h @ p["fc3.weight"].T + p["fc3.bias"]


def triple_layer_forward(x, p):
    """TripleLayerNet (architeture/tripleLayer.py:5-17): l3(sigmoid(l2(l1 x))) -- no activation after l1."""
    h = x @ p["l1.weight"].T + p["l1.bias"]
    h = torch.sigmoid(h @ p["l2.weight"].T + p["l2.bias"])
    return h @ p["l3.weight"].T + p["l3.bias"]


def two_layer_forward(x, p):
    """The two-Linear-layer DoubleLayerNet variant the reference keeps commented out (architeture/doubleLayer.py:38-51):
    fc4(relu(fc1 x)), 136 -> 64 -> 1 (BASELINE.json configs[0]).  Its log_softmax(dim=1) only shifts each slate's scores
    by a constant (irrelevant to every listwise loss here) and is omitted, as in the device module."""
    h = torch.relu(x @ p["fc1.weight"].T + p["fc1.bias"])
    return h @ p["fc4.weight"].T + p["fc4.bias"]
