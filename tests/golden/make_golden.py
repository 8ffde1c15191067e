#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs ONLY in the build container, where the reference checkout is mounted read-only at
/root/reference (it never travels to the GPU box).  For every case it

  1. calls the reference function (losses/*.py, architeture/*.py) on seeded CPU inputs, fp32 and fp64,
  2. asserts that oracle/ltr_oracle.py (autograd AND closed-form flavours) reproduces loss and
     gradient -- this is what pins the oracle,
  3. stores inputs + expected outputs as plain arrays in an .npz next to a JSON manifest.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz, manifest.json)
Seed 2020 is the reference's own (main_batch_execution.py:21-22).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("LTR_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from losses.approxNDCG import approxNDCGLoss          # noqa: E402  (reference)
from losses.listnet import listnetLoss                # noqa: E402
from losses.lambdaL import lambdaLoss, lambdaMask     # noqa: E402
from losses.ordinal import ordinalLoss, with_ordinals  # noqa: E402
from architeture.doubleLayer import DoubleLayerNet    # noqa: E402
from architeture.tripleLayer import TripleLayerNet    # noqa: E402
import ltr_oracle as O                                # noqa: E402

torch.manual_seed(2020)
np.random.seed(2020)
torch.set_num_threads(4)

ARR = {}        # group -> {name: ndarray}
MANIFEST = {}   # group -> [case dicts]
WORST = {}      # loss kind -> worst oracle-vs-reference deviation seen


def relerr(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    if a.numel() == 0:
        return 0.0
    if not torch.isfinite(b).all():
        same = (torch.isfinite(a) == torch.isfinite(b)).all() and torch.allclose(
            torch.nan_to_num(a, 0, 0, 0), torch.nan_to_num(b, 0, 0, 0), rtol=1e-5, atol=1e-12)
        return 0.0 if same else float("inf")
    if not torch.isfinite(a).all():
        return float("inf")
    den = max(float(b.abs().max()), 1e-30)
    return float((a - b).abs().max()) / den


def note(kind, err, tol):
    assert err == err, f"{kind}: NaN deviation"
    WORST[kind] = max(WORST.get(kind, 0.0), err)
    assert err <= tol, f"{kind}: oracle deviates from reference by {err:.3e} > {tol:.1e}"


def put(group, case, **arrays):
    g = ARR.setdefault(group, {})
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        g[f"{case['id']}/{k}"] = np.asarray(v)
    MANIFEST.setdefault(group, []).append(case)


def labels(B, S, skew=False):
    if skew:  # MSLR-like grade skew, SURVEY 8(d)
        p = torch.tensor([0.52, 0.32, 0.13, 0.02, 0.01])
        return torch.multinomial(p, B * S, replacement=True).view(B, S).float()
    return torch.randint(0, 5, (B, S)).float()


def pad_tail(y, counts):
    y = y.clone()
    for b, c in enumerate(counts):
        if c > 0:
            y[b, y.shape[1] - c:] = -1.0
    return y


def run_ref(fn, s, *a, **kw):
    s = s.clone().requires_grad_(True)
    out = fn(s, *a, **kw)
    out.backward()
    return out.detach(), s.grad.detach()


# ------------------------------------------------------------------------------------ approxNDCG
def gen_approx():
    for S in (8, 32, 128, 512):
        B = 4
        variants = []
        s = torch.randn(B, S)
        y = labels(B, S)
        variants.append(("plain", s, y, 1.0))
        variants.append(("alpha", torch.randn(B, S), labels(B, S, skew=True), 2.5))
        variants.append(("padded", torch.randn(B, S), pad_tail(labels(B, S), [0, 1, S // 2, S - 2]), 1.0))
        y0 = labels(B, S)
        y0[1] = 0.0
        variants.append(("zero_labels", torch.randn(B, S), y0, 1.0))
        se = torch.randn(B, S) * 30.0
        se[0, 0] = 100.0
        se[0, 1] = -100.0
        variants.append(("extreme", se, labels(B, S), 1.0))
        if S == 8:
            variants.append(("all_padded_slate", torch.randn(B, S), pad_tail(labels(B, S), [S, 0, 3, 0]), 1.0))
        for name, s, y, alpha in variants:
            cid = f"approx_S{S}_{name}"
            loss, grad = run_ref(lambda p, t: approxNDCGLoss(p, t, alpha=alpha), s, y)
            loss64, grad64 = run_ref(lambda p, t: approxNDCGLoss(p, t, alpha=alpha), s.double(), y.double())
            # pin the oracle
            ol, og = run_ref(lambda p, t: O.approx_ndcg(p, t, alpha=alpha), s, y)
            note("approx/autograd32", max(relerr(ol, loss), relerr(og, grad)), 5e-6)
            cl, cg, _ = O.approx_ndcg_closed_form(s, y, alpha=alpha)
            note("approx/closed32", max(relerr(cl, loss), relerr(cg, grad)), 1e-5)
            cl, cg, _ = O.approx_ndcg_closed_form(s.double(), y.double(), alpha=alpha)
            note("approx/closed64", max(relerr(cl, loss64), relerr(cg, grad64)), 1e-9)
            put("approx", dict(id=cid, S=S, B=B, alpha=alpha, eps=1e-10, pad=-1),
                y_pred=s, y_true=y, loss=loss, grad=grad, loss64=loss64, grad64=grad64)


# --------------------------------------------------------------------------------------- ListNet
def gen_listnet():
    for S in (8, 32, 128, 512):
        B = 4
        for name, s, y, sig in (
            ("plain", torch.randn(B, S), labels(B, S), False),
            ("skew", torch.randn(B, S) * 3.0, labels(B, S, skew=True), False),
            ("sigmoid", torch.randn(B, S), labels(B, S), True),
        ):
            cid = f"listnet_S{S}_{name}"
            loss, grad = run_ref(lambda p, t: listnetLoss(t, p, apply_sigmoid=sig), s, y)
            loss64, grad64 = run_ref(lambda p, t: listnetLoss(t, p, apply_sigmoid=sig), s.double(), y.double())
            ol, og = run_ref(lambda p, t: O.listnet(t, p, sig), s, y)
            note("listnet/autograd32", max(relerr(ol, loss), relerr(og, grad)), 2e-6)
            cl, cg = O.listnet_closed_form(y, s, sig)
            note("listnet/closed32", max(relerr(cl, loss), relerr(cg, grad)), 1e-5)
            cl, cg = O.listnet_closed_form(y.double(), s.double(), sig)
            note("listnet/closed64", max(relerr(cl, loss64), relerr(cg, grad64)), 1e-9)
            put("listnet", dict(id=cid, S=S, B=B, apply_sigmoid=sig),
                y_pred=s, y_true=y, loss=loss, grad=grad, loss64=loss64, grad64=grad64)


# ------------------------------------------------------------------------------------ LambdaLoss
def gen_lambda():
    schemes = list(O.SCHEMES)
    for S in (8, 32, 128, 512):
        B = 4
        combos = []
        if S <= 32:
            for sch in schemes:
                combos.append((sch, None, 1.0, 10.0, "sum", "binary", "plain"))
            combos.append(("ndcgLoss2PP_scheme", 5, 1.0, 10.0, "sum", "binary", "plain"))
            combos.append(("ndcgLoss2PP_scheme", 10, 2.0, 5.0, "mean", "natural", "plain"))
            combos.append(("ndcgLoss1_scheme", 5, 1.0, 10.0, "mean", "binary", "padded"))
            combos.append(("ndcgLoss2_scheme", None, 0.5, 10.0, "mean", "natural", "padded"))
            combos.append(("lamdbaRank_scheme", 10, 1.0, 10.0, "sum", "natural", "padded"))
            combos.append((None, 5, 1.0, 10.0, "mean", "binary", "padded"))
            combos.append(("rankNetWeightedByGTDiffPowed_scheme", None, 1.0, 10.0, "sum", "binary", "extreme"))
            combos.append(("ndcgLoss2PP_scheme", None, 1.0, 10.0, "sum", "binary", "extreme"))
        else:
            combos.append(("ndcgLoss2PP_scheme", None, 1.0, 10.0, "sum", "binary", "plain"))
            combos.append(("ndcgLoss2PP_scheme", 10, 1.0, 10.0, "mean", "binary", "padded"))
            combos.append(("lamdbaRank_scheme", None, 1.0, 10.0, "sum", "natural", "plain"))
            combos.append(("ndcgLoss1_scheme", None, 1.0, 10.0, "sum", "binary", "plain"))
            combos.append((None, None, 2.0, 10.0, "mean", "binary", "plain"))
            combos.append(("ndcgLoss2_scheme", None, 1.0, 10.0, "sum", "binary", "extreme"))
        for n, (sch, k, sigma, mu, red, lg, kind) in enumerate(combos):
            s = torch.randn(B, S)
            y = labels(B, S)
            if kind == "padded":
                y = pad_tail(y, [0, 2, S // 2, S - 3])
            if kind == "extreme":
                s = s * 30.0
                s[0, 0], s[0, 1] = 100.0, -100.0
                y[0, 0], y[0, 1] = 4.0, 0.0      # make the +-100 pair a kept pair both ways round
                y[1, 0], y[1, 1] = 0.0, 4.0
                s[1, 0], s[1, 1] = 100.0, -100.0
            kw = dict(weighing_scheme=sch, k=k, sigma=sigma, mu=mu, reduction=red, reduction_log=lg)
            cid = f"lambda_S{S}_{n:02d}_{sch}_{kind}"
            loss, grad = run_ref(lambda p, t: lambdaLoss(p, t, **kw), s, y)
            loss64, grad64 = run_ref(lambda p, t: lambdaLoss(p, t, **kw), s.double(), y.double())
            ol, og = run_ref(lambda p, t: O.lambda_loss(p, t, **kw), s, y)
            note("lambda/autograd32", max(relerr(ol, loss), relerr(og, grad)), 1e-5)
            cl, cg, _ = O.lambda_loss_closed_form(s, y, **kw)
            note("lambda/closed32", max(relerr(cl, loss), relerr(cg, grad)), 2e-5)
            cl, cg, nk = O.lambda_loss_closed_form(s.double(), y.double(), **kw)
            note("lambda/closed64", max(relerr(cl, loss64), relerr(cg, grad64)), 1e-7)
            arrays = dict(y_pred=s, y_true=y, loss=loss, grad=grad, loss64=loss64, grad64=grad64,
                          n_kept=np.int64(int(nk)))
            if S <= 32 and kind != "extreme":
                # lambdaMask(return_losses=True): full [B,S,S] matrix in pred-rank order + its backward
                sr = s.clone().requires_grad_(True)
                full = lambdaMask(sr, y, return_losses=True, **kw)
                gup = torch.randn_like(full)
                full.backward(gup)
                of, okeep = O.lambda_pairs(s, y, weighing_scheme=sch, k=k, sigma=sigma, mu=mu, reduction_log=lg)
                note("lambda/full32", relerr(of, full.detach()), 1e-5)
                masked = lambdaMask(s, y, **kw)
                assert torch.allclose(of[okeep], masked, rtol=1e-5, atol=1e-6), cid
                arrays.update(full=full.detach(), full_gup=gup, full_grad=sr.grad, masked=masked,
                              keep=okeep.numpy().astype(np.uint8))
            put("lambda", dict(id=cid, S=S, B=B, scheme=sch, k=k, sigma=sigma, mu=mu, reduction=red,
                               reduction_log=lg, eps=1e-10, pad=-1, kind=kind,
                               has_full=bool(S <= 32 and kind != "extreme")), **arrays)


# --------------------------------------------------------------------------------------- ordinal
def gen_ordinal():
    for S in (8, 32):
        B, n = 4, 4
        for name in ("plain", "padded", "saturated"):
            p = torch.rand(B, S, n) * 0.98 + 0.01
            y = labels(B, S)
            if name == "padded":
                y = pad_tail(y, [0, 1, S // 2, S - 1])
            if name == "saturated":
                p[0, 0, :] = torch.tensor([0.0, 1.0, 0.0, 1.0])   # exercises the -100 log clamp
                y[0, 0] = 2.0
            cid = f"ordinal_S{S}_{name}"
            pr = p.clone().requires_grad_(True)
            try:
                loss = ordinalLoss(pr, y, n)
            except RuntimeError as e:
                # torch >= 1.5 BCELoss rejects the -1 targets the reference feeds it for padded docs
                # (ordinal.py:23,44) before its own mask (:45) can zero them: the reference cannot run this
                # case on this torch.  Stored from the oracle (intended semantics) and flagged UNPINNED.
                assert name == "padded" and "between 0 and 1" in str(e)
                cl, cg = O.ordinal_closed_form(p, y, n)
                ol, og = run_ref(lambda q, t: O.ordinal(q, t, n), p, y)
                assert relerr(cl, ol) < 1e-6 and relerr(cg, og) < 1e-5
                put("ordinal", dict(id=cid, S=S, B=B, n=n, pad=-1, pinned=False,
                                    why="reference raises on torch>=1.5: BCELoss target check"),
                    y_pred=p, y_true=y, loss=ol, grad=og, ordinals=with_ordinals(y.clone(), n))
                continue
            loss.backward()
            ol, og = run_ref(lambda q, t: O.ordinal(q, t, n), p, y)
            note("ordinal/autograd32", relerr(ol, loss.detach()), 2e-6)
            note("ordinal/autograd32", relerr(og, pr.grad), 2e-6)
            cl, cg = O.ordinal_closed_form(p, y, n)
            note("ordinal/closed32", max(relerr(cl, loss.detach()), relerr(cg, pr.grad)), 1e-5)
            assert torch.equal(O.with_ordinals(y, n), with_ordinals(y.clone(), n))
            put("ordinal", dict(id=cid, S=S, B=B, n=n, pad=-1),
                y_pred=p, y_true=y, loss=loss.detach(), grad=pr.grad, ordinals=with_ordinals(y.clone(), n))


# --------------------------------------------------------------------------------------- scorers
def gen_scorers():
    F = 136
    B, S = 3, 16
    x = torch.randn(B, S, F)
    y = labels(B, S)
    gs = torch.randn(B, S, 1)

    # ---- TripleLayerNet: forward, backward under a fixed upstream gradient, and end-to-end + approxNDCG
    torch.manual_seed(2020)
    net = TripleLayerNet(F)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    out = net(x, None, None)
    out.backward(gs)
    g_up = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    note("triple/forward", relerr(O.triple_layer_forward(x, sd), out.detach()), 2e-6)
    net.zero_grad()
    loss = approxNDCGLoss(net(x, None, None).squeeze(-1), y)
    loss.backward()
    g_e2e = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    arrays = dict(x=x, y_true=y, gs=gs, out=out.detach(), e2e_loss=loss.detach())
    arrays.update({f"sd.{k}": v for k, v in sd.items()})
    arrays.update({f"gup.{k}": v for k, v in g_up.items()})
    arrays.update({f"ge2e.{k}": v for k, v in g_e2e.items()})
    put("scorers", dict(id="triple", F=F, B=B, S=S, keys=list(sd.keys())), **arrays)

    # ---- DoubleLayerNet: predict()/eval path and the training path with the dropout masks captured
    torch.manual_seed(2020)
    net = DoubleLayerNet(F)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    pred = net.predict(x, None, None)
    note("double/predict", relerr(O.double_layer_forward(x, sd), pred.detach()), 2e-6)
    net.eval()
    assert torch.equal(net(x, None, None), pred)
    pred.backward(gs)
    g_eval = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    net.zero_grad()
    net.train()
    keeps = []
    h = net.dropout.register_forward_hook(lambda m, i, o: keeps.append((o != 0) | (i[0] == 0)))
    torch.manual_seed(7)
    out_tr = net(x, None, None)
    h.remove()
    out_tr.backward(gs)
    g_train = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    k1, k2 = [k.float() for k in keeps]
    note("double/train", relerr(O.double_layer_forward(x, sd, k1, k2), out_tr.detach()), 2e-6)
    net.zero_grad()
    net.eval()
    loss = approxNDCGLoss(net(x, None, None).squeeze(-1), y)
    loss.backward()
    g_e2e = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    arrays = dict(x=x, y_true=y, gs=gs, out_eval=pred.detach(), out_train=out_tr.detach(),
                  keep1=k1.numpy().astype(np.uint8), keep2=k2.numpy().astype(np.uint8), e2e_loss=loss.detach())
    arrays.update({f"sd.{k}": v for k, v in sd.items()})
    arrays.update({f"geval.{k}": v for k, v in g_eval.items()})
    arrays.update({f"gtrain.{k}": v for k, v in g_train.items()})
    arrays.update({f"ge2e.{k}": v for k, v in g_e2e.items()})
    put("scorers", dict(id="double", F=F, B=B, S=S, keys=list(sd.keys())), **arrays)


if __name__ == "__main__":
    gen_approx()
    gen_listnet()
    gen_lambda()
    gen_ordinal()
    gen_scorers()
    total = 0
    for g, arrs in ARR.items():
        path = os.path.join(HERE, f"{g}.npz")
        np.savez_compressed(path, **arrs)
        total += os.path.getsize(path)
        print(f"{g}: {len(MANIFEST[g])} cases, {os.path.getsize(path) / 1024:.0f} KiB")
    MANIFEST["_oracle_vs_reference_worst_relerr"] = WORST
    MANIFEST["_generator"] = dict(torch=torch.__version__, seed=2020, reference="Haiga/nn-with-pytorch-personalized-losses @ v1")
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(MANIFEST, f, indent=1)
    print(f"total {total / 1024:.0f} KiB; worst oracle-vs-reference deviations:")
    for k, v in sorted(WORST.items()):
        print(f"  {k:24s} {v:.3e}")
