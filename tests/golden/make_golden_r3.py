#!/usr/bin/env python3
"""Round-2 golden vectors for SURVEY.md row f-3 (set-transformer scorer, BASELINE config 5) from the REAL reference.

Same rules as make_golden.py / make_golden_r2.py: runs ONLY in the build container against the read-only reference
checkout; builds `architeture.multiLayer.make_model` networks of the reference with seeded weights, runs them (fp32,
CPU) on seeded inputs, takes the reference's own approxNDCGLoss on the scores and back-propagates; asserts that the
oracle restatement (oracle/ltr_encoder_oracle.py) reproduces scores, loss and every parameter gradient -- this pins the
oracle -- and stores inputs, weights and expected outputs as plain arrays:

    tests/golden/encoder.npz, tests/golden/manifest_r3.json

Dropout: the reference draws its masks from torch's global RNG, which no other implementation can reproduce, so the
networks are pinned in eval mode and in train mode with dropout = 0 (identical arithmetic); the dropout sites are pinned
separately through the oracle's explicit keep masks (tests/test_encoder_gpu.py).
Usage:  python tests/golden/make_golden_r3.py
"""
import copy
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

from architeture.multiLayer import make_model                 # noqa: E402  (reference)
from losses.approxNDCG import approxNDCGLoss                   # noqa: E402  (reference)
import ltr_encoder_oracle as EO                                # noqa: E402

import attr                                                    # noqa: E402


@attr.s(auto_attribs=True)
class TransformerConfig:
    """What the reference's config hands to make_model (multiLayer.py:142 calls attr.asdict on it)."""
    N: int
    d_ff: int
    h: int
    dropout: float
    positional_encoding: object = None


torch.manual_seed(2021)
torch.set_num_threads(4)

CASES = [
    # id, n_features, fc_model, transformer, B, S, padded documents per slate
    ("fc32_enc2_S12", 16, dict(sizes=[32], input_norm=False, activation=None, dropout=0.0),
     dict(N=2, d_ff=64, h=4, dropout=0.1, positional_encoding=None), 3, 12, (0, 3, 5)),
    ("enc1_dk8_S20", 24, None, dict(N=1, d_ff=32, h=3, dropout=0.0, positional_encoding=None), 2, 20, (0, 0)),
    ("fc3_norm_noenc_S9", 16, dict(sizes=[24, 40, 16], input_norm=True, activation="Sigmoid", dropout=0.0), None, 4, 9, None),
    ("enc1_dk17_S33", 136, None, dict(N=1, d_ff=64, h=8, dropout=0.1, positional_encoding=None), 2, 33, (1, 7)),
    ("fc128_enc1_S100", 136, dict(sizes=[128, 256, 128], input_norm=False, activation=None, dropout=0.0),
     dict(N=1, d_ff=128, h=8, dropout=0.1, positional_encoding=None), 2, 100, (0, 37)),
    ("fc64_enc3_S256", 136, dict(sizes=[64], input_norm=False, activation=None, dropout=0.0),
     dict(N=3, d_ff=128, h=4, dropout=0.0, positional_encoding=None), 1, 256, (11,)),
]


def relerr(a, b, floor=1e-30):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), floor)


def main():
    arr, manifest, worst, worst64 = {}, [], 0.0, 0.0
    for cid, F, fc, tr, B, S, pads in CASES:
        net = make_model(fc_model=copy.deepcopy(fc), transformer=TransformerConfig(**tr) if tr else None,
                         post_model=dict(d_output=1, output_activation="Sigmoid"), n_features=F)
        # make_model leaves biases / norm parameters at their init (0 / 1): perturb so that every gradient is exercised
        with torch.no_grad():
            for name, p in net.named_parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
        net.eval()
        x = torch.randn(B, S, F)
        y = torch.randint(0, 5, (B, S)).float()
        mask = None
        if pads is not None:
            mask = torch.zeros(B, S, dtype=torch.bool)
            for b, n in enumerate(pads):
                if n:
                    mask[b, S - n:] = True
                    y[b, S - n:] = -1.0
        scores = net(x, mask, None)
        loss = approxNDCGLoss(scores, y)
        net.zero_grad()
        loss.backward()
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        cfg = EO.config_of(dict(fc_model=fc, transformer=tr), F)
        # pin the oracle: fp32 like the reference, and fp64
        for dt, tol in ((torch.float32, 2e-5), (torch.float64, 1e-3)):      # fp64: bounded by the reference's own fp32 noise
            s_o, l_o, g_o = EO.scores_and_grads(sd, x, mask, cfg, lambda s: approxNDCGLoss(s, y.to(s.dtype)), dtype=dt)
            # the key-projection bias has an identically-zero gradient (softmax is shift invariant): what the reference
            # stores for it is rounding noise, so gradients are compared on the scale of the case's largest gradient
            floor = 1e-3 * max(float(g.abs().max()) for g in grads.values())
            e = max([relerr(s_o, scores), relerr(l_o, loss)] + [relerr(g_o[k], grads[k], floor) for k in grads])
            if dt == torch.float32:
                worst = max(worst, e)
            else:
                worst64 = max(worst64, e)
            assert e <= tol, f"{cid} {dt}: oracle deviates from the reference by {e:.3e}"
        arr[f"{cid}/x"], arr[f"{cid}/y"] = x.numpy(), y.numpy()
        if mask is not None:
            arr[f"{cid}/mask"] = mask.numpy()
        arr[f"{cid}/scores"], arr[f"{cid}/loss"] = scores.detach().numpy(), loss.detach().numpy()
        for k, v in sd.items():
            arr[f"{cid}/w/{k}"] = v.numpy()
        for k, v in grads.items():
            arr[f"{cid}/g/{k}"] = v.numpy()
        manifest.append(dict(id=cid, n_features=F, fc_model=fc, transformer=tr, B=B, S=S, has_mask=mask is not None,
                             keys=list(sd.keys())))
        print(f"{cid}: loss {float(loss.detach()):.6f}  params {sum(v.numel() for v in sd.values())}")
    np.savez_compressed(os.path.join(HERE, "encoder.npz"), **arr)
    with open(os.path.join(HERE, "manifest_r3.json"), "w") as f:
        json.dump({"_note": "generated by make_golden_r3.py from the reference (fp32 CPU); oracle pinned at generation "
                            f"time: worst deviation {worst:.2e} in fp32 (same arithmetic as the reference), {worst64:.2e} in fp64 "
                            "(= the reference's own fp32 rounding noise)", "encoder": manifest}, f, indent=1)
    print("worst oracle deviation fp32 / fp64", worst, worst64)


if __name__ == "__main__":
    main()
