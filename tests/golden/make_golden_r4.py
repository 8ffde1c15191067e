#!/usr/bin/env python3
"""Round-3 golden vectors from the REAL reference (build container only; the reference never travels):

  (1) encoder_c5 -- the network `bench.py` / tools/bench_encoder.py time for BASELINE config 5: FC 136 -> 128, 6 encoder
      blocks, 8 heads, d_ff 2048, slate 256, B = 2, eval mode, one slate with a padded tail.
      3.6 M parameters: the weights are NOT stored -- they are drawn from numpy's PCG64 stream (`seeded_state_dict`
      below, re-implemented by the test from the manifest's seed / scales), loaded into the reference's model with
      load_state_dict, and only x, y, mask, scores, loss, every 1-D parameter gradient in full and, of every matrix
      gradient, its first 4 rows + its L2 norm + its sum are stored (300 KB).
  (2) blocks -- every building block of architeture/transformer.py / multiLayer.py called ON ITS OWN:
      LayerNorm, SublayerConnection, MultiHeadedAttention, attention(), PositionwiseFeedForward, EncoderLayer, Encoder,
      FCModel, OutputLayer.forward / .score with d_output = 1 and 3; inputs, state_dict, outputs, and the gradients of
      sum(out * w) w.r.t. the block's parameters and its input.

Both are also run through oracle/ltr_encoder_oracle.py (asserted equal) where the oracle covers them.
Usage:  python tests/golden/make_golden_r4.py
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

from architeture.multiLayer import make_model, FCModel, OutputLayer                 # noqa: E402  (reference)
from architeture import transformer as RT                                           # noqa: E402  (reference)
from losses.approxNDCG import approxNDCGLoss                                         # noqa: E402  (reference)
import ltr_encoder_oracle as EO                                                      # noqa: E402

import attr                                                                          # noqa: E402


@attr.s(auto_attribs=True)
class TransformerConfig:
    N: int
    d_ff: int
    h: int
    dropout: float
    positional_encoding: object = None


def seeded_state_dict(shapes, seed):
    """{key: tensor} from numpy's PCG64: matrices ~ U(-a, a) with the Xavier bound a = sqrt(6 / (fan_in + fan_out)),
    `a_2` / norm weights ~ 1 + 0.1 N(0,1), every other vector ~ 0.1 N(0,1); keys in the given order, float32."""
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape in shapes:
        if len(shape) == 2:
            a = float(np.sqrt(6.0 / (shape[0] + shape[1])))
            v = rng.uniform(-a, a, size=shape)
        else:
            v = 0.1 * rng.standard_normal(size=shape)
            if key.endswith("a_2") or key.endswith("norm.weight"):
                v = 1.0 + v
        out[key] = torch.from_numpy(v.astype(np.float32))
    return out


def relerr(a, b, floor=1e-30):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), floor)


def c5_case(arr):
    torch.manual_seed(404)
    F, B, S = 136, 2, 256
    fc = dict(sizes=[128], input_norm=False, activation=None, dropout=0.1)
    tr = dict(N=6, d_ff=2048, h=8, dropout=0.1, positional_encoding=None)
    net = make_model(fc_model=copy.deepcopy(fc), transformer=TransformerConfig(**tr),
                     post_model=dict(d_output=1, output_activation=None), n_features=F)
    shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    seed = 20261004
    sd = seeded_state_dict(shapes, seed)
    net.load_state_dict(sd)
    net.eval()
    x = torch.randn(B, S, F)
    y = torch.randint(0, 5, (B, S)).float()
    mask = torch.zeros(B, S, dtype=torch.bool)
    mask[1, S - 29:] = True
    y[1, S - 29:] = -1.0
    scores = net(x, mask, None)
    loss = approxNDCGLoss(scores, y)
    net.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    cfg = EO.config_of(dict(fc_model=fc, transformer=tr), F)
    s_o, l_o, g_o = EO.scores_and_grads(sd, x, mask, cfg, lambda s: approxNDCGLoss(s, y.to(s.dtype)), dtype=torch.float32)
    floor = 1e-3 * max(float(g.abs().max()) for g in grads.values())
    e = max([relerr(s_o, scores), relerr(l_o, loss)] + [relerr(g_o[k], grads[k], floor) for k in grads])
    assert e < 5e-5, f"oracle deviates from the reference on the config-5 network by {e:.3e}"
    cid = "c5_fc128_enc6_dff2048_S256"
    arr[f"{cid}/x"], arr[f"{cid}/y"], arr[f"{cid}/mask"] = x.numpy(), y.numpy(), mask.numpy()
    arr[f"{cid}/scores"], arr[f"{cid}/loss"] = scores.detach().numpy(), loss.detach().numpy()
    for k, g in grads.items():
        if g.dim() == 1:
            arr[f"{cid}/g/{k}"] = g.numpy()
        else:
            arr[f"{cid}/grows/{k}"] = g[:4].numpy()
            arr[f"{cid}/gstat/{k}"] = np.array([float(g.double().norm()), float(g.double().sum()), float(g.abs().max())])
    print(f"{cid}: loss {float(loss):.6f}, params {sum(v.numel() for v in sd.values())}, oracle fp32 deviation {e:.2e}")
    return dict(id=cid, n_features=F, fc_model=fc, transformer=tr, B=B, S=S, has_mask=True, weight_seed=seed,
                shapes=[[k, list(s)] for k, s in shapes], oracle_fp32_deviation=e)


def _run_block(mod, inputs, call, arr, cid, extra=None):
    """Forward `call(mod, *inputs)`, backward of sum(out * w); store everything."""
    mod is not None and mod.eval()
    ins = [t.clone().requires_grad_(True) if t.is_floating_point() else t for t in inputs]
    out = call(mod, *ins)
    out_t = out[0] if isinstance(out, tuple) else out
    w = torch.randn_like(out_t)
    params = dict(mod.named_parameters()) if mod is not None else {}
    mod is not None and mod.zero_grad()
    (out_t * w).sum().backward()
    for i, t in enumerate(inputs):
        arr[f"{cid}/in{i}"] = t.detach().clone().numpy()
        if t.is_floating_point():
            arr[f"{cid}/din{i}"] = ins[i].grad.detach().clone().numpy()
    arr[f"{cid}/out"], arr[f"{cid}/w_out"] = out_t.detach().clone().numpy(), w.clone().numpy()
    if isinstance(out, tuple):
        arr[f"{cid}/out1"] = out[1].detach().clone().numpy()
    keys = []
    if mod is not None:
        for k, v in mod.state_dict().items():
            arr[f"{cid}/w/{k}"] = v.detach().clone().numpy()
            keys.append(k)
        for k, p in params.items():
            arr[f"{cid}/g/{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().clone().numpy()   # (a view would alias a .grad that later blocks accumulate into)
    rec = dict(id=cid, keys=keys, n_inputs=len(inputs))
    rec.update(extra or {})
    print(f"{cid}: out {tuple(out_t.shape)}")
    return rec


def perturb(mod):
    with torch.no_grad():
        for p in mod.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            else:
                torch.nn.init.xavier_uniform_(p)
    return mod


def block_cases(arr):
    torch.manual_seed(777)
    recs = []
    B, S, d, h, dff = 3, 20, 32, 4, 64
    x = torch.randn(B, S, d)
    mask = torch.zeros(B, S, dtype=torch.bool)
    mask[1, S - 6:] = True
    m3 = mask.unsqueeze(-2)                                   # what Encoder.forward hands its layers (transformer.py:55)
    recs.append(_run_block(perturb(RT.LayerNorm(d)), [x], lambda m, a: m(a), arr, "LayerNorm", dict(kind="LayerNorm", d=d)))
    ff = perturb(RT.PositionwiseFeedForward(d, dff, 0.1))
    recs.append(_run_block(ff, [x], lambda m, a: m(a), arr, "PositionwiseFeedForward", dict(kind="PositionwiseFeedForward", d=d, d_ff=dff)))
    mha = perturb(RT.MultiHeadedAttention(h, d, 0.1))
    recs.append(_run_block(mha, [x, m3], lambda m, a, mk: m(a, a, a, mk), arr, "MultiHeadedAttention_self",
                           dict(kind="MultiHeadedAttention", d=d, h=h, self_attn=True)))
    xk, xv = torch.randn(B, S, d), torch.randn(B, S, d)
    recs.append(_run_block(mha, [x, xk, xv, m3], lambda m, a, b, c, mk: m(a, b, c, mk), arr, "MultiHeadedAttention_cross",
                           dict(kind="MultiHeadedAttention", d=d, h=h, self_attn=False)))
    q, k, v = (torch.randn(B, h, S, d // h) for _ in range(3))
    recs.append(_run_block(None, [q, k, v, mask.view(B, 1, 1, S)], lambda m, a, b, c, mk: RT.attention(a, b, c, mask=mk, dropout=None),
                           arr, "attention_fn", dict(kind="attention", h=h, dk=d // h)))
    sub = perturb(RT.SublayerConnection(d, 0.1))
    recs.append(_run_block(sub, [x], lambda m, a: m(a, ff), arr, "SublayerConnection_ffn",
                           dict(kind="SublayerConnection", d=d, d_ff=dff, sublayer="PositionwiseFeedForward")))
    mha.attn = None                                           # (holds the last p_attn: not deep-copyable)
    layer = perturb(RT.EncoderLayer(d, copy.deepcopy(mha), copy.deepcopy(ff), 0.1))
    recs.append(_run_block(layer, [x, m3], lambda m, a, mk: m(a, mk), arr, "EncoderLayer", dict(kind="EncoderLayer", d=d, h=h, d_ff=dff)))
    enc = perturb(RT.make_transformer(N=2, d_ff=dff, h=h, dropout=0.1, n_features=d))
    recs.append(_run_block(enc, [x, mask], lambda m, a, mk: m(a, mk, None), arr, "Encoder", dict(kind="Encoder", d=d, h=h, d_ff=dff, N=2)))
    xf = torch.randn(B, S, 24)
    fcm = perturb(FCModel(sizes=[40, 16], input_norm=True, activation=None, dropout=0.2, n_features=24))
    recs.append(_run_block(fcm, [xf], lambda m, a: m(a), arr, "FCModel_norm", dict(kind="FCModel", sizes=[40, 16], input_norm=True, n_features=24)))
    fcm2 = perturb(FCModel(sizes=[32], input_norm=False, activation=None, dropout=0.0, n_features=24))
    recs.append(_run_block(fcm2, [xf], lambda m, a: m(a), arr, "FCModel_plain", dict(kind="FCModel", sizes=[32], input_norm=False, n_features=24)))
    for dout in (1, 3):
        ol = perturb(OutputLayer(d, dout, None))
        recs.append(_run_block(ol, [x], lambda m, a: m(a), arr, f"OutputLayer_d{dout}_forward", dict(kind="OutputLayer", d=d, d_output=dout, method="forward")))
        recs.append(_run_block(ol, [x], lambda m, a: m.score(a), arr, f"OutputLayer_d{dout}_score", dict(kind="OutputLayer", d=d, d_output=dout, method="score")))
    # the whole model with d_output = 3: forward -> [B, S, 3], score -> sum over the outputs (multiLayer.py:113-124)
    net = make_model(dict(sizes=[d], input_norm=False, activation=None, dropout=0.0), TransformerConfig(N=1, d_ff=dff, h=h, dropout=0.0),
                     dict(d_output=3, output_activation=None), 24)
    perturb(net)
    recs.append(_run_block(net, [xf, mask], lambda m, a, mk: m(a, mk, None), arr, "LTRModel_d3_forward",
                           dict(kind="LTRModel", d=d, h=h, d_ff=dff, N=1, d_output=3, method="forward", n_features=24)))
    recs.append(_run_block(net, [xf, mask], lambda m, a, mk: m.score(a, mk, None), arr, "LTRModel_d3_score",
                           dict(kind="LTRModel", d=d, h=h, d_ff=dff, N=1, d_output=3, method="score", n_features=24)))
    recs.append(_run_block(net, [xf, mask], lambda m, a, mk: m.prepare_for_output(a, mk, None), arr, "LTRModel_prepare_for_output",
                           dict(kind="LTRModel", d=d, h=h, d_ff=dff, N=1, d_output=3, method="prepare_for_output", n_features=24)))
    return recs


def main():
    torch.set_num_threads(8)
    arr = {}
    c5 = c5_case(arr)
    np.savez_compressed(os.path.join(HERE, "encoder_c5.npz"), **arr)
    arr2 = {}
    blocks = block_cases(arr2)
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **arr2)
    with open(os.path.join(HERE, "manifest_r4.json"), "w") as f:
        json.dump({"_note": "generated by make_golden_r4.py from the reference (fp32 CPU, eval mode)", "encoder_c5": [c5],
                   "blocks": blocks}, f, indent=1)


if __name__ == "__main__":
    main()
