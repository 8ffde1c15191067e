"""CPU: the encoder oracle against the reference's golden vectors (tests/golden/encoder.npz, made by make_golden_r3.py
from the imported reference), and the host-side contract of the make_model drop-in (names, state_dict keys, shapes,
argument checks) -- no kernel is launched here."""
import copy

import numpy as np
import pytest
import torch

from conftest import golden

import ltr_encoder_oracle as EO
import ltr_oracle as O


@pytest.mark.parametrize("case", golden("encoder").cases, ids=lambda c: c["id"])
def test_oracle_reproduces_reference(case):
    g = golden("encoder")
    sd = {k: torch.from_numpy(g.arr(case, "w/" + k)) for k in case["keys"]}
    x, y = torch.from_numpy(g.arr(case, "x")), torch.from_numpy(g.arr(case, "y"))
    mask = torch.from_numpy(g.arr(case, "mask")) if case["has_mask"] else None
    cfg = EO.config_of(dict(fc_model=case["fc_model"], transformer=case["transformer"]), case["n_features"])
    s, loss, grads = EO.scores_and_grads(sd, x, mask, cfg, lambda v: O.approx_ndcg(v, y.to(v.dtype)), dtype=torch.float32)
    want = g.arr(case, "scores")
    assert np.abs(s.numpy() - want).max() <= 2e-5 * np.abs(want).max()
    assert abs(float(loss) - float(g.arr(case, "loss"))) <= 2e-5 * abs(float(g.arr(case, "loss")))
    gmax = max(float(np.abs(g.arr(case, "g/" + k)).max()) for k in case["keys"])
    for k in case["keys"]:
        ref = g.arr(case, "g/" + k)
        # identically-zero gradients (key-projection bias: softmax is shift invariant) hold rounding noise in the
        # reference: compare on the scale of the case's largest gradient
        e = np.abs(grads[k].numpy() - ref).max() / max(np.abs(ref).max(), 1e-3 * gmax)
        assert e <= 3e-4, (k, e)      # fp32 both sides, different op order in the loss: cancelling sums (norm.b_2) carry ~1e-4


def test_oracle_dropout_masks_and_bf16_rounding_are_wired():
    torch.manual_seed(0)
    fc = dict(sizes=[16], input_norm=False, activation=None, dropout=0.5)
    tr = dict(N=1, d_ff=32, h=2, dropout=0.5, positional_encoding=None)
    from architeture.multiLayer import make_model
    net = make_model(copy.deepcopy(fc), copy.deepcopy(tr), dict(d_output=1), 8)
    sd = {k: v.detach().double() for k, v in net.state_dict().items()}
    cfg = EO.config_of(dict(fc_model=fc, transformer=tr), 8)
    x, mask = torch.randn(2, 6, 8, dtype=torch.float64), torch.zeros(2, 6, dtype=torch.bool)
    base = EO.encoder_scores(sd, x, mask, cfg)
    ones = {("fc", 0): torch.ones(12, 16), ("attn", 0): torch.ones(2, 2, 6, 6), ("attn_out", 0): torch.ones(12, 16),
            ("ffn_hidden", 0): torch.ones(12, 32), ("ffn_out", 0): torch.ones(12, 16)}
    for site in ones:                      # every site, alone, changes the scores; all-ones = scaling by 1/(1-p)
        keep = {site: (torch.rand_like(ones[site]) > 0.5).double()}
        assert not torch.allclose(EO.encoder_scores(sd, x, mask, cfg, keep), base)
    assert not torch.equal(EO.encoder_scores(sd, x, mask, cfg, bf16=True), base)
    assert torch.allclose(EO.encoder_scores(sd, x, mask, cfg, bf16=True), base, rtol=0.1, atol=0.05)


def test_make_model_layout_matches_reference_keys():
    from architeture import multiLayer as ML, transformer as TR
    for case in golden("encoder").cases:
        net = ML.make_model(copy.deepcopy(case["fc_model"]), copy.deepcopy(case["transformer"]),
                            dict(d_output=1, output_activation="Sigmoid"), case["n_features"])
        assert list(net.state_dict().keys()) == case["keys"]
        g = golden("encoder")
        for k, v in net.state_dict().items():
            assert tuple(v.shape) == g.arr(case, "w/" + k).shape, k
        spec = net._ltr_spec(case["n_features"])
        assert len(net._ltr_params()) == spec.n_params()
    for name in ("clones", "Encoder", "LayerNorm", "SublayerConnection", "EncoderLayer", "attention", "MultiHeadedAttention",
                 "PositionwiseFeedForward", "make_transformer"):
        assert hasattr(TR, name)
    for name in ("first_arg_id", "FCModel", "LTRModel", "OutputLayer", "make_model"):
        assert hasattr(ML, name)


def test_make_model_reference_quirks():
    import attr
    from architeture.multiLayer import FCModel, make_model
    from architeture.transformer import make_transformer

    @attr.s(auto_attribs=True)
    class TransformerConfig:            # what the reference's config hands over (multiLayer.py:142: attr.asdict)
        N: int
        d_ff: int
        h: int
        dropout: float
        positional_encoding: object = None

    net = make_model(dict(sizes=[32], input_norm=True, activation="Tanh", dropout=None), TransformerConfig(2, 64, 4, 0.1),
                     dict(d_output=1, output_activation="Sigmoid"), 16)
    assert isinstance(net.input_layer.activation, torch.nn.Identity)          # config activations are ignored (:29, :105)
    assert isinstance(net.output_layer.activation, torch.nn.Identity)
    assert net.input_layer.dropout.p == 0.0                                    # `dropout or 0.0` (:30)
    assert len(net.encoder.layers) == 2 and net.encoder.position is None
    sizes = [24, 8]
    FCModel(sizes, False, None, 0.0, 16)
    assert sizes == [16, 24, 8]                                                # the caller's list is mutated (:27)
    w = net.encoder.layers[0].feed_forward.w_1.weight
    bound = (6.0 / (w.shape[0] + w.shape[1])) ** 0.5
    assert float(w.abs().max()) <= bound + 1e-6                                # Xavier uniform (:146-148)
    enc = make_transformer()                                                   # defaults: N=6, d_ff=2048, h=8, 136 features
    assert len(enc.layers) == 6 and enc.layers[0].self_attn.d_k == 17 and enc.layers[0].feed_forward.w_1.out_features == 2048
    with pytest.raises(AssertionError):
        make_transformer(h=5, n_features=136)                                  # d_model % h (:179)


def test_cpu_tensors_are_refused():
    from architeture.multiLayer import make_model
    from ltr_mi355x._lib import LtrDeviceError
    net = make_model(dict(sizes=[16], input_norm=False, activation=None, dropout=0.0), None, dict(d_output=1), 8)
    with pytest.raises(LtrDeviceError):
        net(torch.randn(2, 5, 8), None, None)


def test_gemm_descriptor_layout_matches_the_header():
    """ctypes mirror of `struct ltr_gemm_desc` (include/ltr_encoder.h): same field order, 144 bytes on LP64."""
    import ctypes
    import re
    from conftest import ROOT
    from ltr_mi355x.encoder import GemmDesc
    import os
    hdr = open(os.path.join(ROOT, "include", "ltr_encoder.h")).read()
    body = hdr[hdr.index("typedef struct ltr_gemm_desc {"):hdr.index("} ltr_gemm_desc;")]
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        m = re.match(r"(?:const\s+)?(?:uint16_t|uint64_t|int64_t|int32_t|float)\s+(.*)", decl.strip())
        if m:
            names += [n.strip(" *") for n in m.group(1).split(",")]
    assert names == [f[0] for f in GemmDesc._fields_], names
    assert ctypes.sizeof(GemmDesc) == 144


def test_round3_fixtures_are_complete():
    """tests/golden/make_golden_r4.py: the benched config-5 network (weights by seed) and the stand-alone blocks -- every
    array the GPU tests read exists with the declared shape; the block modules expose the reference's state_dict keys."""
    from architeture import transformer as TR
    g5 = golden("encoder_c5")
    case = g5.cases[0]
    assert (case["B"], case["S"], case["n_features"]) == (2, 256, 136) and case["transformer"]["N"] == 6
    assert g5.arr(case, "scores").shape == (2, 256) and g5.arr(case, "mask").sum() == 29
    for k, sh in case["shapes"]:
        if len(sh) == 1:
            assert g5.arr(case, "g/" + k).shape == tuple(sh)
        else:
            assert g5.arr(case, "grows/" + k).shape == (min(4, sh[0]), sh[1]) and g5.arr(case, "gstat/" + k).shape == (3,)
    gb = golden("blocks")
    ids = {c["id"] for c in gb.cases}
    assert {"LayerNorm", "SublayerConnection_ffn", "MultiHeadedAttention_self", "MultiHeadedAttention_cross", "attention_fn",
            "PositionwiseFeedForward", "EncoderLayer", "Encoder", "FCModel_norm", "FCModel_plain", "OutputLayer_d1_forward",
            "OutputLayer_d3_forward", "OutputLayer_d3_score", "LTRModel_d3_forward", "LTRModel_d3_score",
            "LTRModel_prepare_for_output"} <= ids
    layer = next(c for c in gb.cases if c["id"] == "EncoderLayer")
    mod = TR.EncoderLayer(32, TR.MultiHeadedAttention(4, 32, 0.1), TR.PositionwiseFeedForward(32, 64, 0.1), 0.1)
    assert list(mod.state_dict().keys()) == layer["keys"]


def test_oracle_reproduces_reference_on_the_benched_config5_network():
    """fp32 oracle vs the stored reference numbers of the 3.6 M-parameter network (scores, loss, vector gradients, matrix rows)."""
    from conftest import seeded_state_dict
    g5 = golden("encoder_c5")
    case = g5.cases[0]
    sd = seeded_state_dict(case["shapes"], case["weight_seed"])
    x, y, mask = (torch.from_numpy(g5.arr(case, n)) for n in ("x", "y", "mask"))
    cfg = EO.config_of(dict(fc_model=case["fc_model"], transformer=case["transformer"]), case["n_features"])
    s, loss, grads = EO.scores_and_grads(sd, x, mask, cfg, lambda v: O.approx_ndcg(v, y.to(v.dtype)), dtype=torch.float32)
    want = g5.arr(case, "scores")
    assert np.abs(s.numpy() - want).max() <= 5e-5 * np.abs(want).max()
    gmax = max(float(np.abs(g5.arr(case, "g/" + k)).max()) for k, sh in case["shapes"] if len(sh) == 1)
    for k, sh in case["shapes"]:
        ref = g5.arr(case, "g/" + k) if len(sh) == 1 else g5.arr(case, "grows/" + k)
        got = grads[k].numpy() if len(sh) == 1 else grads[k][:4].numpy()
        assert np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-3 * gmax) <= 5e-4, k
