"""MI355X drop-in for architeture/transformer.py of the reference (the Annotated-Transformer encoder used as a
slate encoder, :29-257).  Same class names, constructor signatures and state_dict keys; the arithmetic runs in
the kernels of csrc/ltr_encoder.hip, driven as ONE fused forward/backward by `LTRModel` (architeture/multiLayer.py
-> ltr_mi355x/encoder.py).  The classes below are therefore parameter containers with the reference's layout:
calling an inner block on its own (outside an LTRModel) is not part of the HIP path and raises."""
import copy

import torch
import torch.nn as nn


def clones(module, N):
    """N identical (deep-copied) layers (transformer.py:19-26)."""
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def _inner(name):
    raise NotImplementedError(
        f"{name}.forward on its own is not built on the MI355X path: the encoder runs as one fused forward/backward "
        "inside architeture.multiLayer.LTRModel (see ltr_mi355x/encoder.py)")


class LayerNorm(nn.Module):
    """a_2 * (x - mean) / (std + eps) + b_2 with torch's UNBIASED std (transformer.py:64-88)."""

    def __init__(self, features, eps=1e-6):
        super(LayerNorm, self).__init__()
        self.a_2 = nn.Parameter(torch.ones(features))
        self.b_2 = nn.Parameter(torch.zeros(features))
        self.eps = eps

    def forward(self, x):
        """Forward-only standalone use (fp32 in, fp32 out) through ltr_enc_layernorm_fwd."""
        from ltr_mi355x import encoder as _enc
        d = x.shape[-1]
        T = x.numel() // d
        with torch.cuda.device(x.device), torch.no_grad():
            xf = x.detach().to(torch.float32).contiguous().view(T, d)
            _, y = _enc.layernorm_fwd(xf, self.a_2.detach().float().contiguous(), self.b_2.detach().float().contiguous(), T, d,
                                      self.eps, 0, want_f32=True)
        return y.view(x.shape).to(x.dtype)


class SublayerConnection(nn.Module):
    """x + dropout(sublayer(norm(x))) (transformer.py:91-114)."""

    def __init__(self, size, dropout):
        super(SublayerConnection, self).__init__()
        self.norm = LayerNorm(size)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, sublayer):
        _inner("SublayerConnection")


class MultiHeadedAttention(nn.Module):
    """h heads of scaled dot-product attention over the slate; `linears` = Wq, Wk, Wv, Wo (transformer.py:167-212)."""

    def __init__(self, h, d_model, dropout=0.1):
        super(MultiHeadedAttention, self).__init__()
        assert d_model % h == 0
        self.d_k = d_model // h
        self.h = h
        self.linears = clones(nn.Linear(d_model, d_model), 4)
        self.attn = None
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, query, key, value, mask=None):
        _inner("MultiHeadedAttention")


class PositionwiseFeedForward(nn.Module):
    """w_2(dropout(relu(w_1 x))) (transformer.py:215-237)."""

    def __init__(self, d_model, d_ff, dropout=0.1):
        super(PositionwiseFeedForward, self).__init__()
        self.w_1 = nn.Linear(d_model, d_ff)
        self.w_2 = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        _inner("PositionwiseFeedForward")


class EncoderLayer(nn.Module):
    """Self-attention block + feed-forward block, each behind a pre-norm residual (transformer.py:117-142)."""

    def __init__(self, size, self_attn, feed_forward, dropout):
        super(EncoderLayer, self).__init__()
        self.self_attn = self_attn
        self.feed_forward = feed_forward
        self.sublayer = clones(SublayerConnection(size, dropout), 2)
        self.size = size

    def forward(self, x, mask):
        _inner("EncoderLayer")

    def _ltr_params(self):
        at, ff = self.self_attn, self.feed_forward
        out = [self.sublayer[0].norm.a_2, self.sublayer[0].norm.b_2]
        for lin in at.linears:
            out += [lin.weight, lin.bias]
        out += [self.sublayer[1].norm.a_2, self.sublayer[1].norm.b_2, ff.w_1.weight, ff.w_1.bias, ff.w_2.weight, ff.w_2.bias]
        return out


class Encoder(nn.Module):
    """N encoder blocks and a final LayerNorm; `position` is always None in the reference (transformer.py:29-59, :256)."""

    def __init__(self, layer, N, position):
        super(Encoder, self).__init__()
        self.layers = clones(layer, N)
        self.norm = LayerNorm(layer.size)
        self.position = position

    def forward(self, x, mask, indices):
        _inner("Encoder")

    def _ltr_params(self):
        out = []
        for layer in self.layers:
            out += layer._ltr_params()
        return out + [self.norm.a_2, self.norm.b_2]

    def _ltr_shape(self):
        l0 = self.layers[0]
        return dict(n_layers=len(self.layers), heads=l0.self_attn.h, d_ff=l0.feed_forward.w_1.out_features,
                    enc_dropout=l0.sublayer[0].dropout.p)


def attention(query, key, value, mask=None, dropout=None):
    """transformer.py:145-164; runs inside the fused encoder (ltr_enc_attention_fwd/bwd), not as a free function."""
    _inner("attention")


def make_transformer(N=6, d_ff=2048, h=8, dropout=0.1, n_features=136, positional_encoding=None):
    """transformer.py:240-257 (positional_encoding is accepted and ignored there, :256)."""
    c = copy.deepcopy
    attn = MultiHeadedAttention(h, n_features, dropout)
    ff = PositionwiseFeedForward(n_features, d_ff, dropout)
    position = None
    return Encoder(EncoderLayer(n_features, c(attn), c(ff), dropout), N, position)
