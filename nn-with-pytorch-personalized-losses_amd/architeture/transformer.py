"""MI355X drop-in for architeture/transformer.py of the reference (the Annotated-Transformer encoder used as a
slate encoder, :29-257).  Same class names, constructor signatures and state_dict keys; the arithmetic runs in
the kernels of csrc/ltr_encoder.hip.  Inside an `LTRModel` (architeture/multiLayer.py -> ltr_mi355x/encoder.py) the
whole network is ONE fused forward/backward; every block can also be called on its own like the reference's modules
(ltr_mi355x/blocks.py: one autograd node per block over the same C-ABI entry points, gradients to parameters and inputs).
Device tensors only.  `mask`: per-document padding ([batch, slate], optionally with singleton head / query axes)."""
import copy
import math

import torch
import torch.nn as nn


def clones(module, N):
    """N identical (deep-copied) layers (transformer.py:19-26)."""
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def _blocks():
    from ltr_mi355x import blocks
    return blocks


def _enc_spec(d, n_layers, h, d_ff, dropout):
    from ltr_mi355x import encoder as _enc
    return _enc.EncoderSpec(n_features=d, fc_sizes=[], input_norm=False, fc_dropout=0.0, n_layers=n_layers, heads=h, d_ff=d_ff,
                            enc_dropout=dropout, has_encoder=True)


class LayerNorm(nn.Module):
    """a_2 * (x - mean) / (std + eps) + b_2 with torch's UNBIASED std (transformer.py:64-88)."""

    def __init__(self, features, eps=1e-6):
        super(LayerNorm, self).__init__()
        self.a_2 = nn.Parameter(torch.ones(features))
        self.b_2 = nn.Parameter(torch.zeros(features))
        self.eps = eps

    def forward(self, x):
        """transformer.py:78-88 through ltr_enc_layernorm_fwd / _bwd (fp32 in, fp32 out)."""
        return _blocks().layer_norm(x, self.a_2, self.b_2, self.eps).to(x.dtype)


class SublayerConnection(nn.Module):
    """x + dropout(sublayer(norm(x))) (transformer.py:91-114)."""

    def __init__(self, size, dropout):
        super(SublayerConnection, self).__init__()
        self.norm = LayerNorm(size)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, sublayer):
        """transformer.py:106-114.  `sublayer` is an arbitrary callable, so this block is the composition it is in the
        reference: the HIP LayerNorm node, the callable, this module's own dropout, the residual add."""
        return x + self.dropout(sublayer(self.norm(x)))


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
        """transformer.py:187-212.  `self.attn` (the attention map the reference keeps, :207; nothing reads it) is not
        materialised on this path: it stays None -- `attention()` returns the map when it is wanted."""
        B = _blocks()
        nb, S = query.shape[0], query.shape[1]
        if _general_shapes(query, key, value, mask, nb, S, 1):
            # key / value sets unlike the query set, or a mask that is not one flag per document (the reference's own callers pass
            # neither): the reference's formulation (:196-212) on the device, as library GEMMs + ATen softmax -- see attention()
            B.require_device(query, key, value)
            m = mask.unsqueeze(1) if mask is not None else None
            q, k, v = [lin(x).view(nb, -1, self.h, self.d_k).transpose(1, 2) for lin, x in zip(self.linears, (query, key, value))]
            x, self.attn = _attention_general(q, k, v, m, self.dropout)
            return self.linears[-1](x.transpose(1, 2).contiguous().view(nb, -1, self.h * self.d_k))
        m8 = B.slate_mask(mask, nb, S, query.device)
        p = self.dropout.p if self.training else 0.0
        same = key is query and value is query
        prm = [t for lin in self.linears for t in (lin.weight, lin.bias)]
        return B.MultiHeadFn.apply(self.h, p, B.fresh_seed() if p > 0 else 0, same, query, key, value, m8, *prm).to(query.dtype)


class PositionwiseFeedForward(nn.Module):
    """w_2(dropout(relu(w_1 x))) (transformer.py:215-237)."""

    def __init__(self, d_model, d_ff, dropout=0.1):
        super(PositionwiseFeedForward, self).__init__()
        self.w_1 = nn.Linear(d_model, d_ff)
        self.w_2 = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        """transformer.py:231-237."""
        B = _blocks()
        p = self.dropout.p if self.training else 0.0
        return B.FeedForwardFn.apply(p, B.fresh_seed() if p > 0 else 0, x, self.w_1.weight, self.w_1.bias, self.w_2.weight,
                                     self.w_2.bias).to(x.dtype)


class EncoderLayer(nn.Module):
    """Self-attention block + feed-forward block, each behind a pre-norm residual (transformer.py:117-142)."""

    def __init__(self, size, self_attn, feed_forward, dropout):
        super(EncoderLayer, self).__init__()
        self.self_attn = self_attn
        self.feed_forward = feed_forward
        self.sublayer = clones(SublayerConnection(size, dropout), 2)
        self.size = size

    def forward(self, x, mask):
        """transformer.py:134-142 as ONE autograd node (both pre-norm residual sublayers; the fused-FFN kernels where the
        shape allows)."""
        B = _blocks()
        at, ff = self.self_attn, self.feed_forward
        # the fused node runs ONE dropout probability at its four sites (attention probabilities, FFN hidden layer, both
        # residual sublayers) -- what make_model builds (multiLayer.py / transformer.py:240-257).  A layer assembled with different
        # probabilities takes the composed per-block path, where every module applies its own p like the reference.
        ps = {float(self.sublayer[0].dropout.p), float(self.sublayer[1].dropout.p), float(at.dropout.p), float(ff.dropout.p)}
        if len(ps) > 1:
            x = self.sublayer[0](x, lambda t: at(t, t, t, mask))
            return self.sublayer[1](x, ff)
        spec = _enc_spec(self.size, 1, at.h, ff.w_1.out_features, self.sublayer[0].dropout.p)
        m = B.slate_mask(mask, x.shape[0], x.shape[1], x.device)
        if m is None:
            m = torch.zeros(x.shape[:2], dtype=torch.uint8, device=x.device)
        return B.Features.apply(spec, False, x, m, B.fresh_seed() if self.training else 0, self.training, *self._ltr_params()).to(x.dtype)

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
        """transformer.py:45-59 (position is None, :256) as ONE autograd node: N blocks + the final LayerNorm."""
        B = _blocks()
        if self.position:
            raise NotImplementedError("positional encodings are not built on the HIP path (the reference never creates one)")
        shape = self._ltr_shape()
        spec = _enc_spec(self.layers[0].size, shape["n_layers"], shape["heads"], shape["d_ff"], shape["enc_dropout"])
        if mask is None:
            raise AttributeError("'NoneType' object has no attribute 'unsqueeze'")      # transformer.py:55
        m = B.slate_mask(mask, x.shape[0], x.shape[1], x.device)
        return B.Features.apply(spec, True, x, m, B.fresh_seed() if self.training else 0, self.training, *self._ltr_params()).to(x.dtype)

    def _ltr_params(self):
        out = []
        for layer in self.layers:
            out += layer._ltr_params()
        return out + [self.norm.a_2, self.norm.b_2]

    def _ltr_shape(self):
        l0 = self.layers[0]
        return dict(n_layers=len(self.layers), heads=l0.self_attn.h, d_ff=l0.feed_forward.w_1.out_features,
                    enc_dropout=l0.sublayer[0].dropout.p)


def _general_shapes(query, key, value, mask, nb, S, seq_axis):
    """True when the fused attention kernels do not apply: key / value of another shape than the query, or a mask that is not one
    padding flag per document (anything but nb * S elements)."""
    if tuple(key.shape) != tuple(query.shape) or tuple(value.shape) != tuple(query.shape):
        return True
    return mask is not None and mask.numel() != nb * S


def _attention_general(query, key, value, mask, dropout):
    """transformer.py:154-163 literally, on device tensors: rocBLAS batched GEMMs + ATen softmax / dropout with autograd's backward.
    The shapes the reference's own callers use never come here (they run the fused HIP kernels)."""
    d_k = query.size(-1)
    scores = torch.matmul(query, key.transpose(-2, -1)) / math.sqrt(d_k)
    if mask is not None:
        scores = scores.masked_fill(mask == 1, float("-inf"))
    p_attn = torch.softmax(scores, dim=-1)
    if dropout is not None:
        p_attn = dropout(p_attn)
    return torch.matmul(p_attn, value), p_attn


def attention(query, key, value, mask=None, dropout=None):
    """transformer.py:145-164: query / key / value [batch, heads, slate, d_k] -> (output [batch, heads, slate, d_k], p_attn
    [batch, heads, slate, slate]).  `dropout`: None or an nn.Dropout module (applied to p_attn when it is in training mode).
    The output carries the analytic backward (ltr_enc_attention_bwd); p_attn is returned detached.  Key / value sets shaped unlike the
    query set and masks other than one flag per document take the reference's formulation on the device (_attention_general)."""
    B = _blocks()
    nb, h, S, dk = query.shape
    if _general_shapes(query, key, value, mask, nb, S, 2):
        B.require_device(query, key, value)
        return _attention_general(query, key, value, mask, dropout)
    m8 = B.slate_mask(mask, nb, S, query.device)
    p = float(dropout.p) if (dropout is not None and getattr(dropout, "training", False)) else 0.0
    seed = B.fresh_seed() if p > 0 else 0
    out = B.AttentionCoreFn.apply(query, key, value, m8, p, seed)
    return out.to(query.dtype), B.attention_probs(query, key, m8, p, seed).to(query.dtype)


def make_transformer(N=6, d_ff=2048, h=8, dropout=0.1, n_features=136, positional_encoding=None):
    """transformer.py:240-257 (positional_encoding is accepted and ignored there, :256)."""
    c = copy.deepcopy
    attn = MultiHeadedAttention(h, n_features, dropout)
    ff = PositionwiseFeedForward(n_features, d_ff, dropout)
    position = None
    return Encoder(EncoderLayer(n_features, c(attn), c(ff), dropout), N, position)
