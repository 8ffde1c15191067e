"""MI355X drop-in for architeture/multiLayer.py of the reference (allRank-style `make_model`, :13-149): FCModel input
block -> optional transformer Encoder -> OutputLayer, scored by the bf16 MFMA kernels of csrc/ltr_encoder.hip as one
fused forward/backward (ltr_mi355x/encoder.py).  Same names, signatures, state_dict keys and quirks:
activations are hard-wired to Identity (:29, :105), parameters with dim > 1 get Xavier-uniform (:146-148)."""
import torch
import torch.nn as nn

from architeture.transformer import make_transformer
from ltr_mi355x import encoder as _enc


def first_arg_id(x, *y):
    return x


class FCModel(nn.Module):
    """Linear layers with Identity activation and dropout after each (multiLayer.py:13-51)."""

    def __init__(self, sizes, input_norm, activation, dropout, n_features):
        super(FCModel, self).__init__()
        sizes.insert(0, n_features)                 # mutates the caller's list, like the reference (:27)
        layers = [nn.Linear(size_in, size_out) for size_in, size_out in zip(sizes[:-1], sizes[1:])]
        self.input_norm = nn.LayerNorm(n_features) if input_norm else nn.Identity()
        self.activation = nn.Identity()
        self.dropout = nn.Dropout(dropout or 0.0)
        self.output_size = sizes[-1]
        self.layers = nn.ModuleList(layers)

    def forward(self, x):
        """multiLayer.py:36-46 on its own: input norm, then Linear + dropout per layer, as ONE autograd node (gradients to
        the parameters and to x)."""
        from ltr_mi355x import blocks as B
        spec = _enc.EncoderSpec(n_features=x.shape[-1], fc_sizes=[l.out_features for l in self.layers],
                                input_norm=isinstance(self.input_norm, nn.LayerNorm), fc_dropout=self.dropout.p, n_layers=0, heads=1,
                                d_ff=8, enc_dropout=0.0, has_encoder=False)
        x3 = x if x.dim() == 3 else x.reshape(1, -1, x.shape[-1])
        out = B.Features.apply(spec, False, x3, None, B.fresh_seed() if self.training else 0, self.training, *self._ltr_params())
        return out.reshape(*x.shape[:-1], out.shape[-1]).to(x.dtype)

    def _ltr_params(self):
        out = [self.input_norm.weight, self.input_norm.bias] if isinstance(self.input_norm, nn.LayerNorm) else []
        for lin in self.layers:
            out += [lin.weight, lin.bias]
        return out


class OutputLayer(nn.Module):
    """w_1: d_model -> d_output, Identity activation, squeeze(dim=2) (multiLayer.py:94-124)."""

    def __init__(self, d_model, d_output, output_activation=None):
        super(OutputLayer, self).__init__()
        self.activation = nn.Identity()
        self.d_output = d_output
        self.w_1 = nn.Linear(d_model, d_output)

    def forward(self, x):
        """multiLayer.py:107-113: w_1(x).squeeze(dim=2) -> [batch, slate] for d_output = 1, [batch, slate, d_output] otherwise."""
        from ltr_mi355x import blocks as B
        fn = B.score_linear if self.d_output == 1 else B.linear
        return self.activation(fn(x, self.w_1.weight, self.w_1.bias).to(x.dtype).squeeze(dim=2))

    def score(self, x):
        """multiLayer.py:115-124: the individual outputs summed when d_output > 1."""
        if self.d_output > 1:
            return self.forward(x).sum(-1)
        return self.forward(x)


class LTRModel(nn.Module):
    """input_layer -> encoder -> output_layer (multiLayer.py:54-91).  forward / score return scores [batch, slate]
    (d_output = 1: `squeeze(dim=2)`, :113) computed by one fused HIP forward; backward() fills every parameter's .grad.
    `mask` [batch, slate]: 1 / True marks padded documents (transformer.py:158-159); it may be None only without an
    encoder (the reference dereferences it, transformer.py:55).  `indices` is unused (position is None, :256)."""

    def __init__(self, input_layer, encoder, output_layer):
        super(LTRModel, self).__init__()
        self.input_layer = input_layer if input_layer else nn.Identity()
        self.encoder = encoder if encoder else first_arg_id
        self.output_layer = output_layer
        self.ltr_seed = 0x5EED          # base of the counter-based dropout streams; bumped on every training forward
        self._ltr_calls = 0

    def _ltr_spec(self, n_features):
        fc = self.input_layer if isinstance(self.input_layer, FCModel) else None
        enc = self.encoder if isinstance(self.encoder, nn.Module) else None
        shape = enc._ltr_shape() if enc is not None else dict(n_layers=0, heads=1, d_ff=8, enc_dropout=0.0)
        return _enc.EncoderSpec(n_features=n_features,
                                fc_sizes=[l.out_features for l in fc.layers] if fc is not None else [],
                                input_norm=fc is not None and isinstance(fc.input_norm, nn.LayerNorm),
                                fc_dropout=fc.dropout.p if fc is not None else 0.0,
                                has_encoder=enc is not None, **shape)

    def _ltr_params(self):
        out = self.input_layer._ltr_params() if isinstance(self.input_layer, FCModel) else []
        if isinstance(self.encoder, nn.Module):
            out += self.encoder._ltr_params()
        return out + [self.output_layer.w_1.weight, self.output_layer.w_1.bias]

    def _ltr_next_seed(self):
        if self.training:
            self._ltr_calls += 1
        return (self.ltr_seed + 0x9E3779B97F4A7C15 * self._ltr_calls) & (2 ** 64 - 1)

    def prepare_for_output(self, x, mask, indices):
        """Encoder output [batch, slate, d_model] (multiLayer.py:64-72), one autograd node (gradients to every parameter
        below the output layer and to x)."""
        spec = self._ltr_spec(x.shape[-1])
        return _enc.encoder_features(spec, x, mask, self._ltr_next_seed(), self.training, self._ltr_params())

    def forward(self, x, mask, indices):
        """multiLayer.py:74-81.  d_output = 1: the whole network as one fused node -> [batch, slate]; d_output > 1: the
        encoder node followed by the output layer's own node -> [batch, slate, d_output]."""
        if self.output_layer.d_output != 1:
            return self.output_layer(self.prepare_for_output(x, mask, indices))
        spec = self._ltr_spec(x.shape[-1])
        return _enc.EncoderScores.apply(spec, x, mask, self._ltr_next_seed(), self.training, *self._ltr_params())

    def score(self, x, mask, indices):
        """multiLayer.py:83-91: scores [batch, slate] (outputs summed when d_output > 1)."""
        if self.output_layer.d_output != 1:
            return self.output_layer.score(self.prepare_for_output(x, mask, indices))
        return self.forward(x, mask, indices)


def _as_kwargs(cfg):
    if isinstance(cfg, dict):
        return dict(cfg)
    try:
        import attr
        if attr.has(type(cfg)):
            return attr.asdict(cfg, recurse=False)
    except ImportError:
        pass
    return dict(vars(cfg))


def make_model(fc_model, transformer, post_model, n_features):
    """multiLayer.py:127-149.  `fc_model` / `post_model`: dicts; `transformer`: falsy, a dict, or the config's attrs
    object (N, d_ff, h, dropout, positional_encoding)."""
    if fc_model:
        fc_model = FCModel(**fc_model, n_features=n_features)
    d_model = n_features if not fc_model else fc_model.output_size
    if transformer:
        transformer = make_transformer(n_features=d_model, **_as_kwargs(transformer))
    model = LTRModel(fc_model, transformer, OutputLayer(d_model, **post_model))
    for p in model.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)
    return model
