"""Scorers outside the reference's live class surface, for benchmarking.

`TwoLayerNet(136)` is the two-Linear-layer DoubleLayerNet variant the reference keeps commented out
(architeture/doubleLayer.py:38-51: fc1 136 -> 64, ReLU, fc4 64 -> 1) and BASELINE.json names in configs[0]
("doubleLayer FC (136->64->1)").  Its trailing `log_softmax(dim=1)` is omitted: it shifts every slate's scores by one
constant, and every listwise loss of this package is invariant to that (same loss, same gradients)."""
from torch import nn

from . import scorer as _scorer


class TwoLayerNet(nn.Module):
    """fc1 (136 -> 64), ReLU, fc4 (64 -> 1); attribute names as in the reference's commented variant.  One HIP launch
    (fp32 MFMA); no dropout.  Input [batch, slate, 136] fp32 on the device, output [batch, slate, 1]."""
    _ltr_net = _scorer.NET_TWO_LAYER_64H
    _ltr_dropout = False

    def __init__(self, input_size=136):
        super().__init__()
        if input_size != 136:
            raise NotImplementedError("the two-layer bench scorer is compiled for 136 input features")
        self.fc1 = nn.Linear(input_size, 64)
        self.fc4 = nn.Linear(64, 1)

    def _ltr_params(self):
        return [self.fc1.weight, self.fc1.bias, self.fc4.weight, self.fc4.bias]

    def forward(self, x, c1=None, c2=None):
        return _scorer.mlp_scores(self._ltr_net, self._ltr_params(), x)
