"""MI355X drop-in for architeture/doubleLayer.py of the reference (the live DoubleLayerNet, :54-73)."""
from torch import nn

from ltr_mi355x import scorer as _scorer


class DoubleLayerNet(nn.Module):
    """fc1 (n -> n), ReLU, Dropout(0.5), fc2 (n -> n), ReLU, Dropout(0.5), fc3 (n -> 1).

    Same attributes and state_dict keys as the reference (`fc1`, `fc2`, `fc3`, `dropout`), so its
    checkpoints load unchanged (continue_batch_execution.py:100-101).  `forward(x, c1, c2)` applies
    dropout in training mode, `predict(x, c1, c2)` never does; c1/c2 are ignored, as in the reference.
    Input [batch, slate, n] fp32 on the device; output [batch, slate, 1].  The three layers run as one
    HIP launch (fp32 MFMA, activations stay in registers); dropout uses a counter-based keep stream
    seeded from torch's global seed, since the CPU generator stream cannot be reproduced on the device.
    """
    _ltr_net = _scorer.NET_DOUBLE
    _ltr_dropout = True

    def __init__(self, input_size):
        super().__init__()
        self._ltr_net = _scorer.net_id("double", input_size)      # 136 (MSLR-WEB) or 64 (TD2003) features
        self.fc1 = nn.Linear(input_size, input_size)
        self.fc2 = nn.Linear(input_size, input_size)
        self.fc3 = nn.Linear(input_size, 1)
        self.dropout = nn.Dropout(p=0.5)
        self._ltr_calls = 0

    def _ltr_params(self):
        return [self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self.fc3.weight, self.fc3.bias]

    def forward(self, x, c1, c2, keep1=None, keep2=None):
        train = self.training and self.dropout.p > 0       # (p is 0.5 in the reference, doubleLayer.py:60; any p in [0, 1) runs)
        if self._ltr_net == _scorer.NET_WIDE:               # more than 136 features: library GEMMs (scorer.wide_forward)
            return _scorer.wide_forward("double", self._ltr_params(), x, train, self.dropout.p, keep1, keep2)
        self._ltr_calls += 1
        return _scorer.mlp_scores(self._ltr_net, self._ltr_params(), x, dropout=_scorer.drop_code(train, self.dropout.p),
                                  seed=_scorer.next_seed(self._ltr_calls), keep1=keep1, keep2=keep2)

    def predict(self, x, c1, c2):
        if self._ltr_net == _scorer.NET_WIDE:
            return _scorer.wide_forward("double", self._ltr_params(), x, False)
        return _scorer.mlp_scores(self._ltr_net, self._ltr_params(), x, dropout=False)
