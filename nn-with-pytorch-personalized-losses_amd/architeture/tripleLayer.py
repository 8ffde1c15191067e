"""MI355X drop-in for architeture/tripleLayer.py of the reference (TripleLayerNet, :5-17)."""
from torch import nn

from ltr_mi355x import scorer as _scorer


class TripleLayerNet(nn.Module):
    """l1 (n -> 64, no activation), l2 (64 -> 32), sigmoid, l3 (32 -> 1); state_dict keys `l1`, `l2`, `l3`.
    `forward(x, mask, indices)` ignores mask/indices like the reference.  Input [batch, slate, n] fp32 on
    the device, output [batch, slate, 1]; one HIP launch, fp32 MFMA."""
    _ltr_net = _scorer.NET_TRIPLE
    _ltr_dropout = False

    def __init__(self, N_features):
        super(TripleLayerNet, self).__init__()
        self._ltr_net = _scorer.net_id("triple", N_features)      # 136 (MSLR-WEB) or 64 (TD2003) features
        self.l1 = nn.Linear(N_features, 64)
        self.l2 = nn.Linear(64, 32)
        self.l3 = nn.Linear(32, 1)

    def _ltr_params(self):
        return [self.l1.weight, self.l1.bias, self.l2.weight, self.l2.bias, self.l3.weight, self.l3.bias]

    def forward(self, x, mask, indices):
        if self._ltr_net == _scorer.NET_WIDE:               # more than 136 features: library GEMMs, l2 . l1 folded (scorer.wide_forward)
            return _scorer.wide_forward("triple", self._ltr_params(), x)
        return _scorer.mlp_scores(self._ltr_net, self._ltr_params(), x)
