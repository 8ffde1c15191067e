"""Drop-in for the reference's `architeture` package [sic]: the FC scorers of the listwise-LTR hot path,
as nn.Modules whose forward/backward run in the gfx950 slate-pipeline kernels (csrc/ltr_scorer.hip)."""
