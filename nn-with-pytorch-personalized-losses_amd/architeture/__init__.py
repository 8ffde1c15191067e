"""Drop-in for the reference's `architeture` package [sic]: the FC scorers of the listwise-LTR hot path,
as nn.Modules whose forward/backward run in the gfx950 slate-pipeline kernels (csrc/ltr_scorer.hip).

OVERLAY, not shadow (see losses/__init__.py): `architeture.doubleLayer` / `architeture.tripleLayer` resolve
here; modules this package does not provide (e.g. the caller's `architeture.multiLayer`,
main_batch_execution.py:11) fall through to the same-named package directories later on sys.path."""
import pkgutil as _pkgutil

__path__ = _pkgutil.extend_path(__path__, __name__)
