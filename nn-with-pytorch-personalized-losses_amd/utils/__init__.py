"""Drop-in overlay for the reference's `utils` package (a namespace package there: no __init__.py).  `utils.metrics`
(NDCG@k, GeoRisk) and `utils.dataset` (LETOR loader, baseline files) resolve here and run on the MI355X / the native
parser; every other module (`utils.computeMetrics`, `utils.runSklearn`, ...) falls through to the same-named
directories later on sys.path -- regular packages and namespace portions alike (see losses/__init__.py)."""
import pkgutil as _pkgutil

__path__ = _pkgutil.extend_path(__path__, __name__)
