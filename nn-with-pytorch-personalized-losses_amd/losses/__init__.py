"""Drop-in for the reference's `losses` package (losses/__init__.py:1-5 imports approxNDCG, exactNDCG,
lambdaL, orderScore, riskLosses -- NOT listnet / ordinal, which callers import explicitly).

OVERLAY, not shadow: this directory goes FIRST on PYTHONPATH, in front of the caller's own tree.  The
hot-path modules (approxNDCG, listnet, lambdaL, ordinal, riskLosses/) resolve here and run as HIP kernels on
the MI355X; every other `losses.*` module the caller's tree has (exactNDCG, orderScore, ...) falls through
to the same-named package directories found later on sys.path, so `from losses import *` and
`from losses.<anything> import ...` of main_batch_execution.py:14-17 keep working unchanged."""
import importlib as _importlib
import pkgutil as _pkgutil

__path__ = _pkgutil.extend_path(__path__, __name__)

from losses import approxNDCG  # noqa: F401,E402
from losses import lambdaL  # noqa: F401,E402

# the reference's __init__ also binds these; exactNDCG / orderScore are outside the hot path and come from the
# caller's tree when it has them (absent -> simply not bound, as a tree without them would behave)
for _name in ("exactNDCG", "orderScore", "riskLosses.riskLosses"):
    try:
        globals()[_name.split(".")[-1]] = _importlib.import_module("losses." + _name)
    except ModuleNotFoundError as _e:
        if _e.name not in ("losses." + _name, "losses." + _name.split(".")[0]):
            raise      # the module exists but one of ITS imports is missing: surface that
del _name
