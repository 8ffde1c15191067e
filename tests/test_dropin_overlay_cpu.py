"""CPU: the documented drop-in recipe -- this package FIRST on PYTHONPATH, the
caller's tree behind it -- must let the caller's own import block resolve: hot-path modules from here, every
other `losses.*` / `architeture.*` module from the caller's tree (overlay, not shadow).

The caller's tree is a throw-away STUB built in tmp_path (dummy modules with the names the reference's
drivers import at main_batch_execution.py:10-19); nothing of the reference is read."""
import os
import subprocess
import sys
import textwrap

from conftest import PKG

STUB = {
    "architeture/__init__.py": "",
    "architeture/multiLayer.py": "STUB = True\ndef make_model(*a, **k):\n    return 'stub-make_model'\n",     # shadowed (row f-3)
    "architeture/customNet.py": "def make_custom(*a, **k):\n    return 'stub-make_custom'\n",               # falls through
    "architeture/doubleLayer.py": "class DoubleLayerNet:\n    STUB = True\n",          # must be shadowed
    "config.py": "class Config:\n    pass\n",
    "losses/__init__.py": "from losses import approxNDCG\nfrom losses import exactNDCG\nfrom losses import lambdaL\n"
                          "from losses import orderScore\nfrom losses.riskLosses import riskLosses\n",
    "losses/approxNDCG.py": "STUB = True\n",                                               # must be shadowed
    "losses/lambdaL.py": "STUB = True\n",
    "losses/exactNDCG.py": "def ndcgLoss(*a):\n    return 'stub-exact'\n",
    "losses/orderScore.py": "def orderScoreLoss(*a):\n    return 'stub-order'\n",
    "losses/extraExperiment.py": "VALUE = 41\n",
    "losses/riskLosses/riskFunctions.py": "STUB = True\ndef geoRisk(*a, **k): pass\ndef zRisk(*a, **k): pass\n",
    "losses/riskLosses/riskLosses.py": "STUB = True\ndef geoRiskListnetLoss(*a, **k): pass\n",
    # `utils` is a NAMESPACE package in the reference (no __init__.py): the overlay must merge with that too
    "utils/dataset.py": "STUB = True\ndef get_data(*a): pass\ndef svmDataset(*a): pass\ndef get_baseline_data(*a): pass\n",
    "utils/metrics.py": "STUB = True\ndef mNdcg(*a): pass\n",
    "utils/computeMetrics.py": "VALUE = 7\n",
}

# the exact import block of the reference's batch driver (main_batch_execution.py:10-19), then checks
CHILD = textwrap.dedent("""
    import os, sys
    from architeture.doubleLayer import DoubleLayerNet
    from architeture.multiLayer import make_model
    from architeture.tripleLayer import TripleLayerNet
    from config import Config
    from losses import *
    from losses.lambdaL import lambdaLoss
    from losses.listnet import listnetLoss
    from losses.riskLosses.riskFunctions import geoRisk
    from utils.dataset import get_data, svmDataset, get_baseline_data
    from utils.metrics import mNdcg
    from architeture.customNet import make_custom
    import losses, architeture, losses.extraExperiment, utils.computeMetrics, utils.metrics, utils.dataset
    import architeture.multiLayer, architeture.transformer
    pkg, stub = sys.argv[1], sys.argv[2]
    inside = lambda m, d: os.path.abspath(sys.modules[m].__file__).startswith(os.path.abspath(d) + os.sep)
    for m in ("losses", "losses.approxNDCG", "losses.lambdaL", "losses.listnet", "architeture",
              "architeture.doubleLayer", "architeture.tripleLayer", "architeture.multiLayer", "architeture.transformer",
              "utils.metrics", "utils.dataset"):
        assert inside(m, pkg), (m, sys.modules[m].__file__)
    # the risk losses (SURVEY.md row f-1) resolve here once this package provides them, else in the caller's tree
    ours = os.path.exists(os.path.join(pkg, "losses", "riskLosses", "riskLosses.py"))
    for m in ("losses.riskLosses.riskLosses", "losses.riskLosses.riskFunctions"):
        assert inside(m, pkg if ours else stub), (m, sys.modules[m].__file__)
        assert hasattr(sys.modules[m], "STUB") != ours
    for m in ("architeture.customNet", "losses.exactNDCG", "losses.orderScore", "losses.extraExperiment", "config",
              "utils.computeMetrics"):
        assert inside(m, stub), (m, sys.modules[m].__file__)
    assert not hasattr(architeture.multiLayer, "STUB") and make_custom() == "stub-make_custom"
    assert losses.extraExperiment.VALUE == 41 and utils.computeMetrics.VALUE == 7
    assert not hasattr(utils.metrics, "STUB") and not hasattr(utils.dataset, "STUB")
    assert not hasattr(DoubleLayerNet, "STUB") and not hasattr(losses.approxNDCG, "STUB")
    # `from losses import *` binds what the reference's losses/__init__.py binds (:1-5)
    for name in ("approxNDCG", "exactNDCG", "lambdaL", "orderScore", "riskLosses"):
        assert name in globals(), name
    assert exactNDCG.ndcgLoss() == "stub-exact" and callable(riskLosses.geoRiskListnetLoss) and callable(geoRisk)
    print("OVERLAY-OK")
""")


def _write_tree(base, files):
    for rel, body in files.items():
        path = os.path.join(base, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(body)


def _run(pythonpath, cwd, args):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(pythonpath))
    return subprocess.run([sys.executable, "-c", CHILD] + args, cwd=cwd, env=env, capture_output=True, text=True,
                          timeout=300)


def test_reference_import_block_resolves_through_the_overlay(tmp_path):
    stub = str(tmp_path / "caller_tree")
    _write_tree(stub, STUB)
    # cwd is NOT the caller's tree: `python -c` puts the cwd at sys.path[0], ahead of PYTHONPATH (second test)
    r = _run([PKG, stub], str(tmp_path), [PKG, stub])
    assert r.returncode == 0 and "OVERLAY-OK" in r.stdout, r.stderr[-3000:]


def test_overlay_also_works_when_the_script_dir_comes_first(tmp_path):
    """`python main_batch_execution.py` puts the script's directory at sys.path[0], AHEAD of PYTHONPATH: the caller's
    own `losses/` would win.  INTEGRATION.md therefore launches through `python -m ltr_mi355x.run script.py ...`,
    which puts this package first; check that launcher."""
    stub = str(tmp_path / "caller_tree")
    _write_tree(stub, STUB)
    script = os.path.join(stub, "driver.py")
    with open(script, "w") as f:
        f.write(CHILD)
    env = dict(os.environ, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, "-m", "ltr_mi355x.run", script, PKG, stub], cwd=stub, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OVERLAY-OK" in r.stdout, r.stderr[-3000:]


def test_standalone_import_without_a_caller_tree():
    """With no caller tree behind it the package still imports (the fall-through names are simply absent)."""
    code = "import losses, architeture; from losses import *; assert 'approxNDCG' in dir() and 'lambdaL' in dir(); print('OK')"
    env = dict(os.environ, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd="/")
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-3000:]
