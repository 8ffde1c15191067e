"""pytest plumbing: markers, import paths, golden-fixture access.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI export check, gloo world_size-2 DP tests.
`-m gpu`      : the parity tests proper -- HIP path (through the C-ABI) vs oracle / golden vectors.
The oracle (oracle/) is put on sys.path HERE only: it is test infrastructure, never product code.
"""
import json
import os
import sys

import numpy as np
import pytest

# The product picks the GEMM FFN path below 8 193 tokens per step (ltr_mi355x.encoder.fused_ffn_enabled); most test networks are that
# small, so the suite forces the fused FFN kernels on (they are what the benched configuration runs) and
# test_small_steps_take_the_gemm_ffn_path_by_default covers the default selection and the agreement of the two paths.
os.environ.setdefault("LTR_ENC_FUSED_FFN", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """One fixture group: manifest cases + lazily loaded arrays (numpy.load, allow_pickle=False)."""

    def __init__(self, group):
        self.cases = _manifest()[group]
        self.npz = np.load(os.path.join(GOLDEN, f"{group}.npz"), allow_pickle=False)

    def arr(self, case, name):
        return self.npz[f"{case['id']}/{name}"]

    def has(self, case, name):
        return f"{case['id']}/{name}" in self.npz.files


_cache = {}


def golden(group):
    if group not in _cache:
        _cache[group] = Golden(group)
    return _cache[group]


def _manifest():
    """manifest.json (round 1: approx / listnet / lambda / ordinal / scorers) + manifest_r2.json (risk / metrics) +
    manifest_r3.json (encoder) + manifest_r4.json (encoder_c5: the benched config-5 network; blocks: every building block
    of transformer.py / multiLayer.py called on its own)."""
    out = {}
    for name in ("manifest.json", "manifest_r2.json", "manifest_r3.json", "manifest_r4.json"):
        with open(os.path.join(GOLDEN, name)) as f:
            out.update({k: v for k, v in json.load(f).items() if not k.startswith("_")})
    return out


def golden_cases(group):
    return _manifest()[group]


def relerr(a, b):
    """max|a-b| / max|b| -- the parity metric fixed in SURVEY 8(c) / BASELINE.md (1e-5 bar)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max()) / max(float(np.abs(b).max()), 1e-30)


def seeded_state_dict(shapes, seed):
    """The weight recipe of tests/golden/make_golden_r4.py (numpy PCG64): the config-5 fixture stores the seed, not 3.6 M
    weights."""
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape in shapes:
        shape = tuple(shape)
        if len(shape) == 2:
            a = float(np.sqrt(6.0 / (shape[0] + shape[1])))
            v = rng.uniform(-a, a, size=shape)
        else:
            v = 0.1 * rng.standard_normal(size=shape)
            if key.endswith("a_2") or key.endswith("norm.weight"):
                v = 1.0 + v
        out[key] = __import__('torch').from_numpy(v.astype(np.float32))
    return out


@pytest.fixture(scope="session")
def root():
    return ROOT


# ---------------------------------------------------------------------------------------------- parity ledger
# The -m gpu scorer tests record, per test and per quantity, the worst deviation from the fp64 oracle / golden
# vector (max|delta| / max|ref|), the oracle's OWN fp32-vs-fp64 deviation for the same quantity and whether the
# relaxed bar max(1e-5, 4 x fp32 noise) was needed.  Written at session end to gpurun_out/parity_report.json on
# the GPU box (copied to profiles/ afterwards) -- VERDICT r1 item 5.
_LEDGER = []
_current_test = [None]


@pytest.fixture(autouse=True)
def _ledger_test_name(request):
    _current_test[0] = request.node.nodeid
    yield
    _current_test[0] = None


def ledger_record(quantity, err, noise=None, tol=1e-5, note=None, asserted=True):
    """asserted=False: a quantity the test records but does NOT gate (a parameter tensor whose exact gradient is identically
    zero holds rounding noise on both sides: its relative error is meaningless).  Such rows are kept apart from `over_1e-5`."""
    e = {"test": _current_test[0], "quantity": quantity, "rel_err": float(err), "bar": float(tol),
         "over_1e-5": bool(err > 1e-5), "asserted": bool(asserted)}
    if noise is not None:
        e["oracle_fp32_noise"] = float(noise)
        e["relaxed_bar_needed"] = bool(err > tol)
    if note:
        e["note"] = note
    _LEDGER.append(e)


def pytest_sessionfinish(session, exitstatus):
    if not _LEDGER:
        return
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    worst = {}
    for e in _LEDGER:
        k = e["quantity"]
        if k not in worst or e["rel_err"] > worst[k]["rel_err"]:
            worst[k] = e
    # `over_1e-5` lists ASSERTED quantities only; what a test records without gating goes to its own list
    over = sorted((e for e in _LEDGER if e["over_1e-5"] and e.get("asserted", True)), key=lambda e: -e["rel_err"])
    unasserted = sorted((e for e in _LEDGER if not e.get("asserted", True)), key=lambda e: -e["rel_err"])
    with open(os.path.join(out_dir, "parity_report.json"), "w") as f:
        json.dump({"metric": "max|delta| / max|ref| (SURVEY 8c)", "entries": len(_LEDGER), "asserted_entries": len(_LEDGER) - len(unasserted),
                   "exitstatus": int(exitstatus), "over_1e-5_count": len(over), "over_1e-5_needing_the_relaxed_bar": sum(1 for e in over if e.get("relaxed_bar_needed")),
                   "worst_per_quantity": {k: v for k, v in worst.items() if v.get("asserted", True)}, "over_1e-5": over,
                   "recorded_not_asserted (exact value ~0: relative error meaningless)": unasserted, "all": _LEDGER}, f, indent=1)
