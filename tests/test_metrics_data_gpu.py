"""GPU parity tests for the on-device evaluation metrics (NDCG@k, GeoRisk; utils/metrics.py of the
reference) against the reference's golden vectors and the numpy oracle, and for the device side of the data path
(row f-2: the per-epoch gather, main_batch_execution.py:112-117)."""
import numpy as np
import pytest
import torch

import ltr_metrics_oracle as MO
from conftest import golden, golden_cases, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import ltr_mi355x
    ltr_mi355x.lib()
    return torch.device("cuda:0")


def _inputs(g, case):
    src = next(c for c in g.cases if c["id"] == case["inputs"])
    return g.arr(src, "y_true"), g.arr(src, "y_score")


@pytest.mark.parametrize("case", [c for c in golden_cases("metrics") if c["kind"] == "ndcg"], ids=lambda c: c["id"])
def test_mndcg_golden(case, dev):
    """mNdcg through the reference's own signature, fed what its drivers feed it (numpy arrays)."""
    from utils.metrics import mNdcg
    g = golden("metrics")
    y, s = _inputs(g, case)
    s32 = s.astype(np.float32)                 # the drivers pass the scorer's fp32 outputs
    got = mNdcg(y, s32, k=case["k"], no_relevant=case["no_relevant"], gains=case["gains"], use_numpy=case["use_numpy"])
    assert isinstance(got, list) and len(got) == case["Q"] and isinstance(got[0], float)
    if case["variant"] == "ties" or np.array_equal(s32.astype(np.float64).argsort(axis=1), s.argsort(axis=1)):
        ref = g.arr(case, "per_query") if np.array_equal(s32.astype(np.float64), s) else \
            MO.ndcg_per_query(y, s32, k=case["k"], gains=case["gains"], no_relevant=case["no_relevant"], stable=not case["use_numpy"])
    else:                                      # fp32 rounding merged two scores: the order is the fp32 one
        ref = MO.ndcg_per_query(y, s32, k=case["k"], gains=case["gains"], no_relevant=case["no_relevant"], stable=not case["use_numpy"])
    assert relerr(got, ref) < 1e-12
    assert abs(np.mean(got) - np.mean(ref)) < 1e-12


@pytest.mark.parametrize("Q,S,k", [(1, 1, 5), (3, 7, 10), (1000, 128, 10), (64, 1000, 100), (5, 3000, 5)])
def test_ndcg_vs_oracle_any_shape(Q, S, k, dev):
    from ltr_mi355x.metrics import ndcg_at_k
    from utils.metrics import dcg, ndcg, torchNdcg
    rng = np.random.default_rng(Q + S)
    y = rng.integers(0, 5, size=(Q, S)).astype(np.float64)
    y[0] = 0.0
    s = rng.standard_normal((Q, S)).astype(np.float32)
    if S > 4:
        s[:, 3] = s[:, 1]                      # exact ties
    yd, sd = torch.from_numpy(y).to(dev), torch.from_numpy(s).to(dev)
    for gains in ("linear", "exponential"):
        for nr in (True, False):
            for rev in (False, True):
                got = ndcg_at_k(yd, sd, k=k, no_relevant=nr, gains=gains, reverse_ties=rev)
                assert got.dtype == torch.float64 and got.device.type == "cuda"
                ref = MO.ndcg_per_query(y, s, k=k, gains=gains, no_relevant=nr, stable=not rev)
                assert relerr(got.cpu().numpy(), ref) < 1e-12
    # single-query wrappers and the torch flavour (exponential gains, ideal DCG 0 -> 0 / NaN)
    assert abs(ndcg(y[1 % Q], s[1 % Q], k=k) - MO.ndcg_per_query(y[[1 % Q]], s[[1 % Q]], k=k)[0]) < 1e-12
    assert abs(dcg(y[1 % Q], s[1 % Q], k=k, gains="exponential") - MO.dcg_at_k(y[[1 % Q]], s[[1 % Q]], k, "exponential")[0]) < 1e-9
    t = torchNdcg(yd, sd.unsqueeze(-1), k=k, return_type="tensor")
    ref = MO.ndcg_per_query(y, s, k=k, gains="exponential", no_relevant=False)
    assert t.dtype == torch.float32 and relerr(t.cpu().numpy(), ref) < 1e-6 and float(t[0]) == 0.0
    lst = torchNdcg(yd, sd, k=None, return_type="list")
    assert np.isnan(lst[0]) and relerr(lst[1:], MO.ndcg_per_query(y, s, k=S, gains="exponential", no_relevant=False)[1:]) < 1e-12
    with pytest.raises(ValueError, match="Invalid gains option."):
        ndcg_at_k(yd, sd, gains="quadratic")


@pytest.mark.parametrize("case", [c for c in golden_cases("metrics") if c["kind"] == "georisk"], ids=lambda c: c["id"])
def test_georisk_metric_golden(case, dev):
    from utils.metrics import getGeoRiskDefault
    g = golden("metrics")
    got = getGeoRiskDefault(g.arr(case, "mat"), case["alpha"])
    assert isinstance(got, np.ndarray) and got.shape == g.arr(case, "value").shape
    assert relerr(got, g.arr(case, "value")) < 1e-5
    if case["id"].startswith("georisk_kat_a3"):
        assert abs(got[0] - 0.31438308416523303) < 1e-6        # the reference's KAT, through the numpy-metric flavour too


def test_georisk_metric_zero_guard(dev):
    """A query on which every system scores 0 has e = 0: the numpy metric counts 0 there (metrics.py:30-33)."""
    from utils.metrics import getGeoRiskDefault
    rng = np.random.default_rng(1)
    m = rng.random((40, 4)) * 0.8 + 0.1
    m[7] = 0.0
    m[11] = 0.0
    assert relerr(getGeoRiskDefault(m, 5.0), MO.geo_risk_all_systems(m, 5.0)) < 1e-5


@pytest.mark.parametrize("shape", [(1000, 128, 136), (37, 5, 3), (4096, 32), (513, 128, 2), (10, 7)])
def test_gather_rows(shape, dev):
    from ltr_mi355x.data import EpochShuffler, gather_rows
    gen = torch.Generator(device=dev).manual_seed(shape[0])
    src = torch.randn(shape, device=dev, generator=gen)
    idx = torch.randperm(shape[0], device=dev, generator=gen)
    assert torch.equal(gather_rows(src, idx), src[idx])
    sub = idx[: shape[0] // 3]
    assert torch.equal(gather_rows(src, sub), src[sub])
    assert gather_rows(src, idx[:0]).shape == (0,) + tuple(shape[1:])
    misaligned = torch.randn(shape[0] * int(np.prod(shape[1:])) + 1, device=dev, generator=gen)[1:].view(shape)
    assert torch.equal(gather_rows(misaligned, idx), misaligned[idx])
    # two epochs of the reference's shuffle: every tensor permuted by the same idx, buffers swap
    y = torch.randn(shape[0], 11, device=dev, generator=gen)
    sh = EpochShuffler(src, y)
    cur_x, cur_y = src, y
    for _ in range(2):
        idx2, (nx, ny) = sh.shuffle(generator=gen)
        assert torch.equal(nx, cur_x[idx2]) and torch.equal(ny, cur_y[idx2])
        cur_x, cur_y = nx.clone(), ny.clone()
    with pytest.raises(ValueError):
        gather_rows(src, idx, out=src)
    # negative indices count from the end (torch semantics); an index still out of range gives a zero row (torch asserts)
    neg = torch.tensor([-1, 0, -shape[0], shape[0] - 1], device=dev)
    assert torch.equal(gather_rows(src, neg), src[neg])
    bad = torch.tensor([0, shape[0], -shape[0] - 1, 1 % shape[0]], device=dev)
    got = gather_rows(src, bad)
    assert torch.equal(got[0], src[0]) and torch.equal(got[3], src[1 % shape[0]])
    assert float(got[1].abs().max()) == 0.0 and float(got[2].abs().max()) == 0.0


def test_gather_bandwidth_smoke(dev):
    """C2-sized shard of the epoch gather (25 000 slates x 128 x 136 fp32 = 1.74 GB): correctness on a sample of rows."""
    from ltr_mi355x.data import gather_rows
    gen = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(25_000, 128, 136, device=dev, generator=gen)
    idx = torch.randperm(25_000, device=dev, generator=gen)
    out = gather_rows(X, idx)
    probe = torch.tensor([0, 1, 12_345, 24_999], device=dev)
    assert torch.equal(out[probe], X[idx[probe]])
