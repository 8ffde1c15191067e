"""CPU: the C-ABI library builds, loads and exports every symbol include/*.h declares (no compute calls),
and the host layer refuses CPU tensors instead of falling back."""
import glob
import os
import re

import pytest
import torch


def _declared(root):
    names = set()
    for h in glob.glob(os.path.join(root, "include", "*.h")):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(ltr_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


@pytest.fixture(scope="module")
def handle():
    from ltr_mi355x.build import build
    build(force=False, verbose=False)
    import ltr_mi355x
    return ltr_mi355x.lib()


def test_exports_every_declared_symbol(handle, root):
    import ctypes
    from ltr_mi355x import library_path
    raw = ctypes.CDLL(library_path())
    names = _declared(root)
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing
    assert handle.ltr_abi_version() == 1
    assert handle.ltr_error_string(0) == b"ok"
    assert b"NULL" in handle.ltr_error_string(-1)


def test_bound_prototypes_cover_header(handle, root):
    from ltr_mi355x import _lib, _scorer_protos
    bound = set(_lib._PROTOTYPES) | set(_scorer_protos.PROTOTYPES)
    assert set(_declared(root)) == bound


def test_argument_rejection_without_gpu(handle):
    # launchers validate before touching the device: NULL pointers / bad shapes come back as LTR_ERR_*
    assert handle.ltr_approxndcg_fwd_bwd(None, None, 1, 8, 1.0, 1e-10, -1.0, 1.0, None, None, None) == -1
    assert handle.ltr_reduce_sum_f32(None, 4, 1.0, None, None) == -1
    assert handle.ltr_ordinal_num_blocks(1000) == 4


def test_cpu_tensors_are_refused():
    from ltr_mi355x import LtrDeviceError
    from losses.approxNDCG import approxNDCGLoss
    from losses.listnet import listnetLoss
    from losses.lambdaL import lambdaLoss
    from losses.ordinal import ordinalLoss
    s, y = torch.randn(2, 8), torch.randint(0, 5, (2, 8)).float()
    for call in (lambda: approxNDCGLoss(s, y), lambda: listnetLoss(y, s), lambda: lambdaLoss(s, y),
                 lambda: ordinalLoss(torch.rand(2, 8, 4), y, 4)):
        with pytest.raises(LtrDeviceError):
            call()


def test_host_argument_contract():
    from ltr_mi355x.functional import SCHEME_IDS, _lambda_args, slate_2d
    with pytest.raises(ValueError, match="Reduction logarithm base can be either natural or binary"):
        _lambda_args(1e-10, -1, None, None, 1.0, 10.0, "decimal")
    with pytest.raises(KeyError):
        _lambda_args(1e-10, -1, "nope_scheme", None, 1.0, 10.0, "binary")
    assert _lambda_args(1e-10, -1, "lamdbaRank_scheme", 7, 1.0, 10.0, "natural")[:2] == (3, 7)
    assert len(SCHEME_IDS) == 8
    assert slate_2d(torch.zeros(3, 5, 1), "x").shape == (3, 5)
    with pytest.raises(ValueError):
        slate_2d(torch.zeros(3), "x")


def test_scorer_modules_construct_on_cpu():
    """state_dict keys/shapes of SURVEY 3.3 for the compiled input sizes; other sizes refuse loudly."""
    from architeture.doubleLayer import DoubleLayerNet
    from architeture.tripleLayer import TripleLayerNet
    for F in (136, 64):
        d, t = DoubleLayerNet(F), TripleLayerNet(F)
        assert {k: tuple(v.shape) for k, v in d.state_dict().items()} == {
            "fc1.weight": (F, F), "fc1.bias": (F,), "fc2.weight": (F, F), "fc2.bias": (F,), "fc3.weight": (1, F), "fc3.bias": (1,)}
        assert {k: tuple(v.shape) for k, v in t.state_dict().items()} == {
            "l1.weight": (64, F), "l1.bias": (64,), "l2.weight": (32, 64), "l2.bias": (32,), "l3.weight": (1, 32), "l3.bias": (1,)}
        assert isinstance(d.dropout, torch.nn.Dropout) and d.dropout.p == 0.5
    d = DoubleLayerNet(100)                       # any input size up to 136 runs zero-padded on a compiled geometry
    assert tuple(d.fc2.weight.shape) == (100, 100) and tuple(TripleLayerNet(46).l1.weight.shape) == (64, 46)
    w = DoubleLayerNet(220)                       # wider than the fused kernels' tile: library-GEMM path (scorer.wide_forward), still device-only
    assert tuple(w.fc1.weight.shape) == (220, 220) and tuple(TripleLayerNet(300).l1.weight.shape) == (64, 300)
    from ltr_mi355x import LtrDeviceError
    with pytest.raises(LtrDeviceError):
        TripleLayerNet(136)(torch.zeros(2, 4, 136), None, None)
    with pytest.raises(LtrDeviceError):
        w(torch.zeros(2, 4, 220), None, None)     # no CPU path for the wide networks either


def test_module_surface_matches_reference():
    """Names/signatures of SURVEY.md section 8(b)."""
    import inspect
    import losses
    from losses import approxNDCG, lambdaL, listnet, ordinal
    assert hasattr(losses, "approxNDCG") and hasattr(losses, "lambdaL")
    sig = inspect.signature
    assert list(sig(approxNDCG.approxNDCGLoss).parameters) == ["y_pred", "y_true", "eps", "padded_value_indicator", "alpha"]
    assert list(sig(listnet.listnetLoss).parameters) == ["y_true", "y_predicted", "apply_sigmoid"]
    assert list(sig(lambdaL.lambdaLoss).parameters) == ["y_pred", "y_true", "eps", "padded_value_indicator",
                                                         "weighing_scheme", "k", "sigma", "mu", "reduction", "reduction_log"]
    assert list(sig(lambdaL.lambdaMask).parameters)[-1] == "return_losses"
    for n in ("ndcgLoss1_scheme", "ndcgLoss2_scheme", "lamdbaRank_scheme", "ndcgLoss2PP_scheme", "rankNet_scheme",
              "rankNetWeightedByGTDiff_scheme", "rankNetWeightedByGTDiffPowed_scheme"):
        assert callable(getattr(lambdaL, n))
    assert ordinal.PADDED_Y_VALUE == -1
    assert list(sig(ordinal.ordinalLoss).parameters) == ["y_pred", "y_true", "n", "padded_value_indicator"]
    y = torch.tensor([[0., 2., -1.]])
    assert ordinal.with_ordinals(y, 2).tolist() == [[[0., 0.], [1., 1.], [-1., -1.]]]
