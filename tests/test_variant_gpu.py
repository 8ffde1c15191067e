"""GPU: the f16 x 2 split-precision variant library (libltr_mi355x_f16x2.so, same C ABI, selected with LTR_LIB) runs the scorer
parity tests AT THE SAME BARS as the default exact-fp32 library -- in a child process, because a process binds one library.
(The evidence visits run the WHOLE -m gpu suite under LTR_LIB: profiles/r03_parity_report_f16x2_variant.json.)"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_f16x2_variant_passes_the_scorer_suite_at_the_fp32_bars():
    from ltr_mi355x.build import variant_path
    so = variant_path("f16x2")
    assert os.path.exists(so), f"{so} not built: __graft_entry__.build() builds it next to the default library"
    env = dict(os.environ, LTR_LIB=so)
    tests = ["tests/test_scorer_gpu.py", "tests/test_two_layer_gpu.py", "tests/test_fused_gaps_gpu.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", *tests], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
    # the child really ran the variant: its kernels are f16 MFMAs, and the library says so
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, sys.argv[1]); import ltr_mi355x; print(ltr_mi355x.library_path())",
                          os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd")], env=env, capture_output=True, text=True, timeout=120)
    assert out.stdout.strip().endswith("libltr_mi355x_f16x2.so"), out.stdout + out.stderr
