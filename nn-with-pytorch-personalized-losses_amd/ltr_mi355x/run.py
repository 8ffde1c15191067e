"""Launcher for unmodified reference scripts:  python -m ltr_mi355x.run main_batch_execution.py <args...>

`python script.py` puts the SCRIPT's directory at sys.path[0], ahead of PYTHONPATH, so a driver started from
inside the reference tree would import its own `losses/` and `architeture/` no matter what PYTHONPATH says.
This launcher runs the script with this package's directory first and the script's directory second: the
hot-path modules resolve here, everything else (architeture.multiLayer, config, utils, losses.exactNDCG ...)
falls through to the script's tree (see losses/__init__.py: overlay, not shadow).

Device placement: the reference never moves anything to a device.  With
LTR_DEFAULT_DEVICE set (e.g. "cuda") the launcher installs it as torch's default device before the script
runs, so `torch.tensor(...)` and `nn.Linear(...)` inside the driver land on the MI355X.
"""
import os
import runpy
import sys


def main(argv):
    if len(argv) < 2:
        raise SystemExit("usage: python -m ltr_mi355x.run <script.py> [args...]")
    script = os.path.abspath(argv[1])
    pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sdir = os.path.dirname(script)
    sys.path[:] = [pkg, sdir] + [p for p in sys.path if p not in ("", pkg, sdir)]
    dev = os.environ.get("LTR_DEFAULT_DEVICE")
    if dev:
        import torch
        torch.set_default_device(dev)
    sys.argv = [script] + list(argv[2:])
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main(sys.argv)
