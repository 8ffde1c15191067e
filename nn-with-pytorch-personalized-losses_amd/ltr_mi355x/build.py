"""Build libltr_mi355x.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
OUT = os.path.join(HERE, "libltr_mi355x.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-fno-gpu-rdc"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(
        os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not force and not is_stale():
        return OUT
    cmd = [hipcc] + FLAGS + sources() + ["-o", OUT + ".tmp"]
    if verbose:
        print("[ltr build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    build(force=True)
