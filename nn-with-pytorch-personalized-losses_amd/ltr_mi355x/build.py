"""Build libltr_mi355x.so (and its split-precision variant) in-tree with hipcc for gfx950 (cross-compiles without a GPU).

  libltr_mi355x.so        exact fp32 slate pipeline (v_mfma_f32_16x16x4_f32)                    -- the default library
  libltr_mi355x_f16x2.so  same C ABI; the FC scorer GEMMs of the slate pipeline as two-piece f16 splits (hi + lo, three of
                          the four piece products) on v_mfma_f32_16x16x32_f16 with fp32 accumulation (-DLTR_F16X2=1);
                          selected with LTR_LIB=<path> (ltr_mi355x._lib).  Only csrc/ltr_scorer.hip differs.
"""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
OUT = os.path.join(HERE, "libltr_mi355x.so")
# -fno-slp-vectorize: the SLP vectoriser turns independent fp32 chains into v_pk_*_f32 plus operand-pair moves; packed fp32
# has no rate advantage on CDNA4, so the moves are pure cost (measured: approxNDCG S=128 115 -> 132 M slates/s, the fused
# two-layer step 23.1 -> 25.7 M; profiles/r03_variant_ab.json)
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-fno-gpu-rdc", "-fno-slp-vectorize"]
VARIANTS = {"": [], "f16x2": ["-DLTR_F16X2=1"]}


def variant_path(variant=""):
    return OUT if not variant else os.path.join(HERE, f"libltr_mi355x_{variant}.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale(out=OUT):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(
        os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, variant=""):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = variant_path(variant)
    if not force and not is_stale(out):
        return out
    cmd = [hipcc] + FLAGS + VARIANTS[variant] + sources() + ["-o", out + ".tmp"]
    if verbose:
        print("[ltr build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    return out


def build_all(force=False, verbose=True):
    return [build(force, verbose, v) for v in VARIANTS]


if __name__ == "__main__":
    build_all(force=True)
