#!/usr/bin/env python3
"""Resource usage (VGPRs, AGPRs, scratch bytes per lane, occupancy) of every kernel of one translation unit, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.   tools/kres.py csrc/ltr_scorer.hip [filter] [-- extra hipcc flags]"""
import re, subprocess, sys
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
src = args[0]; flt = args[1] if len(args) > 1 else ""
cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-fno-slp-vectorize", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
if "error:" in out:
    print("\n".join(l for l in out.splitlines() if "error" in l)); sys.exit(1)
cur = None
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        continue
    for k in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(re.escape(k) + r": (\d+)", l)
        if m and cur is not None:
            cur[k] = int(m.group(1))
            if k.startswith("LDS"):
                if flt in cur["name"]:
                    n = re.sub(r"\(anonymous namespace\)::", "", cur["name"]).split("(")[0]
                    print(f'{cur.get("VGPRs"):4d} v {cur.get("AGPRs"):4d} a {cur.get("ScratchSize [bytes/lane]"):5d} B scratch  occ {cur.get("Occupancy [waves/SIMD]")}  {n[:150]}')
                cur = None
