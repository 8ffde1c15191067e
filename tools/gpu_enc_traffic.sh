#!/bin/bash
# HBM-side traffic of one BASELINE config 5 training step: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel
# trace only) over tools/bench_encoder.py, summed over every kernel of the timed steps.  FETCH_SIZE is doubled (gfx950 counts
# wide coalesced reads at half their size, MI355X_MICROARCH.md section HBM).  usage: bash tools/gpu_enc_traffic.sh TAG
TAG=${1:-enct}; REPO=$(pwd); OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $REPO/$OUT/${TAG}_$c -- python3 $REPO/tools/bench_encoder.py --batch 256 --steps 4 --warmup 2 > $REPO/$OUT/${TAG}_$c.log 2>&1
  echo "[$c] exit $?"
done
cd $REPO
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, json, collections
out, tag = sys.argv[1:3]
tot = {}
per = collections.defaultdict(lambda: collections.defaultdict(float))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{tag}_{c}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    tot[c] = sum(float(r["Counter_Value"]) for r in rows)
    for r in rows:
        per[r["Kernel_Name"].split("(")[0][-44:]][c] += float(r["Counter_Value"])
steps = 6          # 2 warm-up + 4 timed steps, all profiled
kib = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / steps
res = {"steps_profiled": steps, "FETCH_SIZE_KiB_per_step": tot["FETCH_SIZE"] / steps, "WRITE_SIZE_KiB_per_step": tot["WRITE_SIZE"] / steps,
       "hbm_GB_per_step": kib * 1024 / 1e9,
       "top_kernels_GB_per_step": {k: round((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / 1e9 / steps, 3)
                                   for k, v in sorted(per.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))[:10]}}
print(json.dumps(res, indent=1))
open(f"{out}/{tag}_traffic.json", "w").write(json.dumps(res, indent=1))
PY
