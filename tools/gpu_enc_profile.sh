#!/bin/bash
# rocprofv3 kernel stats of the BASELINE config 5 step (tools/bench_encoder.py); summary printed + left under gpurun_out/.
# usage (on the GPU box, from the repo root): bash tools/gpu_enc_profile.sh TAG [bench_encoder args]
TAG=${1:-enc}; shift
REPO=$(pwd); OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/${TAG}_prof -- python3 $REPO/tools/bench_encoder.py "$@" > $REPO/$OUT/${TAG}_prof.log 2>&1
echo "[prof] exit $?"
cd $REPO
grep '^{' $OUT/${TAG}_prof.log
python3 - "$OUT/${TAG}_prof" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel_stats.csv"); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time ms", tot / 1e6)
for r in rows[:24]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {r["Percentage"]}')
import shutil; shutil.copy(f[0], f"gpurun_out/{sys.argv[1].split('/')[-1]}_kernel_stats.csv")
PY
