#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary python tool; prints the top kernels.  usage: bash tools/gpu_prof_cmd.sh TAG tools/x.py [args]
TAG=$1; shift
REPO=$(pwd); OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/${TAG}_prof -- python3 "$REPO/$1" "${@:2}" > $REPO/$OUT/${TAG}_prof.log 2>&1
echo "[prof] exit $?"
cd $REPO
python3 - "$OUT/${TAG}_prof" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {r["Percentage"]}')
PY
