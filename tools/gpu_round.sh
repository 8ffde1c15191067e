#!/bin/bash
# One GPU-box visit: parity tests, smoke, headline bench, rocprofv3 kernel stats of the same bench command.
# A step that times out / is killed stops the visit (no further GPU step after a hang).
set -o pipefail
TAG=${1:-r}
OUT=gpurun_out
mkdir -p $OUT
run() {  # name, timeout, cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/${TAG}_$name.log 2>&1
  local rc=$?
  echo "[$name] exit $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] TIMED OUT - stopping"; tail -5 $OUT/${TAG}_$name.log; exit 9; fi
  return $rc
}
run tests 900 python -m pytest tests -m gpu -x -q; tail -25 $OUT/${TAG}_tests.log
run smoke 300 python __graft_entry__.py smoke; tail -3 $OUT/${TAG}_smoke.log
run bench 600 python bench.py --steps 20 --warmup 3; tail -2 $OUT/${TAG}_bench.log
run bench_triple 300 python bench.py --steps 20 --warmup 3 --net triple --no-cpu-baseline; tail -1 $OUT/${TAG}_bench_triple.log
run bench_losses 300 python tools/bench_losses.py; tail -9 $OUT/${TAG}_bench_losses.log
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run_prof() {
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/${TAG}_prof -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $REPO/$OUT/${TAG}_prof.log 2>&1
  echo "[prof] exit $?"
}
run_prof
cd $REPO
find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -r head -4 | cut -c1-200
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-24)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $REPO/$OUT/${TAG}_pmc_$n -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$OUT/${TAG}_pmc_$n.log 2>&1
  echo "[pmc $n] exit $?"
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections, json, sys
agg = collections.defaultdict(list)
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for f in glob.glob("gpurun_out/*_pmc_*/**/*counter_collection.csv", recursive=True):
    if "/r" not in f: continue
    for r in csv.DictReader(open(f)):
        if "slate_pipeline" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({k: sum(v)/len(v) for k, v in sorted(agg.items())}))
PY
