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
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run_prof() {
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/${TAG}_prof -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $REPO/$OUT/${TAG}_prof.log 2>&1
  echo "[prof] exit $?"
}
run_prof
cd $REPO
find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12
