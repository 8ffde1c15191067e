#!/bin/bash
# Round-3 visit: whole -m gpu suite, smoke, headline bench + secondary nets, loss micro-bench.
set -o pipefail
TAG=${1:-r3v}; OUT=gpurun_out; mkdir -p $OUT
run() { local name=$1 to=$2; shift 2; timeout -k 10 $to "$@" > $OUT/${TAG}_$name.log 2>&1; local rc=$?; echo "[$name] exit $rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMED OUT"; tail -5 $OUT/${TAG}_$name.log; exit 9; fi; return $rc; }
run tests 1000 python -m pytest tests -m gpu -q; grep -E "passed|failed" $OUT/${TAG}_tests.log | tail -1; grep -E "^FAILED|^ERROR" $OUT/${TAG}_tests.log | head -20
cp $OUT/parity_report.json $OUT/${TAG}_parity_report.json 2>/dev/null
run smoke 300 python __graft_entry__.py smoke; tail -2 $OUT/${TAG}_smoke.log
run bench 600 python bench.py --steps 20 --warmup 3; tail -1 $OUT/${TAG}_bench.log | cut -c1-400
for net in two64 triple; do
  run bench_$net 200 python bench.py --steps 20 --warmup 3 --net $net --no-cpu-baseline --no-extras; tail -1 $OUT/${TAG}_bench_$net.log | cut -c1-200
done
run bench_losses 300 python tools/bench_losses.py; grep -E "approx|lambda" $OUT/${TAG}_bench_losses.log
