#!/bin/bash
# Quick visit: parity tests, phase timings, bench (both nets), loss micro-bench, optional stamped diagnostic build.
TAG=${1:-q}; OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_tests.log 2>&1; echo "[tests] exit $?"; grep -E "passed|failed" $OUT/${TAG}_tests.log | tail -1
timeout -k 10 120 python tools/bench_phases.py > $OUT/${TAG}_phases.log 2>&1 || { echo "phases failed"; tail -5 $OUT/${TAG}_phases.log; exit 9; }
grep net $OUT/${TAG}_phases.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.log 2>&1; tail -1 $OUT/${TAG}_bench.log | cut -c1-200
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --net triple --no-cpu-baseline > $OUT/${TAG}_bench_triple.log 2>&1; tail -1 $OUT/${TAG}_bench_triple.log | cut -c1-200
timeout -k 10 200 python tools/bench_losses.py > $OUT/${TAG}_bench_losses.log 2>&1; grep -E "approx|lambda" $OUT/${TAG}_bench_losses.log
if [ -f variants/lib_stamps.so ]; then LTR_LIB=$PWD/variants/lib_stamps.so timeout -k 10 120 python tools/phase_stamps.py > $OUT/${TAG}_stamps.log 2>&1; grep net $OUT/${TAG}_stamps.log | cut -c1-330; fi
