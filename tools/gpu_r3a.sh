#!/bin/bash
# Round-3 quick visit: loss / scorer parity tests, loss micro-bench, fused bench lines (double / two64 / triple).
TAG=${1:-r3a}; OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_losses_gpu.py tests/test_scorer_gpu.py tests/test_fused_gaps_gpu.py tests/test_two_layer_gpu.py -q -x > $OUT/${TAG}_tests.log 2>&1; rc=$?; echo "[tests] exit $rc"; tail -3 $OUT/${TAG}_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/bench_losses.py > $OUT/${TAG}_bench_losses.log 2>&1 && grep -E "approx" $OUT/${TAG}_bench_losses.log
for net in double two64 triple; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --net $net --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_$net.log 2>&1 && tail -1 $OUT/${TAG}_bench_$net.log | cut -c1-160
done
