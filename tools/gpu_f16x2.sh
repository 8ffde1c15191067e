#!/bin/bash
# f16 x 2 variant visit: scorer / fused parity tests with LTR_LIB=<variant>, phase stamps of both builds, bench lines (double / two64 / triple).
TAG=${1:-f16}; V=${2:-f16x2}; OUT=gpurun_out; mkdir -p $OUT
LTR_LIB=$PWD/variants/$V.so timeout -k 10 600 python -m pytest tests/test_scorer_gpu.py tests/test_fused_gaps_gpu.py tests/test_two_layer_gpu.py tests/test_dp_gpu.py -m gpu -q > $OUT/${TAG}_tests.log 2>&1; echo "[tests $V] exit $?"; grep -E "^FAILED|passed|failed" $OUT/${TAG}_tests.log | cut -c1-250 | head -30
cp $OUT/parity_report.json $OUT/${TAG}_parity_report.json 2>/dev/null
LTR_LIB=$PWD/variants/${V}_stamps.so timeout -k 10 200 python tools/phase_stamps.py > $OUT/${TAG}_stamps.jsonl 2>$OUT/${TAG}_stamps.err; echo "[stamps] exit $?"
for net in double two64 triple; do
  LTR_LIB=$PWD/variants/$V.so timeout -k 10 200 python bench.py --steps 20 --warmup 3 --net $net --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_$net.log 2>&1; tail -1 $OUT/${TAG}_bench_$net.log | cut -c1-180
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --net $net --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_${net}_fp32.log 2>&1; tail -1 $OUT/${TAG}_bench_${net}_fp32.log | cut -c1-180
done
