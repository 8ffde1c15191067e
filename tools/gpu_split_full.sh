#!/bin/bash
# Whole -m gpu suite against the split-precision variant (same C ABI, LTR_LIB), then stamps + bench lines for both libraries.
TAG=${1:-sf}; OUT=gpurun_out; mkdir -p $OUT
V=$PWD/nn-with-pytorch-personalized-losses_amd/ltr_mi355x/libltr_mi355x_bf16x3.so
LTR_LIB=$V timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/${TAG}_tests_bf16x3.log 2>&1; echo "[tests bf16x3] exit $?"; grep -E "passed|failed" $OUT/${TAG}_tests_bf16x3.log | tail -1
cp $OUT/parity_report.json $OUT/${TAG}_parity_report_bf16x3.json
LTR_LIB=$PWD/variants/lib_split_stamps.so timeout -k 10 120 python tools/phase_stamps.py > $OUT/${TAG}_stamps.log 2>&1; grep net $OUT/${TAG}_stamps.log | python3 -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['net'], r['total_cycles'], r['phases'], r.get('fc1_detail'))"
LTR_LIB=$V timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_bf16x3.log 2>&1; tail -1 $OUT/${TAG}_bench_bf16x3.log | cut -c1-200
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_fp32.log 2>&1; tail -1 $OUT/${TAG}_bench_fp32.log | cut -c1-200
