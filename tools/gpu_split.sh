#!/bin/bash
# Split-precision variant visit: scorer parity tests, phase stamps, bench (variant vs default library).
TAG=${1:-s}; OUT=gpurun_out; mkdir -p $OUT
V=$PWD/nn-with-pytorch-personalized-losses_amd/ltr_mi355x/libltr_mi355x_bf16x3.so
LTR_LIB=$V timeout -k 10 900 python -m pytest tests/test_scorer_gpu.py tests/test_fused_gaps_gpu.py tests/test_dp_gpu.py -q -x > $OUT/${TAG}_tests.log 2>&1; echo "[tests] exit $?"; grep -E "passed|failed" $OUT/${TAG}_tests.log | tail -1
if [ -f variants/lib_split_stamps.so ]; then LTR_LIB=$PWD/variants/lib_split_stamps.so timeout -k 10 120 python tools/phase_stamps.py > $OUT/${TAG}_stamps.log 2>&1; grep net $OUT/${TAG}_stamps.log | python3 -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['net'], r['total_cycles'], r['phases'])"; fi
LTR_LIB=$V timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench.log 2>&1; tail -1 $OUT/${TAG}_bench.log | cut -c1-250
LTR_LIB=$V timeout -k 10 200 python bench.py --steps 20 --warmup 3 --net triple --no-cpu-baseline > $OUT/${TAG}_bench_triple.log 2>&1; tail -1 $OUT/${TAG}_bench_triple.log | cut -c1-250
