#!/bin/bash
# fp32 feature-partitioned two-layer kernel: parity tests, then the two64 bench line against the previous build
OUT=gpurun_out; mkdir -p $OUT; V=${1:-fp32w}
LTR_LIB=$PWD/variants/$V.so timeout -k 10 400 python -m pytest tests/test_two_layer_gpu.py tests/test_scorer_gpu.py -m gpu -q -x > $OUT/fcw32_tests.log 2>&1; echo "[tests] $?"; tail -3 $OUT/fcw32_tests.log | cut -c1-250
for round in 1 2; do
  for v in ${BASE:-fp32q} $V; do
    LTR_LIB=$PWD/variants/$v.so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --net two64 --no-cpu-baseline --no-extras > $OUT/fcw32_${v}_$round.log 2>&1 || exit 1
    echo "$v two64 r$round $(tail -1 $OUT/fcw32_${v}_$round.log | python3 -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"], r.get("roofline"))')"
  done
done
