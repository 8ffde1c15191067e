#!/bin/bash
# A/B of variant libraries on the headline bench line (two interleaved rounds) + scorer parity tests on the LAST variant named.
OUT=gpurun_out; TAG=$1; shift; mkdir -p $OUT
for round in 1 2; do
  for v in "$@"; do
    LTR_LIB=$PWD/variants/$v.so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --net double --no-cpu-baseline --no-extras > $OUT/${TAG}_${v}_$round.log 2>&1
    echo "$v r$round $(tail -1 $OUT/${TAG}_${v}_$round.log | python3 -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"])' 2>&1 | tail -1)"
  done
done
last="${@: -1}"
LTR_LIB=$PWD/variants/$last.so timeout -k 10 600 python -m pytest tests/test_scorer_gpu.py tests/test_fused_gaps_gpu.py -m gpu -q -x > $OUT/${TAG}_tests.log 2>&1; echo "[tests $last] exit $?"; tail -2 $OUT/${TAG}_tests.log | cut -c1-200
