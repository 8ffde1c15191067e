#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for round in 1 2; do
  for v in f16x2_slp f16x2_noslp; do
    for net in double two64 triple; do
      LTR_LIB=$PWD/variants/$v.so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --net $net --no-cpu-baseline --no-extras > $OUT/slp_${v}_${net}_$round.log 2>&1 || exit 1
      echo "$v $net r$round $(tail -1 $OUT/slp_${v}_${net}_$round.log | python3 -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])')"
    done
  done
done
LTR_LIB=$PWD/variants/f16x2_noslp.so timeout -k 10 800 python -m pytest tests -m gpu -q > $OUT/slp_tests_f16x2.log 2>&1; echo "[tests f16x2 noslp] exit $?"; tail -3 $OUT/slp_tests_f16x2.log | cut -c1-200
