#!/bin/bash
# A/B: SLP vectoriser on/off for every translation unit (loss microbench + headline + two64 + triple lines), then parity tests on the no-SLP build.
OUT=gpurun_out; mkdir -p $OUT
for v in fp32_slp fp32_noslp; do
  LTR_LIB=$PWD/variants/$v.so timeout -k 10 200 python tools/bench_losses.py > $OUT/slp_losses_$v.jsonl 2>$OUT/slp_losses_$v.err || exit 1
  echo "== $v"; cat $OUT/slp_losses_$v.jsonl | cut -c1-400
done
for round in 1 2; do
  for v in fp32_slp fp32_noslp; do
    for net in double two64 triple; do
      LTR_LIB=$PWD/variants/$v.so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --net $net --no-cpu-baseline --no-extras > $OUT/slp_${v}_${net}_$round.log 2>&1 || exit 1
      echo "$v $net r$round $(tail -1 $OUT/slp_${v}_${net}_$round.log | python3 -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])')"
    done
  done
done
LTR_LIB=$PWD/variants/fp32_noslp.so timeout -k 10 700 python -m pytest tests -m gpu -q -x > $OUT/slp_tests.log 2>&1; echo "[tests noslp] exit $?"; tail -2 $OUT/slp_tests.log | cut -c1-200
