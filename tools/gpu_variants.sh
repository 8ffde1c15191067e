#!/bin/bash
# A/B the pipeline-kernel build variants under variants/*.so (same ABI) on one box, interleaved twice;
# the first variant also runs the scorer parity tests.
OUT=gpurun_out; TAG=${1:-v}; mkdir -p $OUT
first=$(ls variants/*.so | tail -2 | head -1)
LTR_LIB=$PWD/$first timeout -k 10 600 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_vtests.log 2>&1; echo "[tests on $first] exit $?"; tail -3 $OUT/${TAG}_vtests.log
for round in 1 2; do
for so in variants/*.so; do
  echo "== $so" >> $OUT/${TAG}_variants.log
  LTR_LIB=$PWD/$so timeout -k 10 120 python tools/bench_phases.py >> $OUT/${TAG}_variants.log 2>&1 || { echo FAILED $so; exit 9; }
done; done
grep -E "==|double" $OUT/${TAG}_variants.log | paste - - | sed 's/{"net": "double", "skip": 0, //' | sort
grep -E "==|triple" $OUT/${TAG}_variants.log | paste - - | sed 's/{"net": "triple", "skip": 0, //' | sort | awk 'NR%2==1'
