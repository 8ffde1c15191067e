#!/bin/bash
# A/B: default in-tree library vs every variants/*.so (same ABI), two interleaved rounds of the phase timing;
# scorer parity tests on each variant first.
OUT=gpurun_out; TAG=${1:-ab}; mkdir -p $OUT
for so in variants/*.so; do
  LTR_LIB=$PWD/$so timeout -k 10 600 python -m pytest tests/test_scorer_gpu.py -m gpu -q -x > $OUT/${TAG}_$(basename $so).tests.log 2>&1; echo "[tests $so] exit $?"; tail -1 $OUT/${TAG}_$(basename $so).tests.log
done
for round in 1 2; do
  echo "== default"; timeout -k 10 120 python tools/bench_phases.py 2>&1 | grep net
  for so in variants/*.so; do echo "== $so"; LTR_LIB=$PWD/$so timeout -k 10 120 python tools/bench_phases.py 2>&1 | grep net; done
done
