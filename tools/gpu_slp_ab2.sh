#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for v in fp32r fp32_noslp; do
LTR_LIB=$PWD/variants/$v.so timeout -k 10 300 python -m pytest tests/test_dp_gpu.py -m gpu -q -s -k make_model > $OUT/slp_dp_$v.log 2>&1; echo "[dp $v] exit $?"; grep "dp make_model\|passed\|failed" $OUT/slp_dp_$v.log | cut -c1-200
done
LTR_LIB=$PWD/variants/fp32_noslp.so timeout -k 10 800 python -m pytest tests -m gpu -q > $OUT/slp_tests.log 2>&1; echo "[tests noslp] exit $?"; tail -5 $OUT/slp_tests.log | cut -c1-200
