#!/bin/bash
# Round-4 fused-kernel visit: parity of the fused kernels with the library under test, interleaved A/B bench lines, phase stamps.
#   usage: TESTS="v1 v2" AB="base new ..." NETS="two64 double triple" STAMPS="v_stamps ..." bash tools/gpu_r4_loss.sh TAG
TAG=$1; OUT=gpurun_out; mkdir -p $OUT
for v in $TESTS; do
  LTR_LIB=$PWD/variants/$v.so timeout -k 10 900 python -m pytest tests/test_two_layer_gpu.py tests/test_scorer_gpu.py tests/test_fused_gaps_gpu.py -m gpu -q -x > $OUT/${TAG}_tests_$v.log 2>&1
  rc=$?; echo "[tests $v] exit $rc: $(tail -n 1 $OUT/${TAG}_tests_$v.log | cut -c1-200)"
  [ $rc -ne 0 ] && { grep -E "^E |FAILED" $OUT/${TAG}_tests_$v.log | head -n 12 | cut -c1-250; exit $rc; }
done
for net in ${NETS:-two64 double triple}; do
  for round in 1 2; do
    for v in $AB; do
      LTR_LIB=$PWD/variants/$v.so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --net $net --no-cpu-baseline --no-extras > $OUT/${TAG}_${net}_${v}_$round.log 2>&1 || { echo "bench $net $v failed"; tail -n 5 $OUT/${TAG}_${net}_${v}_$round.log; exit 1; }
      echo "$net $v r$round $(tail -n 1 $OUT/${TAG}_${net}_${v}_$round.log | python3 -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["roofline"]["hbm_frac_of_8TBps"])')"
    done
  done
done
for v in $STAMPS; do
  LTR_LIB=$PWD/variants/$v.so timeout -k 10 120 python tools/fcw_stamps.py > $OUT/${TAG}_fcw_stamps_$v.jsonl 2>&1; echo "== fcw stamps $v"; grep "^{" $OUT/${TAG}_fcw_stamps_$v.jsonl | cut -c1-700
  LTR_LIB=$PWD/variants/$v.so timeout -k 10 120 python tools/phase_stamps.py > $OUT/${TAG}_stamps_$v.jsonl 2>&1; echo "== pipeline stamps $v"; grep "^{" $OUT/${TAG}_stamps_$v.jsonl | python3 -c '
import sys, json
for ln in sys.stdin:
    r = json.loads(ln); print(r["net"], "total", r["total_cycles"], "loss_detail", r["loss_detail"], "phases", r["phases"])'
done
