#!/bin/bash
# Phase stamps of the split-precision pipeline under the LTR_DIAG ablation switches (results wrong, timing only).
TAG=${1:-d}; OUT=gpurun_out; mkdir -p $OUT
for skip in ${SKIPS:-0 32 64 96 128}; do
  LTR_DEBUG_SKIP=$skip LTR_LIB=$PWD/variants/lib_split_diag.so timeout -k 10 120 python tools/phase_stamps.py > $OUT/${TAG}_diag_$skip.log 2>&1 || { echo "skip $skip failed"; tail -3 $OUT/${TAG}_diag_$skip.log; exit 9; }
  grep '"double"' $OUT/${TAG}_diag_$skip.log | python3 -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('skip', $skip, r['total_cycles'], list(r['phases'].values()))"
done
