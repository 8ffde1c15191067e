#!/bin/bash
# Phase breakdown + PMC counters of the slate pipeline kernel.
set -o pipefail
TAG=${1:-p}
OUT=gpurun_out
mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/${TAG}_tests.log 2>&1; echo "[tests] exit $?"; tail -12 $OUT/${TAG}_tests.log
rc=$?
for skip in 0 1; do
  LTR_DEBUG_SKIP=$skip timeout -k 10 120 python tools/bench_phases.py >> $OUT/${TAG}_phases.log 2>&1 || { echo "phase run $skip failed"; tail -3 $OUT/${TAG}_phases.log; exit 9; }
done
grep -h net $OUT/${TAG}_phases.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.log 2>&1; tail -1 $OUT/${TAG}_bench.log
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --net triple --no-cpu-baseline > $OUT/${TAG}_bench_triple.log 2>&1; tail -1 $OUT/${TAG}_bench_triple.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $REPO/$OUT/${TAG}_counters.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $REPO/$OUT/${TAG}_pmc_$n -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$OUT/${TAG}_pmc_$n.log 2>&1
  echo "[pmc $n] exit $?"
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/*_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "slate_pipeline" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
