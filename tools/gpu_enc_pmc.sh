#!/bin/bash
# PMC counters of one encoder kernel at config-5 shapes.  usage: bash tools/gpu_enc_pmc.sh TAG "<--only filter>" "<kernel name substring>"
TAG=$1; ONLY=$2; KERN=$3
REPO=$(pwd); OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp
n=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  n=$((n+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $REPO/$OUT/${TAG}_pmc_$n -- python3 $REPO/tools/bench_encoder_kernels.py --only "$ONLY" > $REPO/$OUT/${TAG}_pmc_$n.log 2>&1 || { echo "pmc set $n failed"; tail -3 $REPO/$OUT/${TAG}_pmc_$n.log; }
done
cd $REPO
python3 - "$OUT" "$TAG" "$KERN" <<'PY'
import csv, glob, sys, collections
out, tag, kern = sys.argv[1:4]
agg = collections.defaultdict(list)
for f in glob.glob(f"{out}/{tag}_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
