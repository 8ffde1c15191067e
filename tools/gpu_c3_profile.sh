#!/bin/bash
# BASELINE config 3 evidence: rocprofv3 kernel stats + PMC for the standalone lambdaLoss kernel (ndcgLoss2PP, 8192 x 512).
TAG=${1:-c3}; OUT=gpurun_out; mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/${TAG}_stats -- python3 $REPO/tools/bench_losses.py --only lambda512 > $REPO/$OUT/${TAG}_stats.log 2>&1; echo "[stats] exit $?"
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_WAVES"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-30)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $REPO/$OUT/${TAG}_pmc_$n -- python3 $REPO/tools/bench_losses.py --only lambda512 > $REPO/$OUT/${TAG}_pmc_$n.log 2>&1
  echo "[pmc $n] exit $?"
done
cd $REPO
python3 - "$TAG" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/{tag}_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lambda_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
for f in glob.glob(f"gpurun_out/{tag}_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lambda_kernel" in r["Name"]:
            out["kernel_stats"] = {k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs") if k in r}
print(json.dumps(out))
open(f"gpurun_out/{tag}_summary.json", "w").write(json.dumps(out, indent=1))
PY
