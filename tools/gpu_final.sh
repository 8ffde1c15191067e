#!/bin/bash
# Round-end evidence visit: whole -m gpu suite, smoke, headline bench, other BASELINE shapes, rocprofv3 kernel stats of the
# bench command (warm-up launches excluded from the average by the summary step), FETCH_SIZE / WRITE_SIZE passes.
set -o pipefail
TAG=${1:-fin}; OUT=gpurun_out; mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run() { local name=$1 to=$2; shift 2; timeout -k 10 $to "$@" > $OUT/${TAG}_$name.log 2>&1; local rc=$?; echo "[$name] exit $rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMED OUT"; exit 9; fi; return $rc; }
run tests 1000 python -m pytest tests -m gpu -q; grep -E "passed|failed" $OUT/${TAG}_tests.log | tail -1
cp $OUT/parity_report.json $OUT/${TAG}_parity_report.json 2>/dev/null
run smoke 300 python __graft_entry__.py smoke; tail -2 $OUT/${TAG}_smoke.log
run bench 600 python bench.py --steps 20 --warmup 3; tail -1 $OUT/${TAG}_bench.log | cut -c1-300
run bench_c3 300 python bench.py --steps 10 --warmup 2 --loss lambdaLoss --slate 512 --queries 8192 --batch 8192 --no-cpu-baseline; tail -1 $OUT/${TAG}_bench_c3.log | cut -c1-200
run bench_c1 300 python bench.py --steps 20 --warmup 3 --loss listnet --slate 32 --queries 400000 --batch 100000 --no-cpu-baseline; tail -1 $OUT/${TAG}_bench_c1.log | cut -c1-200
run bench_l128 300 python bench.py --steps 20 --warmup 3 --loss lambdaLoss --no-cpu-baseline; tail -1 $OUT/${TAG}_bench_l128.log | cut -c1-200
run bench_triple 300 python bench.py --steps 20 --warmup 3 --net triple --no-cpu-baseline; tail -1 $OUT/${TAG}_bench_triple.log | cut -c1-200
run bench_losses 300 python tools/bench_losses.py; grep -E "approx|lambda" $OUT/${TAG}_bench_losses.log
run bench_c5 300 python tools/bench_encoder.py --batch 256 --steps 10 --warmup 3; tail -1 $OUT/${TAG}_bench_c5.log | cut -c1-400
run bench_c5_kernels 300 python tools/bench_encoder_kernels.py; grep -c "^{" $OUT/${TAG}_bench_c5_kernels.log
bash tools/gpu_enc_profile.sh ${TAG}_c5 --batch 256 --steps 5 --warmup 2 > $OUT/${TAG}_c5_profile.txt 2>&1; head -12 $OUT/${TAG}_c5_profile.txt | cut -c1-160
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/${TAG}_prof -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $REPO/$OUT/${TAG}_prof.log 2>&1; echo "[prof] exit $?"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-24)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $REPO/$OUT/${TAG}_pmc_$n -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $REPO/$OUT/${TAG}_pmc_$n.log 2>&1
  echo "[pmc $n] exit $?"
done
cd $REPO
python3 - "$TAG" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/{tag}_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "slate_pipeline" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"pmc_per_launch_mean": {k: sum(v) / len(v) for k, v in sorted(agg.items())}}
# kernel-trace durations of the pipeline kernel: drop the warm-up launches (first 2), average the timed ones
for f in glob.glob(f"gpurun_out/{tag}_prof/**/*kernel_trace.csv", recursive=True):
    d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "slate_pipeline" in r["Kernel_Name"]]
    d.sort()
    timed = [x[1] for x in d[2:]]
    out["kernel_trace"] = {"launches": len(d), "timed_launches": len(timed), "avg_ns_timed": sum(timed) / max(len(timed), 1),
                           "min_ns": min(timed), "max_ns": max(timed), "avg_ns_all": sum(x[1] for x in d) / len(d)}
print(json.dumps(out))
open(f"gpurun_out/{tag}_summary.json", "w").write(json.dumps(out, indent=1))
PY
