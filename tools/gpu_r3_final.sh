#!/bin/bash
# Round-3 evidence visit: whole -m gpu suite with the default (exact fp32) library AND with the f16 x 2 variant (LTR_LIB), smoke,
# headline bench (+ secondary lines + CPU baseline), the other BASELINE shapes on both libraries, loss micro-bench, config 5,
# rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE / SQ counters of the bench command for both libraries, phase stamps.
set -o pipefail
TAG=${1:-r3fin}; OUT=gpurun_out; mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
V=$PWD/nn-with-pytorch-personalized-losses_amd/ltr_mi355x/libltr_mi355x_f16x2.so
run() { local name=$1 to=$2; shift 2; timeout -k 10 $to "$@" > $OUT/${TAG}_$name.log 2>&1; local rc=$?; echo "[$name] exit $rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMED OUT"; exit 9; fi; return $rc; }
PART=${2:-all}      # tests | bench | prof | all  (three gpurun calls fit the 1200 s limit comfortably)
if [ $PART = tests ] || [ $PART = all ]; then
run tests 1000 python -m pytest tests -m gpu -q; grep -E "passed|failed" $OUT/${TAG}_tests.log | tail -1; grep -E "^FAILED" $OUT/${TAG}_tests.log | head
cp $OUT/parity_report.json $OUT/${TAG}_parity_report.json 2>/dev/null
LTR_LIB=$V run tests_f16x2 1000 python -m pytest tests -m gpu -q; grep -E "passed|failed" $OUT/${TAG}_tests_f16x2.log | tail -1; grep -E "^FAILED" $OUT/${TAG}_tests_f16x2.log | head
cp $OUT/parity_report.json $OUT/${TAG}_parity_report_f16x2.json 2>/dev/null
run smoke 300 python __graft_entry__.py smoke; tail -2 $OUT/${TAG}_smoke.log
fi
[ $PART = tests ] && exit 0
if [ $PART = bench ] || [ $PART = all ]; then
run bench 900 python bench.py --steps 20 --warmup 3; tail -1 $OUT/${TAG}_bench.log | cut -c1-300
for lib in fp32 f16x2; do
  if [ $lib = f16x2 ]; then export LTR_LIB=$V; else unset LTR_LIB; fi
  run bench_c3_$lib 300 python bench.py --steps 10 --warmup 2 --loss lambdaLoss --slate 512 --queries 8192 --batch 8192 --no-cpu-baseline --no-extras; tail -1 $OUT/${TAG}_bench_c3_$lib.log | cut -c1-160
  run bench_c1_$lib 300 python bench.py --steps 20 --warmup 3 --loss listnet --slate 32 --queries 400000 --batch 100000 --no-cpu-baseline --no-extras; tail -1 $OUT/${TAG}_bench_c1_$lib.log | cut -c1-160
  run bench_l128_$lib 300 python bench.py --steps 20 --warmup 3 --loss lambdaLoss --no-cpu-baseline --no-extras; tail -1 $OUT/${TAG}_bench_l128_$lib.log | cut -c1-160
  run bench_triple_$lib 300 python bench.py --steps 20 --warmup 3 --net triple --no-cpu-baseline --no-extras; tail -1 $OUT/${TAG}_bench_triple_$lib.log | cut -c1-160
  run bench_two64_$lib 300 python bench.py --steps 20 --warmup 3 --net two64 --no-cpu-baseline --no-extras; tail -1 $OUT/${TAG}_bench_two64_$lib.log | cut -c1-160
done
unset LTR_LIB
run bench_losses 300 python tools/bench_losses.py; grep -E "approx|lambda" $OUT/${TAG}_bench_losses.log
run bench_c5 300 python tools/bench_encoder.py --batch 256 --steps 10 --warmup 3; tail -1 $OUT/${TAG}_bench_c5.log | cut -c1-300
run bench_aux 300 python tools/bench_aux.py; grep -c . $OUT/${TAG}_bench_aux.log
run bench_enc_kernels 200 python tools/bench_encoder_kernels.py; grep -c kernel $OUT/${TAG}_bench_enc_kernels.log
run enc_parity 300 python tools/enc_parity_report.py; grep -c case $OUT/${TAG}_enc_parity.log
for v in fp32_stamps f16x2_stamps; do        # tools/build_variant.sh fp32_stamps -DLTR_STAMPS ; ... f16x2_stamps -DLTR_F16X2=1 -DLTR_STAMPS
  [ -f variants/$v.so ] && LTR_LIB=$PWD/variants/$v.so timeout -k 10 200 python tools/phase_stamps.py > $OUT/${TAG}_stamps_$v.jsonl 2>/dev/null; echo "[stamps $v] exit $?"
  [ -f variants/$v.so ] && LTR_LIB=$PWD/variants/$v.so timeout -k 10 200 python tools/fcw_stamps.py > $OUT/${TAG}_fcw_stamps_$v.jsonl 2>/dev/null; echo "[fcw stamps $v] exit $?"
done
fi
[ $PART = bench ] && exit 0
cd /tmp && export TMPDIR=/tmp
for lib in fp32 f16x2; do
  if [ $lib = f16x2 ]; then export LTR_LIB=$V; else unset LTR_LIB; fi
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/${TAG}_prof_$lib -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $REPO/$OUT/${TAG}_prof_$lib.log 2>&1; echo "[prof $lib] exit $?"
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
    n=$(echo $set | tr ' ' '_' | cut -c1-24)
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $REPO/$OUT/${TAG}_pmc_${lib}_$n -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $REPO/$OUT/${TAG}_pmc_${lib}_$n.log 2>&1
    echo "[pmc $lib $n] exit $?"
  done
done
unset LTR_LIB
cd $REPO
python3 - "$TAG" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
out = {}
for lib in ("fp32", "f16x2"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/{tag}_pmc_{lib}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "slate_pipeline" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    rec = {"pmc_per_launch_mean": {k: sum(v) / len(v) for k, v in sorted(agg.items())}}
    for f in glob.glob(f"gpurun_out/{tag}_prof_{lib}/**/*kernel_trace.csv", recursive=True):
        d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "slate_pipeline" in r["Kernel_Name"]]
        d.sort()
        timed = [x[1] for x in d[2:]]
        if timed:
            rec["kernel_trace"] = {"launches": len(d), "timed_launches": len(timed), "avg_ns_timed": sum(timed) / len(timed), "min_ns": min(timed),
                                   "max_ns": max(timed), "avg_ns_all": sum(x[1] for x in d) / len(d)}
    out[lib] = rec
print(json.dumps(out))
open(f"gpurun_out/{tag}_summary.json", "w").write(json.dumps(out, indent=1))
PY
