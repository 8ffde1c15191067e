#!/bin/bash
# time + FETCH_SIZE/WRITE_SIZE of the fused kernel: default library vs variants/*.so
OUT=gpurun_out; TAG=${1:-tr}; REPO=$PWD; mkdir -p $OUT
for so in default variants/*.so; do
  [ "$so" = default ] && unset LTR_LIB || export LTR_LIB=$REPO/$so
  echo "== $so"; timeout -k 10 120 python tools/bench_phases.py 2>&1 | grep net
  n=$(basename $so .so)
  (cd /tmp && export TMPDIR=/tmp && for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $REPO/$OUT/${TAG}_${n}_$c -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; done)
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/tr*_*_SIZE")):
    v=[float(r["Counter_Value"]) for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "slate_pipeline" in r["Kernel_Name"]]
    if v: print(d.split("/")[-1], f"{sum(v)/len(v):.4g} KiB")
PY
