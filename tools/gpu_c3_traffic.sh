#!/bin/bash
# HBM-side traffic of the three launches of a config-3 step (lambdaLoss, slate 512: forward+save, loss kernel, backward-from-saved):
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (kernel trace only), mean per launch and kernel.
TAG=${1:-c3t}; REPO=$(pwd); OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $REPO/$OUT/${TAG}_$c -- python3 $REPO/bench.py --steps 3 --warmup 1 --loss lambdaLoss --slate 512 --queries 8192 --batch 8192 --no-cpu-baseline --no-extras > $REPO/$OUT/${TAG}_$c.log 2>&1
  echo "[pmc $c] exit $?"
done
cd $REPO
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections, json
out, tag = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{tag}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "slate_pipeline" in k or "lambda_blocked" in k:
                name = ("forward + save" if "Li0E" in k.split("EEE")[-1][:12] or "ELi0ELi" in k else "backward from saved") if "slate_pipeline" in k else "lambda_blocked_kernel"
                agg[k[:140]][c].append(float(r["Counter_Value"]))
res = {}
for k, v in agg.items():
    res[k] = {c: round(sum(x) / len(x), 1) for c, x in v.items()}
    res[k]["launches_sampled"] = len(next(iter(v.values())))
    f, w = res[k].get("FETCH_SIZE", 0), res[k].get("WRITE_SIZE", 0)
    res[k]["GB_per_launch (2 x FETCH + WRITE, KiB -> GB)"] = round((2 * f + w) * 1024 / 1e9, 3)
print(json.dumps(res, indent=1))
open(f"{out}/{tag}_summary.json", "w").write(json.dumps(res, indent=1))
PY
