#!/bin/bash
# kernel trace of the graphed config-5 step at 16 slates per step: how much of the replay is kernel time, how much dispatch gaps
set -o pipefail
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for b in 16 256; do
rm -rf $ROOT/gpurun_out/r4_graph_prof_b$b
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r4_graph_prof_b$b -- python3 $ROOT/tools/bench_encoder.py --batch $b --steps 10 --warmup 3 --graph > $ROOT/gpurun_out/r4_graph_prof_b$b.log 2>&1 || exit 1
tail -n 1 $ROOT/gpurun_out/r4_graph_prof_b$b.log
done
cd $ROOT
python3 - <<'PY'
import csv, glob, json
for b in (16, 256):
    f = glob.glob(f"gpurun_out/r4_graph_prof_b{b}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last replay: find the last seed_epoch_kernel
    idx = [i for i, r in enumerate(rows) if "seed_epoch" in r["Kernel_Name"]]
    a, e = idx[-2], idx[-1]
    step = rows[a:e]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
    span = int(rows[e]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
    by = {}
    for r in step:
        k = r["Kernel_Name"].split("(")[0][-60:]
        d = by.setdefault(k, [0, 0]); d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    top = sorted(by.items(), key=lambda kv: -kv[1][1])[:25]
    out = {"batch": b, "kernels_per_step": len(step), "span_us": span / 1e3, "kernel_busy_us": busy / 1e3, "gap_us": (span - busy) / 1e3,
           "gap_per_kernel_us": (span - busy) / 1e3 / len(step), "top": [{"kernel": k, "calls": v[0], "us": v[1] / 1e3} for k, v in top]}
    json.dump(out, open(f"gpurun_out/r4_graph_prof_b{b}.json", "w"), indent=1)
    print(json.dumps({k: out[k] for k in out if k != "top"}))
    for t in out["top"]:
        print("   ", t)
PY
