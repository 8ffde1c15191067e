#!/bin/bash
# kernel trace of the graphed config-5 step: how much of a replay is kernel time, how much dispatch gaps, which kernels    usage: [batches...]
set -o pipefail
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
BATCHES=${@:-16}
cd /tmp
for b in $BATCHES; do
rm -rf $ROOT/gpurun_out/r4_graph_prof_b$b
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r4_graph_prof_b$b -- python3 $ROOT/tools/bench_encoder.py --batch $b --steps 10 --warmup 3 --graph > $ROOT/gpurun_out/r4_graph_prof_b$b.log 2>&1 || exit 1
tail -n 1 $ROOT/gpurun_out/r4_graph_prof_b$b.log | cut -c1-200
done
cd $ROOT
python3 - $BATCHES <<'PY'
import csv, glob, json, re, sys
for b in sys.argv[1:]:
    f = glob.glob(f"gpurun_out/r4_graph_prof_b{b}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "seed_epoch" in r["Kernel_Name"]]
    a, e = idx[-2], idx[-1]
    step = rows[a:e]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
    span = int(rows[e]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
    by = {}
    for r in step:
        n = re.sub(r"^void ", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
        k = n.split("<")[0].split("(")[0]
        if n.startswith("at::"):
            k += ":" + ",".join(sorted(set(re.findall(r"(\w+Functor|\w+_kernel_cuda|Lerp\w*|Sqrt\w*|FusedAdam\w*|adam\w*)", n))))[:60]
        d = by.setdefault(k, [0, 0]); d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    top = sorted(by.items(), key=lambda kv: -kv[1][1])
    out = {"batch": int(b), "kernels_per_step": len(step), "span_us": span / 1e3, "kernel_busy_us": busy / 1e3, "gap_us": (span - busy) / 1e3,
           "gap_per_kernel_us": (span - busy) / 1e3 / len(step), "kernels": [{"kernel": k, "calls": v[0], "us": round(v[1] / 1e3, 1)} for k, v in top]}
    json.dump(out, open(f"gpurun_out/r4_graph_prof_b{b}.json", "w"), indent=1)
    print(json.dumps({k: out[k] for k in out if k != "kernels"}))
    for t in out["kernels"][:24]:
        print("   ", t)
PY
