#!/bin/bash
# Round-4 final visit: whole GPU suite, smoke, the default bench line (with secondaries), config 3, loss micro-bench.
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/r4f_tests.log 2>&1; rc=$?
echo "[tests] exit $rc: $(tail -n 1 $OUT/r4f_tests.log | cut -c1-200)"
[ $rc -ne 0 ] && { grep -E "^E |FAILED" $OUT/r4f_tests.log | head -n 12 | cut -c1-250; exit $rc; }
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | grep "smoke" || exit 1
timeout -k 10 900 python bench.py > $OUT/r4f_bench_default.log 2>$OUT/r4f_bench_default.err || { tail -n 5 $OUT/r4f_bench_default.err; exit 1; }
tail -n 1 $OUT/r4f_bench_default.log | python3 -c '
import sys, json
j = json.loads(sys.stdin.read())
print("HEADLINE", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["kernel_ms"], j["roofline"]["traffic"], "cpu", j["cpu_baseline"]["value"])
for k, v in j.get("secondary", {}).items():
    print("  ", k, v.get("value"), v.get("ms_per_step"), json.dumps(v.get("roofline"))[:260], v.get("error", ""))'
timeout -k 10 300 python bench.py --loss lambdaLoss --slate 512 --queries 16384 --batch 8192 --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -n 1 > $OUT/r4f_bench_c3.json || exit 1
python3 -c 'import json; j=json.load(open("gpurun_out/r4f_bench_c3.json")); print("C3 double", j["value"], j["ms_per_step"], j["roofline"]["frac"])'
timeout -k 10 300 python tools/bench_losses.py 2>/dev/null | grep "^{" > $OUT/r4f_bench_losses.jsonl || exit 1
python3 -c '
import json
for l in open("gpurun_out/r4f_bench_losses.jsonl"):
    j = json.loads(l); print("loss", {k: j[k] for k in list(j)[:6]})'
