#!/bin/bash
# Encoder-path A/B of two builds of the library (variants/<a>.so, variants/<b>.so): per-kernel micro-bench + the config-5 step, interleaved.
A=${1:-base}; B=${2:-new}; OUT=gpurun_out; mkdir -p $OUT
for v in $A $B; do
  LTR_LIB=$PWD/variants/$v.so timeout -k 10 300 python3 tools/bench_encoder_kernels.py > $OUT/r4_encab_kernels_$v.jsonl 2>/dev/null || { echo "kernels $v failed"; exit 1; }
done
python3 - <<PY
import json
def load(v):
    out = {}
    for l in open("$OUT/r4_encab_kernels_" + v + ".jsonl"):
        if l.startswith("{"):
            j = json.loads(l); out[j.get("kernel", j.get("name", "?"))] = j
    return out
a, b = load("$A"), load("$B")
for k in a:
    if k in b:
        ka = [x for x in ("us", "ms", "time_us") if x in a[k]]
        if ka:
            print(f"{k:44s} $A {a[k][ka[0]]:10.2f}  $B {b[k][ka[0]]:10.2f}  {b[k][ka[0]] / a[k][ka[0]]:.3f}")
PY
for r in 1 2; do for v in $A $B; do
  echo "config5 $v r$r $(LTR_LIB=$PWD/variants/$v.so timeout -k 10 300 python3 tools/bench_encoder.py --batch 256 --steps 20 --warmup 5 --graph 2>/dev/null | tail -n 1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["slates_per_s"], j["ms_per_step"])')"
done; done
