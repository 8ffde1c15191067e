#!/bin/bash
# Config-5 training step: eager launches vs one hipGraph, at 16 / 64 / 128 / 256 slates per step (the C++ orchestrator and the fused scoring tail were measured under the graph before they were removed: profiles/r04_c5_graph_step.jsonl).
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r4_graph_step.jsonl
: > $OUT
for b in 16 64 128 256; do
  timeout -k 10 300 python3 tools/bench_encoder.py --batch $b --steps 20 --warmup 5 | tail -n 1 >> $OUT || exit 1
  timeout -k 10 300 python3 tools/bench_encoder.py --batch $b --steps 20 --warmup 5 --graph | tail -n 1 >> $OUT || exit 1
done
timeout -k 10 300 python3 tools/bench_encoder.py --batch 256 --steps 20 --warmup 5 --adam foreach | tail -n 1 >> $OUT || exit 1
timeout -k 10 300 python3 tools/bench_encoder.py --batch 16 --steps 20 --warmup 5 --adam foreach | tail -n 1 >> $OUT || exit 1
cat $OUT
