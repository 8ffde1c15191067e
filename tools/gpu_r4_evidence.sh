#!/bin/bash
# Round-4 evidence visit: rocprofv3 kernel-trace + PMC summaries (tools/pmc_summary.py: four passes per item) for the kernels the
# metric rests on.  usage: bash tools/gpu_r4_evidence.sh [items...]   (default: all)
OUT=gpurun_out; mkdir -p $OUT
H=nn-with-pytorch-personalized-losses_amd/ltr_mi355x/libltr_mi355x_f16x2.so
B="--steps 4 --warmup 1 --no-cpu-baseline --no-extras"
items=${@:-two64 two64_h double double_h triple triple_h approx128 approx512 lambda128 lambda512}
for it in $items; do
  case $it in
    two64)     python3 tools/pmc_summary.py --tag r04_two64_fp32 --kernel fcw_fused_kernel --skip 1 -- bench.py --net two64 $B ;;
    two64_h)   python3 tools/pmc_summary.py --tag r04_two64_f16x2 --kernel fcw_fused_kernel --lib $H --skip 1 -- bench.py --net two64 $B ;;
    double)    python3 tools/pmc_summary.py --tag r04_double_fp32 --kernel slate_pipeline_kernel --skip 1 -- bench.py --net double $B ;;
    double_h)  python3 tools/pmc_summary.py --tag r04_double_f16x2 --kernel slate_pipeline_kernel --lib $H --skip 1 -- bench.py --net double $B ;;
    triple)    python3 tools/pmc_summary.py --tag r04_triple_fp32 --kernel fcw_fused_kernel --skip 1 -- bench.py --net triple $B ;;
    triple_h)  python3 tools/pmc_summary.py --tag r04_triple_f16x2 --kernel fcw_fused_kernel --lib $H --skip 1 -- bench.py --net triple $B ;;
    approx128) python3 tools/pmc_summary.py --tag r04_loss_approxndcg_S128 --kernel approxndcg_kernel --skip 3 -- tools/bench_losses.py --only approxndcg128 ;;
    approx512) python3 tools/pmc_summary.py --tag r04_loss_approxndcg_S512 --kernel approxndcg_kernel --skip 3 -- tools/bench_losses.py --only approxndcg512 ;;
    lambda128) python3 tools/pmc_summary.py --tag r04_loss_lambda2pp_S128 --kernel lambda_kernel --skip 3 -- tools/bench_losses.py --only lambda128 ;;
    lambda512) python3 tools/pmc_summary.py --tag r04_loss_lambda2pp_S512 --kernel lambda_blocked_kernel --skip 3 -- tools/bench_losses.py --only lambda512 ;;
  esac > $OUT/r04_ev_$it.log 2>&1
  echo "[$it] $(tail -n 1 $OUT/r04_ev_$it.log | cut -c1-400)"
done
