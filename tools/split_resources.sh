#!/bin/bash
# Register / scratch usage of the split-precision pipeline kernels (hipcc remarks), DoubleNet rows first.
cd /root/repo/nn-with-pytorch-personalized-losses_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-gpu-rdc -DLTR_SPLIT_BF16=1 $EXTRA -c ltr_scorer.hip -o /tmp/isa/scorer_split.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|ScratchSize" | grep -A2 "slate_pipeline\|error" | grep -E "error|Name|VGPRs|Scratch" | sed 's/.*remark: //' | paste - - - | sed 's/_ZN12_GLOBAL__N_121slate_pipeline_kernelINS_4NetTI//; s/\[-Rpass[^]]*\]//g' | head -${1:-5}
