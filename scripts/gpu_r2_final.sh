#!/bin/bash
# Round-2 evidence pass: rocprofv3 kernel stats of the bench command, the four PMC passes for the
# two-kernel backward and for the one-launch backward (separate passes, kernel trace only).
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r02 -- python $R/bench.py --steps 1 --warmup 0 --layers 4 --no-cpu-baseline --no-full-model 2>&1 | tail -2) > $R/gpurun_out/prof.log
PMC_DIR=pmc bash $R/scripts/gpu_pmc_attention.sh
PMC_DIR=pmc_fused PMC_BENCH_ARGS=--fused-bwd bash $R/scripts/gpu_pmc_attention.sh
cd $R
head -8 $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) | cut -c1-140
ls gpurun_out/pmc gpurun_out/pmc_fused | head
