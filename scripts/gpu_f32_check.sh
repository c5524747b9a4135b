#!/bin/bash
# The float32 flavour on the GPU: its tests, then a timing of the three f32 attention kernels at S = 4096 and 8192
# (32 heads), and of the f32 2-layer slice of BASELINE configs[0].
#   gpurun --timeout 900 -- 'TAG=r06_f32 bash scripts/gpu_f32_check.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp
O=$R/gpurun_out/${TAG:-f32}; mkdir -p "$O"
cd $R
timeout 800 python -m pytest tests/test_gpu_f32.py -q -x --durations=10 2>&1 | tail -40 > $O/tests.txt
tail -15 $O/tests.txt
timeout 300 python scripts/gpu_f32_bench.py 2>&1 | tail -20 | tee $O/bench.txt
