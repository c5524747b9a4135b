#!/bin/bash
# Same-box ABAB of bench.py's model_full leg with the weight gradients through the library (LWM_WGRAD_HIP=0: narrow operand
# transposed + hipBLASLt, round 6's first half) and through lwm_wgrad_bf16.
#   gpurun --timeout 900 -- 'bash scripts/gpu_model_full_wgrad_ab.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r06_wgrad_ab; mkdir -p $O; : > $O/model_full_wgrad_ab.txt
for rep in 1 2; do
  for hip in 0 1; do
    (cd $R && LWM_WGRAD_HIP=$hip python -c "
import json, torch, bench
d = bench.model_full_leg(torch)
print('LWM_WGRAD_HIP=$hip', round(d['ms_per_step'], 1), 'ms', round(d['tokens_per_s']), 'tokens/s  loss', d['loss'], ' peak GiB', round(d['peak_hbm_gib'], 1))
" 2>/dev/null | grep -v amdgpu.ids) | tee -a $O/model_full_wgrad_ab.txt
  done
done
