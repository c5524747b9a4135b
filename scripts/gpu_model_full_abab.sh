#!/bin/bash
# Same-box ABAB of bench.py's model_full leg: the tree of round 5 (git archive of 3325d1c built under build/r05_tree, which
# travels with gpurun) against this tree.   gpurun --timeout 900 -- 'bash scripts/gpu_model_full_abab.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r06_abab; mkdir -p $O; : > $O/model_full_abab.txt
for rep in 1 2; do
  for tree in build/r05_tree .; do
    (cd $R/$tree && python -c "
import json, torch, bench
d = bench.model_full_leg(torch)
print('$tree', round(d['ms_per_step'], 1), 'ms', round(d['tokens_per_s']), 'tokens/s  loss', d['loss'], ' peak GiB', round(d['peak_hbm_gib'], 1))
" 2>/dev/null | grep -v amdgpu.ids) | tee -a $O/model_full_abab.txt
  done
done
