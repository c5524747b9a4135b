#!/bin/bash
# rocprofv3 kernel trace + statistics of bench.py's model_full leg ALONE (all 32 layers of LWM-7B, S = 32768, fwd+bwd: one
# warm-up step + one timed step), then the per-class table:  gpurun --timeout 900 -- 'TAG=r06 bash scripts/gpu_model_full_prof.sh'
#   -> gpurun_out/$TAG/model_full_kernel_stats.csv, model_full_trace_step.csv (the timed step's launches), model_full.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp
O=$R/gpurun_out/${TAG:-mfull}; mkdir -p "$O"
cd /tmp; rm -rf /tmp/prof_mfull
timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mfull -o mf -- \
    python -c "
import sys, json; sys.path.insert(0, '$R')
import torch, bench
print('MODEL_FULL ' + json.dumps(bench.model_full_leg(torch, **json.loads('''${MODEL_KW:-{\}}'''))))
" > "$O/model_full.log" 2>&1
grep '^MODEL_FULL ' "$O/model_full.log" | sed 's/^MODEL_FULL //' > "$O/model_full.json"
f=$(find /tmp/prof_mfull -name '*kernel_stats.csv' | head -1)
t=$(find /tmp/prof_mfull -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] || { echo "no kernel stats"; tail -20 "$O/model_full.log"; exit 1; }
cp "$f" "$O/model_full_kernel_stats.csv"
python "$R/scripts/model_full_table.py" "$t" "$O/model_full.json" | tee "$O/model_full_table.txt"
