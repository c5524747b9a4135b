#!/bin/bash
# round 2: first contact of the one-launch backward with the hardware -- parity, determinism, A/B timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_attention.py -q -x -k "fwd_bwd_vs_oracle or fused or carries" > gpurun_out/fused_tests.log 2>&1
echo "rc=$?" >> gpurun_out/fused_tests.log
tail -15 gpurun_out/fused_tests.log
for mode in "" "--two-kernel-bwd"; do
  timeout 300 python bench.py --steps 2 --warmup 1 --layers 8 --no-cpu-baseline --no-vqgan $mode > gpurun_out/ab_fused$mode.json 2> gpurun_out/ab_fused$mode.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_fused$mode.json").read().strip().splitlines()[-1])
print("mode='$mode' tok/s(32L-equiv) %.0f" % (d['value']*8/32), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()}, d['roofline']['kernel'], round(d['roofline']['frac'],3))
PY
done
