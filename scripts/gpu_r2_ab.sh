#!/bin/bash
# timing-only A/B of the fused backward (results of the variant builds are WRONG by construction)
cd $GRAFT_REPO_ROOT
for lib in ${AB_LIBS:-""}; do
  [ "$lib" = "product" ] && lib=""
  LWM_HIP_LIB=$lib timeout 200 python bench.py --steps 2 --warmup 1 --layers 4 --no-cpu-baseline --no-vqgan $AB_ARGS 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done
