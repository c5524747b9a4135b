#!/bin/bash
# The evidence pass of a round, one GPU call:  gpurun --timeout 1500 -- 'TAG=r04 bash scripts/gpu_round_check.sh'
#   1. pytest -m gpu (with the 30 slowest tests)        -> gpurun_out/$TAG/tests.txt
#   2. __graft_entry__.smoke()                           -> gpurun_out/$TAG/smoke.txt
#   3. the driver's bench command                        -> gpurun_out/$TAG/bench.json (+ bench.err)
#   4. rocprofv3 --kernel-trace --stats, 4 layers, main workload only (every launch of a kernel has the bench line's
#      shape, so the average durations are comparable)   -> gpurun_out/$TAG/kernel_stats.csv
# STEPS="tests smoke bench prof" selects (default all).  Copy what should be judged into profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp
T=${TAG:-round}; O=$R/gpurun_out/$T; mkdir -p $O
STEPS=${STEPS:-tests smoke bench prof}
cd $R
if [[ " $STEPS " == *" tests "* ]]; then
  s=$(date +%s)
  timeout ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -q -x --durations=30 2>&1 | tail -60 > $O/tests.txt
  echo "wall seconds: $(( $(date +%s) - s ))" >> $O/tests.txt
  tail -5 $O/tests.txt
fi
if [[ " $STEPS " == *" smoke "* ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.txt
fi
if [[ " $STEPS " == *" bench "* ]]; then
  s=$(date +%s)
  timeout 900 python bench.py --gpus 1 --steps ${BENCH_STEPS:-20} --warmup ${BENCH_WARMUP:-5} > $O/bench.json 2> $O/bench.err
  echo "bench wall seconds: $(( $(date +%s) - s ))" | tee -a $O/bench.err
  python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, {k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()}, d["roofline"]["frac"])
PY
fi
if [[ " $STEPS " == *" prof "* ]]; then
  cd /tmp; rm -rf /tmp/prof_$T
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$T -o ks -- \
      python $R/bench.py --steps 3 --warmup 1 --layers 4 --no-cpu-baseline --no-vqgan > $O/prof.log 2>&1
  f=$(find /tmp/prof_$T -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -8 $O/kernel_stats.csv
fi
