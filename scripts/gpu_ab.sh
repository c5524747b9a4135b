#!/bin/bash
# One GPU call, parameterised by the environment (replaces the per-call scripts of earlier rounds):
#   AB_LIBS   "product old a4 ..."   libraries to time with scripts/micro/attn_bench (product = lwm_amd/liblwm_hip.so,
#                                    anything else = build/ab/liblwm_<name>.so from scripts/ab_build.sh)
#   AB_S / AB_H / AB_REPS            problem, default 32768 / 32 / 3;  AB_PROF_DUMP=1 | -1: kernel laps of -DLWM_PROF builds
#   AB_TESTS  pytest arguments run afterwards (e.g. "tests/test_gpu_attention.py -m gpu -x -q"), empty = none
#   AB_TAG    output directory under gpurun_out/ (default ab)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/${AB_TAG:-ab}; rm -rf $O; mkdir -p $O
for name in ${AB_LIBS:-product}; do
  lib=$R/build/ab/liblwm_$name.so; [ "$name" = product ] && lib=$R/lwm_amd/liblwm_hip.so
  for rep in $(seq 1 ${AB_ROUNDS:-1}); do
    LWM_PROF_DUMP=${AB_PROF_DUMP:-0} timeout 180 $R/scripts/micro/attn_bench $lib ${AB_S:-32768} ${AB_H:-32} ${AB_REPS:-3} 2>&1 | sed "s|$R/||" >> $O/timing.txt
  done
done
cat $O/timing.txt
if [ -n "$AB_TESTS" ]; then
  cd $R && timeout ${AB_TEST_TIMEOUT:-900} python -m pytest $AB_TESTS 2>&1 | tail -${AB_TEST_TAIL:-25} | tee $O/tests.txt
fi
