#!/bin/bash
# Shader clock and socket power while the attention kernels run on random and on all-zero operands
# (profiles/r04_power_clock.txt): rocm-smi sampled every 0.5 s beside scripts/micro/attn_bench.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp
O=$R/gpurun_out/power_clock.txt; : > $O
for amp in 1.7 0; do
  echo "### operand amplitude $amp" >> $O
  ( for i in $(seq 1 14); do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | egrep -i "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' ' ; echo; sleep 0.5; done ) >> $O &
  S=$!
  LWM_BENCH_AMP=$amp $R/scripts/micro/attn_bench $R/lwm_amd/liblwm_hip.so 32768 32 60 2>&1 | sed "s|$R/||" > /tmp/ab_$amp.txt
  wait $S
  cat /tmp/ab_$amp.txt >> $O
done
cat $O
