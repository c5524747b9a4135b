#!/bin/bash
# scripts/micro/mfma_power over the orderings, rocm-smi sampled beside each run -> gpurun_out/mfma_power.txt
# MFMA_CFGS="chains same amplitude;..." selects the runs (see mfma_power.cpp)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp
O=$R/gpurun_out/mfma_power.txt; : > $O
IFS=";" read -ra CFGS <<< "${MFMA_CFGS:-1 0 1;2 0 1;4 0 1;8 0 1;1 1 1;8 1 1;8 0 0;1 0 0}"
for cfg in "${CFGS[@]}"; do
  ( sleep 0.7; for i in 1 2 3; do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | egrep -i "sclk|Socket Graphics Package Power" | sed 's/.*(\([0-9]*Mhz\)).*/\1/; s/.*Power (W): //' | tr '\n' ' '; echo; sleep 0.6; done ) > /tmp/smi.txt &
  S=$!
  $R/scripts/micro/mfma_power $cfg ${MFMA_ITERS:-700000} >> $O
  wait $S
  sed 's/^/      smi: /' /tmp/smi.txt >> $O
done
cat $O
