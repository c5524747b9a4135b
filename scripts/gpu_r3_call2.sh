# Round 3, second GPU call: the one-wave-per-SIMD forward (attn_fwd64.h) -- smoke under a short timeout, timing against
# the 8-wave kernel and the prescaled variant, parity tests; bisect of the atomic fused backward; atomic probe.
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r3c2; rm -rf $O; mkdir -p $O
# 1. smoke: small problem first (a hang here must not eat the budget)
timeout 60 $R/scripts/micro/fused_bench $R/lwm_amd/liblwm_hip.so 2048 8 1 two > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
cat $O/smoke.txt
if grep -q "smoke rc=0" $O/smoke.txt; then
  for lib in $R/lwm_amd/liblwm_hip.so $R/build/ab/liblwm_f4pre.so $R/build/ab/liblwm_fwdold.so $R/build/ab/liblwm_fbnoatom.so $R/build/ab/liblwm_fbnodq.so; do
    timeout 120 $R/scripts/micro/fused_bench $lib 32768 32 4 all >> $O/timing.txt 2>&1
  done
  cat $O/timing.txt
  cd $R
  timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_ring_sim.py -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt
  cat $O/pytest.txt
  cd /tmp
fi
timeout 120 $R/scripts/micro/atomic_probe 2048 > $O/atomic_probe.txt 2>&1
cat $O/atomic_probe.txt
