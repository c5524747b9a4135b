# round 3, call 17: forward with the row sums on the matrix pipe
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r3c17; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
  timeout 100 $R/scripts/micro/fused_bench $R/lwm_amd/liblwm_hip.so 32768 32 5 two 2>&1 < /dev/null | cut -c1-90 >> $O/fwd_timing.txt
done
cat $O/fwd_timing.txt
LWM_PROF_DUMP=1 timeout 100 $R/scripts/micro/fused_bench $R/build/ab/liblwm_prof.so 32768 32 2 two 2>&1 < /dev/null | cut -c1-150 | head -6 >> $O/phase_clocks.txt
cat $O/phase_clocks.txt
cd $R
timeout 400 python -m pytest tests/test_gpu_attention.py -q 2>&1 < /dev/null | tail -12
