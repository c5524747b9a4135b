# round 3, call 23: forward, LDS fragment prefetch distance 3 / 5 / 6 / 7 (register rings of eight)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r3c23; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for v in a3 a5 a6 a7; do
  timeout 100 $R/scripts/micro/fused_bench $R/build/ab/liblwm_$v.so 32768 32 5 two 2>&1 < /dev/null | cut -c1-100 | sed "s#.*/build/ab/##" >> $O/fwd_timing.txt
done
done
cat $O/fwd_timing.txt
LWM_PROF_DUMP=1 timeout 100 $R/scripts/micro/fused_bench $R/build/ab/liblwm_a6prof.so 32768 32 2 two 2>&1 < /dev/null | cut -c1-150 | head -5 > $O/phase_clocks.txt
cat $O/phase_clocks.txt
cd $R
LWM_HIP_LIB=$R/build/ab/liblwm_a6.so timeout 400 python -m pytest tests/test_gpu_attention.py tests/test_gpu_ring_sim.py -q -x -k "oracle or fused or exact or rescale or flavour" 2>&1 < /dev/null | tail -3
