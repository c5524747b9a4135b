# round 3, call 24: forward, the vector work of phase 2b deferred to the top of the next iteration (LWM_F4_DEFER)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r3c24; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for v in d0 d1; do
  timeout 100 $R/scripts/micro/fused_bench $R/build/ab/liblwm_$v.so 32768 32 5 two 2>&1 < /dev/null | cut -c1-100 | sed "s#.*/build/ab/##" >> $O/fwd_timing.txt
done
done
cat $O/fwd_timing.txt
LWM_PROF_DUMP=1 timeout 100 $R/scripts/micro/fused_bench $R/build/ab/liblwm_d1prof.so 32768 32 2 two 2>&1 < /dev/null | cut -c1-150 | head -5 > $O/phase_clocks.txt
cat $O/phase_clocks.txt
cd $R
LWM_HIP_LIB=$R/build/ab/liblwm_d1.so timeout 300 python -m pytest tests/test_gpu_attention.py -q -x 2>&1 < /dev/null | tail -3
