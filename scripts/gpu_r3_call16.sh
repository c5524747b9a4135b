# round 3, call 16: forward LDS-DMA pieces: bare M0 write (no save/restore, no wait states) and odd-gap placement
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r3c16; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for lib in lwm_amd/liblwm_hip.so build/ab/liblwm_m0.so build/ab/liblwm_odd.so build/ab/liblwm_m0odd.so; do
  timeout 100 $R/scripts/micro/fused_bench $R/$lib 32768 32 5 two 2>&1 < /dev/null | cut -c1-90 >> $O/fwd_timing.txt
done
done
cat $O/fwd_timing.txt
for lib in build/ab/liblwm_baseprof.so build/ab/liblwm_m0prof.so; do
  LWM_PROF_DUMP=1 timeout 100 $R/scripts/micro/fused_bench $R/$lib 32768 32 2 two 2>&1 < /dev/null | cut -c1-150 | head -6 >> $O/phase_clocks.txt
done
cat $O/phase_clocks.txt
cd $R
LWM_HIP_LIB=$R/build/ab/liblwm_m0odd.so timeout 400 python -m pytest tests/test_gpu_attention.py -x -q 2>&1 < /dev/null | tail -3
