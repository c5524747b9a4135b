# round 3, call 18: dK/dV and fused backward with bare M0 writes in glds_load_* (no save / restore)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r3c18; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for lib in lwm_amd/liblwm_hip.so build/ab/liblwm_barem0.so; do
  timeout 100 $R/scripts/micro/fused_bench $R/$lib 32768 32 3 all 2>&1 < /dev/null | cut -c1-130 >> $O/timing.txt
done
done
cat $O/timing.txt
cd $R
LWM_HIP_LIB=$R/build/ab/liblwm_barem0.so timeout 400 python -m pytest tests/test_gpu_attention.py -q -x 2>&1 < /dev/null | tail -3
