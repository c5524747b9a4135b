R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c5; rm -rf $O; mkdir -p $O
LWM_PROF_DUMP=1 timeout 120 $R/scripts/micro/fused_bench $R/build/ab/liblwm_f4prof.so 32768 32 2 two > $O/prof.txt 2>&1
timeout 120 $R/scripts/micro/fused_bench $R/lwm_amd/liblwm_hip.so 32768 32 4 two >> $O/prof.txt 2>&1
cat $O/prof.txt
timeout 300 python scripts/gpu_fwd_determinism.py 2>&1 | grep "skip=False" > $O/determinism.txt
cat $O/determinism.txt
