# gemv_bench over the product library and every build/ab/liblwm_*.so variant
R=$GRAFT_REPO_ROOT; cd /tmp
for lib in $R/lwm_amd/liblwm_hip.so $R/build/ab/liblwm_*.so; do
  timeout 60 $R/scripts/micro/gemv_bench $lib ${AB_ROWS:-1} 50 2>&1
done | tee $R/gpurun_out/gemv_ab.txt
