# conv_bench over the product library and every build/ab/liblwmv_*.so variant (scripts/ab_build_vqgan.sh)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/conv_ab; rm -rf $O; mkdir -p $O
for lib in $R/lwm_amd/liblwm_hip.so $R/build/ab/liblwmv_*.so; do
  timeout 200 $R/scripts/micro/conv_bench $lib ${AB_FRAMES:-32} ${AB_REPS:-3} >> $O/timing.txt 2>&1
done
cat $O/timing.txt
