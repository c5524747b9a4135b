# decode_bench over the product library and every build/ab/liblwm_*.so variant, at several split counts
R=$GRAFT_REPO_ROOT; cd /tmp
O=$R/gpurun_out/decode_ab.txt; rm -f $O
for lib in $R/lwm_amd/liblwm_hip.so $R/build/ab/liblwm_*.so; do
  for sp in ${AB_SPLITS:-256 512 1024}; do
    timeout 100 $R/scripts/micro/decode_bench $lib ${AB_K:-131072} $sp 20 >> $O 2>&1
  done
done
cat $O
