R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c4; rm -rf $O; mkdir -p $O
for name in product f4dmatop; do
  lib=$R/build/ab/liblwm_$name.so; [ "$name" = product ] && lib=$R/lwm_amd/liblwm_hip.so
  echo "== $name" >> $O/determinism.txt
  LWM_HIP_LIB=$lib timeout 300 python scripts/gpu_fwd_determinism.py >> $O/determinism.txt 2>&1
done
cat $O/determinism.txt
