R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c10; rm -rf $O; mkdir -p $O
for name in loslack loslack1; do
  echo "== $name" >> $O/determinism.txt
  LWM_HIP_LIB=$R/build/ab/liblwm_$name.so timeout 300 python scripts/gpu_fwd_determinism.py 2>&1 | grep "S=8192 H=4 packed=True skip=False rep=0" >> $O/determinism.txt
done
cat $O/determinism.txt
