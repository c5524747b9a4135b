# Round 3, first GPU call: atomic-add probe, fused (atomic) backward vs two-kernel timing for both atomic scopes,
# parity tests of the attention kernels, RCCL first contact.
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r3c1; rm -rf $O; mkdir -p $O
timeout 120 $R/scripts/micro/atomic_probe 2048 > $O/atomic_probe.txt 2>&1
for lib in $R/lwm_amd/liblwm_hip.so $R/build/ab/liblwm_l2atom.so; do
  timeout 120 $R/scripts/micro/fused_bench $lib 32768 32 4 all >> $O/timing.txt 2>&1
done
cat $O/atomic_probe.txt $O/timing.txt
cd $R
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_ring_c.py tests/test_gpu_probe.py -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt
cat $O/pytest.txt
cat gpurun_out/rccl_first_contact.txt 2>/dev/null | tail -5
LWM_HIP_LIB=$R/build/ab/liblwm_l2atom.so timeout 600 python -m pytest tests/test_gpu_attention.py -m gpu -x -q -k "fused or vs_oracle or packed" 2>&1 | tail -8 > $O/pytest_l2atom.txt
cat $O/pytest_l2atom.txt
