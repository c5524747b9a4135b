# round 3, call 20: the C ring driver between real processes over the IPC transport (one GPU)
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c20; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_ring_ipc.py -x -q -m gpu 2>&1 < /dev/null | tail -40 > $O/pytest.txt
cat $O/pytest.txt
cat gpurun_out/ipc_ring8.txt 2>/dev/null
