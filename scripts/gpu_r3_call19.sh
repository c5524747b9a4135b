# round 3, call 19: the C ring driver with zigzag ownership and the direct schedule (thread-played ranks)
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c19; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_ring_c.py -x -q -m gpu 2>&1 < /dev/null | tail -25 > $O/pytest.txt
cat $O/pytest.txt
