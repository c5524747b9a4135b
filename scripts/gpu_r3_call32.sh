R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
timeout 100 python -m pytest "tests/test_ring_c.py::test_c_ring8_at_config3_shard_shapes_vs_oracle[direct-True]" "tests/test_gpu_ring_sim.py::test_ring8_at_config3_shard_shapes_vs_oracle[mesh-True]" -q -x 2>&1 < /dev/null | tail -3
