# round 3, call 21: bench.py N > 1 through the C driver on ONE GPU (IPC transport, ranks share the device), the Python
# dry run, then the whole GPU suite
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c21; rm -rf $O; mkdir -p $O
timeout 900 python bench.py --gpus 4 --backend gloo --transport ipc --steps 1 --warmup 1 --layers 2 > $O/bench_ipc4.json 2> $O/bench_ipc4.err < /dev/null
tail -c 2500 $O/bench_ipc4.json; tail -3 $O/bench_ipc4.err
timeout 600 python bench.py --gpus 2 --backend gloo --steps 1 --warmup 1 --layers 1 --no-configs2 > $O/bench_gloo2.json 2> $O/bench_gloo2.err < /dev/null
tail -c 600 $O/bench_gloo2.json; tail -3 $O/bench_gloo2.err
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 < /dev/null | tail -8 > $O/pytest.txt
cat $O/pytest.txt
