# round 3, call 26: bench.py N = 2 through the C driver over IPC on one GPU (the collective first-layer vote included)
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c26; rm -rf $O; mkdir -p $O
timeout 300 python bench.py --gpus 2 --backend gloo --transport ipc --steps 1 --warmup 0 --layers 1 --no-configs2 > $O/bench_ipc2.json 2> $O/bench_ipc2.err < /dev/null
tail -c 900 $O/bench_ipc2.json; grep -v "socket.cpp\|Gloo\|amdgpu.ids" $O/bench_ipc2.err | tail -5
