# mesh/ring schedules with the real kernels (thread-simulated ranks) + dry run of bench.py's N>1 path
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_ring_sim.py -q 2>&1 | tail -5) > gpurun_out/mesh_tests.log
for cfg in "2 ring" "4 mesh" "4 ring"; do set -- $cfg
  (timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 2951$1 bench.py --gpus $1 --steps 1 --warmup 1 --layers 2 --seq 16384 --backend gloo --schedule $2 2>&1 | tail -3) > gpurun_out/dry_$1_$2.log
done
for f in gpurun_out/mesh_tests.log gpurun_out/dry_*; do echo "== $f"; cat $f | cut -c1-1500; done
