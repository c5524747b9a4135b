# rocprofv3 kernel statistics of the bench's generate leg (4-layer slice, hipGraph decode)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/gprof
cat > /tmp/gen.py <<'PY'
import sys, json
sys.path.insert(0, "/root/repo")
import torch, bench
print(json.dumps(bench.generate_leg(torch)))
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gprof -o g -- python /tmp/gen.py > $R/gpurun_out/gprof.log 2>&1
find $R/gpurun_out/gprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/generate_kernel_stats.csv
find $R/gpurun_out/gprof -name "*kernel_trace.csv" -size +1M -delete
head -24 $R/gpurun_out/generate_kernel_stats.csv | cut -c1-120
tail -2 $R/gpurun_out/gprof.log | cut -c1-300
