#!/bin/bash
# round 2 verification pass: GPU tests, smoke, bench, dry run of the plain N=2 command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/tests_r2.log 2>&1
echo "rc=$?" >> gpurun_out/tests_r2.log
tail -12 gpurun_out/tests_r2.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r2.log 2>&1; tail -2 gpurun_out/smoke_r2.log
