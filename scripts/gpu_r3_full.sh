# full GPU suite + the default bench line (what the driver runs at round end)
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3full; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json; tail -5 $O/bench.err
