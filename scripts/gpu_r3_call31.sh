R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c31; rm -rf $O; mkdir -p $O
timeout 250 python -m pytest tests/test_gpu_hf_anchor.py tests/test_gpu_llama_ops.py tests/test_gpu_llama_model.py tests/test_gpu_infer.py tests/test_cli.py -q -x 2>&1 < /dev/null | tail -5 > $O/pytest.txt
cat $O/pytest.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | tail -2
