# generate leg with the GEMV projections and with the library GEMMs (llama_ops._decode_rows forced to 0), same box
cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch, bench
import lwm_amd.llama_ops as LO
r = bench.generate_leg(torch)
print("gemv     eager %.3f  hipgraph %.3f ms/token  same_tokens %s" % (r["eager_ms_per_token"], r["hipgraph_ms_per_token"], r["same_tokens"]), flush=True)
LO._decode_rows = lambda x, k: 0
r = bench.generate_leg(torch)
print("library  eager %.3f  hipgraph %.3f ms/token  same_tokens %s" % (r["eager_ms_per_token"], r["hipgraph_ms_per_token"], r["same_tokens"]), flush=True)
PY
