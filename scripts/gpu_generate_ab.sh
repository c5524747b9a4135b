cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch, bench, json, cProfile, pstats, io
r = bench.generate_leg(torch)
print("gemv    ", round(r["eager_ms_per_token"],3), round(r["hipgraph_ms_per_token"],3))
import lwm_amd.llama_ops as LO
from lwm_amd.llama import LLaMAConfig, LLaMAForCausalLM
cfg = LLaMAConfig.load_config("7b", num_hidden_layers=4, max_sequence_length=32768, theta=1e7)
with torch.device("cuda"):
    model = LLaMAForCausalLM(cfg)
ids = torch.randint(0, cfg.vocab_size, (1, 2048), device="cuda")
model.generate(ids, max_new_tokens=3, max_length=32768, graph=False)
pr = cProfile.Profile(); pr.enable()
model.generate(ids, max_new_tokens=20, max_length=32768, graph=False)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
orig = LO._decode_rows
LO._decode_rows = lambda x, k: 0
r = bench.generate_leg(torch)
print("library ", round(r["eager_ms_per_token"],3), round(r["hipgraph_ms_per_token"],3))
PY
