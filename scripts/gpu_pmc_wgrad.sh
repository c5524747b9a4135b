# Counter passes of lwm_wgrad_bf16 on one weight-gradient shape (default wqkv: K = 4096, N = 12288, S = 32768) and, for
# comparison, the library GEMM torch.matmul picks for the same product with the narrow operand transposed.
#   gpurun --timeout 600 -- 'TAG=r06i bash scripts/gpu_pmc_wgrad.sh'
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
TAG=${TAG:-wgrad}; O=$R/gpurun_out/$TAG; mkdir -p $O; rm -rf $O/pmc_wgrad
cat > /tmp/one_wgrad.py <<'PY'
import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from lwm_amd import _capi
from lwm_amd._lib import lib
from lwm_amd.llama_ops import transpose2d
S, K, N = (int(v) for v in os.environ.get("WG_SHAPE", "32768,4096,12288").split(","))
L = lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = (torch.randn(S, K, device="cuda") * 0.5).to(torch.bfloat16); g = (torch.randn(S, N, device="cuda") * 0.5).to(torch.bfloat16)
dw = torch.empty(K, N, device="cuda", dtype=torch.bfloat16)
n = L.lwm_wgrad_workspace_bytes(S, K, N); ws = torch.empty(max(n, 16), dtype=torch.uint8, device="cuda")
for _ in range(3):
    _capi.check(L, L.lwm_wgrad_bf16(x.data_ptr(), K, g.data_ptr(), N, dw.data_ptr(), N, S, K, N, ws.data_ptr(), n, st), "wgrad")
    torch.matmul(transpose2d(x), g) if K <= N else torch.matmul(x.t(), transpose2d(g).t())
torch.cuda.synchronize()
PY
i=1
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  (timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_wgrad -o pass$i -- python /tmp/one_wgrad.py 2>&1 | tail -2) > $O/pmc_wgrad_pass$i.log
  i=$((i+1))
done
python - $O/pmc_wgrad <<'PY' | tee $O/pmc_wgrad_summary.txt
import csv, glob, sys, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for fn in sorted(glob.glob(sys.argv[1] + "/**/pass*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(fn, newline="")):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "wgrad" not in k and "Cijk" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
        acc[k]["dur_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k]["dur_ns"] += 1
        acc[k]["regs"] = int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count") or 0); cnt[k]["regs"] = 1
for k, a in acc.items():
    v = {c: a[c] / max(cnt[k][c], 1) * (1 if c in ("dur_ns", "regs") else 1) for c in a}
    n = cnt[k]["GRBM_GUI_ACTIVE"] or 1
    gui = v.get("GRBM_GUI_ACTIVE", 0)
    print(k, {c: round(x, 1) for c, x in v.items()})
    if gui:
        print("   mfma_util", round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / 8 * 1024), 3), "lds_idx_active_frac", round(v.get("SQ_LDS_IDX_ACTIVE", 0) / (gui / 8 * 256), 3),
              "bank_conflict/idx", round(v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3),
              "wait_any/wave", round(v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), 3), "wait_inst_any/wave", round(v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), 3),
              "wait_inst_lds/wave", round(v.get("SQ_WAIT_INST_LDS", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), 3),
              "fetch_GB", round(v.get("FETCH_SIZE", 0) * 2048 / 1e9, 3), "write_GB", round(v.get("WRITE_SIZE", 0) * 1024 / 1e9, 3), "eff_clock_GHz", round(gui / 8 / max(v["dur_ns"], 1), 3))
PY
