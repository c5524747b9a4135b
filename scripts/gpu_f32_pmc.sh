#!/bin/bash
# Counter passes over the float32 attention kernels (one launch each at S = 8192, 32 heads):
#   gpurun --timeout 600 -- 'TAG=r06_f32 bash scripts/gpu_f32_pmc.sh'   -> gpurun_out/$TAG/pmc_f32.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp
O=$R/gpurun_out/${TAG:-f32}; mkdir -p "$O"; cd /tmp
cat > /tmp/f32_once.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from lwm_amd import ops
S, H = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 32
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v, do = (torch.randn(1, S, H, 128, device="cuda", generator=g) for _ in range(4))
o, lse = ops.attn_fwd_block(q, k, v, causal=True)
delta = ops.attn_bwd_delta(o, do, lse)
ops.attn_bwd_dq_block(q, k, v, do, lse, delta, causal=True)
ops.attn_bwd_dkdv_block(q, k, v, do, lse, delta, causal=True)
torch.cuda.synchronize()
PY
i=1
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pmc_f32_$i
  (timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmc_f32_$i -o p -- python /tmp/f32_once.py 2>&1 | tail -2) > $O/pmc_f32_pass$i.log
  i=$((i+1))
done
python - <<'PY' | tee $O/pmc_f32.txt
import csv, glob, collections
csv.field_size_limit(1 << 30)
k = collections.defaultdict(dict)
for fn in glob.glob("/tmp/pmc_f32_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn, newline="")):
        n = r["Kernel_Name"].split("(")[0]
        if "f32" not in n or "attn" not in n: continue
        d = k[n]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        d["regs"] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("LDS_Block_Size"))
for n, d in k.items():
    print(n, d.pop("regs"), "dur_us", d.pop("dur_us"))
    g = d.get("GRBM_GUI_ACTIVE", 0) / 8
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in d: print("   mfma_util", round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 1024), 3))
    for c, v in sorted(d.items()): print(f"   {c:32s} {v:.4g}")
PY
