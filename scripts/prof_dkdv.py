"""Temporary: per-phase cycle accounting of the dkdv kernel (instrumented build scripts/liblwm_prof.so)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lwm_amd import _capi, ops
lib = C.CDLL(os.path.join(ROOT, "scripts", "liblwm_prof.so"))
lib.lwm_attn_bwd_dkdv.argtypes = [C.POINTER(_capi.LwmAttnArgs), C.c_void_p]
S, H = 32768, 32
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda: torch.randn(1, S, H, 128, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
q, k, v, do = mk(), mk(), mk(), mk()
out, lse = ops.attn_fwd_block(q, k, v, causal=True)
delta = ops.attn_bwd_delta(out, do)
a = ops._bwd_base(q, k, v, do, lse, delta, dict(q_start=0, k_start=0, causal=True, seg_q=None, seg_k=None, key_valid=None, scale=None))
dk = torch.empty_like(k); dv = torch.empty_like(k)
a.dk, a.dv = ops._t4(dk, "dk"), ops._t4(dv, "dv")
a.final_out = 1
dbg = torch.zeros(8 * 8, dtype=torch.int64, device="cuda")
a.out_acc = dbg.data_ptr()
for _ in range(2):
    rc = lib.lwm_attn_bwd_dkdv(C.byref(a), None)
    torch.cuda.synchronize()
assert rc == 0
d = dbg.cpu().numpy().reshape(8, 8)
names = ["S+dP mfma", "softmax", "dV+dK mfma", "dma issue", "kernel total", "vmcnt wait", "barrier", "tiles(1 of 2 bufs)"]
print("wave " + " ".join(f"{n:>14s}" for n in names))
for w in range(8):
    n = max(d[w, 7], 1)
    print(f"{w:4d} " + " ".join(f"{(d[w,i]/n if i not in (4,7) else d[w,i]):14.1f}" for i in range(8)))
