"""Per-phase cycle accounting of the forward and dK/dV kernels (S=32768, H=32, causal).

Needs the instrumented library: `bash scripts/build_prof.sh` (here), then run this on the GPU box.
Prints, per wave of the longest-running workgroup of head 0, average 100 MHz-clock ticks per tile
for each phase.  s_memtime drains the scalar queue, so the build runs ~10 % slower than the
product; use the numbers to rank phases only.
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lwm_amd import _capi, ops

lib = C.CDLL(os.environ.get("LWM_PROF_LIB") or os.path.join(ROOT, "scripts", "liblwm_prof.so"))
for n in ("lwm_attn_fwd", "lwm_attn_bwd_dkdv"):
    getattr(lib, n).argtypes = [C.POINTER(_capi.LwmAttnArgs), C.c_void_p]
S, H = 32768, 32
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda: torch.randn(1, S, H, 128, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
q, k, v, do = mk(), mk(), mk(), mk()


def show(title, dbg, names):
    d = dbg.cpu().numpy().reshape(8, 8)
    print(title)
    print("wave " + " ".join(f"{n:>12s}" for n in names) + "        tiles        total")
    for w in range(8):
        n = max(int(d[w, 5]), 1)
        per = [d[w, 0] / n, d[w, 1] / n, d[w, 2] / n, d[w, 3] / (n / 2), d[w, 4] / (n / 2)]
        print(f"{w:4d} " + " ".join(f"{x:12.1f}" for x in per) + f" {int(d[w, 5]):12d} {int(d[w, 6]):12d}")


# forward
a = ops._base(q, k, v, q_start=0, k_start=0, causal=True, seg_q=None, seg_k=None, key_valid=None, scale=None)
out = torch.empty_like(q)
lse = torch.empty(1, H, S, dtype=torch.float32, device="cuda")
a.out = ops._t4(out, "out"); a.lse = lse.data_ptr(); a.final_out = 1
dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
a.out_acc = dbg.data_ptr()
for _ in range(2):
    assert lib.lwm_attn_fwd(C.byref(a), None) == 0
    torch.cuda.synchronize()
show("forward (q tile = last, head 0)", dbg, ["S mfma", "softmax", "PV mfma", "stage write", "barrier"])

# dK/dV
out, lse = ops.attn_fwd_block(q, k, v, causal=True)
delta = ops.attn_bwd_delta(out, do)
a = ops._bwd_base(q, k, v, do, lse, delta, dict(q_start=0, k_start=0, causal=True, seg_q=None, seg_k=None,
                                                  key_valid=None, scale=None))
dk = torch.empty_like(k); dv = torch.empty_like(k)
a.dk, a.dv = ops._t4(dk, "dk"), ops._t4(dv, "dv")
a.final_out = 1
dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
a.out_acc = dbg.data_ptr()
for _ in range(2):
    assert lib.lwm_attn_bwd_dkdv(C.byref(a), None) == 0
    torch.cuda.synchronize()
show("dK/dV (key block 0, head 0)", dbg, ["S,dP mfma", "exp/dS", "dV,dK mfma", "stat+dma wait", "barrier"])
