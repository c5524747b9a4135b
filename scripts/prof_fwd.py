import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lwm_amd import _capi, ops
lib = C.CDLL(os.path.join(ROOT, "scripts", "liblwm_prof.so"))
lib.lwm_attn_fwd.argtypes = [C.POINTER(_capi.LwmAttnArgs), C.c_void_p]
S, H = 32768, 32
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda: torch.randn(1, S, H, 128, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
q, k, v = mk(), mk(), mk()
a = ops._base(q, k, v, q_start=0, k_start=0, causal=True, seg_q=None, seg_k=None, key_valid=None, scale=None)
out = torch.empty_like(q); lse = torch.empty(1, H, S, dtype=torch.float32, device="cuda")
a.out = ops._t4(out, "out"); a.lse = lse.data_ptr(); a.final_out = 1
dbg = torch.zeros(8 * 8, dtype=torch.int64, device="cuda")
a.out_acc = dbg.data_ptr()
for _ in range(2):
    assert lib.lwm_attn_fwd(C.byref(a), None) == 0
    torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(8, 8)
print("wave   S-mfma  softmax  PV-mfma  stage-write  barrier | tiles  total")
for w in range(8):
    n = max(d[w, 5], 1)
    print(w, [round(float(d[w, i]) / n, 1) for i in (0, 1, 2)], round(float(d[w,3])/ (n/2),1), round(float(d[w,4])/(n/2),1), d[w, 5], d[w, 6])
