"""GPU debug helper: dq / dk / dv of the product library against the fp64 oracle, error per block of 32 rows.
    python scripts/gpu_dbg_dkdv.py Sq Sk causal [H=1] [zero_do_from zero_do_to]   (rows of dO zeroed: isolates query units)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from lwm_amd import ops
from oracle import attention_ref as R

Sq, Sk, causal = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1
g = torch.Generator().manual_seed(0)
q, do = (torch.randn(1, Sq, H, 128, generator=g).to(torch.bfloat16) for _ in range(2))
k, v = (torch.randn(1, Sk, H, 128, generator=g).to(torch.bfloat16) for _ in range(2))
if len(sys.argv) > 6:
    do[:, int(sys.argv[5]):int(sys.argv[6])] = 0
qd, kd, vd, dod = (t.cuda() for t in (q, k, v, do))
out, lse = ops.attn_fwd_block(qd, kd, vd, causal=causal)
delta = ops.attn_bwd_delta(out, dod, lse)
dk, dv = ops.attn_bwd_dkdv_block(qd, kd, vd, dod, lse, delta, causal=causal)
dq = ops.attn_bwd_dq_block(qd, kd, vd, dod, lse, delta, causal=causal)
torch.cuda.synchronize()
f = lambda t: t.float().cpu().numpy()
rq, rk, rv = R.dense_attention_bwd(f(q), f(k), f(v), f(do), causal=causal)
np.set_printoptions(linewidth=220, precision=3, suppress=True)
for name, got, ref in (("dk", f(dk), rk), ("dv", f(dv), rv), ("dq", f(dq), rq)):
    e = np.abs(got - ref)
    nan = np.isnan(got)
    e[nan] = 1e9
    blk = e.max(axis=(0, 2, 3))
    blk = np.pad(blk, (0, (-len(blk)) % 32)).reshape(-1, 32).max(axis=1)
    print(name, sys.argv[1:], "nan", int(nan.sum()), "ref max %.3f" % np.abs(ref).max(), "err per 32 rows:", blk[:24])
