import sys
sys.path.insert(0, ".")
import numpy as np, torch, pytest
from tests.test_ring_c import _run_c_ring
from oracle import attention_ref as R
n, S, H = 8, 131072, 2
bounds = [0, 40000, 70000, 100000, S]
f = lambda t, rows, h: t[:, rows, h:h + 1].float().cpu().numpy()
def go(schedule, packed):
    seg_fn = (lambda S_: torch.bucketize(torch.arange(S_), torch.tensor(bounds[1:-1]), right=True)) if packed else False
    got, (q, k, v, do, seg, kv), sent = _run_c_ring(n, S, H, True, seg_fn, False, layout="zigzag", schedule=schedule)
    out, dq, dk, dv = got
    if not packed:
        return
    for i, (a, b) in enumerate(zip(bounds[:-1], bounds[1:])):
        h = i & 1
        rows, keys = slice(b - 256, b), slice(a, b)
        ro, _ = R.dense_attention(f(q, rows, h), f(k, keys, h), f(v, keys, h), causal=True, q_start=b - 256 - a)
        e = np.abs(f(out, rows, h) - ro).max(axis=(0, 2, 3))
        print(schedule, "doc", i, "out err per 32 rows:", np.round(e.reshape(-1, 32).max(1), 3), flush=True)
go("ring", False)
go("direct", True)
go("direct", True)
