import sys, json, torch
sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
import bench
from lwm_amd.ring_c import CRing
orig_null = CRing.null.__func__
for fill in (False, True):
    def null(cls, rank, size, **kw):
        ring = orig_null(cls, rank, size, **kw)
        ring._fill = fill
        return ring
    CRing.null = classmethod(null)
    fwd0 = CRing.forward
    def fwd(self, q, *a, **kw):
        if getattr(self, "_fill", False) and not getattr(self, "_filled", False):
            B, c, H, D = q.shape
            self._workspace(B, c, H, D, True)
            n = self._ws.numel() // 2 * 2
            self._ws[:n].view(torch.bfloat16).normal_()
            self._filled = True
        return fwd0(self, q, *a, **kw)
    CRing.forward = fwd
    for n, S in ((2, 131072), (4, 32768), (8, 32768)):
        m = bench.ring_model_leg(torch, n=n, S=S, reps=2)
        print("fill" if fill else "as allocated", n, S, m["per_rank_ms_per_layer"], round(m["compute_bound_tflops_per_gpu"], 1), flush=True)
    CRing.forward = fwd0
