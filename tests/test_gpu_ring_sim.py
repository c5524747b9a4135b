"""The N>1 ring path with the REAL HIP block kernels on one GPU: n Python threads play
the n ranks of the "sp" group (same device, same stream) and exchange K/V and the
travelling dK/dV carries through in-process queues with the semantics of
TorchRingComm.rotate (send to rank+1, receive from rank-1).  This exercises what the
gloo tests (CPU stand-in kernels) and the single-rank GPU tests cannot: strided segment
views, f32 carries across ring steps, zigzag ownership and the causal pair skipping,
through lwm_attn_fwd / lwm_attn_bwd_* themselves.  Result must equal ring = 1."""
import queue
import threading

import numpy as np
import pytest

from oracle import attention_ref as R

pytestmark = pytest.mark.gpu


class _Handle:
    def __init__(self, q, n):
        self.q, self.n = q, n

    def wait(self):
        return self.q.get(timeout=60)


class ThreadComm:
    """rank r puts into rank (r+1)%n's inbox, takes from its own."""

    def __init__(self, rank, size, inboxes):
        self.rank, self.size, self.inboxes = rank, size, inboxes

    def rotate(self, tensors):
        self.inboxes[(self.rank + 1) % self.size].put([t.clone() for t in tensors])
        return _Handle(self.inboxes[self.rank], len(tensors))


def _run_ring(n, layout_kind, S, H, packed):
    import torch
    from lwm_amd.ring import HipBlockOps, SeqLayout, ring_attention, ring_backward, ring_forward
    g = torch.Generator().manual_seed(0)
    mk = lambda: torch.randn(1, S, H, 128, generator=g).to(torch.bfloat16).cuda()
    q, k, v, do = mk(), mk(), mk(), mk()
    seg = None
    if packed:
        seg = torch.zeros(1, S, dtype=torch.int32)
        seg[:, S // 3:] = 1
        seg[:, (5 * S) // 8:] = 2
        seg = seg.cuda()
    lay = SeqLayout(layout_kind, n, S)
    inboxes = [queue.Queue() for _ in range(n)]
    res, errs = [None] * n, []

    def worker(r):
        try:
            idx = lay.global_index(r).cuda()
            ql, kl, vl, dol = (t[:, idx].clone() for t in (q, k, v, do))
            comm = ThreadComm(r, n, inboxes)
            # the driver functions directly: torch runs every backward() of a device on ONE
            # autograd thread, which would serialise (deadlock) the n simulated ranks
            out, lses = ring_forward(HipBlockOps, comm, ql, kl, vl, layout=lay, causal=True, segment_ids=seg)
            dq, dk, dv = ring_backward(HipBlockOps, comm, ql, kl, vl, out, lses, dol, layout=lay, causal=True,
                                       segment_ids=seg)
            torch.cuda.synchronize()
            res[r] = (idx.cpu(), out, dq, dk, dv)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not errs, errs
    # single-device result with the same kernels
    q1, k1, v1 = (t.clone().requires_grad_(True) for t in (q, k, v))
    o1 = ring_attention(q1, k1, v1, causal=True, segment_ids=seg)
    o1.backward(do)
    full = [torch.zeros_like(q) for _ in range(4)]
    for idx, o, gq, gk, gv in res:
        for dst, src in zip(full, (o, gq, gk, gv)):
            dst[:, idx.cuda()] = src
    return full, (o1.detach(), q1.grad, k1.grad, v1.grad), (q, k, v, do, seg)


@pytest.mark.parametrize("n,layout_kind,packed", [(2, "contiguous", False), (2, "zigzag", True),
                                                  (4, "zigzag", False), (8, "zigzag", True)])
def test_ring_n_equals_ring_1_on_gpu(n, layout_kind, packed):
    S, H = 256 * n, 2
    got, ref, (q, k, v, do, seg) = _run_ring(n, layout_kind, S, H, packed)
    for name, a, b in zip(("out", "dq", "dk", "dv"), got, ref):
        a, b = a.float(), b.float()
        err = ((a - b).abs().max() / b.abs().max()).item()
        assert err <= 1.6e-2, (name, err)          # both are bf16 roundings of the same sums
    # and against the fp64 oracle
    f = lambda t: t.float().cpu().numpy()
    sg = None if seg is None else seg.cpu().numpy()
    ro, _ = R.dense_attention(f(q), f(k), f(v), causal=True, seg_q=sg, seg_k=sg)
    rq, rk, rv = R.dense_attention_bwd(f(q), f(k), f(v), f(do), causal=True, seg_q=sg, seg_k=sg)
    for name, a, b in zip(("out", "dq", "dk", "dv"), got, (ro, rq, rk, rv)):
        err = np.abs(f(a) - b).max() / np.abs(b).max()
        assert err <= 2e-2, (name, err)
