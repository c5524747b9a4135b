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
from tests._parity import check as _check, check_dq as _check_dq

pytestmark = pytest.mark.gpu


class _Handle:
    def __init__(self, q, n):
        self.q, self.n = q, n

    def wait(self):
        return self.q.get(timeout=60)


class _PairHandle:
    def __init__(self, links, me, recvs):
        self.links, self.me, self.recvs = links, me, recvs

    def wait(self):
        for peer, buf in self.recvs:
            buf.copy_(self.links[(peer, self.me)].get(timeout=60))
        self.recvs = []


class ThreadComm:
    """rank r puts into rank (r+1)%n's inbox, takes from its own; exchange_async (the mesh
    schedule's transport) uses one FIFO per ordered (src, dst) pair -- messages between a pair
    match in posting order, as grouped RCCL send/recv do."""

    def __init__(self, rank, size, inboxes, schedule="ring", links=None):
        self.rank, self.size, self.inboxes = rank, size, inboxes
        self.schedule, self.links = schedule, links
        self.sent_bytes = 0

    def rotate(self, tensors):
        self.sent_bytes += sum(t.numel() * t.element_size() for t in tensors)
        self.inboxes[(self.rank + 1) % self.size].put([t.clone() for t in tensors])
        return _Handle(self.inboxes[self.rank], len(tensors))

    def exchange_async(self, sends, recvs):
        for peer, t in sends:
            self.sent_bytes += t.numel() * t.element_size()
            self.links[(self.rank, peer)].put(t.clone())
        return _PairHandle(self.links, self.rank, list(recvs))


def _run_ring(n, layout_kind, S, H, packed, schedule="ring", B=1):
    import torch
    from lwm_amd.ring import HipBlockOps, SeqLayout, ring_attention, ring_backward, ring_forward
    g = torch.Generator().manual_seed(0)
    mk = lambda: torch.randn(B, S, H, 128, generator=g).to(torch.bfloat16).cuda()
    q, k, v, do = mk(), mk(), mk(), mk()
    seg = None
    if callable(packed):
        seg = packed(S).to(torch.int32)[None].expand(B, S).contiguous().cuda()
    elif packed:
        seg = torch.zeros(B, S, dtype=torch.int32)
        seg[:, S // 3:] = 1
        seg[:, (5 * S) // 8:] = 2
        seg = seg.cuda()
    lay = layout_kind if not isinstance(layout_kind, str) else SeqLayout(layout_kind, n, S)      # (or a SeqLayout object)
    inboxes = [queue.Queue() for _ in range(n)]
    links = {(a, b): queue.Queue() for a in range(n) for b in range(n)}
    res, errs, sent = [None] * n, [], [0] * n

    def worker(r):
        try:
            idx = lay.global_index(r).cuda()
            ql, kl, vl, dol = (t[:, idx].clone() for t in (q, k, v, do))
            comm = ThreadComm(r, n, inboxes, schedule, links)
            # the driver functions directly: torch runs every backward() of a device on ONE
            # autograd thread, which would serialise (deadlock) the n simulated ranks
            out, lses = ring_forward(HipBlockOps, comm, ql, kl, vl, layout=lay, causal=True, segment_ids=seg)
            dq, dk, dv = ring_backward(HipBlockOps, comm, ql, kl, vl, out, lses, dol, layout=lay, causal=True,
                                       segment_ids=seg)
            torch.cuda.synchronize()
            sent[r] = comm.sent_bytes
            res[r] = (idx.cpu(), out, dq, dk, dv)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=600)
    assert not errs, errs
    # single-device result with the same kernels
    q1, k1, v1 = (t.clone().requires_grad_(True) for t in (q, k, v))
    o1 = ring_attention(q1, k1, v1, causal=True, segment_ids=seg)
    o1.backward(do)
    full = [torch.zeros_like(q) for _ in range(4)]
    for idx, o, gq, gk, gv in res:
        for dst, src in zip(full, (o, gq, gk, gv)):
            dst[:, idx.cuda()] = src
    _run_ring.sent_bytes = sum(sent)
    return full, (o1.detach(), q1.grad, k1.grad, v1.grad), (q, k, v, do, seg)


@pytest.mark.parametrize("n,layout_kind,packed,schedule", [
    (2, "contiguous", False, "ring"), (2, "zigzag", True, "ring"), (4, "zigzag", False, "ring"),
    (8, "zigzag", True, "ring"),
    (2, "zigzag", True, "mesh"), (4, "contiguous", False, "mesh"), (8, "zigzag", True, "mesh")])
def test_ring_n_equals_ring_1_on_gpu(n, layout_kind, packed, schedule):
    S, H = 256 * n, 2
    got, ref, (q, k, v, do, seg) = _run_ring(n, layout_kind, S, H, packed, schedule)
    f = lambda t: t.float().cpu().numpy()
    for name, a, b in zip(("out", "dq", "dk", "dv"), got, ref):     # both are bf16 roundings of the same sums
        # (dq globally only: the two runs save outputs that differ in a last bit here and there, and a row whose
        # gradient cancels is made of exactly that; its own row check follows, against the oracle)
        _check(f"{name} ring{n} vs ring1", f(a), f(b), **({"row_tol": None} if name == "dq" else {}))
    # and against the fp64 oracle
    sg = None if seg is None else seg.cpu().numpy()
    ro, _ = R.dense_attention(f(q), f(k), f(v), causal=True, seg_q=sg, seg_k=sg)
    rq, rk, rv, rqx = R.dense_attention_bwd(f(q), f(k), f(v), f(do), causal=True, seg_q=sg, seg_k=sg, out_saved=f(got[0]))
    for name, a, b in zip(("out", "dk", "dv"), (got[0], got[2], got[3]), (ro, rk, rv)):
        _check(f"{name} ring{n}", f(a), b)
    _check_dq(f"dq ring{n}", f(got[1]), rq, rqx)


@pytest.mark.parametrize("n,layout_kind,packed,S", [(2, "zigzag", True, 1024), (4, "zigzag", True, 2048), (4, "contiguous", False, 1024),
                                                     (8, "zigzag", True, 4096), (4, "balanced", True, 4096)])
def test_mesh_gathered_form_on_gpu(n, layout_kind, packed, S, monkeypatch):
    """lwm_amd/ring.py's own gathered form of the mesh schedule (segments of a multiple of 256 rows, B = 1, causal): the
    fetched segments in position order in one buffer, piecewise position maps, two launches per kernel -- against the
    single device, the fp64 oracle, and the per-pair form of the same driver (LWM_RING_FORM=pairs): same bytes."""
    import torch
    from lwm_amd import ring as ring_mod
    H = 2
    lay_kind = layout_kind
    if layout_kind == "balanced":
        lens = [S // 3, S // 4 + 7, S - S // 3 - S // 4 - 7]
        lay_kind = ring_mod.balanced_layout(n, S, lens, chunks_per_rank=4)
        assert lay_kind.kind == "table"
        seg_fn = lambda S_: torch.bucketize(torch.arange(S_), torch.tensor(np.cumsum(lens)[:-1]), right=True)
        packed = seg_fn
    calls = {"fwd": 0}
    real_fwd = ring_mod.HipBlockOps.fwd
    monkeypatch.setattr(ring_mod.HipBlockOps, "fwd", staticmethod(lambda *a, **kw: (calls.__setitem__("fwd", calls["fwd"] + 1), real_fwd(*a, **kw))[1]))
    got, ref, (q, k, v, do, seg) = _run_ring(n, lay_kind, S, H, packed, "mesh")
    assert calls["fwd"] <= 2 * n + 1, calls            # two launches per rank (+ the single-device reference)
    gathered_bytes = _run_ring.sent_bytes
    f = lambda t: t.float().cpu().numpy()
    sg = None if seg is None else seg.cpu().numpy()
    ro, _ = R.dense_attention(f(q), f(k), f(v), causal=True, seg_q=sg, seg_k=sg)
    rq, rk, rv, rqx = R.dense_attention_bwd(f(q), f(k), f(v), f(do), causal=True, seg_q=sg, seg_k=sg, out_saved=f(got[0]))
    for name, a, b in zip(("out", "dk", "dv"), (got[0], got[2], got[3]), (ro, rk, rv)):
        _check(f"{name} mesh gathered n={n} {layout_kind}", f(a), b)
    _check_dq(f"dq mesh gathered n={n} {layout_kind}", f(got[1]), rq, rqx)
    monkeypatch.setenv("LWM_RING_FORM", "pairs")
    calls["fwd"] = 0
    pairs, _, _ = _run_ring(n, lay_kind, S, H, packed, "mesh")
    assert calls["fwd"] > 2 * n + 1 or n == 2, calls
    assert _run_ring.sent_bytes == gathered_bytes
    for name, a, b in zip(("out", "dq", "dk", "dv"), got, pairs):
        assert ((a.float() - b.float()).abs().max() / b.float().abs().max()).item() <= 8e-3, name


def _doc_windows(bounds, w=256):
    """[(doc_start, doc_end, window_start)]: the last `w` rows of every document."""
    return [(a, b, b - w) for a, b in zip(bounds[:-1], bounds[1:])]


@pytest.mark.parametrize("schedule,packed", [("ring", False), ("mesh", True)])
def test_ring8_at_config3_shard_shapes_vs_oracle(schedule, packed):
    """BASELINE configs[2]: S = 131072 over an 8-rank zigzag ring, c = 16384 per rank (two half-chunks
    of 8192 at global offsets r*8192 and (15-r)*8192) -- the shard shapes and (q_start, k_start)
    offsets the metric names, through the real kernels and the real driver with thread-played ranks,
    checked against the fp64 oracle in windows that land on the first, a middle and the last rank.
    Unpacked run (reference "ring" schedule): out / dq windows anywhere cost 256 x (keys so far); the
    complete dk / dv of the LAST keys need only the last queries.  Packed run ("mesh" schedule, four
    documents ending inside the shards of ranks 4, 7, 3 and 0): every gradient of a document's last
    rows is checkable against that document alone."""
    import torch
    n, S, H = 8, 131072, 2
    bounds = [0, 40000, 70000, 100000, S]
    seg_fn = (lambda S_: torch.bucketize(torch.arange(S_), torch.tensor(bounds[1:-1]), right=True)) if packed else False
    got, ref, (q, k, v, do, seg) = _run_ring(n, "zigzag", S, H, seg_fn, schedule)
    for name, a, b in zip(("out", "dq", "dk", "dv"), got, ref):      # ring 8 == ring 1 (same kernels)
        a, b = a.float(), b.float()
        assert ((a - b).abs().max() / b.abs().max()).item() <= 8e-3, name
    f = lambda t, rows, h: t[:, rows, h:h + 1].float().cpu().numpy()
    out, dq, dk, dv = got
    if not packed:
        # owner of row x: half-chunk x // 8192 -> rank min(c, 15 - c).  96: rank 0 (early chunk);
        # 65408: crosses rank 7's two half-chunks; S-256: rank 0 (late chunk); 36000: rank 4
        for h, r0 in ((0, 96), (1, 65408), (0, S - 256), (1, 36000)):
            rows, keys = slice(r0, r0 + 256), slice(0, r0 + 256)
            ro, _ = R.dense_attention(f(q, rows, h), f(k, keys, h), f(v, keys, h), causal=True, q_start=r0)
            rq, _, _, rqx = R.dense_attention_bwd(f(q, rows, h), f(k, keys, h), f(v, keys, h), f(do, rows, h),
                                                  causal=True, q_start=r0, out_saved=f(out, rows, h))
            _check(f"out ring8 row {r0}", f(out, rows, h), ro)
            _check_dq(f"dq ring8 row {r0}", f(dq, rows, h), rq, rqx)
        K0, h = S - 512, 1
        rows, allk = slice(K0, S), slice(0, S)
        _, rk, rv = R.dense_attention_bwd(f(q, rows, h), f(k, allk, h), f(v, allk, h), f(do, rows, h),
                                          causal=True, q_start=K0)
        _check("dk ring8 last keys", f(dk, slice(K0, K0 + 256), h), rk[:, K0:K0 + 256])
        _check("dv ring8 last keys", f(dv, slice(K0 + 256, S), h), rv[:, K0 + 256:])
        return
    for i, (a, b, w0) in enumerate(_doc_windows(bounds)):
        # a document longer than 30000 rows is too much for a dense fp64 oracle: its last rows only need
        # the keys of the document, its last KEYS only the queries after them -- take the last 2048 rows
        # as queries against the whole document (complete for out/dq of these rows and dk/dv of keys >= w0)
        h = i & 1
        qa = w0          # (the window's own rows are all the queries its out / dq / dk / dv need: 8x less oracle time)
        rows, keys = slice(qa, b), slice(a, b)
        ro, _ = R.dense_attention(f(q, rows, h), f(k, keys, h), f(v, keys, h), causal=True, q_start=qa - a)
        rq, rk, rv, rqx = R.dense_attention_bwd(f(q, rows, h), f(k, keys, h), f(v, keys, h), f(do, rows, h),
                                                causal=True, q_start=qa - a, out_saved=f(out, rows, h))
        win = slice(w0, b)
        _check(f"out ring8 doc {i}", f(out, win, h), ro[:, w0 - qa:])
        _check_dq(f"dq ring8 doc {i}", f(dq, win, h), rq[:, w0 - qa:], rqx[:, w0 - qa:])
        _check(f"dk ring8 doc {i}", f(dk, win, h), rk[:, w0 - a:])
        _check(f"dv ring8 doc {i}", f(dv, win, h), rv[:, w0 - a:])


def test_mesh8_at_1m_token_offsets_vs_oracle():
    """BASELINE configs[4]'s offsets: S = 1,048,576 over an 8-rank zigzag ring under the mesh schedule
    (c = 131072 per rank, half-chunks at global offsets up to 983040), one head, packed 4096-token
    documents so that the fp64 oracle is a 4096 x 4096 problem wherever it is asked.  Documents are
    sampled at the start, across the middle seam (rank 7's two half-chunks) and at the very end."""
    import torch
    n, S, H, doc = 8, 1 << 20, 1, 4096
    got, ref, (q, k, v, do, seg) = _run_ring(n, "zigzag", S, H, lambda S_: torch.arange(S_) // doc, "mesh")
    for name, a, b in zip(("out", "dq", "dk", "dv"), got, ref):
        assert torch.equal(a, b) or ((a.float() - b.float()).abs().max() / b.float().abs().max()).item() <= 8e-3, name
    f = lambda t, rows: t[:, rows, 0:1].float().cpu().numpy()
    for d0 in (0, (S // 2) - doc, S // 2, 131072 * 5 + 8 * doc, S - doc):
        rows = slice(d0, d0 + doc)
        ro, _ = R.dense_attention(f(q, rows), f(k, rows), f(v, rows), causal=True)
        rq, rk, rv, rqx = R.dense_attention_bwd(f(q, rows), f(k, rows), f(v, rows), f(do, rows), causal=True,
                                                out_saved=f(got[0], rows))
        for name, a, b in zip(("out", "dk", "dv"), (got[0], got[2], got[3]), (ro, rk, rv)):
            _check(f"{name} mesh8@1M doc {d0 // doc}", f(a, rows), b)
        _check_dq(f"dq mesh8@1M doc {d0 // doc}", f(got[1], rows), rq, rqx)


@pytest.mark.parametrize("schedule", ["ring", "mesh"])
def test_ring_batch_2_on_gpu(schedule):
    """B = 2: every per-segment view handed to the kernels is strided in the batch dimension."""
    got, ref, (q, k, v, do, seg) = _run_ring(4, "zigzag", 1024, 2, True, schedule, B=2)
    f = lambda t: t.float().cpu().numpy()
    for name, a, b in zip(("out", "dq", "dk", "dv"), got, ref):
        _check(f"{name} ring4 B=2", f(a), f(b), **({"row_tol": None} if name == "dq" else {}))


def test_mesh_schedule_moves_fewer_bytes_than_the_ring():
    """zigzag + causal: a rank's early half-chunk is invisible to every earlier rank's queries and
    its late half-chunk to every later rank's early queries, so the direct fetch ships ~3/4 of
    the K/V bytes the rotating ring does, and no f32 carry is forwarded through bystanders."""
    n, S, H = 8, 2048, 2
    _run_ring(n, "zigzag", S, H, False, "ring")
    ring_bytes = _run_ring.sent_bytes
    _run_ring(n, "zigzag", S, H, False, "mesh")
    mesh_bytes = _run_ring.sent_bytes
    assert mesh_bytes < 0.8 * ring_bytes, (mesh_bytes, ring_bytes)


class ThreadGroupComm(ThreadComm):
    """adds all_gather / exchange (rank-major, barrier-synchronised) for the inference drivers."""

    def __init__(self, rank, size, inboxes, shared, barrier):
        super().__init__(rank, size, inboxes)
        self.shared, self.barrier = shared, barrier

    def all_gather(self, t):
        import torch
        self.shared[self.rank] = t.clone()
        self.barrier.wait()
        out = torch.stack([self.shared[r] for r in range(self.size)])
        self.barrier.wait()
        return out

    def exchange(self, sends, recvs):
        for peer, t in sends:
            self.shared.setdefault(("x", self.rank, peer), []).append(t.clone())
        self.barrier.wait()
        cursor = {}
        for peer, buf in recvs:
            i = cursor.get(peer, 0)
            buf.copy_(self.shared[("x", peer, self.rank)][i])
            cursor[peer] = i + 1
        self.barrier.wait()
        for peer, _ in sends:
            self.shared.pop(("x", self.rank, peer), None)
        self.barrier.wait()


@pytest.mark.parametrize("n", [2, 4])
def test_sharded_cache_and_decode_on_gpu(n):
    """cache_update + ring_inference with the real kernels, n simulated ranks: prefill rows land
    in the owning shards, then one decode step over the sharded cache equals dense attention."""
    import torch
    from lwm_amd.ring import HipBlockOps, cache_update, ring_inference
    B, H, D, c, P, start = 2, 4, 128, 384, 64 * n, 100
    max_len = c * n
    g = torch.Generator().manual_seed(1)
    mk = lambda s: torch.randn(B, s, H, D, generator=g).to(torch.bfloat16).cuda()
    k_new, v_new, k_dec, v_dec, q_dec = mk(P), mk(P), mk(1), mk(1), mk(1)
    am = torch.ones(B, max_len, dtype=torch.bool)
    am[:, 130:140] = False
    inboxes = [queue.Queue() for _ in range(n)]
    shared, barrier = {}, threading.Barrier(n)
    res, errs = [None] * n, []

    def worker(r):
        try:
            comm = ThreadGroupComm(r, n, inboxes, shared, barrier)
            ck = torch.zeros(B, c, H, D, dtype=torch.bfloat16, device="cuda")
            cv = torch.zeros_like(ck)
            p = P // n
            idx = cache_update(HipBlockOps, comm, ck, cv, k_new[:, r * p:(r + 1) * p].contiguous(),
                               v_new[:, r * p:(r + 1) * p].contiguous(), start, new_sharded=True)
            idx = cache_update(HipBlockOps, comm, ck, cv, k_dec, v_dec, idx, new_sharded=False)
            mask = ((torch.arange(max_len)[None, None, :] <= idx - 1) & am[:, None, :]).to(torch.uint8).cuda()
            out = ring_inference(HipBlockOps, comm, q_dec, ck, cv, mask, q_sharded=False)
            torch.cuda.synchronize()
            res[r] = (idx, ck.cpu(), cv.cpu(), out.float().cpu())
        except Exception as e:  # pragma: no cover
            errs.append(e)
            barrier.abort()

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not errs, errs
    ck = torch.zeros(B, max_len, H, D)
    cv = torch.zeros(B, max_len, H, D)
    ck[:, start:start + P], cv[:, start:start + P] = k_new.float().cpu(), v_new.float().cpu()
    ck[:, start + P], cv[:, start + P] = k_dec[:, 0].float().cpu(), v_dec[:, 0].float().cpu()
    assert all(r[0] == start + P + 1 for r in res)
    assert torch.equal(torch.cat([r[1] for r in res], 1).float(), ck)
    assert torch.equal(torch.cat([r[2] for r in res], 1).float(), cv)
    rmask = R.decode_mask(B, 1, max_len, start + P, am.numpy())
    ro, _ = R.dense_attention(q_dec.float().cpu().numpy(), ck.numpy(), cv.numpy(), causal=False, dense_mask=rmask)
    for r in res:
        _check("out sharded decode", r[3].numpy(), ro)


def test_reserved_cu_stream_runs_the_attention_kernels_bit_for_bit():
    """lwm_amd.ring_c.reserved_cu_stream (hipExtStreamCreateWithCUMask): the attention forward + backward on a stream that
    leaves 16 CUs alone give the bits of the default stream (the knob bench.py prices as LWM_RING_RESERVE_CUS)."""
    import torch
    from lwm_amd.ring import ring_attention
    from lwm_amd.ring_c import reserved_cu_stream
    g = torch.Generator().manual_seed(5)
    q, k, v, do = (torch.randn(1, 2048, 4, 128, generator=g).to(torch.bfloat16).cuda() for _ in range(4))

    def run():
        qd, kd, vd = (t.clone().requires_grad_(True) for t in (q, k, v))
        out = ring_attention(qd, kd, vd, causal=True)
        out.backward(do)
        return out.detach(), qd.grad, kd.grad, vd.grad

    ref = run()
    torch.cuda.synchronize()
    st = reserved_cu_stream(16)
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        got = run()
    st.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    import pytest as _pytest
    with _pytest.raises(ValueError):
        reserved_cu_stream(0)
