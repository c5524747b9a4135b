"""Ring driver: the sequence-parallel schedule around the per-block kernels.

Reference behaviour being replaced (lwm/llama.py:539-569 + the `ringattention`
package): the sequence is sharded over mesh axis "sp";
for ring step t rank r holds the K/V block of rank (r - t) mod n, runs the
blockwise update of its local queries against it, then K/V move i -> i+1
(lax.ppermute).  The backward rotates k, v, dk, dv the same way.

MI355X-first differences (results identical):
  * the exchange is posted BEFORE the step's kernels (RCCL send/recv over xGMI
    on its own stream) into a second buffer, and waited for after them, so the
    transfer of block t+1 overlaps the MFMA work on block t;
  * K/V blocks wholly in the causal future of the local queries launch nothing;
  * an optional "zigzag" ownership (rank r owns half-chunks r and 2n-1-r)
    balances causal work across ranks; ownership is a property of the layout
    object, the kernels only ever see (q_start, k_start) global offsets.

Two exchange schedules produce the same results:
  * "ring"  -- the reference's: K/V (and, in the backward, the f32 dK/dV carries) hop
    i -> i+1 once per step; every byte crosses ONE xGMI link per step, serially.
  * "mesh"  -- MI355X's xGMI is a full mesh (a direct link to each of the 7 peers), so
    nothing needs forwarding: every rank posts, up front and in one grouped launch, the
    direct transfer of exactly those K/V segments each peer's queries can see (all links
    busy at once, ~1/4 fewer bytes under zigzag+causal), works through the blocks as in
    the ring, returns each block's f32 dK/dV PARTIAL straight to its owner while the
    next block is computed, and the owner reduces the partials in fixed rank order
    (lwm_sum_f32_to_bf16).  Default; LWM_RING_SCHEDULE=ring|mesh overrides.

The driver is written against two small interfaces so the schedule can be
exercised on CPU (gloo) in tests with a stand-in block backend:
  block ops : fwd / bwd_delta / bwd_dq / bwd_dkdv / cast / zeros / empty
  block ops (mesh) : + sum_cast
  comm      : rank, size, schedule, rotate(list[tensor]) -> handle.wait() -> list[tensor],
              exchange_async([(peer, tensor)], [(peer, buffer)]) -> handle
The product block ops (`HipBlockOps`) call liblwm_hip.so and nothing else.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import ops as _ops


# ----------------------------------------------------------------- layouts
class SeqLayout:
    """Which global token positions a rank owns, as contiguous segments
    (local_offset, length, global_start) of its local shard.

    kind "contiguous" (the reference's, lwm/llama.py:560-562), "zigzag" (half-chunks r and 2n-1-r), or "table": the
    sequence cut into len(owner) = n * P equal chunks, chunk j owned by rank owner[j], P chunks per rank held in ascending
    position order -- any ownership a loader computes (balanced_layout for packed batches); include/lwm_hip.h
    LWM_RING_LAYOUT_TABLE."""

    def __init__(self, kind: str, n: int, seq_len: int, owner=None):
        if kind not in ("contiguous", "zigzag", "table"):
            raise ValueError(f"unknown layout {kind!r}")
        if kind == "zigzag" and n == 1:
            kind = "contiguous"
        self.owner = None
        if kind == "table":
            owner = [int(r) for r in owner]
            if len(owner) % n or seq_len % len(owner) or any(owner.count(r) != len(owner) // n for r in range(n)):
                raise ValueError(f"ownership table: {len(owner)} chunks must be n * P, divide {seq_len}, every rank owning P")
            self.owner = owner
        div = n if kind == "contiguous" else 2 * n if kind == "zigzag" else len(owner)
        if seq_len % div:
            raise ValueError(f"seq_len {seq_len} not divisible by {div} for layout {kind}")
        self.kind, self.n, self.seq_len = kind, n, seq_len
        self.local_len = seq_len // n

    def segments(self, rank: int):
        c = self.local_len
        if self.kind == "contiguous":
            return [(0, c, rank * c)]
        if self.kind == "table":
            cs = self.seq_len // len(self.owner)
            # (one segment per chunk, also where two chunks are adjacent: every rank has the same number of segments,
            #  which the neighbour-ring schedule's travelling carries rely on)
            return [(i * cs, cs, j * cs) for i, j in enumerate(j for j, r in enumerate(self.owner) if r == rank)]
        h = c // 2
        return [(0, h, rank * h), (h, h, (2 * self.n - 1 - rank) * h)]

    def global_index(self, rank: int) -> torch.Tensor:
        """Global positions of the rank's local rows (for sharding tensors in tests/loaders)."""
        return torch.cat([torch.arange(g, g + ln) for _, ln, g in self.segments(rank)])


def balanced_layout(n, seq_len, doc_lengths=None, chunks_per_rank=4, align=256):
    """Ownership that balances CAUSAL, DOCUMENT-MASKED work over n ranks: the sequence is cut into n * chunks_per_rank
    chunks, each weighed by its visible (query, key) pairs -- a query at offset p of its document sees p + 1 keys; without
    documents the whole sequence is one -- and handed out heaviest first to the least loaded rank that still has room
    (every rank gets chunks_per_rank chunks).  Zigzag is the two-chunk answer for ONE document; for BASELINE configs[4]'s
    packing (1,048,576 tokens in 15 documents, n = 8) it leaves the slowest rank at 1.9x the mean, this at ~1.03 with four
    chunks.  The loader knows the document lengths (lwm/data.py packs them); all ranks must build the same table.
    -> SeqLayout("table"); falls back to fewer chunks per rank while a chunk would not be a multiple of `align` rows.
    More than 8 chunks per rank (LWM_MAX_PIECES of the kernels' piecewise position maps) is accepted: ring_attention then
    runs the table through this module's pair-form driver instead of the C driver's gathered form."""
    import numpy as np
    P = int(chunks_per_rank)
    while P > 1 and seq_len % (n * P * align):
        P -= 1
    if seq_len % (n * P):
        raise ValueError(f"seq_len {seq_len} is not divisible by {n * P}")
    if n == 1:
        return SeqLayout("contiguous", 1, seq_len)
    lens = [seq_len] if not doc_lengths else [int(x) for x in doc_lengths]
    if sum(lens) != seq_len or min(lens) <= 0:
        raise ValueError("doc_lengths must be positive and sum to seq_len")
    m, cs = n * P, seq_len // (n * P)
    # pairs seen by the queries of [a, b) of a document that starts at s: sum_{p=a-s}^{b-s-1} (p + 1)
    tri = lambda x: x * (x + 1) // 2
    w = np.zeros(m, dtype=np.float64)
    s = 0
    for ln in lens:
        for j in range(s // cs, (s + ln - 1) // cs + 1):
            a, b = max(s, j * cs), min(s + ln, (j + 1) * cs)
            w[j] += tri(b - s) - tri(a - s)
        s += ln
    owner, load, room = [0] * m, [0.0] * n, [P] * n
    for j in sorted(range(m), key=lambda j_: (-w[j_], j_)):
        r = min((r_ for r_ in range(n) if room[r_]), key=lambda r_: (load[r_], r_))
        owner[j], load[r], room[r] = r, load[r] + w[j], room[r] - 1
    return SeqLayout("table", n, seq_len, owner=owner)


def pair_visible(qseg, kseg, causal: bool) -> bool:
    """False when the (q segment, k segment) pair is wholly above the causal diagonal."""
    if not causal:
        return True
    _, qlen, qg = qseg
    _, _, kg = kseg
    return kg <= qg + qlen - 1


# ----------------------------------------------------------------- comm
class _Handle:
    def __init__(self, reqs, bufs, keep=None):
        self.reqs, self.bufs, self.keep = reqs, bufs, keep

    def wait(self):
        for r in self.reqs:
            r.wait()
        self.reqs = []
        return self.bufs


class TorchRingComm:
    """send to (rank+1)%n, receive from (rank-1)%n -- RCCL (backend "nccl") on
    GPUs, gloo in the CPU tests."""

    def __init__(self, group=None, schedule=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.schedule = schedule or os.environ.get("LWM_RING_SCHEDULE") or "mesh"
        if self.schedule not in ("ring", "mesh"):
            raise ValueError(f"unknown exchange schedule {self.schedule!r}")
        self._pool, self._flip = {}, {}
        self._dst = dist.get_global_rank(group, (self.rank + 1) % self.size) if group is not None \
            else (self.rank + 1) % self.size
        self._src = dist.get_global_rank(group, (self.rank - 1) % self.size) if group is not None \
            else (self.rank - 1) % self.size

    def pooled(self, tag, shape, dtype, device):
        """Exchange buffers are allocated ONCE per (role, shape) and reused by every layer and step: a
        receive into a pooled buffer is enqueued behind everything already on the compute stream (RCCL
        work waits for the issuing stream) and every driver call waits for its own sends before it
        returns, so a buffer is never overwritten while a kernel or a send still reads it."""
        key = (tag, tuple(shape), dtype, str(device))
        buf = self._pool.get(key)
        if buf is None:
            buf = self._pool[key] = torch.empty(tuple(shape), dtype=dtype, device=device)
        return buf

    def rotate(self, tensors):
        # double buffer per travelling tensor: the block received at step t is sent on at step t+1
        # while the next one lands in the other buffer
        # (one parity per travelling SET -- the backward rotates K/V and the f32 dK/dV carries alternately)
        sig = tuple((tuple(t.shape), t.dtype) for t in tensors)
        flip = self._flip[sig] = self._flip.get(sig, 0) ^ 1
        bufs = [self.pooled(("rot", sig, i, flip), t.shape, t.dtype, t.device) for i, t in enumerate(tensors)]
        p2p = []
        for t, b in zip(tensors, bufs):
            p2p.append(dist.P2POp(dist.isend, t, self._dst, self.group))
            p2p.append(dist.P2POp(dist.irecv, b, self._src, self.group))
        reqs = dist.batch_isend_irecv(p2p)
        return _Handle(reqs, bufs)

    def all_gather(self, t):
        """-> (n, *t.shape), rank-major."""
        out = torch.empty((self.size,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous(), group=self.group) if t.is_cuda else \
            dist.all_gather(list(out.unbind(0)), t.contiguous(), group=self.group)
        return out

    def exchange_async(self, sends, recvs):
        """sends: [(peer, tensor)], recvs: [(peer, empty tensor)] -- ONE grouped P2P launch
        (on RCCL the transfers of a group run concurrently, one xGMI link per peer).
        Messages between one pair of ranks match in list order.  -> handle; the sent
        tensors are kept alive by it."""
        ops_ = []
        for peer, t in sends:
            dst = dist.get_global_rank(self.group, peer) if self.group is not None else peer
            ops_.append(dist.P2POp(dist.isend, t, dst, self.group))
        for peer, t in recvs:
            src = dist.get_global_rank(self.group, peer) if self.group is not None else peer
            ops_.append(dist.P2POp(dist.irecv, t, src, self.group))
        reqs = dist.batch_isend_irecv(ops_) if ops_ else []
        return _Handle(reqs, [t for _, t in recvs], keep=[t for _, t in sends])

    def exchange(self, sends, recvs):
        self.exchange_async(sends, recvs).wait()


class SingleComm:
    rank, size = 0, 1
    schedule = "ring"

    def rotate(self, tensors):
        return _Handle([], list(tensors))

    def all_gather(self, t):
        return t.unsqueeze(0)

    def exchange_async(self, sends, recvs):
        assert not sends and not recvs
        return _Handle([], [])

    def exchange(self, sends, recvs):
        assert not sends and not recvs


# ----------------------------------------------------------------- block ops
class HipBlockOps:
    """The product backend: hand-written HIP kernels through the C ABI."""

    piece_align = 256        # piecewise position maps: cuts on multiples of 256 rows (include/lwm_hip.h)

    fwd = staticmethod(_ops.attn_fwd_block)
    bwd_delta = staticmethod(_ops.attn_bwd_delta)
    bwd_dq = staticmethod(_ops.attn_bwd_dq_block)
    bwd_dkdv = staticmethod(_ops.attn_bwd_dkdv_block)
    cast = staticmethod(_ops.cast_f32_to_bf16)
    sum_cast = staticmethod(_ops.sum_f32_to_bf16)
    fwd_splitk = staticmethod(_ops.attn_fwd_splitk)
    combine = staticmethod(_ops.attn_combine)
    cache_write = staticmethod(_ops.kv_cache_write)

    @staticmethod
    def empty(shape, dtype, like):
        return torch.empty(shape, dtype=dtype, device=like.device)

    @staticmethod
    def zeros(shape, dtype, like):
        return torch.zeros(shape, dtype=dtype, device=like.device)


# The backward is deterministic: lwm_attn_bwd_delta + lwm_attn_bwd_dkdv + lwm_attn_bwd_dq, no atomics, fixed summation
# orders (7 GEMM units executed: S and dP are recomputed by the dq kernel).  A 5-unit form that stored bf16 dq partials
# was measured in rounds 2-3 and retired in round 4 (profiles/r04_backward.md).


# ----------------------------------------------------------------- helpers
class _MaskSlices:
    """Per-segment views of the replicated (B, S_global) masks, cut ONCE per driver call: the forward,
    dQ and dK/dV launches of a (q segment, k segment) pair get the same tensor objects, so whatever a
    block backend derives from them (the HIP backend's segment-block hint tables) is derived once."""

    def __init__(self, segment_ids, key_valid):
        self.segment_ids, self.key_valid = segment_ids, key_valid
        self._seg, self._kv = {}, {}

    def _cut(self, memo, src, g, ln):
        if src is None:
            return None
        got = memo.get((g, ln))
        if got is None:
            got = memo[(g, ln)] = src[:, g:g + ln].contiguous()
        return got

    def __call__(self, qseg, kseg):
        _, qlen, qg = qseg
        _, klen, kg = kseg
        return (self._cut(self._seg, self.segment_ids, qg, qlen), self._cut(self._seg, self.segment_ids, kg, klen),
                self._cut(self._kv, self.key_valid, kg, klen))


def _rows(t, seg):
    off, ln, _ = seg
    return t[:, off:off + ln]


def _from_acc(block, acc, dtype, dst=None):
    """an f32 carry -> the operands' dtype: the cast kernel for bf16 operands, the carry itself (a copy into `dst`) when
    the operands ARE f32 -- the fp32 flavour of the op (the reference's --dtype=fp32) rounds nothing on the way out"""
    if dtype == torch.float32:
        if dst is None:
            return acc
        dst.copy_(acc.reshape(dst.shape))
        return dst
    return block.cast(acc) if dst is None else block.cast(acc, dst)


def _xbuf(comm, block, tag, shape, dtype, like):
    """An exchange buffer: from the communicator's pool when it has one (TorchRingComm), else fresh."""
    pooled = getattr(comm, "pooled", None)
    if pooled is not None:
        return pooled(tag, shape, dtype, like.device)
    return block.empty(tuple(shape), dtype, like)


def _fwd_plan(layout, rank, n, causal):
    """[(t, qi, ki)] in execution order + index of the last entry per q segment."""
    qsegs = layout.segments(rank)
    plan = []
    for t in range(n):
        ksegs = layout.segments((rank - t) % n)
        for qi, qs in enumerate(qsegs):
            for ki, ks in enumerate(ksegs):
                if pair_visible(qs, ks, causal):
                    plan.append((t, qi, ki))
    return plan


def _needed_ksegs(layout, q_rank, k_rank, causal):
    """Indices of k_rank's segments that at least one query segment of q_rank can see."""
    qs_, ks_ = layout.segments(q_rank), layout.segments(k_rank)
    return [ki for ki, ks in enumerate(ks_) if any(pair_visible(qs, ks, causal) for qs in qs_)]


def _is_mesh(comm):
    return comm.size > 1 and getattr(comm, "schedule", "ring") == "mesh"


class _MeshBlocks:
    """Direct fetch of every peer's visible K/V segments (the mesh schedule's transport).

    All transfers are posted at construction, `wave` distances per grouped launch (default:
    all of them in one, i.e. every xGMI link of this GPU busy at once).  get(t) -> {ki: [k_seg,
    v_seg]} of rank (r - t) mod n, waiting for that wave only."""

    def __init__(self, comm, layout, tensors, causal, wave=None, block=None, tag="fwd"):
        n, r = comm.size, comm.rank
        own = layout.segments(r)
        B = tensors[0].shape[0]
        self.local = {ki: [_rows(x, s) for x in tensors] for ki, s in enumerate(own)}
        self.remote, self.handles = {}, {}
        wave = wave or int(os.environ.get("LWM_MESH_WAVE", "0")) or (n - 1)
        ts = list(range(1, n))
        for w0 in range(0, len(ts), wave):
            sends, recvs = [], []
            for t in ts[w0:w0 + wave]:
                dst, src = (r + t) % n, (r - t) % n
                for ki in _needed_ksegs(layout, dst, r, causal):
                    sends += [(dst, _rows(x, own[ki]).contiguous()) for x in tensors]
                segs = layout.segments(src)
                got = {}
                for ki in _needed_ksegs(layout, r, src, causal):
                    shp = lambda x: (B, segs[ki][1]) + tuple(x.shape[2:])
                    got[ki] = [_xbuf(comm, block, ("mesh", tag, t, ki, j), shp(x), x.dtype, x) if block is not None
                               else torch.empty(shp(x), dtype=x.dtype, device=x.device)
                               for j, x in enumerate(tensors)]
                    recvs += [(src, b) for b in got[ki]]
                self.remote[t] = got
            h = comm.exchange_async(sends, recvs)
            for t in ts[w0:w0 + wave]:
                self.handles[t] = h

    def get(self, t):
        if t == 0:
            return self.local
        self.handles[t].wait()
        return self.remote[t]

    def finish(self):
        for h in self.handles.values():
            h.wait()


# ----------------------------------------------------------------- the mesh schedule's gathered form
# What the C driver does since round 5 (lwm_amd/csrc/ring_driver.inc, "gathered form"), restated over
# torch.distributed: the K/V segments a rank can see of its peers -- every segment that begins below the end of its own
# last one -- are received in POSITION order into one buffer and read by the kernels through piecewise position maps
# (ops: q_piece2 / k_piece2), the local shard likewise; per kernel TWO launches per call (the local block while the
# fetch is in flight, then everything that arrived) instead of one per (q segment, k segment) pair.
MAX_PIECES = 8


class _Gathered:
    """The plan of one rank: local pieces, the fetched segments in position order, who sends what to whom."""

    def __init__(self, layout, rank, n):
        self.own = layout.segments(rank)                    # [(off, len, g)], ascending g, local rows in that order
        self.end = {r_: max(g + ln for _, ln, g in layout.segments(r_)) for r_ in range(n)}   # end of a rank's last segment
        self.q_cuts = [(off, g) for off, ln, g in self.own[1:]]
        self.q_start = self.own[0][2]
        # fetched segments: (g, len, src rank, ki), ascending position
        need = sorted((g, ln, s_, ki) for s_ in range(n) if s_ != rank for ki, (_, ln, g) in enumerate(layout.segments(s_))
                      if g < self.end[rank])
        self.rows, self.k_cuts, at, prev_end = {}, [], 0, None
        for g, ln, s_, ki in need:
            if prev_end is not None and g != prev_end:
                self.k_cuts.append((at, g))
            self.rows[(s_, ki)] = (at, ln)
            at, prev_end = at + ln, g + ln
        self.Lg = at
        self.k_start = need[0][0] if need else 0
        self.pos_runs = []                                  # (row, len, g) runs of the gathered buffer, for mask slices
        for g, ln, s_, ki in need:
            if self.pos_runs and self.pos_runs[-1][2] + self.pos_runs[-1][1] == g:
                self.pos_runs[-1] = (self.pos_runs[-1][0], self.pos_runs[-1][1] + ln, self.pos_runs[-1][2])
            else:
                self.pos_runs.append((self.rows[(s_, ki)][0], ln, g))

    @staticmethod
    def applies(block, comm, layout, q, causal):
        if not (_is_mesh(comm) and causal and q.shape[0] == 1 and getattr(block, "piece_align", 0)):
            return False
        if q.dtype == torch.float32 and q.is_cuda:
            return False        # (piecewise position maps are the bf16 kernels'; the f32 flavour runs the pair form)
        if os.environ.get("LWM_RING_FORM") == "pairs":
            return False
        n = comm.size
        segs = [layout.segments(r_) for r_ in range(n)]
        if any(ln % block.piece_align or off % block.piece_align for ss in segs for off, ln, _ in ss):
            return False
        if any(len(ss) > MAX_PIECES or any(a[2] >= b[2] for a, b in zip(ss, ss[1:])) for ss in segs):
            return False
        return all(len(_Gathered(layout, r_, n).k_cuts) < MAX_PIECES for r_ in range(n))


def _gathered_masks(g, segment_ids, key_valid):
    """(seg local, key_valid local, seg gathered, key_valid gathered) in row order"""
    cut = lambda src, runs: None if src is None else torch.cat([src[:, p:p + ln] for _, ln, p in runs], dim=1).contiguous()
    loc = [(off, ln, gpos) for off, ln, gpos in g.own]
    return cut(segment_ids, loc), cut(key_valid, loc), cut(segment_ids, g.pos_runs), cut(key_valid, g.pos_runs)


def _gathered_fetch(comm, block, layout, g, tensors, tag):
    """posts the exchange of the K / V segments in one grouped launch -> (handle, [gathered buffer per tensor])"""
    n, r = comm.size, comm.rank
    bufs = [_xbuf(comm, block, ("gath", tag, j), (1, max(g.Lg, 1)) + tuple(x.shape[2:]), x.dtype, x) for j, x in enumerate(tensors)]
    sends, recvs = [], []
    for t in range(1, n):
        dst, src = (r + t) % n, (r - t) % n
        for off, ln, gpos in g.own:                               # ours that dst can see, ascending
            if gpos < g.end[dst]:
                sends += [(dst, x[:, off:off + ln]) for x in tensors]
        for ki in range(len(layout.segments(src))):               # src's that we can see, ascending
            if (src, ki) in g.rows:
                at, ln = g.rows[(src, ki)]
                recvs += [(src, b[:, at:at + ln]) for b in bufs]
    return comm.exchange_async(sends, recvs), [b[:, :g.Lg] for b in bufs]


def _gathered_forward(block, comm, q, k, v, *, layout, segment_ids, key_valid, scale):
    n, r = comm.size, comm.rank
    B, c, H, D = q.shape
    g = _Gathered(layout, r, n)
    k = k if k.is_contiguous() else k.contiguous()
    v = v if v.is_contiguous() else v.contiguous()
    handle, (kg, vg) = _gathered_fetch(comm, block, layout, g, [k, v], "fwd")
    sq, kvl, sg, kvg = _gathered_masks(g, segment_ids, key_valid)
    out = block.empty((B, c, H, D), q.dtype, q)
    lse = block.empty((B, H, c), torch.float32, q)
    remote = g.Lg > 0
    acc_o = block.empty((B, c, H, D), torch.float32, q) if remote else None
    acc_l = block.empty((B, H, c), torch.float32, q) if remote else None
    qkw = dict(q_start=g.q_start, q_piece2=g.q_cuts or None, causal=True, scale=scale)
    # the local block, while the fetch is in flight
    block.fwd(q, k, v, k_start=g.q_start, k_piece2=g.q_cuts or None, seg_q=sq, seg_k=sq, key_valid=kvl,
              out=None if remote else out, lse=None if remote else lse, out_acc=acc_o, lse_acc=acc_l, carry_in=False,
              final=not remote, **qkw)
    handle.wait()
    if remote:      # everything that arrived, in one launch
        block.fwd(q, kg, vg, k_start=g.k_start, k_piece2=g.k_cuts or None, seg_q=sq, seg_k=sg, key_valid=kvg,
                  out=out, lse=lse, out_acc=acc_o, lse_acc=acc_l, carry_in=True, final=True, **qkw)
    return out, [lse]


def _gathered_backward(block, comm, q, k, v, out, lses, dout, *, layout, segment_ids, key_valid, scale):
    n, r = comm.size, comm.rank
    B, c, H, D = q.shape
    g = _Gathered(layout, r, n)
    k = k if k.is_contiguous() else k.contiguous()
    v = v if v.is_contiguous() else v.contiguous()
    dout = dout if dout.is_contiguous() else dout.contiguous()
    lse = lses[0]
    delta = block.bwd_delta(out, dout, lse)
    handle, (kg, vg) = _gathered_fetch(comm, block, layout, g, [k, v], "bwd")
    sq, kvl, sg, kvg = _gathered_masks(g, segment_ids, key_valid)
    remote = g.Lg > 0
    qkw = dict(q_start=g.q_start, q_piece2=g.q_cuts or None, causal=True, scale=scale)
    loc = dict(k_start=g.q_start, k_piece2=g.q_cuts or None, seg_q=sq, seg_k=sq, key_valid=kvl, **qkw)
    rem = dict(k_start=g.k_start, k_piece2=g.k_cuts or None, seg_q=sq, seg_k=sg, key_valid=kvg, **qkw)
    dq = block.empty((B, c, H, D), q.dtype, q)
    dq_acc = block.empty((B, c, H, D), torch.float32, q) if remote else None
    block.bwd_dq(q, k, v, dout, lse, delta, dq=None if remote else dq, dq_acc=dq_acc, carry_in=False, final=not remote, **loc)
    handle.wait()
    part = None
    if remote:
        block.bwd_dq(q, kg, vg, dout, lse, delta, dq=dq, dq_acc=dq_acc, carry_in=True, final=True, **rem)
        part = [_xbuf(comm, block, ("gpart", w), (1, g.Lg, H, D), torch.float32, q) for w in (0, 1)]
        block.bwd_dkdv(q, kg, vg, dout, lse, delta, dk_acc=part[0], dv_acc=part[1], carry_in=False, final=False, **rem)
    # every owner gets its pieces straight back; what the peers computed for OUR rows arrives in the same exchange
    sends, recvs, got = [], [], {}
    for t in range(1, n):
        owner, giver = (r - t) % n, (r + t) % n
        for ki in range(len(layout.segments(owner))):
            if (owner, ki) in g.rows:
                at, ln = g.rows[(owner, ki)]
                sends += [(owner, part[0][:, at:at + ln]), (owner, part[1][:, at:at + ln])]
        for ki, (off, ln, gpos) in enumerate(g.own):
            if gpos < g.end[giver]:
                got[(t, ki)] = tuple(_xbuf(comm, block, ("gret", t, ki, w), (1, ln, H, D), torch.float32, q) for w in (0, 1))
                recvs += [(giver, got[(t, ki)][0]), (giver, got[(t, ki)][1])]
    ret = comm.exchange_async(sends, recvs)
    dk_loc, dv_loc = (block.empty((B, c, H, D), torch.float32, q) for _ in range(2))
    block.bwd_dkdv(q, k, v, dout, lse, delta, dk_acc=dk_loc, dv_acc=dv_loc, carry_in=False, final=False, **loc)
    ret.wait()
    dk = block.empty((B, c, H, D), q.dtype, q)
    dv = block.empty((B, c, H, D), q.dtype, q)
    for ki, (off, ln, _) in enumerate(g.own):       # fixed order: own partial, then distance 1, 2, ...
        for w, (loc_p, dst) in enumerate(((dk_loc, dk), (dv_loc, dv))):
            srcs = [loc_p[:, off:off + ln]] + [got[(t, ki)][w] for t in range(1, n) if (t, ki) in got]
            block.sum_cast([s_.contiguous() for s_ in srcs], dst[:, off:off + ln])
    return dq, dk, dv


# ----------------------------------------------------------------- forward
def ring_forward(block, comm, q, k, v, *, layout, causal=True, segment_ids=None, key_valid=None,
                 scale=None):
    """Returns (out bf16 (B,c,H,D), [lse per q segment (B,H,len)]) -- or, in the mesh schedule's gathered form, one lse
    piece for the whole shard (an opaque residual between ring_forward and ring_backward of one geometry)."""
    if _Gathered.applies(block, comm, layout, q, causal):
        return _gathered_forward(block, comm, q, k, v, layout=layout, segment_ids=segment_ids, key_valid=key_valid, scale=scale)
    n, r = comm.size, comm.rank
    B, c, H, D = q.shape
    qsegs = layout.segments(r)
    masks = _MaskSlices(segment_ids, key_valid)
    plan = _fwd_plan(layout, r, n, causal)
    first = {}
    last = {}
    for idx, (_, qi, _) in enumerate(plan):
        first.setdefault(qi, idx)
        last[qi] = idx
    out = block.empty((B, c, H, D), q.dtype, q)
    lses = [block.empty((B, H, ln), torch.float32, q) for _, ln, _ in qsegs]
    acc_o = [None] * len(qsegs)
    acc_l = [None] * len(qsegs)
    for qi, (off, ln, _) in enumerate(qsegs):
        if qi not in first:  # no visible key anywhere: defined as out = 0, lse = -inf
            out[:, off:off + ln].zero_()
            lses[qi].fill_(float("-inf"))
        elif first[qi] != last[qi]:
            acc_o[qi] = block.empty((B, ln, H, D), torch.float32, q)
            acc_l[qi] = block.empty((B, H, ln), torch.float32, q)

    # (one rank: nothing travels, and the kernels take strided views as they are -- the harness hands over views of its
    #  fused (B,S,3,H,D) projection buffer; a copy here would be a pass over K and V per layer for nothing)
    k_cur = k if (n == 1 or k.is_contiguous()) else k.contiguous()
    v_cur = v if (n == 1 or v.is_contiguous()) else v.contiguous()
    mesh = _MeshBlocks(comm, layout, [k_cur, v_cur], causal, block=block, tag="fwd") if _is_mesh(comm) else None
    keep = []
    idx = 0
    for t in range(n):
        handle = comm.rotate([k_cur, v_cur]) if (mesh is None and t < n - 1) else None
        ksegs = layout.segments((r - t) % n)
        held = None
        while idx < len(plan) and plan[idx][0] == t:
            _, qi, ki = plan[idx]
            qs, ks = qsegs[qi], ksegs[ki]
            sq, sk, kv = masks(qs, ks)
            fin = idx == last[qi]
            if mesh is not None:
                held = held if held is not None else mesh.get(t)
                k_blk, v_blk = held[ki]
            else:
                k_blk, v_blk = _rows(k_cur, ks), _rows(v_cur, ks)
            block.fwd(_rows(q, qs), k_blk, v_blk, q_start=qs[2], k_start=ks[2],
                      causal=causal, seg_q=sq, seg_k=sk, key_valid=kv, scale=scale,
                      out=_rows(out, qs) if fin else None, lse=lses[qi] if fin else None,
                      out_acc=acc_o[qi], lse_acc=acc_l[qi], carry_in=idx != first[qi], final=fin)
            idx += 1
        if handle is not None:
            keep.append((k_cur, v_cur))
            k_cur, v_cur = handle.wait()
    if mesh is not None:
        mesh.finish()   # our sends must have left before k/v may be reused by the caller
    return out, lses


# ----------------------------------------------------------------- backward
def ring_backward(block, comm, q, k, v, out, lses, dout, *, layout, causal=True,
                  segment_ids=None, key_valid=None, scale=None):
    """Returns (dq, dk, dv) bf16, each (B,c,H,D), for the local shard."""
    if _Gathered.applies(block, comm, layout, q, causal):
        return _gathered_backward(block, comm, q, k, v, out, lses, dout, layout=layout, segment_ids=segment_ids,
                                  key_valid=key_valid, scale=scale)
    n, r = comm.size, comm.rank
    B, c, H, D = q.shape
    qsegs = layout.segments(r)
    masks = _MaskSlices(segment_ids, key_valid)
    if not dout.is_contiguous():
        dout = dout.contiguous()
    deltas = [block.bwd_delta(_rows(out, qs), _rows(dout, qs), lses[qi]) for qi, qs in enumerate(qsegs)]

    if n == 1 and len(qsegs) == 1:
        # single block: write bf16 results straight from the accumulators
        qs = qsegs[0]
        sq, sk, kv = masks(qs, qs)
        kw = dict(q_start=qs[2], k_start=qs[2], causal=causal, seg_q=sq, seg_k=sk, key_valid=kv,
                  scale=scale)
        dk, dv = block.bwd_dkdv(q, k, v, dout, lses[0], deltas[0], final=True, **kw)
        dq = block.bwd_dq(q, k, v, dout, lses[0], deltas[0], final=True, **kw)
        return dq, dk, dv

    if _is_mesh(comm):
        return _mesh_backward(block, comm, q, k, v, lses, dout, deltas, layout=layout, causal=causal,
                              segment_ids=segment_ids, key_valid=key_valid, scale=scale)

    # f32 carries.  None is zeroed: the first contribution to a carry is written, not accumulated
    # (every key segment has one at step 0 -- its own diagonal -- and every query segment likewise).
    dq_acc = [block.empty((B, ln, H, D), torch.float32, q) for _, ln, _ in qsegs]
    dq_seen = [False] * len(qsegs)
    # dk/dv accumulators of the block currently held; they travel with it
    ksegs0 = layout.segments(r)
    dk_acc = [block.empty((B, ln, H, D), torch.float32, q) for _, ln, _ in ksegs0]
    dv_acc = [block.empty((B, ln, H, D), torch.float32, q) for _, ln, _ in ksegs0]
    nks = len(ksegs0)

    k_cur = k if (n == 1 or k.is_contiguous()) else k.contiguous()
    v_cur = v if (n == 1 or v.is_contiguous()) else v.contiguous()
    keep = []
    dkv_handle = None
    for t in range(n):
        kv_handle = comm.rotate([k_cur, v_cur]) if t < n - 1 else None
        ksegs = layout.segments((r - t) % n)
        pairs = [(qi, ki) for qi, qs in enumerate(qsegs) for ki, ks in enumerate(ksegs)
                 if pair_visible(qs, ks, causal)]
        # dq first: it does not need the travelling dk/dv accumulators
        for qi, ki in pairs:
            qs, ks = qsegs[qi], ksegs[ki]
            sq, sk, kv = masks(qs, ks)
            block.bwd_dq(_rows(q, qs), _rows(k_cur, ks), _rows(v_cur, ks), _rows(dout, qs), lses[qi],
                         deltas[qi], q_start=qs[2], k_start=ks[2], causal=causal, seg_q=sq, seg_k=sk,
                         key_valid=kv, scale=scale, dq_acc=dq_acc[qi], carry_in=dq_seen[qi], final=False)
            dq_seen[qi] = True
        if dkv_handle is not None:
            got = dkv_handle.wait()
            dk_acc, dv_acc = got[:nks], got[nks:]
        fresh = set(range(nks)) if t == 0 else set()     # at step 0 the carries hold nothing yet
        for qi, ki in pairs:
            qs, ks = qsegs[qi], ksegs[ki]
            sq, sk, kv = masks(qs, ks)
            block.bwd_dkdv(_rows(q, qs), _rows(k_cur, ks), _rows(v_cur, ks), _rows(dout, qs), lses[qi],
                           deltas[qi], q_start=qs[2], k_start=ks[2], causal=causal, seg_q=sq,
                           seg_k=sk, key_valid=kv, scale=scale, dk_acc=dk_acc[ki], dv_acc=dv_acc[ki],
                           carry_in=ki not in fresh, final=False)
            fresh.discard(ki)
        assert not fresh, "a key segment without a step-0 contribution"
        keep.append((dk_acc, dv_acc))
        dkv_handle = comm.rotate(list(dk_acc) + list(dv_acc))  # n rotations bring them home
        if kv_handle is not None:
            keep.append((k_cur, v_cur))
            k_cur, v_cur = kv_handle.wait()
    got = dkv_handle.wait()
    dk_acc, dv_acc = got[:nks], got[nks:]

    dq = block.empty((B, c, H, D), q.dtype, q)
    dk = block.empty((B, c, H, D), q.dtype, q)
    dv = block.empty((B, c, H, D), q.dtype, q)
    if len(qsegs) == 1:
        _from_acc(block, dq_acc[0], q.dtype, dq)
        _from_acc(block, dk_acc[0], q.dtype, dk)
        _from_acc(block, dv_acc[0], q.dtype, dv)
    else:
        for i, (off, ln, _) in enumerate(qsegs):
            dq[:, off:off + ln].copy_(_from_acc(block, dq_acc[i], q.dtype))
            dk[:, off:off + ln].copy_(_from_acc(block, dk_acc[i], q.dtype))
            dv[:, off:off + ln].copy_(_from_acc(block, dv_acc[i], q.dtype))
    return dq, dk, dv


def _mesh_backward(block, comm, q, k, v, lses, dout, deltas, *, layout, causal, segment_ids, key_valid,
                   scale):
    """Backward under the mesh schedule (module docstring).  Order on the compute stream:
    dq(local) | for each remote block: dq, dk/dv partial -> posted to its owner | dk/dv(local),
    so the first arrivals are covered by local work and the last partial's flight by the
    local dK/dV launch."""
    n, r = comm.size, comm.rank
    B, c, H, D = q.shape
    qsegs = layout.segments(r)
    masks = _MaskSlices(segment_ids, key_valid)
    k_c = k if k.is_contiguous() else k.contiguous()
    v_c = v if v.is_contiguous() else v.contiguous()
    mesh = _MeshBlocks(comm, layout, [k_c, v_c], causal, block=block, tag="bwd")
    dq = block.empty((B, c, H, D), q.dtype, q)
    dk = block.empty((B, c, H, D), q.dtype, q)
    dv = block.empty((B, c, H, D), q.dtype, q)

    def pairs_at(t):
        ksegs = layout.segments((r - t) % n)
        return ksegs, [(qi, ki) for qi, qs in enumerate(qsegs) for ki, ks in enumerate(ksegs)
                       if pair_visible(qs, ks, causal)]

    # dq: chained through an f32 carry per q segment; the last contribution writes bf16
    n_dq = [0] * len(qsegs)
    for t in range(n):
        for qi, _ in pairs_at(t)[1]:
            n_dq[qi] += 1
    acc_shape = lambda ln: (B, ln, H, D)
    dq_acc = [block.empty(acc_shape(ln), torch.float32, q) if n_dq[qi] > 1 else None
              for qi, (_, ln, _) in enumerate(qsegs)]
    done_dq = [0] * len(qsegs)
    for qi, (off, ln, _) in enumerate(qsegs):
        if n_dq[qi] == 0:
            dq[:, off:off + ln].zero_()

    def run_dq(t, held):
        ksegs, pairs = pairs_at(t)
        for qi, ki in pairs:
            qs, ks = qsegs[qi], ksegs[ki]
            sq, sk, kv = masks(qs, ks)
            done_dq[qi] += 1
            fin = done_dq[qi] == n_dq[qi]
            block.bwd_dq(_rows(q, qs), held[ki][0], held[ki][1], _rows(dout, qs), lses[qi], deltas[qi],
                         q_start=qs[2], k_start=ks[2], causal=causal, seg_q=sq, seg_k=sk, key_valid=kv,
                         scale=scale, dq=_rows(dq, qs) if fin else None, dq_acc=dq_acc[qi],
                         carry_in=done_dq[qi] > 1, final=fin)

    def run_dkdv(t, held):
        """-> {ki: (dk_part, dv_part)} f32, for the key segments that got a contribution."""
        ksegs, pairs = pairs_at(t)
        part = {}
        for qi, ki in pairs:
            qs, ks = qsegs[qi], ksegs[ki]
            sq, sk, kv = masks(qs, ks)
            first = ki not in part
            if first:
                # a REMOTE block's partial is sent to its owner: pooled (the send is waited for before
                # this call returns); the local block's partial (t == 0) feeds the final reduction only
                mk = (lambda w: _xbuf(comm, block, ("part", t, ki, w), (B, ks[1], H, D), torch.float32, q)) if t else \
                     (lambda w: block.empty((B, ks[1], H, D), torch.float32, q))
                part[ki] = (mk(0), mk(1))
            block.bwd_dkdv(_rows(q, qs), held[ki][0], held[ki][1], _rows(dout, qs), lses[qi], deltas[qi],
                           q_start=qs[2], k_start=ks[2], causal=causal, seg_q=sq, seg_k=sk, key_valid=kv,
                           scale=scale, dk_acc=part[ki][0], dv_acc=part[ki][1], carry_in=not first,
                           final=False)
        return part

    own = layout.segments(r)
    run_dq(0, mesh.get(0))
    returned = {}     # t -> ({ki: (dk, dv)} received from rank r+t for MY segments, handle)
    for t in range(1, n):
        held = mesh.get(t)
        run_dq(t, held)
        part = run_dkdv(t, held)
        owner, giver = (r - t) % n, (r + t) % n
        sends = []
        for ki in _needed_ksegs(layout, r, owner, causal):   # == sorted(part)
            sends += [(owner, part[ki][0]), (owner, part[ki][1])]
        got, recvs = {}, []
        for ki in _needed_ksegs(layout, giver, r, causal):
            ln = own[ki][1]
            got[ki] = tuple(_xbuf(comm, block, ("ret", t, ki, w), (B, ln, H, D), torch.float32, q) for w in (0, 1))
            recvs += [(giver, got[ki][0]), (giver, got[ki][1])]
        returned[t] = (got, comm.exchange_async(sends, recvs))
    local = run_dkdv(0, mesh.get(0))
    mesh.finish()
    for t in returned:
        returned[t][1].wait()
    for ki, (off, ln, _) in enumerate(own):
        for which, dst in ((0, dk), (1, dv)):
            srcs = ([local[ki][which]] if ki in local else []) + \
                   [returned[t][0][ki][which] for t in sorted(returned) if ki in returned[t][0]]
            if not srcs:
                dst[:, off:off + ln].zero_()
            elif len(own) == 1 or B == 1:
                block.sum_cast(srcs, dst[:, off:off + ln] if len(own) > 1 else dst)
            elif dst.dtype == torch.float32:
                dst[:, off:off + ln].copy_(block.sum_cast(srcs, block.empty(srcs[0].shape, torch.float32, q)))
            else:
                dst[:, off:off + ln].copy_(block.sum_cast(srcs))
    return dq, dk, dv


# ----------------------------------------------------------------- autograd
class _RingAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, segment_ids, key_valid, cfg):
        block, comm, layout, causal, scale = cfg
        out, lses = ring_forward(block, comm, q, k, v, layout=layout, causal=causal,
                                 segment_ids=segment_ids, key_valid=key_valid, scale=scale)
        ctx.save_for_backward(q, k, v, out, *lses)
        ctx.cfg = cfg
        ctx.masks = (segment_ids, key_valid)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, *lses = ctx.saved_tensors
        block, comm, layout, causal, scale = ctx.cfg
        segment_ids, key_valid = ctx.masks
        dq, dk, dv = ring_backward(block, comm, q, k, v, out, lses, dout, layout=layout,
                                   causal=causal, segment_ids=segment_ids, key_valid=key_valid,
                                   scale=scale)
        return dq, dk, dv, None, None, None


_TORCH_COMMS = {}


def _torch_comm(group):
    """one communicator per process group: its exchange-buffer pool (TorchRingComm.pooled) is what makes the buffers
    allocate-once across layers and steps"""
    cm = _TORCH_COMMS.get(id(group))
    if cm is None or cm[0] is not group:
        cm = _TORCH_COMMS[id(group)] = (group, TorchRingComm(group))
    return cm[1]


_C_RINGS = {}
_C_RINGS_PARKED = []        # rings replaced by a larger one: kept alive for graphs that still refer to them
_C_RING_REFUSED = set()     # groups whose first contact with the C driver failed on some rank: they stay on this module's driver


def _c_driver_transport(group, block_ops, q):
    """Which transport the C-ABI ring driver (lwm_ring_attn_fwd / _bwd) would use for this call, or None = this module's
    driver over torch.distributed.  The C driver is the DEFAULT on GPUs when the group's backend is RCCL ("nccl");
    on a gloo group (the CPU tests; a dry run of N processes on one GPU) it runs only when LWM_RING_TRANSPORT=ipc
    selects the library's own IPC transport -- gloo itself cannot carry device memory.  LWM_RING_DRIVER=python opts out,
    =c insists (a refusal is then an error instead of a fallback)."""
    want = os.environ.get("LWM_RING_DRIVER") or "auto"
    if block_ops is not None or want == "python" or group in _C_RING_REFUSED or not q.is_cuda:
        return None
    if q.dtype != torch.bfloat16:
        return None          # (the C driver moves bf16 blocks: the fp32 flavour of the op rides this module's driver)
    try:
        backend, size = dist.get_backend(group), dist.get_world_size(group)
    except Exception:
        return None
    if size < 2:
        return None
    transport = os.environ.get("LWM_RING_TRANSPORT") or ("rccl" if backend == "nccl" else None)
    if transport == "rccl" and backend != "nccl":
        return None          # (an RCCL communicator of our own beside a gloo job: not what anybody asked for)
    return transport if transport in ("rccl", "ipc") else None


def _vote(group, ok):
    """min over the group of a 0/1 flag (on the device for RCCL, on the host for gloo)"""
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    v = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(v, op=dist.ReduceOp.MIN, group=group)
    return int(v.item()) == 1


def _c_ring_for(group, layout_kind, schedule, transport, slot_bytes, slots=8):
    """One C ring object (communicator / mailboxes, side stream, workspace) per (group, layout, schedule, transport),
    reused by every layer; None when the set-up fails on ANY rank: every rank then takes this module's driver.
    Every collective of the set-up is entered by every rank whatever happened on it before (a rank that
    failed ahead of the bootstrap broadcast used to meet the others in a different collective): (1) a capability
    vote -- the library loads, and for RCCL its run-time symbol table resolves -- (2) the bootstrap inside CRing, whose
    own exchanges carry an ok flag, (3) a vote on the finished object.
    The direct schedule's owner-side reduction takes at most 16 sources (lwm_sum_f32_to_bf16): larger groups get the
    neighbour ring."""
    from .ring_c import CRing
    if schedule in ("mesh", "direct") and dist.get_world_size(group) > 16:
        schedule = "ring"
    key = (group, layout_kind, schedule, transport, torch.cuda.current_device())      # (the group object itself: an id() can be recycled)
    ring = _C_RINGS.get(key)
    if ring is not None and transport == "ipc" and (ring.ipc_slot_bytes < slot_bytes or ring._ipc_slots < slots):
        # a larger shard than the mailboxes were cut for: every rank sees the same shapes, so every rank comes through
        # here together.  The old ring is PARKED, not closed: an autograd graph of an earlier forward may still hold it
        # (ctx.cfg) and would run its backward on a destroyed handle.
        _C_RINGS_PARKED.append(ring)
        ring = None
    if ring is None:
        err = None
        try:
            CRing.probe(transport)
        except Exception as e:       # noqa: BLE001 -- whatever it is, the vote decides
            err = e
        if _vote(group, err is None):
            try:
                # (ncclCommInitRank / the IPC handle exchange: under the first-contact watchdog, lwm_amd/ring_c.py)
                from .ring_c import first_contact
                with first_contact(f"set-up of the C ring driver over {transport}"):
                    ring = CRing(group, layout=layout_kind, schedule=schedule, transport=transport,
                                 ipc_slot_bytes=slot_bytes if transport == "ipc" else None, ipc_slots=slots)
            except Exception as e:       # noqa: BLE001
                err = e
            if not _vote(group, err is None):
                err = err or RuntimeError("another rank failed")
        else:
            err = err or RuntimeError("another rank failed")
        if err is not None:
            if ring is not None:
                ring.close()
            _C_RING_REFUSED.add(group)
            _C_RINGS.pop(key, None)
            if os.environ.get("LWM_RING_DRIVER") == "c":
                raise RuntimeError(f"LWM_RING_DRIVER=c, but the C ring driver could not be set up on every rank: {err!r}")
            import warnings
            warnings.warn(f"lwm_amd.ring: the C ring driver could not be set up on every rank of this group ({err!r} here); "
                          "falling back to the torch.distributed driver")
            return None
        _C_RINGS[key] = ring
    return ring


def ring_driver_info(group):
    """What carried the last ring_attention calls of this group: {"driver": "c" | "python", ...} (logging / tests)."""
    for (g, lay, sched, transport, _dev), ring in _C_RINGS.items():
        if g is group:
            return {"driver": "c", "layout": lay, "schedule": sched, "transport": transport, "bytes_sent": ring.bytes_sent}
    return {"driver": "python", "refused_c_driver": group in _C_RING_REFUSED}


def ring_attention(q, k, v, *, group=None, causal=True, segment_ids=None, key_valid=None,
                   scale=None, layout="contiguous", block_ops=None, comm=None):
    """Differentiable ring attention on the local (B, S/n, H, D) shards.

    segment_ids / key_valid are the FULL-length (B, S_global) tensors, replicated
    on every rank, exactly as the reference passes attn_bias / segment_ids
    un-sharded (lwm/llama.py:563-564).

    `layout` names which positions the local rows ARE (the caller sharded the sequence that way): "contiguous" =
    the reference's, "zigzag" = balanced causal work.  This low-level entry keeps the reference's rule as its
    default; the operator surface above it (lwm_amd.ringattention: set_sp_group / sp_shard / ringattention) defaults to
    zigzag for more than one rank.
    """
    if comm is None:
        if group is None and not (dist.is_available() and dist.is_initialized()):
            comm = SingleComm()
        elif group is None and dist.get_world_size() == 1:
            comm = SingleComm()
        elif group is None:
            # never default to WORLD: a data-parallel job would silently ring its replicas
            raise RuntimeError("ring_attention in a multi-process job needs the sequence-parallel group "
                               "(group=... or comm=...; torch.distributed.group.WORLD for a pure ring)")
        else:
            transport = _c_driver_transport(group, block_ops, q)
            if transport is not None:
                # N > 1 on GPUs: the exchange is driven by the C-ABI ring driver (lwm_ring_attn_fwd / _bwd: RCCL -- or the
                # library's IPC transport -- on a side HIP stream, the same layouts, the direct schedule = this module's
                # "mesh").
                from .ring_c import ring_attention_c
                kind = layout.kind if isinstance(layout, SeqLayout) else layout
                sched = os.environ.get("LWM_RING_SCHEDULE") or "mesh"
                # (an ownership table travels with the call: one ring object serves every table; it posts up to
                #  2 x chunks-per-rank messages per pair and group)
                table = layout if kind == "table" else None
                # an ownership table runs in the C driver's GATHERED form only (lwm_ring_attn_*: B = 1, causal, at most
                # LWM_MAX_PIECES = 8 chunks per rank, the direct schedule = at most 16 ranks); any other table call is
                # this module's driver's, whose pair form has none of those limits
                c_ok = table is None or (q.shape[0] == 1 and causal and dist.get_world_size(group) <= 16 and
                                         len(layout.owner) // layout.n <= 8 and sched in ("mesh", "direct"))
                slots = max(8, 4 * q.shape[0], 2 * (len(layout.owner) // layout.n) if table is not None else 0)
                c_ring = _c_ring_for(group, "zigzag" if table is not None else kind, sched, transport, q.numel() * 4,
                                     slots) if c_ok else None
                if c_ring is not None:
                    return ring_attention_c(q, k, v, c_ring, causal=causal, segment_ids=segment_ids,
                                            key_valid=key_valid, scale=scale, layout=table)
            comm = _torch_comm(group)
    block = block_ops if block_ops is not None else HipBlockOps
    lay = layout if isinstance(layout, SeqLayout) else SeqLayout(layout, comm.size,
                                                                 q.shape[1] * comm.size)
    if segment_ids is not None and segment_ids.dtype != torch.int32:
        segment_ids = segment_ids.to(torch.int32)
    if key_valid is not None and key_valid.dtype != torch.uint8:
        key_valid = (key_valid != 0).to(torch.uint8)
    return _RingAttention.apply(q, k, v, segment_ids, key_valid, (block, comm, lay, causal, scale))


# ----------------------------------------------------------------- inference (dense mask)
def _pick_splits(B, Q, H, Sk):
    """Enough workgroups to fill 256 CUs a few times over.  Q == 1 runs the streaming
    decode kernel (one workgroup = all heads of a key range): ~512 workgroups of >= 128
    keys; otherwise the MFMA kernel (one workgroup per head and piece), >= 256 keys each."""
    if Q == 1:
        return max(1, min(-(-512 // B), Sk // 128))
    base = ((Q + 255) // 256) * H * B
    want = max(1, -(-512 // base))
    return max(1, min(want, max(1, Sk // 256)))


def ring_inference(block, comm, q, k, v, mask, *, q_sharded, scale=None):
    """ringattention_inference (lwm/llama.py:571-614).

    q: (B,Q,H,D) replicated over the group when `q_sharded` is False (decode,
    lwm/llama.py:599) else this rank's (B,Q/n,H,D) shard; k, v: this rank's
    contiguous (B,K/n,H,D) shard of the cache; mask: u8 (B,Q_local,K_global).
    MI355X-first: nothing rotates.  Decode computes each rank's normalised partial
    over its own cache shard (split-K inside the launch) and all-gathers the tiny
    (out, lse) partials; the short-prefill case all-gathers K/V (S <= chunk size)."""
    n, r = comm.size, comm.rank
    B, Q, H, D = q.shape
    Kl = k.shape[1]
    if mask.shape[-1] != Kl * n:
        raise ValueError(f"attn_mask covers {mask.shape[-1]} keys but the cache has {Kl * n}")
    if q_sharded and n > 1:
        kf = comm.all_gather(k).transpose(0, 1).reshape(B, n * Kl, H, D).contiguous()
        vf = comm.all_gather(v).transpose(0, 1).reshape(B, n * Kl, H, D).contiguous()
        o_parts, l_parts = block.fwd_splitk(q, kf, vf, k_splits=_pick_splits(B, Q, H, n * Kl),
                                            causal=False, dense_mask=mask, scale=scale)
        return block.combine(o_parts, l_parts, want_bf16=True)[0]
    local_mask = mask[:, :, r * Kl:(r + 1) * Kl]
    o_parts, l_parts = block.fwd_splitk(q, k, v, k_splits=_pick_splits(B, Q, H, Kl), causal=False,
                                        dense_mask=local_mask, scale=scale, k_start=r * Kl)
    if n == 1:
        return block.combine(o_parts, l_parts, want_bf16=True)[0]
    o_loc, l_loc = block.combine(o_parts, l_parts, want_bf16=False)
    return block.combine(comm.all_gather(o_loc), comm.all_gather(l_loc), want_bf16=True)[0]


def cache_update(block, comm, cache_k, cache_v, key, value, cache_index, *, new_sharded):
    """FlaxLLaMAAttention._concatenate_to_cache (lwm/llama.py:440-492) for a cache
    sharded contiguously over the group: rank r holds global rows [r*c, (r+1)*c).

    new_sharded False: key/value (B,P,H,D) are replicated (decode, P = 1:
    only the owning shard writes, :454-467).  True: they are this rank's
    (B,P/n,H,D) shard of a P-row update that lands at global rows
    [cache_index, cache_index+P) (prefill, dynamic_update_slice :485-487) -- rows
    that belong to another rank's shard are sent there.  Returns cache_index + P."""
    n, r = comm.size, comm.rank
    c = cache_k.shape[1]
    p = key.shape[1]
    P = p * n if new_sharded else p
    if cache_index < 0 or cache_index + P > c * n:
        raise ValueError("cache update out of range")
    lo, hi = r * c, (r + 1) * c

    def clip(a0, a1, b0, b1):
        s, e = max(a0, b0), min(a1, b1)
        return (s, e) if e > s else None

    if not new_sharded:
        rng = clip(cache_index, cache_index + P, lo, hi)
        if rng is not None:
            for cache, new in ((cache_k, key), (cache_v, value)):
                block.cache_write(cache, new, dst_row0=rng[0] - lo, src_row0=rng[0] - cache_index,
                                  nrows=rng[1] - rng[0])
        return cache_index + P
    # sharded update: my rows are global [cache_index + r*p, cache_index + (r+1)*p)
    my0 = cache_index + r * p
    sends, recvs, local = [], [], []
    for peer in range(n):
        out_rng = clip(my0, my0 + p, peer * c, (peer + 1) * c)          # my rows that peer owns
        in_rng = clip(cache_index + peer * p, cache_index + (peer + 1) * p, lo, hi)  # peer's rows I own
        if peer == r:
            if out_rng is not None:
                local.append(out_rng)
            continue
        if out_rng is not None:
            sl = slice(out_rng[0] - my0, out_rng[1] - my0)
            sends.append((peer, key[:, sl].contiguous()))
            sends.append((peer, value[:, sl].contiguous()))
        if in_rng is not None:
            shape = (key.shape[0], in_rng[1] - in_rng[0]) + tuple(key.shape[2:])
            bk, bv = block.empty(shape, key.dtype, key), block.empty(shape, key.dtype, key)
            recvs.append((peer, bk, in_rng))
            recvs.append((peer, bv, in_rng))
    comm.exchange(sends, [(peer, t) for peer, t, _ in recvs])
    for g0, g1 in local:
        for cache, new in ((cache_k, key), (cache_v, value)):
            block.cache_write(cache, new, dst_row0=g0 - lo, src_row0=g0 - my0, nrows=g1 - g0)
    for i, (peer, buf, (g0, g1)) in enumerate(recvs):
        block.cache_write(cache_k if i % 2 == 0 else cache_v, buf, dst_row0=g0 - lo, src_row0=0, nrows=g1 - g0)
    return cache_index + P
