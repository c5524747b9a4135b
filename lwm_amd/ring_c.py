"""The sequence ring driven from C (include/lwm_hip.h, lwm_ring_*): RCCL ncclSend / ncclRecv of the K/V
blocks (and, in the backward, of the f32 dK/dV carries) on a side HIP stream, hipEvents handing a double
buffer between the exchange and the attention kernels on the compute stream -- the replacement of the
lax.ppermute under `ringattention` (lwm/llama.py:539-569) for hosts that are not
Python.  This module is the thin torch caller: it creates the ring object, owns the workspace tensor and
wraps the two entry points in an autograd Function.

Ownership: "contiguous" = the reference's (rank r holds positions [r*c, (r+1)*c), lwm/llama.py:560-562) or
"zigzag" (half-chunks r and 2n-1-r: balanced causal work; the caller shards with SeqLayout.global_index).
Schedule: "ring" = the reference's neighbour rotation, or "direct" (alias "mesh"): every rank fetches the K/V
segments its queries can see straight from their owners in one grouped exchange and returns f32 dK/dV partials
to the owners -- lwm_amd/ring.py's mesh schedule, driven from C.  Same results as ring.py with the same layout and
schedule."""
from __future__ import annotations

import ctypes as C

import torch

from . import _capi
from ._lib import lib
from .ops import _t4


import contextlib
import os
import sys
import threading


@contextlib.contextmanager
def first_contact(what, seconds=None):
    """A watchdog around the FIRST time this process does `what` with its peers (the communicator set-up, the first
    forward exchange, the first backward exchange of a ring).  The RCCL path of the C driver has run on one GPU and over
    real processes sharing a GPU, never across GPUs; a peer that never arrives would leave every rank inside
    ncclCommInitRank or a stream wait where no exception can reach it -- the default training path would deadlock
    silently.  If the guarded region is still running after LWM_RING_FIRST_CONTACT_S seconds (default 300,
    0 = no watchdog) the process says what hung and how to take the other driver, and exits with code 75: the launcher
    ends the job instead of a hang."""
    limit = float(os.environ.get("LWM_RING_FIRST_CONTACT_S", "300")) if seconds is None else float(seconds)
    if limit <= 0:
        yield
        return

    def fire():
        print(f"[lwm_amd.ring] rank {os.environ.get('RANK', '?')}: {what} did not finish within {limit:.0f} s -- a peer never "
              "arrived or the transport is stuck.  LWM_RING_DRIVER=python takes the torch.distributed ring driver, "
              "LWM_RING_TRANSPORT=ipc the library's own IPC transport, LWM_RING_FIRST_CONTACT_S the limit.  Exiting (75).",
              file=sys.stderr, flush=True)
        os._exit(75)

    t = threading.Timer(limit, fire)
    t.daemon = True
    t.start()
    try:
        yield
    finally:
        t.cancel()


def reserved_cu_stream(reserve, device=None):
    """A HIP stream whose kernels may use all but `reserve` compute units (hipExtStreamCreateWithCUMask), as a
    torch.cuda.ExternalStream: make it the current stream (torch.cuda.set_stream) and the attention launches -- grids of
    whole-CU workgroups, 128-131 KiB of LDS each -- leave those CUs to whatever else wants them, e.g. RCCL's send/recv
    kernels on the ring driver's side stream.  Whether the exchange is served faster from reserved CUs than from CUs that
    free up between workgroups is a question for a multi-GPU run (bench.py: LWM_RING_RESERVE_CUS prices it in the N > 1
    line); the mask's top `reserve` bits are cleared, which CUs those are is the driver's enumeration."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    n_cu = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    reserve = int(reserve)
    if not 0 < reserve < n_cu:
        raise ValueError(f"reserve must be in 1..{n_cu - 1}")
    words = (n_cu + 31) // 32
    mask = [0xFFFFFFFF] * words
    if n_cu % 32:
        mask[-1] = (1 << (n_cu % 32)) - 1
    for cu in range(n_cu - reserve, n_cu):
        mask[cu // 32] &= ~(1 << (cu % 32))
    hip = C.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    hip.hipExtStreamCreateWithCUMask.restype = C.c_int
    st = C.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), words, (C.c_uint32 * words)(*mask))
    if rc != 0 or not st.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(st.value, device=dev)


class CRing:
    """One ring object per (process group, device).  transport: None / "rccl" = RCCL (a communicator is created
    from an ncclUniqueId broadcast over `group`); "ipc" = the library's CU-free transport (peer mailboxes mapped through
    hipIpcMemHandles, hipMemcpyAsync + stream memory operations; `ipc_slot_bytes` must cover the largest message:
    B * c * H * D * 4 bytes, the f32 dK/dV piece of a whole shard, covers everything); or a _capi.LwmRingTransport
    (tests).  `group` may be a gloo group: only the bootstrap (id / handle exchange) goes through it."""

    @staticmethod
    def probe(transport=None):
        """Can this process create a ring over `transport` at all?  Raises if not: the library does not load, or -- for
        RCCL -- its run-time symbol table does not resolve.  Touches no other rank (callers vote on the outcome before
        they enter the collective bootstrap of __init__)."""
        L = lib()
        if transport in (None, "rccl"):
            _capi.check(L, L.lwm_ring_unique_id((C.c_char * 128)()), "lwm_ring_unique_id")
        elif transport == "ipc":
            if int(L.lwm_ring_ipc_info_bytes()) <= 0:
                raise RuntimeError("lwm_ring_ipc_info_bytes() <= 0")

    @classmethod
    def null(cls, rank, size, *, layout="zigzag", schedule="direct", device=None):
        """Rank `rank` of a `size`-rank ring over a transport that moves NOTHING (receive buffers stay as allocated): the
        launches that rank would make, on this GPU, with every transfer free -- what bench.py prices the exchange with
        and what balance reports time rank by rank.  Results are meaningless."""
        ok = lambda *a: 0
        fns = (_capi.RING_GROUP_FN(ok), _capi.RING_SEND_FN(ok), _capi.RING_SEND_FN(ok), _capi.RING_GROUP_FN(ok))
        ring = cls(rank=rank, size=size, transport=_capi.LwmRingTransport(None, *fns), layout=layout, schedule=schedule,
                   device=device)
        ring._keep = fns
        # what the kernels read as "received" K/V must toggle like real data: on all-zero or stale low-entropy memory the
        # MFMA kernels draw less power and run up to 25 % faster (profiles/r04_backward.md section 3) -- a freshly allocated
        # workspace flattered this model by 2-5 % (profiles/r06_null_transport_fill.txt): it is filled with N(0,1) bf16
        ring._null = True
        return ring

    def __init__(self, group=None, *, rank=None, size=None, transport=None, device=None, layout="contiguous",
                 schedule="ring", ipc_slot_bytes=None, ipc_slots=8):
        import torch.distributed as dist
        L = lib()
        if rank is None:
            if dist.is_available() and dist.is_initialized():
                rank, size = dist.get_rank(group), dist.get_world_size(group)
            else:
                rank, size = 0, 1
        self.rank, self.size = int(rank), int(size)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.side = torch.cuda.Stream(device=self.device) if self.size > 1 else None
        side_ptr = C.c_void_p(self.side.cuda_stream) if self.side is not None else None
        h = C.c_void_p()
        self._transport = transport           # keep the callbacks alive
        self._ipc = None
        self._ipc_slots = int(ipc_slots) if transport == "ipc" else None
        self.ipc_slot_bytes = int(ipc_slot_bytes or 0) if transport == "ipc" else 0
        # Every collective below is entered by EVERY rank, whatever failed on it before: a local failure travels as
        # an ok flag inside the exchange and is raised on all ranks after it (a rank that raised ahead of the exchange
        # would leave the others waiting in it).
        if self.size == 1:
            rc = L.lwm_ring_create(None, 0, 1, None, C.byref(h))
        elif transport == "ipc":
            if not ipc_slot_bytes:
                raise ValueError("CRing(transport='ipc') needs ipc_slot_bytes (>= B*c*H*D*4 covers every message)")
            nb = int(L.lwm_ring_ipc_info_bytes())
            info = (C.c_char * nb)()
            ipc = C.c_void_p()
            err = None
            try:
                with torch.cuda.device(self.device):
                    _capi.check(L, L.lwm_ring_ipc_export(self.rank, self.size, int(ipc_slot_bytes), int(ipc_slots), info, C.byref(ipc)),
                                "lwm_ring_ipc_export")
                self._ipc = ipc
            except Exception as e:      # noqa: BLE001 -- reported to every rank below
                err = repr(e)
            blobs = [None] * self.size
            dist.all_gather_object(blobs, (err, bytes(info)), group=group)
            bad = [(r, e) for r, (e, _) in enumerate(blobs) if e is not None]
            if bad:
                self.close()
                raise RuntimeError(f"lwm_ring_ipc_export failed on rank(s) {bad}")
            allinfo = (C.c_char * (nb * self.size)).from_buffer_copy(b"".join(b for _, b in blobs))
            with torch.cuda.device(self.device):
                _capi.check(L, L.lwm_ring_ipc_connect(ipc, allinfo), "lwm_ring_ipc_connect")
            rc = L.lwm_ring_create_ipc(ipc, side_ptr, C.byref(h))
        elif transport is not None and transport != "rccl":
            rc = L.lwm_ring_create_transport(C.byref(transport), self.rank, self.size, side_ptr, C.byref(h))
        else:
            ident = (C.c_char * 128)()
            err = None
            if self.rank == 0:
                try:
                    _capi.check(L, L.lwm_ring_unique_id(ident), "lwm_ring_unique_id")
                except Exception as e:      # noqa: BLE001 -- broadcast with the id, raised on every rank
                    err = repr(e)
            box = [(err, bytes(ident))]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            if box[0][0] is not None:
                raise RuntimeError(f"lwm_ring_unique_id failed on rank 0 of the group: {box[0][0]}")
            ident = (C.c_char * 128).from_buffer_copy(box[0][1])
            with torch.cuda.device(self.device):          # ncclCommInitRank binds the communicator to the CURRENT device
                rc = L.lwm_ring_create_from_id(ident, self.rank, self.size, side_ptr, C.byref(h))
        _capi.check(L, rc, "lwm_ring_create")
        self._h = h
        self._ws = None
        self._parked = []
        # first-contact watchdog state: a real transport's first forward and first backward are run to completion under
        # first_contact(); test transports (callbacks) and one-rank rings have no peer to wait for
        self._unproven = {"forward", "backward"} if (self.size > 1 and (transport in (None, "rccl", "ipc"))) else set()
        self._kept_bytes = 0
        # layout: "contiguous" | "zigzag" | a lwm_amd.ring.SeqLayout (kind "table": the ownership table travels with every
        # call); forward / backward take another one per call (a packed batch has its own balanced ownership)
        self.layout, self._owner = self._layout_code(layout)
        self.schedule = _capi.RING_SCHEDULE[schedule]

    @staticmethod
    def _layout_code(layout):
        """-> (LWM_RING_LAYOUT_* code, ctypes ownership table or None)"""
        if isinstance(layout, str):
            return _capi.RING_LAYOUT[layout], None
        if layout.kind == "table":
            return _capi.RING_LAYOUT["table"], (C.c_int32 * len(layout.owner))(*layout.owner)
        return _capi.RING_LAYOUT[layout.kind], None

    def close(self):
        if getattr(self, "_h", None):
            lib().lwm_ring_destroy(self._h)
            self._h = None
        if getattr(self, "_ipc", None):
            lib().lwm_ring_ipc_destroy(self._ipc)
            self._ipc = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def bytes_sent(self):
        return int(lib().lwm_ring_bytes_sent(self._h))

    @property
    def last_form(self):
        """0 = the last call launched per (q segment, k segment) pair, 1 = gathered form (lwm_ring_last_form)"""
        return int(lib().lwm_ring_last_form(self._h))

    def set_fetch_groups(self, groups):
        """direct schedule: grouped exchanges of the K/V fetch (lwm_ring_set_fetch_groups)"""
        _capi.check(lib(), lib().lwm_ring_set_fetch_groups(self._h, int(groups)), "lwm_ring_set_fetch_groups")

    def fetch_timeline(self):
        """(kv_ms, kernels_ms) of the last call -- rings created under LWM_RING_TIMING=1 (lwm_ring_fetch_timeline)"""
        kv, ke = (C.c_float * self.size)(), (C.c_float * self.size)()
        _capi.check(lib(), lib().lwm_ring_fetch_timeline(self._h, kv, ke, self.size), "lwm_ring_fetch_timeline")
        return list(kv), list(ke)

    def _workspace(self, B, c, H, D, backward):
        # One buffer for the forward AND the backward of a shape (the larger of the two), so that the backward never
        # replaces the buffer a forward still in flight works in: the side stream's transfers into a workspace are not
        # known to torch's caching allocator, which would hand a freed block to the next allocation (another rank's
        # tensor, a staging buffer) while they are pending.  A buffer that has to grow is parked, not freed.
        L = lib()
        need = max(int(L.lwm_ring_workspace_bytes(B, c, H, D, bw, self.size, self.schedule)) for bw in (0, 1))
        if self._ws is None or self._ws.numel() < need + 256:
            if self._ws is not None:
                self._parked.append(self._ws)
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            if getattr(self, "_null", False):        # (CRing.null: nothing will ever be received into it)
                n = self._ws.numel() // 2 * 2
                self._ws[:n].view(torch.bfloat16).normal_()
        off = (-self._ws.data_ptr()) % 256
        return self._ws.data_ptr() + off

    def _args(self, q, k, v, out, lse, segment_ids, key_valid, scale, causal, backward, layout=None):
        if (self._ipc_slots is not None and self.schedule == _capi.RING_SCHEDULE["direct"] and
                (1 + (self.layout == _capi.RING_LAYOUT["zigzag"])) * 2 * q.shape[0] > self._ipc_slots):
            # the direct schedule posts (segments x {K, V} x B) messages per pair in one group: more than `slots` of
            # them would make a send wait for an acknowledgement of its own group.  The transport refuses that itself
            # (ring_ipc.inc, ipc_send); this is the same rule with the number to pass instead.
            need = (1 + (self.layout == _capi.RING_LAYOUT["zigzag"])) * 2 * q.shape[0]
            raise ValueError(f"CRing(transport='ipc', ipc_slots={self._ipc_slots}): the direct schedule at batch {q.shape[0]} "
                             f"needs ipc_slots >= {need}")
        B, c, H, D = q.shape
        a = _capi.LwmRingArgs()
        a.q, a.k, a.v, a.out = _t4(q, "q"), _t4(k, "k"), _t4(v, "v"), _t4(out, "out")
        a.lse = lse.data_ptr()
        a.B, a.c, a.H, a.D = B, c, H, D
        a.scale = float(scale) if scale is not None else 1.0 / D ** 0.5
        a.causal = int(bool(causal))
        Sg = c * self.size
        if segment_ids is not None:
            if segment_ids.dtype != torch.int32 or tuple(segment_ids.shape) != (B, Sg) or not segment_ids.is_contiguous():
                raise ValueError(f"segment_ids: expected contiguous int32 {(B, Sg)} (replicated, full length)")
            a.segment_ids = segment_ids.data_ptr()
        if key_valid is not None:
            if key_valid.dtype != torch.uint8 or tuple(key_valid.shape) != (B, Sg) or not key_valid.is_contiguous():
                raise ValueError(f"key_valid: expected contiguous uint8 {(B, Sg)} (replicated, full length)")
            a.key_valid = key_valid.data_ptr()
        a.workspace = self._workspace(B, c, H, D, backward)
        code, owner = (self.layout, self._owner) if layout is None else self._layout_code(layout)
        a.layout, a.schedule = code, self.schedule
        if owner is not None:
            a.chunk_owner, a.n_chunks = C.cast(owner, C.c_void_p), len(owner)
            a._keep_owner = owner
        return a

    def kv_keep_buffer(self, q):
        """A buffer for the gathered K/V of one layer (lwm_ring_kv_keep_bytes) when keeping it between the forward and the
        backward is within the budget -- LWM_RING_KEEP_KV_MB per layer, default 1024 (0 = never): the backward then fetches
        no K/V again, a quarter of the layer's xGMI bytes.  A function of the geometry only, so every rank decides alike."""
        if self.size < 2 or self.schedule != _capi.RING_SCHEDULE["direct"]:
            return None          # (the neighbour ring never gathers: nothing to keep)
        B, c, H, D = q.shape
        need = int(lib().lwm_ring_kv_keep_bytes(B, c, H, D, self.size))
        cap = float(os.environ.get("LWM_RING_KEEP_KV_MB", "1024")) * (1 << 20)
        # ... and over the whole model: the buffers of all layers are alive between the forward and the backward of a step
        # (LWM_RING_KEEP_KV_TOTAL_MB, default 32768; 32 layers at S = 32768 over 8 ranks hold 15 GB).  The count goes up
        # in the forward and down in the backward of each layer -- the same sequence on every rank.
        total = float(os.environ.get("LWM_RING_KEEP_KV_TOTAL_MB", "32768")) * (1 << 20)
        if need <= 0 or need > cap or self._kept_bytes + need > total:
            return None
        buf = torch.empty(need + 256, dtype=torch.uint8, device=q.device)
        off = (-buf.data_ptr()) % 256
        return buf[off:off + need]

    def forward(self, q, k, v, *, causal=True, segment_ids=None, key_valid=None, scale=None, layout=None, kv_keep=None):
        """kv_keep: a buffer from kv_keep_buffer() -- the gathered form gathers the fetched K/V into it (hand it to
        backward(kv_keep=...) of the same layer, on EVERY rank or on none)"""
        B, c, H, D = q.shape
        k, v = k.contiguous(), v.contiguous()
        out = torch.empty((B, c, H, D), dtype=torch.bfloat16, device=q.device)
        lse = torch.empty((B, H, c), dtype=torch.float32, device=q.device)
        a = self._args(q, k, v, out, lse, segment_ids, key_valid, scale, causal, False, layout)
        if kv_keep is not None:
            a.kv_keep = kv_keep.data_ptr()
        L = lib()
        with self._guard("forward"):
            _capi.check(L, L.lwm_ring_attn_fwd(self._h, C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                        "lwm_ring_attn_fwd")
        return out, lse

    @contextlib.contextmanager
    def _guard(self, which):
        """the first forward / backward of a ring over a real transport: run to completion under the watchdog"""
        if which not in self._unproven:
            yield
            return
        with first_contact(f"the first ring-attention {which} exchange ({self.size} ranks)"):
            yield
            torch.cuda.current_stream().synchronize()
            if self.side is not None:
                self.side.synchronize()
        self._unproven.discard(which)

    def backward(self, q, k, v, out, lse, dout, *, causal=True, segment_ids=None, key_valid=None, scale=None, layout=None,
                 kv_keep=None):
        B, c, H, D = q.shape
        k, v, dout = k.contiguous(), v.contiguous(), dout.contiguous()
        dq, dk, dv = (torch.empty((B, c, H, D), dtype=torch.bfloat16, device=q.device) for _ in range(3))
        a = self._args(q, k, v, out, lse, segment_ids, key_valid, scale, causal, True, layout)
        if kv_keep is not None:
            a.kv_keep, a.kv_kept = kv_keep.data_ptr(), 1
        a.dout, a.dq, a.dk, a.dv = _t4(dout, "dout"), _t4(dq, "dq"), _t4(dk, "dk"), _t4(dv, "dv")
        L = lib()
        with self._guard("backward"):
            _capi.check(L, L.lwm_ring_attn_bwd(self._h, C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                        "lwm_ring_attn_bwd")
        return dq, dk, dv


class _RingAttentionC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, ring, causal, segment_ids, key_valid, scale, layout):
        # (None beyond the budget -- the backward fetches again -- and when nothing asks for a gradient: a no_grad / eval
        #  forward has no backward to keep the K/V for.  ctx.needs_input_grad is the same on every rank of a model.)
        keep = ring.kv_keep_buffer(q) if any(ctx.needs_input_grad[:3]) else None
        out, lse = ring.forward(q, k, v, causal=causal, segment_ids=segment_ids, key_valid=key_valid, scale=scale, layout=layout,
                                kv_keep=keep)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.cfg = (ring, causal, segment_ids, key_valid, scale, layout)
        # the gathered K/V of this layer, kept for its backward -- only when the call took the gathered form (else the
        # buffer was not written)
        ctx.kv_keep = keep if (keep is not None and ring.last_form == 1) else None
        if ctx.kv_keep is not None:
            ring._kept_bytes += ctx.kv_keep.numel()
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        ring, causal, segment_ids, key_valid, scale, layout = ctx.cfg
        dq, dk, dv = ring.backward(q, k, v, out, lse, dout, causal=causal, segment_ids=segment_ids,
                                   key_valid=key_valid, scale=scale, layout=layout, kv_keep=ctx.kv_keep)
        if ctx.kv_keep is not None:
            ring._kept_bytes = max(0, ring._kept_bytes - ctx.kv_keep.numel())
        ctx.kv_keep = None
        return dq, dk, dv, None, None, None, None, None, None


def ring_attention_c(q, k, v, ring: CRing, *, causal=True, segment_ids=None, key_valid=None, scale=None, layout=None):
    """Differentiable ring attention on the local (B, S/n, H, D) shards through lwm_ring_attn_fwd / _bwd."""
    if segment_ids is not None and segment_ids.dtype != torch.int32:
        segment_ids = segment_ids.to(torch.int32)
    if key_valid is not None and key_valid.dtype != torch.uint8:
        key_valid = (key_valid != 0).to(torch.uint8)
    return _RingAttentionC.apply(q, k, v, ring, causal, segment_ids, key_valid, scale, layout)
