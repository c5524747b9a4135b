"""The C-ABI ring driver (include/lwm_hip.h lwm_ring_*; lwm_amd/csrc/ring_driver.inc).
CPU: symbols, struct mirror, workspace sizing, argument validation (everything that returns before HIP is
touched).  GPU: the n = 1 path against the block ops, and the full schedule -- double buffer, events,
carries that travel with the block, masks on global positions -- with n threads as ranks exchanging through
an in-process mailbox TRANSPORT (lwm_ring_create_transport), checked against ring = 1 and the fp64 oracle.
The RCCL transport itself needs several GPUs; `bench.py --gpus N --c-ring` exercises it."""
import ctypes as C
import os
import queue
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from lwm_amd import _capi
    return _capi.bind(C.CDLL(os.path.join(ROOT, "lwm_amd", "liblwm_hip.so"))), _capi


def test_ring_symbols_sizes_and_validation():
    L, cap = _lib()
    assert L.lwm_sizeof(2) == C.sizeof(cap.LwmRingArgs)
    # workspace: K/V double buffer (bf16) + f32 out carry + lse + mask slices; backward adds delta, dq_acc and
    # the double-buffered f32 dK/dV carries.  All pieces 256-byte aligned.
    B, c, H, D = 1, 16384, 32, 128
    blk16, blk32 = B * c * H * D * 2, B * c * H * D * 4
    fwd = L.lwm_ring_workspace_bytes(B, c, H, D, 0)
    bwd = L.lwm_ring_workspace_bytes(B, c, H, D, 1)
    assert fwd >= 4 * blk16 + blk32 and fwd < 4 * blk16 + blk32 + (1 << 22)
    assert bwd - fwd >= 5 * blk32 and bwd % 256 == 0
    assert L.lwm_ring_workspace_bytes(0, c, H, D, 0) == 0
    h = C.c_void_p()
    for rank, n in ((2, 2), (-1, 2), (0, 0), (0, 65)):
        assert L.lwm_ring_create(None, rank, n, None, C.byref(h)) == cap.LWM_EINVAL and not h.value
    assert L.lwm_ring_create(None, 0, 2, None, C.byref(h)) == cap.LWM_EINVAL          # n > 1 without a communicator
    assert b"nccl_comm" in L.lwm_last_error()
    t = cap.LwmRingTransport()                                                        # no send/recv given
    assert L.lwm_ring_create_transport(C.byref(t), 0, 2, None, C.byref(h)) == cap.LWM_EINVAL
    assert L.lwm_ring_create_transport(None, 0, 2, None, None) == cap.LWM_EINVAL
    a = cap.LwmRingArgs()
    assert L.lwm_ring_attn_fwd(None, C.byref(a), None) == cap.LWM_EINVAL
    assert L.lwm_ring_attn_bwd(None, None, None) == cap.LWM_EINVAL
    assert L.lwm_ring_destroy(None) == cap.LWM_OK and L.lwm_ring_bytes_sent(None) == 0
    assert L.lwm_ring_unique_id(None) == cap.LWM_EINVAL


# ---------------------------------------------------------------- GPU
class _Mailbox:
    """send/recv as the C driver wants them (enqueue on the given stream), implemented with one FIFO per
    ordered pair of ranks: a send copies the bytes into a staging tensor on the sender's side stream and posts
    (tensor, event); the matching recv waits for the post, makes its stream wait for the event and copies."""

    def __init__(self, n):
        import torch
        self.n = n
        self.links = {(a, b): queue.Queue() for a in range(n) for b in range(n)}
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        self.keep = []
        self.torch = torch

    def transport(self, rank):
        from lwm_amd import _capi
        torch = self.torch

        def send(ctx, buf, nbytes, peer, stream):
            try:
                tmp = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
                s = torch.cuda.ExternalStream(stream)
                with torch.cuda.stream(s):
                    assert self.hip.hipMemcpyAsync(tmp.data_ptr(), buf, nbytes, 3, stream) == 0
                    ev = torch.cuda.Event()
                    ev.record(s)
                self.links[(rank, peer)].put((tmp, ev))
                return 0
            except Exception as e:  # pragma: no cover
                print("send failed", e)
                return 1

        def recv(ctx, buf, nbytes, peer, stream):
            try:
                tmp, ev = self.links[(peer, rank)].get(timeout=120)
                assert tmp.numel() == nbytes
                s = torch.cuda.ExternalStream(stream)
                s.wait_event(ev)
                assert self.hip.hipMemcpyAsync(buf, tmp.data_ptr(), nbytes, 3, stream) == 0
                self.keep.append(tmp)
                return 0
            except Exception as e:  # pragma: no cover
                print("recv failed", e)
                return 1

        t = _capi.LwmRingTransport(None, _capi.RING_GROUP_FN(lambda ctx: 0), _capi.RING_SEND_FN(send),
                                   _capi.RING_SEND_FN(recv), _capi.RING_GROUP_FN(lambda ctx: 0))
        return t


def _run_c_ring(n, S, H, causal, packed, padded, B=1):
    import torch
    from lwm_amd.ring_c import CRing
    g = torch.Generator().manual_seed(7)
    mk = lambda: torch.randn(B, S, H, 128, generator=g).to(torch.bfloat16).cuda()
    q, k, v, do = mk(), mk(), mk(), mk()
    seg = kv = None
    if packed:
        seg = torch.zeros(B, S, dtype=torch.int32)
        seg[:, S // 3:] = 1
        seg[:, (5 * S) // 8:] = 2
        seg = seg.cuda()
    if padded:
        kv = torch.ones(B, S, dtype=torch.uint8)
        kv[:, 5:40] = 0
        kv = kv.cuda()
    c = S // n
    box = _Mailbox(n)
    res, errs = [None] * n, []

    def worker(r):
        try:
            ring = CRing(rank=r, size=n, transport=box.transport(r) if n > 1 else None)
            sl = slice(r * c, (r + 1) * c)
            ql, kl, vl, dol = (t[:, sl].contiguous() for t in (q, k, v, do))
            out, lse = ring.forward(ql, kl, vl, causal=causal, segment_ids=seg, key_valid=kv)
            dq, dk, dv = ring.backward(ql, kl, vl, out, lse, dol, causal=causal, segment_ids=seg, key_valid=kv)
            torch.cuda.synchronize()
            res[r] = (out, dq, dk, dv, ring.bytes_sent)
            ring.close()
        except Exception as e:  # pragma: no cover
            import traceback
            traceback.print_exc()
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not errs, errs
    got = [torch.cat([res[r][i] for r in range(n)], 1) for i in range(4)]
    return got, (q, k, v, do, seg, kv), [r[4] for r in res]


@pytest.mark.gpu
@pytest.mark.parametrize("n,causal,packed,padded", [(1, True, True, True), (2, True, False, False), (4, True, True, True),
                                                    (4, False, False, True), (8, True, True, False)])
def test_c_ring_schedule_vs_oracle_and_python_driver(n, causal, packed, padded):
    import torch
    from oracle import attention_ref as R
    from lwm_amd.ring import ring_attention
    from tests._parity import check
    S, H = 512 * max(n // 2, 1) if n > 1 else 640, 2
    got, (q, k, v, do, seg, kv), sent = _run_c_ring(n, S, H, causal, packed, padded)
    f = lambda t: t.float().cpu().numpy()
    sg = None if seg is None else seg.cpu().numpy()
    kvn = None if kv is None else kv.cpu().numpy()
    ro, _ = R.dense_attention(f(q), f(k), f(v), causal=causal, seg_q=sg, seg_k=sg, key_valid=kvn)
    rq, rk, rv = R.dense_attention_bwd(f(q), f(k), f(v), f(do), causal=causal, seg_q=sg, seg_k=sg, key_valid=kvn)
    from tests._parity import dq_row_slack
    slack = dq_row_slack(f(do), ro, f(k))
    for name, a, b in zip(("out", "dq", "dk", "dv"), got, (ro, rq, rk, rv)):
        check(f"{name} c-ring n={n}", f(a), b, row_slack=slack if name == "dq" else None)
    # the single-device Python driver on the same data (same kernels, other association order at most)
    q1, k1, v1 = (t.clone().requires_grad_(True) for t in (q, k, v))
    o1 = ring_attention(q1, k1, v1, causal=causal, segment_ids=seg, key_valid=kv)
    o1.backward(do)
    for name, a, b in zip(("out", "dq", "dk", "dv"), got, (o1.detach(), q1.grad, k1.grad, v1.grad)):
        assert ((a.float() - b.float()).abs().max() / b.float().abs().max()).item() <= 8e-3, name
    if n > 1:
        # forward: n-1 rotations of K and V; backward: n-1 of K and V + n of the two f32 carries
        c = S // n
        blk = c * H * 128
        assert all(s == (n - 1) * 2 * blk * 2 * 2 + n * 2 * blk * 4 for s in sent), sent


_RCCL_FIRST_CONTACT = r'''
import ctypes as C, sys, torch
from lwm_amd import _capi
from lwm_amd._lib import lib
L = lib()
torch.cuda.init()
side = torch.cuda.Stream()
ident = (C.c_char * 128)()
_capi.check(L, L.lwm_ring_unique_id(ident), "lwm_ring_unique_id")
assert any(bytes(ident)), "ncclGetUniqueId left the id empty"
h = C.c_void_p()
_capi.check(L, L.lwm_ring_create_from_id(ident, 0, 1, C.c_void_p(side.cuda_stream), C.byref(h)), "lwm_ring_create_from_id")
nbytes = int(sys.argv[1])
g = torch.Generator(device="cuda").manual_seed(3)
src = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device="cuda", generator=g)
dst = torch.zeros_like(src)
cs = torch.cuda.current_stream()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    dst.zero_()
    ev0.record(cs)
    _capi.check(L, L.lwm_ring_selftest(h, src.data_ptr(), dst.data_ptr(), nbytes, C.c_void_p(cs.cuda_stream)), "lwm_ring_selftest")
    ev1.record(cs)
    torch.cuda.synchronize()
    assert torch.equal(src, dst), "bytes received through RCCL differ from the bytes sent"
assert L.lwm_ring_bytes_sent(h) == 3 * nbytes
print("RCCL_SELF_OK %d bytes, last exchange %.3f ms = %.1f GB/s" % (nbytes, ev0.elapsed_time(ev1), nbytes / ev0.elapsed_time(ev1) / 1e6))
_capi.check(L, L.lwm_ring_destroy(h), "lwm_ring_destroy")
'''


@pytest.mark.gpu
def test_rccl_first_contact_on_one_gpu():
    """The RCCL transport of the C ring driver, executed for real on the one GPU a test box has: the library's
    run-time symbol table (dlsym / dlopen librccl), lwm_ring_unique_id, lwm_ring_create_from_id with n = 1 (the
    128-byte ncclUniqueId passed BY VALUE through a dlsym'd pointer), and a grouped ncclSend + ncclRecv of
    256 MiB to our own rank on the ring's side stream, handed over by the driver's events (lwm_ring_selftest).
    Runs in its own process under a timeout: a communicator that hangs must fail this test, not the session."""
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
    p = subprocess.run([sys.executable, "-c", _RCCL_FIRST_CONTACT, str(256 << 20)], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=240)
    out = p.stdout + p.stderr
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_first_contact.txt"), "w") as f:
        f.write(out)
    assert p.returncode == 0 and "RCCL_SELF_OK" in p.stdout, out[-3000:]
