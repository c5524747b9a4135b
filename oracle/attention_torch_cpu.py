"""CPU port of the reference's blockwise attention (forward + backward) on
PyTorch-CPU fp32 -- TEST/BASELINE INFRASTRUCTURE, NOT PRODUCT.

This is the `cpu_baseline` leg of bench.py ("port"): the same algorithm as
oracle/attention_ref.blockwise_ring_attention(+_bwd) (scan over q chunks x k
chunks, online softmax carry, skip of chunk pairs above the diagonal; call site
lwm/llama.py:539-569, chunk sizes 1024/1024 per lwm/llama.py:155-156), written
with torch.matmul so that it uses every host core through the BLAS the wheel
ships.  It stands in for "the reference's JAX/XLA CPU path", which cannot be
installed here (PARITY UNPINNED, see oracle/attention_ref.py).

tests/test_oracle.py checks it against the float64 dense oracle.
"""
import math

import torch


def blockwise_fwd_bwd(q, k, v, dout, *, q_chunk=1024, k_chunk=1024, causal=True):
    """q,k,v,dout: (B,S,H,D) float32 CPU tensors.  Returns (out, dq, dk, dv)."""
    B, S, H, D = q.shape
    scale = 1.0 / math.sqrt(D)
    qh, kh, vh, doh = (t.permute(0, 2, 1, 3).contiguous() for t in (q, k, v, dout))  # B,H,S,D
    qc, kc = min(q_chunk, S), min(k_chunk, S)
    out = torch.zeros_like(qh)
    lse = torch.empty(B, H, S)
    for q0 in range(0, S, qc):
        qs = qh[:, :, q0:q0 + qc]
        num = torch.zeros(B, H, qs.shape[2], D)
        den = torch.zeros(B, H, qs.shape[2])
        mx = torch.full((B, H, qs.shape[2]), float("-inf"))
        for k0 in range(0, S, kc):
            if causal and k0 > q0 + qs.shape[2] - 1:
                continue
            ks, vs = kh[:, :, k0:k0 + kc], vh[:, :, k0:k0 + kc]
            s = torch.matmul(qs, ks.transpose(-1, -2)) * scale
            if causal and k0 + ks.shape[2] - 1 > q0:
                qi = torch.arange(q0, q0 + qs.shape[2])[:, None]
                ki = torch.arange(k0, k0 + ks.shape[2])[None, :]
                s = s.masked_fill(ki > qi, float("-inf"))
            m_new = torch.maximum(mx, s.amax(dim=-1))
            p = torch.exp(s - m_new[..., None])
            corr = torch.exp(mx - m_new)
            num = num * corr[..., None] + torch.matmul(p, vs)
            den = den * corr + p.sum(dim=-1)
            mx = m_new
        out[:, :, q0:q0 + qc] = num / den[..., None]
        lse[:, :, q0:q0 + qc] = mx + torch.log(den)
    delta = (doh * out).sum(dim=-1)
    dq, dk, dv = torch.zeros_like(qh), torch.zeros_like(kh), torch.zeros_like(vh)
    for q0 in range(0, S, qc):
        qs, dos = qh[:, :, q0:q0 + qc], doh[:, :, q0:q0 + qc]
        for k0 in range(0, S, kc):
            if causal and k0 > q0 + qs.shape[2] - 1:
                continue
            ks, vs = kh[:, :, k0:k0 + kc], vh[:, :, k0:k0 + kc]
            s = torch.matmul(qs, ks.transpose(-1, -2)) * scale
            p = torch.exp(s - lse[:, :, q0:q0 + qc, None])
            if causal and k0 + ks.shape[2] - 1 > q0:
                qi = torch.arange(q0, q0 + qs.shape[2])[:, None]
                ki = torch.arange(k0, k0 + ks.shape[2])[None, :]
                p = p.masked_fill(ki > qi, 0.0)
            dv[:, :, k0:k0 + kc] += torch.matmul(p.transpose(-1, -2), dos)
            dp = torch.matmul(dos, vs.transpose(-1, -2))
            ds = p * (dp - delta[:, :, q0:q0 + qc, None]) * scale
            dq[:, :, q0:q0 + qc] += torch.matmul(ds, ks)
            dk[:, :, k0:k0 + kc] += torch.matmul(ds.transpose(-1, -2), qs)
    back = lambda t: t.permute(0, 2, 1, 3).contiguous()
    return back(out), back(dq), back(dk), back(dv)


# ---- every host thread at once: heads are independent, so a many-core host runs them side by side (bench.py cpu_baseline)
def _heads_worker(idx, S, heads, threads, cpus, barrier, out_q):
    import os
    import time
    if cpus:
        # this process's BLAS / OpenMP threads on ITS OWN logical CPUs: without the mask every process's pool lands on the
        # same few cores (measured on a 128-thread host: 16 processes x 8 threads took 20x the time of one)
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            pass
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(100 + idx)
    mk = lambda s: [torch.randn(1, s, 1, 128, generator=g) for _ in range(4)]
    blockwise_fwd_bwd(*mk(1024))                      # warm this process's BLAS threads
    data = mk(S)
    barrier.wait()
    t0 = time.time()
    for _ in range(heads):
        blockwise_fwd_bwd(*data)
    out_q.put((idx, t0, time.time()))


def heads_in_parallel(S, workers, heads_per_worker, threads_per_worker=8, timeout=600):
    """`workers` processes x `threads_per_worker` BLAS threads -- each pinned to its own slice of the logical CPUs this
    process may use -- each running `heads_per_worker` passes of blockwise_fwd_bwd (1 head, S tokens) after a common
    barrier.  -> (wall seconds from the first start to the last end, heads done).  A worker that dies ends the call
    with an error instead of a wait for `timeout`."""
    import multiprocessing as mp
    import os
    import queue
    import time
    ctx = mp.get_context("spawn")
    barrier, out_q = ctx.Barrier(workers), ctx.Queue()
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = []
    masks = [cpus[i * threads_per_worker:(i + 1) * threads_per_worker] if len(cpus) >= workers * threads_per_worker else []
             for i in range(workers)]
    keep = {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "GOMP_CPU_AFFINITY", "KMP_AFFINITY")}
    os.environ["OMP_PROC_BIND"] = "false"            # (inherited by the workers: the mask above decides, not the runtime)
    os.environ.pop("GOMP_CPU_AFFINITY", None)
    os.environ.pop("KMP_AFFINITY", None)
    procs = [ctx.Process(target=_heads_worker, args=(i, S, heads_per_worker, threads_per_worker, masks[i], barrier, out_q))
             for i in range(workers)]
    try:
        for p in procs:
            p.start()
        spans, t_end = [], time.time() + timeout
        while len(spans) < workers:
            try:
                spans.append(out_q.get(timeout=2.0))
            except queue.Empty:
                dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
                if dead:
                    raise RuntimeError(f"a CPU-baseline worker ended with exit code {dead[0]}")
                if time.time() > t_end:
                    raise TimeoutError(f"CPU-baseline workers did not finish within {timeout} s")
    finally:
        for k, v in keep.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        for p in procs:
            p.join(timeout=5)
            if p.is_alive():
                p.kill()
    return max(e for _, _, e in spans) - min(s for _, s, _ in spans), workers * heads_per_worker
