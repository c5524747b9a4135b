"""One causal attention layer on ONE GPU as several launches per kernel instead of one (VERDICT r05 follow-up, round 6):
the forward and dQ over bands of QUERY rows (each against keys [0, band end)), dK/dV over bands of KEY rows (each against the
whole query tensor: the walk starts at the diagonal).  Within a band every workgroup walks about the same number of steps.
Timing probe (HIP events); results are checked against the single launches bit for bit where the arithmetic is the same.
    gpurun -- 'python scripts/gpu_band_probe.py 32768 131072'"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from lwm_amd import ops  # noqa: E402

H, D = 32, 128


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    for S in [int(a) for a in sys.argv[1:]] or [32768]:
        g = torch.Generator(device="cuda").manual_seed(7)
        mk = lambda: torch.randn(1, S, H, D, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
        q, k, v, do = mk(), mk(), mk(), mk()
        reps = 3 if S <= 32768 else 2
        out, lse = ops.attn_fwd_block(q, k, v, causal=True)
        delta = ops.attn_bwd_delta(out, do, lse)
        dq1 = ops.attn_bwd_dq_block(q, k, v, do, lse, delta, causal=True)
        dk1, dv1 = ops.attn_bwd_dkdv_block(q, k, v, do, lse, delta, causal=True)
        t_f = timed(lambda: ops.attn_fwd_block(q, k, v, causal=True, out=out, lse=lse), reps)
        t_q = timed(lambda: ops.attn_bwd_dq_block(q, k, v, do, lse, delta, causal=True, dq=dq1), reps)
        t_k = timed(lambda: ops.attn_bwd_dkdv_block(q, k, v, do, lse, delta, causal=True, dk=dk1, dv=dv1), reps)
        print(f"S={S}: one launch per kernel   fwd {t_f:8.3f}  dq {t_q:8.3f}  dkdv {t_k:8.3f}  sum {t_f + t_q + t_k:8.3f} ms", flush=True)
        for nb in (2, 4, 8, 16):
            h = S // nb
            o2 = torch.empty_like(out)
            lses = [torch.empty(1, H, h, dtype=torch.float32, device="cuda") for _ in range(nb)]

            def fwd():
                for b in range(nb):
                    ops.attn_fwd_block(q[:, b * h:(b + 1) * h], k[:, :(b + 1) * h], v[:, :(b + 1) * h], q_start=b * h, k_start=0,
                                       causal=True, out=o2[:, b * h:(b + 1) * h], lse=lses[b])
            fwd()
            deltas = [ops.attn_bwd_delta(o2[:, b * h:(b + 1) * h], do[:, b * h:(b + 1) * h], lses[b]) for b in range(nb)]
            dq2 = torch.empty_like(dq1)

            def bdq():
                for b in range(nb):
                    ops.attn_bwd_dq_block(q[:, b * h:(b + 1) * h], k[:, :(b + 1) * h], v[:, :(b + 1) * h], do[:, b * h:(b + 1) * h],
                                          lses[b], deltas[b], q_start=b * h, k_start=0, causal=True, dq=dq2[:, b * h:(b + 1) * h])
            dk2, dv2 = torch.empty_like(dk1), torch.empty_like(dv1)

            def bkv():
                for b in range(nb):
                    ops.attn_bwd_dkdv_block(q, k[:, b * h:(b + 1) * h], v[:, b * h:(b + 1) * h], do, lse, delta, q_start=0, k_start=b * h,
                                            causal=True, dk=dk2[:, b * h:(b + 1) * h], dv=dv2[:, b * h:(b + 1) * h])
            # heaviest band first (the last q band / the first key band walk longest) makes no difference to a serial stream
            tf, tq, tk = timed(fwd, reps), timed(bdq, reps), timed(bkv, reps)
            same = (torch.equal(o2, out), torch.equal(dq2, dq1), torch.equal(dk2, dk1) and torch.equal(dv2, dv1))
            print(f"S={S}: {nb:2d} bands                fwd {tf:8.3f}  dq {tq:8.3f}  dkdv {tk:8.3f}  sum {tf + tq + tk:8.3f} ms   "
                  f"({100 * (tf + tq + tk - t_f - t_q - t_k) / (t_f + t_q + t_k):+.1f} %)  bit-identical out/dq/dkdv: {same}", flush=True)
        del q, k, v, do, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
