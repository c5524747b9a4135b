"""PyTorch TunableOp over the library GEMMs of the LWM-7B training step (S = 32768), in the exact call forms the harness
issues (lwm_amd/llama_ops.py: (out, in) kernels forward, fused QKV / w1|w3, transposed narrow operand for wgrad, beta = 1
epilogues, the lm_head chunk): every rocBLAS / hipBLASLt solution is timed per shape and the fastest recorded.
    gpurun -- 'python scripts/gpu_tune_gemms.py gpurun_out/tune/gemm_tuning_gfx950.csv > gpurun_out/tune/log.txt'
The CSV is then committed as lwm_amd/gemm_tuning_gfx950.csv; lwm_amd/llama_ops.py loads it (tuning off) at import."""
import os
import sys
import time

import torch
import torch.cuda.tunable as T

S, d, f, V = 32768, 4096, 11008, 32000
dev = "cuda"


def rnd(*shape):
    return (torch.randn(*shape, device=dev, dtype=torch.float32) * 0.05).to(torch.bfloat16)


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def cases():
    x, g = rnd(S, d), rnd(S, d)
    out = []
    for tag, K, N in (("wqkv", d, 3 * d), ("wo", d, d), ("w13", d, 2 * f), ("w2", f, d)):
        xx, gg = (x if K == d else rnd(S, K)), (g if N == d else rnd(S, N))
        wt, wcat = rnd(N, K), rnd(K, N)
        res = rnd(S, N) if tag in ("wo", "w2") else None
        y, dx = torch.empty(S, N, device=dev, dtype=torch.bfloat16), torch.empty(S, K, device=dev, dtype=torch.bfloat16)
        flops = 2.0 * S * K * N
        if res is None:
            out.append((f"{tag} fwd   x @ wt.t()", flops, lambda xx=xx, wt=wt: xx @ wt.t()))
        else:
            out.append((f"{tag} fwd   addmm(res, x, wt.t())", flops, lambda xx=xx, wt=wt, res=res: torch.addmm(res, xx, wt.t())))
        out.append((f"{tag} dgrad g @ wcat.t()", flops, lambda gg=gg, wcat=wcat: gg @ wcat.t()))
        if K <= N:
            xt = xx.t().contiguous()
            out.append((f"{tag} wgrad xt @ g", flops, lambda xt=xt, gg=gg: xt @ gg))
        else:
            gt = gg.t().contiguous()
            out.append((f"{tag} wgrad x.t() @ gt.t()", flops, lambda xx=xx, gt=gt: xx.t() @ gt.t()))
    c = 8192
    h, kb, dl = rnd(c, d), rnd(d, V), rnd(c, V)
    fl = 2.0 * c * d * V
    out.append(("lm_head fwd   h @ kb", fl, lambda: h @ kb))
    out.append(("lm_head dgrad dl @ kb.t()", fl, lambda: dl @ kb.t()))
    out.append(("lm_head wgrad h.t() @ dl", fl, lambda: h.t() @ dl))
    return out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "gemm_tuning_gfx950.csv"
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    cs = cases()
    before = [timed(fn) for _, _, fn in cs]
    T.enable(True)
    T.tuning_enable(True)
    T.set_filename(path)
    T.set_max_tuning_duration(int(os.environ.get("TUNE_MS", "60")))
    T.set_max_tuning_iterations(int(os.environ.get("TUNE_ITERS", "20")))
    t0 = time.time()
    for name, _, fn in cs:
        t1 = time.time()
        fn()
        torch.cuda.synchronize()
        print(f"tuned {name}: {time.time() - t1:.1f} s", flush=True)
    print(f"tuning took {time.time() - t0:.1f} s", flush=True)
    T.tuning_enable(False)
    after = [timed(fn) for _, _, fn in cs]
    tot_b = tot_a = 0.0
    for (name, fl, _), b, a in zip(cs, before, after):
        print(f"{name:36s} {b:8.3f} ms {fl / b / 1e9:8.1f} TF/s  ->  {a:8.3f} ms {fl / a / 1e9:8.1f} TF/s  ({100 * (b - a) / b:+5.1f} %)")
        tot_b, tot_a = tot_b + b, tot_a + a
    print(f"sum over the listed calls: {tot_b:.3f} -> {tot_a:.3f} ms")
    try:
        T.write_file(path)
    except Exception:
        pass
    print("results:", len(T.get_results()), "validators:", T.get_validators())


if __name__ == "__main__":
    main()
