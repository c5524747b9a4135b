"""hipBLASLt (torch.matmul) on the GEMM shapes of one LWM-7B training step at S = 32768, per pass (forward / dgrad / wgrad)
and per operand layout, TF/s each -- the library half of bench.py's `model_full` leg (profiles/r06_model_full.md).
    gpurun -- 'python scripts/gpu_gemm_shapes.py > gpurun_out/gemm_shapes.txt'"""
import sys
import torch

S, d, f, V = 32768, 4096, 11008, 32000
dev = "cuda"


def timed(fn, reps=6):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def rnd(*shape):
    return (torch.randn(*shape, device=dev, dtype=torch.float32) * 0.05).to(torch.bfloat16)


def line(name, M, N, K, fn, extra=""):
    ms = timed(fn)
    print(f"{name:46s} M={M:6d} N={N:6d} K={K:6d}  {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TF/s {extra}", flush=True)
    return ms


def main():
    M = S
    for (tag, K, N) in (("wq|wk|wv|wo (d x d)", d, d), ("w1|w3 (d x f)", d, f), ("w2 (f x d)", f, d),
                        ("wqkv fused (d x 3d)", d, 3 * d), ("w13 fused (d x 2f)", d, 2 * f)):
        x, w, g = rnd(M, K), rnd(K, N), rnd(M, N)
        wt = w.t().contiguous()
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
        dw = torch.empty(K, N, device=dev, dtype=torch.bfloat16)
        dwt = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
        print(f"--- {tag}")
        line("fwd   x @ W            (W stored (K,N))", M, N, K, lambda: torch.matmul(x, w, out=y))
        line("fwd   x @ Wt.t()       (W stored (N,K))", M, N, K, lambda: torch.matmul(x, wt.t(), out=y))
        line("dgrad g @ W.t()        (W stored (K,N))", M, K, N, lambda: torch.matmul(g, w.t(), out=dx))
        line("dgrad g @ Wt           (W stored (N,K))", M, K, N, lambda: torch.matmul(g, wt, out=dx))
        line("wgrad x.t() @ g     -> (K,N)", K, N, M, lambda: torch.matmul(x.t(), g, out=dw))
        line("wgrad g.t() @ x     -> (N,K)", N, K, M, lambda: torch.matmul(g.t(), x, out=dwt))
        del x, w, g, wt, y, dx, dw, dwt
    # lm_head, in chunks of 8192 rows (chunked_lm_head_loss)
    print("--- lm_head (d x V), 8192-row chunks")
    c = 8192
    h, k, dl = rnd(c, d), rnd(d, V), rnd(c, V)
    kt = k.t().contiguous()
    line("fwd   h @ K", c, V, d, lambda: h @ k)
    line("fwd   h @ Kt.t()", c, V, d, lambda: h @ kt.t())
    line("dgrad dl @ K.t()", c, d, V, lambda: dl @ k.t())
    line("dgrad dl @ Kt", c, d, V, lambda: dl @ kt)
    line("wgrad h.t() @ dl", d, V, c, lambda: h.t() @ dl)
    line("wgrad dl.t() @ h", V, d, c, lambda: dl.t() @ h)
    # three-way sums the unfused dgrad needs
    a, b, c3 = rnd(S, d), rnd(S, d), rnd(S, d)
    ms = timed(lambda: a + b + c3)
    print(f"a + b + c over (S, d) bf16 (two torch adds): {ms:.3f} ms")
    ms = timed(lambda: torch.add(a, b))
    print(f"a + b over (S, d) bf16: {ms:.3f} ms  ({3 * a.numel() * 2 / ms / 1e6:.0f} GB/s)")


if __name__ == "__main__":
    main()
