"""Second look at the library GEMMs of the LWM-7B step (S = 32768): the weight gradient with both operands transposed first
(reduction dimension contiguous, the layout hipBLASLt runs fastest), the cost of those transposes, and the residual add as
the GEMM's beta = 1 epilogue (torch.addmm).    gpurun -- 'python scripts/gpu_gemm_shapes2.py > gpurun_out/gemm_shapes2.txt'"""
import torch
from gpu_gemm_shapes import timed, rnd, line, S, d, f

dev = "cuda"


def main():
    for (tag, K, N) in (("wo (d x d)", d, d), ("wqkv (d x 3d)", d, 3 * d), ("w13 (d x 2f)", d, 2 * f), ("w2 (f x d)", f, d)):
        x, g = rnd(S, K), rnd(S, N)
        xt, gt = x.t().contiguous(), g.t().contiguous()
        dw = torch.empty(K, N, device=dev, dtype=torch.bfloat16)
        dwt = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
        print(f"--- wgrad {tag}")
        line("x.t() @ g              (today)", K, N, S, lambda: torch.matmul(x.t(), g, out=dw))
        line("xt @ gt.t()            (both S-contiguous)", K, N, S, lambda: torch.matmul(xt, gt.t(), out=dw))
        line("gt @ xt.t() -> (N,K)   (both S-contiguous)", N, K, S, lambda: torch.matmul(gt, xt.t(), out=dwt))
        line("xt @ g                 (x transposed only)", K, N, S, lambda: torch.matmul(xt, g, out=dw))
        line("x.t() @ gt.t()         (g transposed only)", K, N, S, lambda: torch.matmul(x.t(), gt.t(), out=dw))
        ms = timed(lambda: x.t().contiguous())
        print(f"    x.t().contiguous() ({S},{K}): {ms:.3f} ms ({2 * x.numel() * 2 / ms / 1e6:.0f} GB/s)")
        ms = timed(lambda: g.t().contiguous())
        print(f"    g.t().contiguous() ({S},{N}): {ms:.3f} ms ({2 * g.numel() * 2 / ms / 1e6:.0f} GB/s)")
        del x, g, xt, gt, dw, dwt
    print("--- residual add as the GEMM epilogue")
    a, w, r = rnd(S, d), rnd(d, d), rnd(S, d)
    wt = w.t().contiguous()
    line("a @ wt.t()", S, d, d, lambda: a @ wt.t())
    line("addmm(r, a, wt.t())", S, d, d, lambda: torch.addmm(r, a, wt.t()))
    line("a @ wt.t() + r", S, d, d, lambda: (a @ wt.t()) + r)
    a2, w2 = rnd(S, f), rnd(f, d)
    w2t = w2.t().contiguous()
    line("a2 @ w2t.t()", S, d, f, lambda: a2 @ w2t.t())
    line("addmm(r, a2, w2t.t())", S, d, f, lambda: torch.addmm(r, a2, w2t.t()))
    # accumulate into an existing gradient: dx += g @ W.t()  (the residual branch's gradient)
    g = rnd(S, d)
    line("addmm(r, g, w.t())  [dgrad + residual grad]", S, d, d, lambda: torch.addmm(r, g, w.t()))
    # weight re-layouts per step
    ws = [rnd(d, d) for _ in range(3)]
    ms = timed(lambda: torch.cat([k.t() for k in ws], 0))
    print(f"cat([wq.t(), wk.t(), wv.t()]) -> (3d, d): {ms:.3f} ms")
    ms = timed(lambda: torch.cat(ws, 1))
    print(f"cat([wq, wk, wv], 1) -> (d, 3d): {ms:.3f} ms")
    w13 = [rnd(d, f) for _ in range(2)]
    ms = timed(lambda: torch.cat([k.t() for k in w13], 0))
    print(f"cat([w1.t(), w3.t()]) -> (2f, d): {ms:.3f} ms")
    ms = timed(lambda: w2.t().contiguous())
    print(f"w2.t().contiguous() (f,d)->(d,f): {ms:.3f} ms")


if __name__ == "__main__":
    main()
