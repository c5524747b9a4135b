"""Per-shape timing of the VQGAN primitives on cuda:0 (HIP events), for kernel tuning.
Usage: python scripts/bench_vqgan.py [frames]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lwm_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda"


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


rows = []
for name, H, Cin, Cout, k, kw in [
    ("enc L0 conv 128->128 @256", 256, 128, 128, 3, {}),
    ("dec up conv 256->256 128->256", 128, 256, 256, 3, dict(up_shift=1)),
    ("conv 256->256 @128", 128, 256, 256, 3, {}),
    ("conv 256->256 @64", 64, 256, 256, 3, {}),
    ("conv 512->512 @32", 32, 512, 512, 3, {}),
    ("conv 768->768 @16", 16, 768, 768, 3, {}),
    ("down 128->128 256->128", 256, 128, 128, 3, dict(stride=2, pad=0, out_hw=(128, 128))),
    ("conv_in 3->128 @256", 256, 3, 128, 3, {}),
    ("conv_out 128->3 @256", 256, 128, 3, 3, {}),
    ("1x1 128->256 @128", 128, 128, 256, 1, {}),
]:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(k, k, Cin, Cout, device=dev) / (k * k * Cin) ** 0.5
    b = torch.randn(Cout, device=dev)
    t = timeit(lambda: ops.conv2d_nhwc(x, w, b, **kw))
    y = ops.conv2d_nhwc(x, w, b, **kw)
    flops = 2.0 * y.numel() * k * k * Cin
    rows.append((name, t * 1e6, flops / t / 1e12))
for C, H in [(128, 256), (256, 128), (512, 32), (768, 16)]:
    x = torch.randn(B, H, H, C, device=dev)
    g = torch.ones(C, device=dev)
    t = timeit(lambda: ops.groupnorm_silu(x, g, g))
    rows.append((f"groupnorm+silu C={C} @{H}", t * 1e6, 3 * x.numel() * 4 / t / 1e12))  # TB/s (2 reads + 1 write)
for name, us, rate in rows:
    print(f"{name:36s} {us:10.1f} us   {rate:8.2f} {'TB/s' if 'groupnorm' in name else 'TF/s'}")
