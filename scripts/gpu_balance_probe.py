import sys, types, torch, time
sys.path.insert(0, '/root/repo')
from lwm_amd.ring_c import CRing
dev = torch.device('cuda', 0)
B, c, H, D, n = 1, 8192, 32, 128, 8
g = torch.Generator(device=dev).manual_seed(5)
q, k, v, do = (torch.randn(B, c, H, D, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16) for _ in range(4))
rings = [CRing.null(r, n, layout="zigzag", schedule="direct", device=dev) for r in range(n)]
def layer(ring):
    o, l = ring.forward(q, k, v, causal=True)
    ring.backward(q, k, v, o, l, do, causal=True)
for ring in rings: layer(ring)
torch.cuda.synchronize()
for w in range(4):
    row = []
    for r, ring in enumerate(rings):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): layer(ring)
        e1.record(); torch.cuda.synchronize()
        row.append(round(e0.elapsed_time(e1) / 3, 2))
    print("window", w, row, flush=True)
# one ring at a time, fresh
for r in (0, 3, 7):
    ring = CRing.null(r, n, layout="zigzag", schedule="direct", device=dev)
    layer(ring); torch.cuda.synchronize()
    t = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); layer(ring); e1.record(); torch.cuda.synchronize(); t.append(round(e0.elapsed_time(e1), 2))
    print("single ring", r, t, ring.last_form, flush=True)
    ring.close()
