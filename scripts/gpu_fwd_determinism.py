"""Run-to-run determinism and hint/no-hint equality of the forward (a race in the staging shows up as differences)."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from lwm_amd import ops
torch.manual_seed(0)
def mk(S, H): return torch.randn(1, S, H, 128, device="cuda", dtype=torch.bfloat16)
for S, H, packed in ((8192, 4, True), (8192, 4, False), (32768, 8, False), (2048, 32, True)):
    q, k, v = mk(S, H), mk(S, H), mk(S, H)
    seg = None
    if packed:
        rng = np.random.default_rng(5); s = np.zeros((1, S), np.int32); pos = d = 0
        while pos < S:
            ln = int(rng.integers(300, 2000)); s[:, pos:pos + ln] = d; pos += ln; d += 1
        seg = torch.from_numpy(s).cuda()
    outs = []
    for skip in ((True, False) if packed else (True,)):
        ops.SEGMENT_SKIP = skip
        for rep in range(4):
            o, l = ops.attn_fwd_block(q, k, v, causal=True, seg_q=seg, seg_k=seg)
            torch.cuda.synchronize()
            outs.append((skip, rep, o.clone(), l.clone()))
    ops.SEGMENT_SKIP = True
    ref = outs[0]
    for skip, rep, o, l in outs[1:]:
        do = (o.float() - ref[2].float()).abs()
        nbad = int((do > 0).sum())
        rows = torch.nonzero(do.amax(dim=(0, 2, 3)) > 0).flatten()[:8].tolist()
        print(f"S={S} H={H} packed={packed} skip={skip} rep={rep}: out differs in {nbad} elements (max {do.max().item():.3e}), "
              f"lse max diff {(l - ref[3]).abs().nan_to_num(0).max().item():.3e}, first rows {rows}")
