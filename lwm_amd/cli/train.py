"""`python -m lwm_amd.cli.train` -- the flag set of lwm/train.py:31-56 (scripts/run_train_text.sh,
scripts/run_train_vision_text.sh).  Runs --total_steps optimisation steps of the model harness
(lwm_amd/llama.py, lwm_amd/vision_llama.py): forward + backward through the HIP RingAttention hot path,
the loss of lwm/train.py:171-209, AdamW with the reference's warm-up + cosine schedule
(--optimizer.adamw_optimizer.*).  Data: synthetic token batches of the configured shape (the reference's
dataset / tokenizer pipeline, wandb logging and GCS checkpoint streaming are not part of the hot path;
their flags are accepted and listed as unused).

Sequence parallelism (--mesh_dim ...,sp): the ranks of one "sp" group share a batch; each holds the rows its ownership
rule gives it -- ZIGZAG half-chunks by default (lwm_amd.ringattention.set_sp_group; LWM_SP_LAYOUT=contiguous selects the
reference's contiguous blocks, lwm/llama.py:560-562) -- cut with `sp_shard`, positions from `sp_positions`; the K/V
exchange is driven by the C-ABI ring driver (RCCL on a side stream) when the job's backend is RCCL.  Two flags beyond the
reference's set, for tests and diagnosis: --lwm_dump_grads=<file> (rank 0 saves the batch, the parameters, the loss
and every gradient of the last step) and --lwm_balance_report (the attention launches of every sp rank timed in turn on
this rank's GPU; --lwm_balance_seq=<S> / --lwm_balance_heads=<H> time another sequence length / head count than the
job's: a debug model's 2 heads make 64 workgroups for 256 CUs, and the longest workgroup, not the work, sets the time)."""
from __future__ import annotations

import math
import sys
import time

import torch

from . import _common as C
from ._flags import parse

DEFAULTS = dict(
    modality="text", use_data_sharded_loader=True, seed=42, mesh_dim="1,-1,1,1", dtype="bf16", total_steps=10000,
    load_llama_config="", update_llama_config="", load_checkpoint="", load_dataset_state="", log_freq=50,
    save_model_freq=0, save_milestone_freq=0, eval_steps=0, tokenizer="LargeWorldModel/LWM-Text-1M",
    log_all_worker=False, autoresume=False, lwm_dump_grads="", lwm_balance_report=False, lwm_balance_seq=0, lwm_balance_heads=0)
GROUPS = ("train_dataset", "eval_dataset", "optimizer", "checkpointer", "llama", "logger", "jax_distributed")


def _shape_from_dataset(ds: dict, vision: bool):
    """(batch_size, seq_length) from --train_dataset.<type>_dataset.* (lwm/data.py)."""
    kind = ds.get("type", "json_vision" if vision else "json")
    sub = ds.get(f"{kind}_dataset", {})
    return int(sub.get("batch_size", 1)), int(sub.get("seq_length", 1024))


def lr_at(step, opt: dict):
    """optax.warmup_cosine_decay_schedule as configured in tux's AdamW factory: linear 0 -> lr over
    lr_warmup_steps, cosine to end_lr at lr_decay_steps."""
    lr, end = float(opt.get("lr", 0.01)), float(opt.get("end_lr", opt.get("lr", 0.01) * 0.1))
    init, warm, decay = float(opt.get("init_lr", 0.0)), int(opt.get("lr_warmup_steps", 2000)), int(opt.get("lr_decay_steps", 500000))
    if step < warm:
        return init + (lr - init) * step / max(warm, 1)
    t = min(max(step - warm, 0) / max(decay - warm, 1), 1.0)
    return end + 0.5 * (lr - end) * (1 + math.cos(math.pi * t))


def main(argv=None):
    F = parse(DEFAULTS, GROUPS, argv, prog="lwm_amd.cli.train")
    if F.modality not in ("text", "vision,text"):
        raise SystemExit(f"unsupported modality: {F.modality}")        # lwm/train.py:76
    vision = F.modality == "vision,text"
    mesh = C.setup_mesh(F.mesh_dim)
    if not torch.cuda.is_available():
        raise SystemExit("lwm_amd.cli.train needs an MI355X (the hot path has no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = C.build_config(F, vision)
    dtype = C.torch_dtype(F.dtype)
    model = C.load_checkpoint(C.build_model(cfg, vision, dtype, F.seed, dev), F.load_checkpoint)
    from ..llama_ops import use_tuned_gemms
    if use_tuned_gemms():
        C.note("library GEMMs: tuned solutions of lwm_amd/gemm_tuning_gfx950.csv where a shape is listed (LWM_GEMM_TUNING=0: off)")
    for g in ("eval_dataset", "checkpointer", "logger", "jax_distributed"):
        if F[g]:
            C.note(f"--{g}.* accepted, unused here: {sorted(F[g])}")
    batch, seq = _shape_from_dataset(F.train_dataset, vision)
    sp = mesh["sp"]
    if seq % sp:
        raise SystemExit(f"seq_length {seq} is not divisible by the sp axis ({sp})")
    dp = mesh["dp"] * mesh["fsdp"]
    local_b = max(1, batch // max(dp, 1))
    C.note(f"mesh {mesh}; synthetic batches of {local_b} x {seq // sp} tokens per process (global {batch} x {seq})")
    opt_cfg = dict(F.optimizer.get("adamw_optimizer", {}))
    if F.optimizer.get("type", "adamw") != "adamw":
        raise SystemExit(f"--optimizer.type={F.optimizer.get('type')!r}: only adamw")
    accum = int(F.optimizer.get("accumulate_gradient_steps", 1))
    opt = torch.optim.AdamW(model.parameters(), lr=lr_at(0, opt_cfg), betas=(float(opt_cfg.get("b1", 0.9)),
                            float(opt_cfg.get("b2", 0.95))), weight_decay=float(opt_cfg.get("weight_decay", 1e-4)))
    clip = float(opt_cfg.get("clip_gradient", 1.0))
    import torch.distributed as dist
    from ..ringattention import sp_all_reduce_sum, sp_layout, sp_shard
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    world_rank = dist.get_rank() if multi else 0
    gloo = multi and dist.get_backend() != "nccl"
    if sp > 1:
        C.note(f"sp axis: {sp} ranks, layout {sp_layout('sp', seq // sp)}")
    # the ranks of one sp group share a batch (each holds a slice of its sequences); data-parallel replicas draw
    # DIFFERENT batches: seed by the replica's coordinate (sp is the fastest axis of the mesh, lwm_amd/mesh.py)
    gen = torch.Generator(device=dev).manual_seed(F.seed + 17 * (world_rank // max(sp, 1)))
    history = []
    for step in range(int(F.total_steps)):
        for pg in opt.param_groups:
            pg["lr"] = lr_at(step, opt_cfg)
        t0 = time.perf_counter()
        for _ in range(accum):
            # a global batch of seq+1 tokens; this process holds ITS rows of every sequence (sp_shard: the ownership
            # rule bound to the "sp" axis).  Each rank's loss is its share of the per-sequence mean (the count of targets
            # is summed over the ring inside the loss operator), so the shares add up to the 1-process loss.
            full = torch.randint(0, cfg.vocab_size, (local_b, seq + 1), device=dev, generator=gen)
            if vision:
                vm_full = torch.zeros(local_b, seq + 1, dtype=torch.bool, device=dev)
                vm_full[:, seq // 4: seq // 4 + (seq // 2)] = True       # a block of vision tokens mid-sequence
                full = torch.where(vm_full, full % cfg.vision_vocab_size, full)
                loss, metrics = model.loss(sp_shard(full[:, :-1]), sp_shard(vm_full[:, :-1]), sp_shard(full[:, 1:]),
                                           sp_shard(vm_full[:, 1:]))
            else:
                loss, acc = model.loss(sp_shard(full[:, :-1]), sp_shard(full[:, 1:]))
                metrics = dict(accuracy=acc)
            (loss / accum).backward()
        if sp > 1:          # the whole sequence's figures, for the log
            loss = sp_all_reduce_sum(loss.detach().clone())
            metrics = {k: sp_all_reduce_sum(v.detach().clone()) for k, v in metrics.items()}
        if multi:           # parameters are replicated: SUM of the sp shares, MEAN over the data-parallel replicas, in buckets
            grads = [p.grad for p in model.parameters() if p.grad is not None]
            bucket, size = [], 0
            for g in grads + [None]:
                if g is None or (bucket and (size + g.numel() > (1 << 26) or g.dtype != bucket[0].dtype)):
                    flat = torch.cat([x.reshape(-1) for x in bucket])
                    if gloo:        # gloo moves host memory and has no bf16: staged as f32 (a dry run on shared devices)
                        host = flat.float().cpu()
                        dist.all_reduce(host)
                        flat.copy_(host.div_(max(dp, 1)))
                    else:
                        red = flat.float() if flat.dtype != torch.float32 else flat
                        dist.all_reduce(red)
                        flat.copy_(red.div_(max(dp, 1)))
                    off = 0
                    for x in bucket:
                        x.copy_(flat[off:off + x.numel()].view_as(x))
                        off += x.numel()
                    bucket, size = [], 0
                if g is not None:
                    bucket.append(g)
                    size += g.numel()
        if F.lwm_dump_grads and step == int(F.total_steps) - 1 and world_rank == 0:
            torch.save({"loss": float(loss.detach()), "metrics": {k: float(v) for k, v in metrics.items()}, "tokens": full.cpu(),
                        "params": {n: p.detach().float().cpu() for n, p in model.named_parameters()},
                        "grads": {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}},
                       F.lwm_dump_grads)
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
        opt.step()
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rec = dict(step=step, loss=float(loss.detach()), learning_rate=lr_at(step, opt_cfg), tokens_per_s=batch * seq * accum / dt,
                   **{k: float(v.detach()) if hasattr(v, "detach") else float(v) for k, v in metrics.items()})
        history.append(rec)
        if F.log_freq and step % int(F.log_freq) == 0 and (world_rank == 0 or F.log_all_worker):
            print(rec, flush=True)
    if sp > 1:
        from ..ring import ring_driver_info
        from ..ringattention import _resolve_axis
        info = ring_driver_info(_resolve_axis("sp"))
        C.note(f"ring driver: {info}")
        history.append(dict(ring=info, layout=sp_layout("sp", seq // sp)))
        if F.lwm_balance_report:
            history.append(dict(balance=balance_report(cfg, local_b, (int(F.lwm_balance_seq) or seq) // sp, sp, dev,
                                                       heads=int(F.lwm_balance_heads) or None)))
            if world_rank == 0:
                print("LWM_BALANCE " + __import__("json").dumps(history[-1]["balance"]), flush=True)
    return history


def balance_report(cfg, B, c, n, dev, reps=3, heads=None, windows=3, layout=None):
    """Per-rank attention time of ONE layer of this job's shard shape, measured without the other ranks in the way: every
    rank of the sp ring is played in turn on the GPU of sp rank 0 by the C ring driver over a transport that moves nothing
    (CRing.null) -- the launch list of rank r under the ownership rule in force, forward + backward -- while the other
    processes wait at a barrier (they may share that GPU).  The chip runs at its power limit under these kernels and the
    clock wanders: a rank's figure is the best of `windows` timing windows of `reps` layers.
    CAVEAT: when the job's processes SHARE a GPU (a dry run), their idle contexts still cost the measuring one scheduler
    time slices -- 43 ms and erratic where the same launch lists take 15 ms alone (scripts/gpu_balance_probe.py): inside such a
    job the figures are indicative only; call this function from a process that has the GPU to itself (layout="zigzag" |
    "contiguous" names the rule when no "sp" axis is bound) for the real ones, as tests/test_gpu_cli_ring.py does.
    -> {"ms_per_rank": [...], "max_over_mean"} on every rank"""
    import torch.distributed as dist
    from ..ring_c import CRing
    from ..ringattention import sp_layout, sp_size_rank
    H, D = heads or cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
    kind = layout or sp_layout("sp", c)
    me = sp_size_rank("sp")[1]
    multi = dist.is_available() and dist.is_initialized()
    ms = [0.0] * n
    if multi:
        torch.cuda.synchronize()
        dist.barrier()
    if me == 0:
        g = torch.Generator(device=dev).manual_seed(5)
        q, k, v, do = (torch.randn(B, c, H, D, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16) for _ in range(4))
        rings = [CRing.null(r, n, layout=kind, schedule="direct", device=dev) for r in range(n)]

        def layer(ring):
            o, l = ring.forward(q, k, v, causal=True)
            ring.backward(q, k, v, o, l, do, causal=True)

        best = [float("inf")] * n
        for ring in rings:
            layer(ring)
        for _ in range(windows):          # (the ranks interleaved: a slow spell of the clock hits all of them alike)
            for r, ring in enumerate(rings):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    layer(ring)
                e1.record()
                torch.cuda.synchronize()
                best[r] = min(best[r], e0.elapsed_time(e1) / reps)
        for ring in rings:
            ring.close()
        ms = best
    if multi:
        torch.cuda.synchronize()
        from ..ringattention import _resolve_axis
        grp = _resolve_axis("sp")
        box = [ms]
        dist.broadcast_object_list(box, src=dist.get_global_rank(grp, 0) if grp is not None else 0, group=grp)
        ms = box[0]
    return {"layout": kind, "shape": [B, c, H, D], "ms_per_rank": [round(x, 4) for x in ms],
            "max_over_mean": max(ms) / (sum(ms) / len(ms))}


if __name__ == "__main__":
    main(sys.argv[1:])
