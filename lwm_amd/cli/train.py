"""`python -m lwm_amd.cli.train` -- the flag set of lwm/train.py:31-56 (scripts/run_train_text.sh,
scripts/run_train_vision_text.sh).  Runs --total_steps optimisation steps of the model harness
(lwm_amd/llama.py, lwm_amd/vision_llama.py): forward + backward through the HIP RingAttention hot path,
the loss of lwm/train.py:171-209, AdamW with the reference's warm-up + cosine schedule
(--optimizer.adamw_optimizer.*).  Data: synthetic token batches of the configured shape (the reference's
dataset / tokenizer pipeline, wandb logging and GCS checkpoint streaming are not part of the hot path;
their flags are accepted and listed as unused)."""
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
    log_all_worker=False, autoresume=False)
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
    sp_rank, world_rank = 0, 0
    if dist.is_available() and dist.is_initialized():
        world_rank = dist.get_rank()
        if sp > 1:
            from ..ringattention import sp_size_rank
            sp_rank = sp_size_rank("sp")[1]
    # the ranks of one sp group share a batch (each holds a slice of its sequences); data-parallel replicas draw
    # DIFFERENT batches: seed by the replica's coordinate (sp is the fastest axis of the mesh, lwm_amd/mesh.py)
    gen = torch.Generator(device=dev).manual_seed(F.seed + 17 * (world_rank // max(sp, 1)))
    c = seq // sp
    history = []
    for step in range(int(F.total_steps)):
        for pg in opt.param_groups:
            pg["lr"] = lr_at(step, opt_cfg)
        t0 = time.perf_counter()
        for _ in range(accum):
            # a global batch of seq+1 tokens; this process holds the rows [sp_rank*c, (sp_rank+1)*c) of it
            full = torch.randint(0, cfg.vocab_size, (local_b, seq + 1), device=dev, generator=gen)
            sl = slice(sp_rank * c, (sp_rank + 1) * c)
            inp, tgt = full[:, :-1][:, sl], full[:, 1:][:, sl]
            if vision:
                vm_full = torch.zeros(local_b, seq + 1, dtype=torch.bool, device=dev)
                vm_full[:, seq // 4: seq // 4 + (seq // 2)] = True       # a block of vision tokens mid-sequence
                full = torch.where(vm_full, full % cfg.vision_vocab_size, full)
                inp, tgt = full[:, :-1][:, sl], full[:, 1:][:, sl]
                loss, metrics = model.loss(inp, vm_full[:, :-1][:, sl], tgt, vm_full[:, 1:][:, sl])
            else:
                loss, acc = model.loss(inp, tgt)
                metrics = dict(accuracy=acc)
            (loss / accum).backward()
        if sp > 1 or dp > 1:        # parameters are replicated: average the gradients over the job, in buckets
            grads = [p.grad for p in model.parameters() if p.grad is not None]
            bucket, size = [], 0
            for g in grads + [None]:
                if g is None or (bucket and (size + g.numel() > (1 << 26) or g.dtype != bucket[0].dtype)):
                    flat = torch.cat([x.reshape(-1) for x in bucket])
                    dist.all_reduce(flat, op=dist.ReduceOp.AVG)
                    off = 0
                    for x in bucket:
                        x.copy_(flat[off:off + x.numel()].view_as(x))
                        off += x.numel()
                    bucket, size = [], 0
                if g is not None:
                    bucket.append(g)
                    size += g.numel()
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
        opt.step()
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rec = dict(step=step, loss=float(loss), learning_rate=lr_at(step, opt_cfg), tokens_per_s=batch * seq * accum / dt,
                   **{k: float(v) for k, v in metrics.items()})
        history.append(rec)
        if F.log_freq and step % int(F.log_freq) == 0:
            print(rec, flush=True)
    return history


if __name__ == "__main__":
    main(sys.argv[1:])
