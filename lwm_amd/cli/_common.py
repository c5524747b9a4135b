"""What the three entry points share: mesh / process-group setup from --mesh_dim, the model from
--load_llama_config / --update_llama_config / --llama.*, checkpoints from --load_checkpoint, the
tokenizer from --tokenizer."""
from __future__ import annotations

import os
import sys

import torch

from .. import mesh as _mesh
from ..llama import LLaMAConfig, LLaMAForCausalLM, parse_config_updates
from ..ringattention import set_sp_group
from ..vision_llama import VideoLLaMAConfig, VideoLLaMAForCausalLM

# fields the reference copies from --llama.* onto a loaded size (lwm/train.py:104-116, lwm/vision_chat.py:152-163)
_SCAN_FIELDS = ("scan_attention", "scan_mlp", "scan_query_chunk_size", "scan_key_chunk_size", "scan_mlp_chunk_size",
                "scan_layers", "param_scan_axis")


def note(msg):
    print(f"[lwm_amd.cli] {msg}", file=sys.stderr, flush=True)


def setup_mesh(mesh_dim: str):
    """--mesh_dim (lwm/train.py:35; tux.get_jax_mesh) -> this process's place in the (dp, fsdp, tp, sp)
    mesh; binds mesh axis "sp" to its process group.  One process per GPU: under torch.distributed.run
    the world is the mesh; a plain `python -m ...` run is a 1-device mesh."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        # LWM_DIST_BACKEND=gloo: a dry run of N processes on fewer GPUs than ranks (ranks share devices; with
        # LWM_RING_TRANSPORT=ipc the K/V exchange then goes through the library's IPC transport, everything else
        # through host memory) -- the whole multi-process code path on a 1-GPU box
        backend = os.environ.get("LWM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            n_dev = torch.cuda.device_count()
            if backend == "nccl" and local >= n_dev:
                raise SystemExit(f"local rank {local} has no GPU of its own ({n_dev} visible): one process per GPU "
                                 f"(LWM_DIST_BACKEND=gloo is the dry run on shared devices)")
            torch.cuda.set_device(local % n_dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    m = _mesh.parse_mesh_dim(mesh_dim, world)
    if m["tp"] != 1 or m["fsdp"] != 1:
        # tensor / fsdp sharding of the PARAMETERS is XLA-SPMD's job in the reference; the hot path built
        # here shards the sequence.  Refuse rather than silently replicate.
        if world > 1:
            raise NotImplementedError(f"mesh_dim {mesh_dim!r}: only the dp and sp axes are executed here "
                                      f"(got fsdp={m['fsdp']}, tp={m['tp']} over {world} processes)")
    if world > 1:
        set_sp_group(_mesh.sp_group(m))
    return m


def build_config(flags, vision: bool):
    """lwm/train.py:101-121 / lwm/vision_chat.py:150-170."""
    cls = VideoLLaMAConfig if vision else LLaMAConfig
    group = dict(flags.get("llama", {}))
    if flags.load_llama_config:
        cfg = cls.load_config(flags.load_llama_config)
        cfg.update({k: group[k] for k in _SCAN_FIELDS if k in group})
    else:
        cfg = cls(**group)
    if flags.update_llama_config:
        cfg.update(parse_config_updates(flags.update_llama_config))
    cfg.update(dict(mesh_dim=flags.mesh_dim))
    return cfg


def torch_dtype(name: str, inference: bool = False):
    """--dtype (fp32 | bf16 | fp16, tux.get_float_dtype_by_name; the reference's default and every launcher's value
    is fp32, lwm/train.py:36, scripts/run_train_text.sh:21).

    bf16 is the headline dtype (bf16 operands, f32 logits / softmax / accumulation = the reference's bf16 run with
    float32_logits=True, BASELINE configs 2-5) and the DEFAULT of the entry points here (a documented divergence from
    the reference's default: README.md).  fp32 is what the command line says it is: float32
    parameters, activations and attention operands through the f32 flavour of every kernel (csrc/attn_f32.h on the
    exact-f32 matrix instruction, csrc/elem_f32.h; library GEMMs in f32) -- the reference's shipped launch lines run
    as they are.  `inference=True` (vision_chat, vision_generation) only changes the note: cached decoding in fp32 runs
    on ONE rank (prefill and decode steps through the f32 training-op kernel with the mask handed over as its
    structure; the sharded decode / dense-mask kernels take bf16 and lwm_amd.llama refuses an sp axis > 1 in fp32).
    fp16 has no path, and a run is never silently computed in another precision than its command line says."""
    if name not in ("fp32", "bf16", "fp16", "float32", "bfloat16", "float16"):
        raise SystemExit(f"unknown --dtype {name!r}")
    if name in ("fp32", "float32"):
        if inference:
            note("--dtype=fp32: cached decoding runs through the float32 flavour of the attention op, one rank, eager steps "
                 "(the bf16 path has the streaming decode kernel and the hipGraph step)")
        return torch.float32
    if name not in ("bf16", "bfloat16"):
        raise SystemExit(f"--dtype={name}: not supported -- there is no fp16 path (the kernels take bf16 operands with f32 "
                         f"logits, softmax and accumulation, or float32 throughout).  The reference's launch scripts pass "
                         f"--dtype='fp32' (scripts/run_train_text.sh:21, run_eval_needle.sh:17, lwm/train.py:36), which runs "
                         f"as it is; for the headline path replace that ONE flag by\n    --dtype='bf16'\n"
                         f"and keep the rest of the command line (BASELINE configs 2-5: bf16 operands, float32_logits=True).")
    return torch.bfloat16


def build_model(cfg, vision: bool, dtype, seed: int, device):
    torch.manual_seed(seed)
    with torch.device(device):
        return (VideoLLaMAForCausalLM if vision else LLaMAForCausalLM)(cfg, dtype)


def load_checkpoint(model, spec: str):
    """--load_checkpoint 'params::<path>' / 'flax_params::<path>' (tux StreamingCheckpointer,
    lwm/train.py:337, scripts/run_vision_chat.sh:27) or 'hf::<dir>' (the PyTorch release, README.md:74)."""
    if not spec:
        note("no --load_checkpoint: randomly initialised weights (seeded)")
        return model
    if "::" not in spec:
        raise SystemExit(f"--load_checkpoint {spec!r}: expected '<type>::<path>' (params, flax_params, hf)")
    kind, path = spec.split("::", 1)
    if not path:
        raise SystemExit(f"--load_checkpoint {spec!r}: empty path")
    from .. import weights as W
    if kind in ("params", "flax_params", "trainstate_params"):
        flat = W.read_flax_stream(path)
        prefix = "params/params/" if any(k.startswith("params/params/") for k in flat) else "params/"
        params = W.flax_llama_to_lwm(flat, prefix=prefix, param_scan_axis=getattr(model.cfg, "param_scan_axis", 0))
        return W.load_params(model, params, strict=False, report=note)
    if kind == "hf":
        sd, _ = W.read_hf_checkpoint(path)
        return W.load_params(model, W.hf_to_lwm(sd, model.cfg.num_attention_heads), strict=False, report=note)
    raise SystemExit(f"--load_checkpoint: unsupported type {kind!r}")


class ByteTokenizer:
    """--tokenizer=synthetic: bytes + 3 (0 pad, 1 bos, 2 eos).  For runs without the LLaMA
    sentencepiece model (no network); `<vision>` / `</vision>` get two reserved ids."""
    pad_token_id, bos_token_id, eos_token_id = 0, 1, 2
    eos_token = "</s>"
    _special = {"<s>": 1, "</s>": 2, "<vision>": 259, "</vision>": 260}

    def encode(self, text):
        ids, i = [], 0
        while i < len(text):
            for tok, tid in self._special.items():
                if text.startswith(tok, i):
                    ids.append(tid)
                    i += len(tok)
                    break
            else:
                ids.extend(b + 3 for b in text[i].encode("utf-8"))
                i += 1
        return ids

    def decode(self, ids, skip_special_tokens=True):
        return bytes(int(t) - 3 for t in ids if 3 <= int(t) < 259).decode("utf-8", errors="replace")

    def batch_decode(self, batch, skip_special_tokens=True):
        return [self.decode(row, skip_special_tokens) for row in batch]


def load_tokenizer(name: str):
    if name == "synthetic":
        return ByteTokenizer()
    try:
        from transformers import AutoTokenizer
        return AutoTokenizer.from_pretrained(name, local_files_only=os.path.isdir(name) or None)
    except Exception as e:
        raise SystemExit(f"--tokenizer={name!r} could not be loaded ({type(e).__name__}: {e}); give a local directory "
                         f"with the LLaMA tokenizer files, or --tokenizer=synthetic for a byte-level stand-in")


def load_vqgan(path: str, seed: int):
    from ..vqgan import VQGAN, VQGANConfig, random_params
    if path:
        return VQGAN(path, replicate=False)
    note("no --vqgan_checkpoint: VQGAN with seeded random weights")
    cfg = VQGANConfig.get_default_config()
    return VQGAN(params=random_params(cfg, seed), config=cfg)
