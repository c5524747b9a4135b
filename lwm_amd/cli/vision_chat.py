"""`python -m lwm_amd.cli.vision_chat` -- the flag set of lwm/vision_chat.py:22-37
(scripts/run_vision_chat.sh): frames -> VQGAN codes (HIP tokeniser) -> the prompt of
lwm/vision_chat.py:110-145 -> sampled continuation through the cached-decode hot path -> text."""
from __future__ import annotations

import math
import sys

import numpy as np
import torch

from . import _common as C
from ._flags import parse

DEFAULTS = dict(prompt="", input_file="", vqgan_checkpoint="", temperature=0.2, max_n_frames=8, seed=1234,
                mesh_dim="1,-1,1,1", dtype="bf16", load_llama_config="", update_llama_config="", load_checkpoint="",
                tokenizer="LargeWorldModel/LWM-Text-1M")
GROUPS = ("llama", "jax_distributed")


def _process_frame(image, size):
    """lwm/vision_chat.py:59-74: resize the short side to `size`, centre crop, scale to [-1, 1]."""
    w, h = image.size
    if w < h:
        nw, nh = size, int(size * h / w)
    else:
        nh, nw = size, int(size * w / h)
    image = image.resize((nw, nh))
    left, top = (nw - size) / 2, (nh - size) / 2
    image = image.crop((left, top, (nw + size) / 2, (nh + size) / 2))
    return np.array(image, dtype=np.float32) / 127.5 - 1


def read_frames(path, max_n_frames):
    """-> (T,256,256,3) f32 in [-1,1].  .png/.jpg (PIL); .npy of (T,H,W,3) uint8 frames (the container has
    no decord: convert a video with any tool to an array of frames); 'synthetic:<T>' = seeded noise."""
    from PIL import Image
    if path.startswith("synthetic:"):
        T = int(path.split(":", 1)[1])
        return np.random.default_rng(0).uniform(-1, 1, (min(T, max_n_frames), 256, 256, 3)).astype(np.float32)
    if path.endswith((".png", ".jpg", ".jpeg")):
        return _process_frame(Image.open(path).convert("RGB"), 256)[None]
    if path.endswith(".npy"):
        video = np.load(path)
        if video.ndim != 4 or video.shape[-1] != 3:
            raise SystemExit(f"{path}: expected (T,H,W,3) frames")
        ids = list(range(len(video))) if len(video) <= max_n_frames else \
            np.linspace(0, len(video) - 1, max_n_frames, dtype=int).tolist()          # lwm/vision_chat.py:85-88
        return np.stack([_process_frame(Image.fromarray(video[i].astype(np.uint8)), 256) for i in ids])
    raise SystemExit(f"{path}: unsupported input (png / jpg / npy frames; video containers need decord, which "
                     f"this image does not have)")


class Sampler:
    """lwm/vision_chat.py:39-233."""

    def __init__(self, F):
        self.F = F
        self.mesh = C.setup_mesh(F.mesh_dim)
        if not torch.cuda.is_available():
            raise SystemExit("lwm_amd.cli.vision_chat needs an MI355X (the hot path has no CPU fallback)")
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.vqgan = C.load_vqgan(F.vqgan_checkpoint, F.seed)
        self.tokenizer = C.load_tokenizer(F.tokenizer)
        self.n_tokens_per_frame = 257
        self.min_buffer_size = 256
        cfg = C.build_config(F, vision=True)
        cfg.update(dict(bos_token_id=self.tokenizer.bos_token_id, eos_token_id=self.tokenizer.eos_token_id))
        self.config = cfg
        # lwm/vision_chat.py:51-53: the prompt buffer is padded to, and at most this many tokens are generated:
        # max(scan_query_chunk_size, scan_key_chunk_size) * mesh.shape['sp']
        self.block_size = max(cfg.scan_query_chunk_size, cfg.scan_key_chunk_size) * int(self.mesh["sp"])
        self.model = C.load_checkpoint(C.build_model(cfg, True, C.torch_dtype(F.dtype, inference=True), F.seed, self.dev),
                                       F.load_checkpoint)
        self.gen = torch.Generator(device=self.dev).manual_seed(F.seed)

    def _read_process_vision(self, path, max_n_frames):
        """frames -> [256 codes, 8192] per frame, 8193 after the last (lwm/vision_chat.py:76-108)."""
        vision = read_frames(path, max_n_frames)
        _, idx = self.vqgan.encode(torch.from_numpy(vision))             # all frames in one call (independent)
        enc = idx.reshape(len(vision), -1).cpu().numpy().astype(int)
        out = []
        for t in range(len(enc)):
            out.extend(enc[t].tolist())
            out.append(8193 if t == len(enc) - 1 else 8192)
        return out

    def construct_input(self, prompts, max_n_frames):
        max_len = max_n_frames * self.n_tokens_per_frame + self.min_buffer_size
        max_len = int(math.ceil(max_len / self.block_size) * self.block_size)
        tk = self.tokenizer
        vision_start, vision_end = tk.encode("<vision>"), tk.encode("</vision>")
        ids = np.zeros((len(prompts), max_len), dtype=np.int64)
        vms = np.zeros((len(prompts), max_len), dtype=bool)
        att = np.zeros((len(prompts), max_len), dtype=np.int32)
        for i, p in enumerate(prompts):
            vision = self._read_process_vision(p["input_path"], max_n_frames)
            text_1 = tk.encode(f"<s>You are a helpful assistant. USER: {p['question']}\n")
            tail = tk.encode(" ASSISTANT:")
            tokens = text_1 + vision_start + vision + vision_end + tail
            vm = [False] * (len(text_1) + len(vision_start)) + [True] * len(vision) + [False] * (len(vision_end) + len(tail))
            assert len(tokens) < max_len, (len(tokens), max_len)
            ids[i, -len(tokens):] = tokens                                   # left padding (:140-142)
            vms[i, -len(tokens):] = vm
            att[i, -len(tokens):] = 1
        return dict(input_ids=ids, vision_masks=vms, attention_mask=att)

    def __call__(self, prompts, max_n_frames, max_new_tokens=None):
        b = self.construct_input(prompts, max_n_frames)
        t = lambda a, dt: torch.from_numpy(a).to(self.dev, dt)
        new = self.model.generate(t(b["input_ids"], torch.int64), t(b["vision_masks"], torch.bool),
                                  t(b["attention_mask"], torch.int32),
                                  max_new_tokens=max_new_tokens or self.block_size,
                                  temperature=self.F.temperature, do_sample=True,
                                  eos_token_id=self.tokenizer.eos_token_id, pad_token_id=self.tokenizer.pad_token_id or 0,
                                  generator=self.gen)
        texts = []
        for text in self.tokenizer.batch_decode(new.cpu().numpy(), skip_special_tokens=True):
            eos = getattr(self.tokenizer, "eos_token", None)
            if eos and eos in text:
                text = text.split(eos, maxsplit=1)[0]
            texts.append(text)
        return texts


def main(argv=None, max_new_tokens=None):
    F = parse(DEFAULTS, GROUPS, argv, prog="lwm_amd.cli.vision_chat")
    assert F.prompt != ""                # lwm/vision_chat.py:236-237
    assert F.input_file != ""
    torch.manual_seed(F.seed)
    sampler = Sampler(F)
    output = sampler([{"input_path": F.input_file, "question": F.prompt}], F.max_n_frames, max_new_tokens)[0]
    print(f"Question: {F.prompt}\nAnswer: {output}")
    return output


if __name__ == "__main__":
    main(sys.argv[1:])
