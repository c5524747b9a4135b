"""`python -m lwm_amd.cli.vision_generation` -- the flag set of lwm/vision_generation.py:21-41
(scripts/run_sample_image.sh, scripts/run_sample_video.sh): classifier-free-guided sampling of VQGAN
codes through the cached-decode hot path, then VQGAN decode (HIP) to pixels."""
from __future__ import annotations

import sys

import numpy as np
import torch

from . import _common as C
from ._flags import parse

DEFAULTS = dict(prompt="Fireworks over the city", output_file="", temperature_image=1.0, temperature_video=1.0,
                top_k_image=8192, top_k_video=100, cfg_scale_image=1.0, cfg_scale_video=1.0, vqgan_checkpoint="",
                n_frames=1, seed=1234, mesh_dim="1,-1,1,1", dtype="bf16", load_llama_config="", update_llama_config="",
                load_checkpoint="", tokenizer="LargeWorldModel/LWM-Text-1M")
GROUPS = ("llama", "jax_distributed")
TOKENS_PER_FRAME = 257


def _left_pad(tok, prompts, max_len, dev):
    """prefix_tokenizer(padding='max_length', truncation=True, padding_side/truncation_side='left')."""
    ids = np.zeros((len(prompts), max_len), np.int64)
    att = np.zeros((len(prompts), max_len), np.int32)
    for i, p in enumerate(prompts):
        t = tok.encode(p)[-max_len:]
        ids[i, -len(t):] = t
        att[i, -len(t):] = 1
    return torch.from_numpy(ids).to(dev), torch.from_numpy(att).to(dev)


def main(argv=None):
    F = parse(DEFAULTS, GROUPS, argv, prog="lwm_amd.cli.vision_generation")
    assert F.output_file != ""                                            # lwm/vision_generation.py:45-51
    if F.output_file.endswith("mp4") or F.output_file.endswith("npy"):
        assert F.n_frames > 1 or F.output_file.endswith("npy")
    elif F.output_file.endswith("png") or F.output_file.endswith("jpg"):
        assert F.n_frames == 1
    else:
        raise ValueError(f"Unsupported output file extension: {F.output_file}")
    C.setup_mesh(F.mesh_dim)
    if not torch.cuda.is_available():
        raise SystemExit("lwm_amd.cli.vision_generation needs an MI355X (the hot path has no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(F.seed)
    vqgan = C.load_vqgan(F.vqgan_checkpoint, F.seed)
    tok = C.load_tokenizer(F.tokenizer)
    cfg = C.build_config(F, vision=True)
    cfg.update(dict(bos_token_id=tok.bos_token_id, eos_token_id=tok.eos_token_id))
    model = C.load_checkpoint(C.build_model(cfg, True, C.torch_dtype(F.dtype, inference=True), F.seed, dev), F.load_checkpoint)
    gen = torch.Generator(device=dev).manual_seed(F.seed)

    def sample(prompts, images, n_tokens, cfg_scale, top_k, temperature, max_input_length):
        """generate_first_frame / generate_video_pred (:139-165, :186-227): conditional + unconditional halves."""
        ids, att = _left_pad(tok, prompts + ["<s><vision>"] * len(prompts), max_input_length, dev)
        vm = torch.zeros_like(ids, dtype=torch.bool)
        if images is not None:
            img = torch.as_tensor(np.concatenate([images, images], 0)).to(dev)
            ids = torch.cat([ids, img], 1)
            att = torch.cat([att, torch.ones_like(img, dtype=att.dtype)], 1)
            vm = torch.cat([vm, torch.ones_like(img, dtype=torch.bool)], 1)
        return model.generate_vision(ids, [cfg_scale] * len(prompts), attention_mask=att, vision_masks=vm,
                                     max_new_tokens=n_tokens, temperature=temperature, top_k=top_k,
                                     generator=gen).cpu().numpy()

    def to_u8(pix):
        return ((pix.cpu().numpy() + 1) * 127.5).astype(np.uint8)

    prompt = f"<s>You are a helpful assistant. USER: Generate an image of {F.prompt} ASSISTANT: <vision>"
    first = sample([prompt], None, TOKENS_PER_FRAME, F.cfg_scale_image, F.top_k_image, F.temperature_image, 128)
    first = first.reshape(1, TOKENS_PER_FRAME)
    image = to_u8(vqgan.decode(np.clip(first[:, :-1], 0, 8191).reshape(-1, 16, 16)))[0]
    if F.n_frames == 1:
        if F.output_file.endswith("npy"):
            np.save(F.output_file, image[None])
        else:
            from PIL import Image
            Image.fromarray(image).save(F.output_file)
        return image[None]
    vprompt = f"<s>You are a helpful assistant. USER: Generate a video of {F.prompt} ASSISTANT: <vision>"
    rest = sample([vprompt], first, (F.n_frames - 1) * TOKENS_PER_FRAME, F.cfg_scale_video, F.top_k_video,
                  F.temperature_video, 128)
    rest = rest.reshape(1, F.n_frames - 1, TOKENS_PER_FRAME)
    codes = np.concatenate([first[:, None], rest], 1)[:, :, :-1].reshape(-1, F.n_frames, 16, 16)
    video = to_u8(vqgan.decode(np.clip(codes[0], 0, 8191)))
    if F.output_file.endswith("npy"):
        np.save(F.output_file, video)
    else:
        raise SystemExit("writing mp4 needs imageio/ffmpeg, which this image does not have: use an .npy output file "
                         "(frames (T,256,256,3) uint8)")
    return video


if __name__ == "__main__":
    main(sys.argv[1:])
