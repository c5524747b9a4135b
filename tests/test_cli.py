"""The entry points keep the reference's command lines (SURVEY.md section 8b "CLI"): every flag that the
reference's own launch scripts pass -- scripts/run_train_text.sh, run_train_vision_text.sh,
run_vision_chat.sh, run_sample_image.sh, run_sample_video.sh, restated here argument for argument --
must parse, land in the right place, and drive the same configuration objects."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TRAIN_TEXT = [  # scripts/run_train_text.sh:19-55
    "--modality=text", "--mesh_dim=!1,-1,2,2", "--dtype=fp32", "--total_steps=200", "--log_freq=1",
    "--save_model_freq=0", "--save_milestone_freq=10", "--load_llama_config=debug",
    "--update_llama_config=dict(theta=10000,max_sequence_length=2048,scan_attention=True,scan_query_chunk_size=256,"
    "scan_key_chunk_size=512,scan_mlp=True,scan_mlp_chunk_size=1024,scan_layers=True)",
    "--tokenizer=LargeWorldModel/LWM-Text-1M", "--optimizer.type=adamw", "--optimizer.accumulate_gradient_steps=1",
    "--optimizer.adamw_optimizer.weight_decay=0.1", "--optimizer.adamw_optimizer.lr=8e-5",
    "--optimizer.adamw_optimizer.end_lr=8e-5", "--optimizer.adamw_optimizer.lr_warmup_steps=5",
    "--optimizer.adamw_optimizer.lr_decay_steps=200", "--use_data_sharded_loader=True", "--train_dataset.type=json",
    "--train_dataset.text_processor.fields=text", "--train_dataset.json_dataset.path=",
    "--train_dataset.json_dataset.seq_length=2048", "--train_dataset.json_dataset.batch_size=1024",
    "--train_dataset.json_dataset.tokenizer_processes=16", "--train_dataset.json_dataset.use_data_sharded_loader=True",
    "--checkpointer.save_optimizer_state=True", "--autoresume=False", "--logger.append_uuid=False",
    "--logger.online=False", "--logger.project_id=lwm", "--logger.experiment_id=example-text-train",
    "--logger.experiment_note=", "--logger.output_dir=", "--logger.wandb_dir=/root/experiment_output/lwm"]

TRAIN_VISION = [  # scripts/run_train_vision_text.sh:19-60 (the flags that differ)
    "--modality=vision,text", "--mesh_dim=!1,-1,2,2", "--load_llama_config=debug",
    "--update_llama_config=dict(theta=50000000,max_sequence_length=2048,scan_attention=True,scan_query_chunk_size=512,"
    "scan_key_chunk_size=1024,scan_mlp=True,scan_mlp_chunk_size=8192,scan_layers=True)",
    "--train_dataset.type=json_vision", "--train_dataset.vision_text_processor.fields_from_example=fields",
    "--train_dataset.vision_text_processor.max_n_frames=4", "--train_dataset.json_vision_dataset.mode=no_pad",
    "--train_dataset.json_vision_dataset.seq_length=2048", "--train_dataset.json_vision_dataset.batch_size=8",
    "--train_dataset.json_vision_dataset.tokenizer_parallel_chunk_size=2"]

VISION_CHAT = [  # scripts/run_vision_chat.sh:16-28
    "--prompt=What is the video about?", "--input_file=clip.npy", "--vqgan_checkpoint=", "--mesh_dim=!1,1,-1,1",
    "--dtype=fp32", "--load_llama_config=7b", "--max_n_frames=8",
    "--update_llama_config=dict(sample_mode='text',theta=50000000,max_sequence_length=131072,scan_attention=False,"
    "scan_query_chunk_size=128,scan_key_chunk_size=128,scan_mlp=False,scan_mlp_chunk_size=2048,scan_layers=True)",
    "--load_checkpoint=params::/ckpt/params", "--tokenizer=LargeWorldModel/LWM-Text-1M"]

SAMPLE_VIDEO = [  # scripts/run_sample_video.sh:15-33 (run_sample_image.sh is the n_frames=1 subset)
    "--prompt=Fireworks over the city", "--output_file=fireworks.mp4", "--temperature_image=1.0",
    "--temperature_video=1.0", "--top_k_image=8192", "--top_k_video=1000", "--cfg_scale_image=5.0",
    "--cfg_scale_video=1.0", "--vqgan_checkpoint=", "--n_frames=8", "--mesh_dim=!1,1,-1,1", "--dtype=fp32",
    "--load_llama_config=7b",
    "--update_llama_config=dict(sample_mode='vision',theta=50000000,max_sequence_length=32768,scan_attention=False,"
    "scan_query_chunk_size=128,scan_key_chunk_size=128,scan_mlp=False,scan_mlp_chunk_size=8192,scan_layers=True)",
    "--load_checkpoint=params::/ckpt/params", "--tokenizer=LargeWorldModel/LWM-Text-1M"]


def _script_flags(name):
    """Flag NAMES the reference's launch script passes (read from /root/reference when it is there)."""
    import re
    path = os.path.join("/root/reference/scripts", name)
    if not os.path.exists(path):
        return None
    return set(re.findall(r"^\s+--([A-Za-z_.]+)=", open(path).read(), flags=re.M))


def test_every_flag_of_the_reference_scripts_parses():
    from lwm_amd.cli import train, vision_chat, vision_generation
    from lwm_amd.cli._flags import parse
    from lwm_amd.cli._common import build_config
    F = parse(train.DEFAULTS, train.GROUPS, TRAIN_TEXT, "train")
    assert F.modality == "text" and F.total_steps == 200 and F.autoresume is False and F.log_freq == 1
    assert F.optimizer["adamw_optimizer"] == dict(weight_decay=0.1, lr=8e-5, end_lr=8e-5, lr_warmup_steps=5, lr_decay_steps=200)
    assert F.train_dataset["json_dataset"]["seq_length"] == 2048 and F.train_dataset["text_processor"]["fields"] == "text"
    assert train._shape_from_dataset(F.train_dataset, False) == (1024, 2048)
    cfg = build_config(F, vision=False)
    assert (cfg.hidden_size, cfg.num_hidden_layers, cfg.theta, cfg.max_sequence_length) == (256, 2, 10000, 2048)
    assert cfg.scan_query_chunk_size == 256 and cfg.scan_layers is True and cfg.mesh_dim == "!1,-1,2,2"
    # schedule: linear warm-up to lr over 5 steps, then flat (end_lr == lr)
    opt = F.optimizer["adamw_optimizer"]
    assert train.lr_at(0, opt) == 0.0 and abs(train.lr_at(5, opt) - 8e-5) < 1e-12 and abs(train.lr_at(150, opt) - 8e-5) < 1e-12

    F = parse(train.DEFAULTS, train.GROUPS, TRAIN_VISION, "train")
    assert F.modality == "vision,text" and train._shape_from_dataset(F.train_dataset, True) == (8, 2048)
    cfg = build_config(F, vision=True)
    assert cfg.vision_vocab_size == 8448 and cfg.theta == 50000000 and cfg.scan_mlp_chunk_size == 8192

    F = parse(vision_chat.DEFAULTS, vision_chat.GROUPS, VISION_CHAT, "vision_chat")
    assert F.prompt == "What is the video about?" and F.max_n_frames == 8 and F.temperature == 0.2
    cfg = build_config(F, vision=True)
    assert cfg.sample_mode == "text" and cfg.max_sequence_length == 131072 and cfg.hidden_size == 4096

    F = parse(vision_generation.DEFAULTS, vision_generation.GROUPS, SAMPLE_VIDEO, "vision_generation")
    assert F.n_frames == 8 and F.top_k_video == 1000 and F.cfg_scale_image == 5.0 and F.output_file == "fireworks.mp4"
    assert build_config(F, vision=True).sample_mode == "vision"

    # the flag NAMES of the scripts themselves (when the reference tree is at hand) are all known
    for script, mod in (("run_train_text.sh", train), ("run_train_vision_text.sh", train),
                        ("run_vision_chat.sh", vision_chat), ("run_sample_image.sh", vision_generation),
                        ("run_sample_video.sh", vision_generation)):
        names = _script_flags(script)
        if names is None:
            continue
        for n in names:
            assert n in mod.DEFAULTS or n.split(".")[0] in mod.GROUPS, (script, n)

    with pytest.raises(SystemExit):
        parse(train.DEFAULTS, train.GROUPS, ["--no_such_flag=1"], "train")
    # absl forms: `--flag value`, bare booleans, --noflag
    F = parse(train.DEFAULTS, train.GROUPS, ["--seed", "7", "--autoresume", "--nouse_data_sharded_loader"], "train")
    assert F.seed == 7 and F.autoresume is True and F.use_data_sharded_loader is False


def test_mesh_dim_strings_of_the_scripts():
    from lwm_amd.mesh import parse_mesh_dim
    assert parse_mesh_dim("!1,-1,2,2", 8) == dict(dp=1, fsdp=2, tp=2, sp=2)
    assert parse_mesh_dim("!1,1,-1,1", 1) == dict(dp=1, fsdp=1, tp=1, sp=1)
    assert parse_mesh_dim("1,-1,1,1", 1)["fsdp"] == 1


def test_byte_tokenizer_round_trip():
    from lwm_amd.cli._common import ByteTokenizer
    t = ByteTokenizer()
    ids = t.encode("<s>You are. USER: hé\n<vision></vision> ASSISTANT:")
    assert ids[0] == 1 and 259 in ids and 260 in ids
    assert t.decode(ids) == "You are. USER: hé\n ASSISTANT:"


@pytest.mark.gpu
def test_one_step_of_each_entry_point_on_the_debug_model(tmp_path):
    import numpy as np
    from lwm_amd.cli import train, vision_chat, vision_generation
    small = ["--load_llama_config=debug", "--mesh_dim=1,-1,1,1", "--dtype=bf16", "--tokenizer=synthetic"]
    hist = train.main(small + ["--modality=text", "--total_steps=2", "--log_freq=1",
                               "--update_llama_config=dict(theta=10000,max_sequence_length=2048,scan_query_chunk_size=256)",
                               "--train_dataset.json_dataset.seq_length=1024", "--train_dataset.json_dataset.batch_size=2",
                               "--optimizer.adamw_optimizer.lr=1e-3", "--optimizer.adamw_optimizer.lr_warmup_steps=1"])
    assert len(hist) == 2 and all(np.isfinite(h["loss"]) for h in hist) and 8.0 < hist[0]["loss"] < 13.0
    hist = train.main(small + ["--modality=vision,text", "--total_steps=1",
                               "--train_dataset.type=json_vision", "--train_dataset.json_vision_dataset.seq_length=512",
                               "--train_dataset.json_vision_dataset.batch_size=1"])
    assert np.isfinite(hist[0]["loss"]) and "vision_loss" in hist[0]
    # vision chat: 2 synthetic frames -> VQGAN codes -> prompt -> 4 sampled tokens
    ans = vision_chat.main(small + ["--prompt=What is the video about?", "--input_file=synthetic:2", "--max_n_frames=2",
                                    "--update_llama_config=dict(sample_mode='text',max_sequence_length=2048,vocab_size=32000)"],
                           max_new_tokens=4)
    assert isinstance(ans, str)
    out = str(tmp_path / "img.npy")
    img = vision_generation.main(small + ["--prompt=Fireworks", f"--output_file={out}", "--n_frames=1", "--top_k_image=50",
                                          "--cfg_scale_image=5.0",
                                          "--update_llama_config=dict(sample_mode='vision',max_sequence_length=2048)"])
    assert img.shape == (1, 256, 256, 3) and img.dtype == np.uint8 and np.load(out).shape == (1, 256, 256, 3)


def test_dtype_flag_refuses_what_the_kernels_do_not_compute():
    """The reference's launchers pass --dtype=fp32 (scripts/run_train_text.sh:21, lwm/train.py:36): it runs as written
    (the f32 flavour of every kernel: csrc/attn_f32.h, csrc/elem_f32.h), in training and -- on one rank -- in cached
    inference; fp16 has no path and is refused, loudly, instead of being run as something else."""
    from lwm_amd.cli._common import torch_dtype
    import torch
    assert torch_dtype("bf16") is torch.bfloat16 and torch_dtype("bfloat16") is torch.bfloat16
    assert torch_dtype("fp32") is torch.float32 and torch_dtype("float32") is torch.float32
    assert torch_dtype("bf16", inference=True) is torch.bfloat16 and torch_dtype("fp32", inference=True) is torch.float32
    for kw in ({}, dict(inference=True)):
        with pytest.raises(SystemExit) as e:
            torch_dtype("fp16", **kw)
        # the message names what does run, and the launcher lines it is about
        assert "--dtype='bf16'" in str(e.value) and "run_train_text.sh:21" in str(e.value)
    with pytest.raises(SystemExit):
        torch_dtype("int8")
    # ... and an invocation WITHOUT --dtype must run: the entry points default to the headline dtype
    from lwm_amd.cli import train, vision_chat, vision_generation
    for mod in (train, vision_chat, vision_generation):
        assert torch_dtype(mod.DEFAULTS["dtype"], inference=mod is not train) is torch.bfloat16, mod.__name__


def test_vision_checkpoint_round_trip_through_load_checkpoint(tmp_path):
    """cli.vision_chat / vision_generation / train --modality=vision,text load FlaxVideoLLaMA checkpoints: besides the
    text model's leaves those carry `transformer/vte/embedding` and `vision_head/kernel` (lwm/vision_llama.py:264-270,
    :354-360).  A tiny scan_layers stream written in the reference's layout must land in every parameter of the
    harness model, and a checkpoint that lacks some must say so."""
    import torch
    from lwm_amd import weights as W
    from lwm_amd.cli import _common
    from lwm_amd.vision_llama import VideoLLaMAConfig, VideoLLaMAForCausalLM
    cfg = VideoLLaMAConfig(vocab_size=64, vision_vocab_size=48, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                           num_attention_heads=1, max_sequence_length=256)
    torch.manual_seed(0)
    src = VideoLLaMAForCausalLM(cfg, torch.float32)
    own = dict(src.named_parameters())
    flat = {"params/params/transformer/wte/embedding": own["wte"].detach(),
            "params/params/transformer/vte/embedding": own["vte"].detach(),
            "params/params/transformer/ln_f/kernel": own["ln_f.kernel"].detach(),
            "params/params/lm_head/kernel": own["lm_head"].detach(),
            "params/params/vision_head/kernel": own["vision_head"].detach()}
    for part, names in (("attention", ("wq", "wk", "wv", "wo")), ("feed_forward", ("w1", "w2", "w3"))):
        for n in names:
            flat[f"params/params/transformer/h/scan_decoder/{part}/{n}/kernel"] = torch.stack(
                [own[f"h.{i}.{part}.{n}"].detach() for i in range(2)])
    for n in ("attention_norm", "ffn_norm"):
        flat[f"params/params/transformer/h/scan_decoder/{n}/kernel"] = torch.stack(
            [own[f"h.{i}.{n}.kernel"].detach() for i in range(2)])
    path = str(tmp_path / "params")
    W.write_flax_stream(path, flat)
    torch.manual_seed(1)
    dst = VideoLLaMAForCausalLM(cfg, torch.float32)
    notes = []
    old = _common.note
    _common.note = notes.append
    try:
        _common.load_checkpoint(dst, f"params::{path}")
        assert not notes, notes                          # nothing missing, nothing unused
        for n, p in dst.named_parameters():
            assert torch.equal(p, own[n]), n
        # a text-only checkpoint into the vision model: loads, and reports what stayed at its initial value
        text_only = {k: v for k, v in flat.items() if "vte" not in k and "vision_head" not in k}
        W.write_flax_stream(path, text_only)
        _common.load_checkpoint(VideoLLaMAForCausalLM(cfg, torch.float32), f"params::{path}")
        assert len(notes) == 1 and "vte" in notes[0] and "vision_head" in notes[0]
    finally:
        _common.note = old
