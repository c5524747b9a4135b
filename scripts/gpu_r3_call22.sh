# round 3, call 22: VQGAN frames per call (tile quantisation of the 512- / 768-channel layers): 32 vs 64 vs 128
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c22; rm -rf $O; mkdir -p $O
timeout 600 python - > $O/vqgan_frames.txt 2>&1 < /dev/null <<'PY'
import json, torch, bench
for frames in (32, 64, 128):
    r = bench.vqgan_leg(torch, frames=frames, reps=3, config4_frames=1020 if frames == 32 else 1024)
    print(frames, json.dumps({k: r[k] for k in ("encode_frames_per_s", "decode_frames_per_s", "encode_tflops", "decode_tflops")}),
          "cfg4", round(r["config4_tokenisation"]["seconds"], 3), "match", r.get("indices_match_oracle"), flush=True)
    torch.cuda.empty_cache()
PY
cat $O/vqgan_frames.txt | grep -v Warning
