R=$GRAFT_REPO_ROOT; cd $R
for S in 4096 8192 16384 32768 65536; do
python bench.py --steps 1 --warmup 1 --layers 4 --seq $S --no-cpu-baseline --no-vqgan 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
S=d['config']['seq_len']; unit=S*S*4096.0
ex={'attn_fwd_kernel':2,'attn_bwd_dkdv_kernel':4,'attn_bwd_dq_kernel':3}
print(S, {k: (round(v['avg_ms'],3), round(ex[k]*unit/(v['avg_ms']*1e-3)/1e12) ) for k,v in d['kernels'].items() if k in ex})"
done
