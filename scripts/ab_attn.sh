# A/B of attention kernel shapes: each variant = env settings, 1 step of 8 layers.
R=$GRAFT_REPO_ROOT; cd $R
run() { echo "== $*"; env "$@" python bench.py --steps 1 --warmup 1 --layers 8 --no-cpu-baseline --no-vqgan $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tok/s(32L-equiv) %.0f' % (d['value']*8/32), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"; }
for v in "$@"; do run $v; done
