cd $GRAFT_REPO_ROOT
# optimizer step folded into the weight-gradient launch (default) against its own launch, alternating on one box
timeout -k 10 500 python -m pytest tests/test_gpu_fused.py -x -q -k "folded or deterministic or golden or reference_vectors" 2>&1 | tail -5
for r in 1 2 3; do for f in 1 0; do for m in "" "--model M2_info --y-dim 1"; do
  DVAE_FOLD_APPLY=$f python bench.py --no-extras --steps 300 --warmup 30 --no-cpu-baseline $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('fold=$f $m', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()})"
done; done; done
