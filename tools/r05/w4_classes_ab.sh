#!/bin/bash
# GPU box: class-sliced weight-gradient schedule (default) against the uniform one (DVAE_W4_UNIFORM=1), alternating, several configurations
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-extras --steps 300 --warmup 30 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', '$2', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()}, flush=True)"; }
for r in 1 2 3; do
  one classes ""
  DVAE_W4_UNIFORM=1 one uniform ""
done
for cfg in "--model M1" "--model M2_info" "--y-dim 1" "--precision fp32" "--precision bf16" "--batch 65536 --steps 50" "--batch 262144 --steps 20"; do
  one classes "$cfg"
  DVAE_W4_UNIFORM=1 one uniform "$cfg"
done
