#!/bin/bash
# GPU box: what the two-launch weight-gradient pass (DVAE_EXCHANGE_GROUPS=2) costs on ONE GPU against the one launch (the part of the multi-GPU
# step that can be measured here; the exchange it is meant to hide cannot)
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-extras --steps 300 --warmup 30 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', '$2', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()}, flush=True)"; }
for r in 1 2 3; do
  one one-launch ""
  DVAE_EXCHANGE_GROUPS=2 one two-launches ""
done
one one-launch "--model M2_info"
DVAE_EXCHANGE_GROUPS=2 one two-launches "--model M2_info"
