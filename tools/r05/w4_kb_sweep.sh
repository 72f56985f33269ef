#!/bin/bash
# GPU box: class-sliced weight-gradient table under bf16x3 with the cost model's delivery term swept (clocks per 1 KB fragment and wave), against uniform
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-extras --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()}, flush=True)"; }
for r in 1 2; do
  one uniform
  for k in 70 110 150 200 300; do DVAE_W4_CLASSES=1 DVAE_W4_KB_CLOCKS=$k one classes_kb$k; done
done
