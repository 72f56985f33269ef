#!/bin/bash
# GPU box: which change of the weight-gradient launch costs / gains what: committed kernel (head), item table with / without the BIAS = false bodies,
# uniform / class-sliced schedule -- alternating, 2 rounds
cd $GRAFT_REPO_ROOT
one() { lib=$PWD/disentangled-vae_amd/build/variants/$1.so; [ "$1" = base ] && lib=$PWD/disentangled-vae_amd/libdvae_hip.so
  DVAE_LIB=$lib python bench.py --no-extras --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', '$2', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()}, flush=True)"; }
for r in 1 2; do
  one head committed
  DVAE_W4_UNIFORM=1 one base uniform
  one base classes
  DVAE_W4_UNIFORM=1 one allbias uniform
  one allbias classes
done
