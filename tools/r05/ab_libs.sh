#!/bin/bash
# usage (GPU box): tools/r05/ab_libs.sh "<bench args>" <lib name> ...  -- alternate prebuilt variants (build/variants/<name>.so; "base" = the product library) on this box, 3 rounds
cd $GRAFT_REPO_ROOT
args="$1"; shift
for r in 1 2 3; do for v in "$@"; do
  lib=$PWD/disentangled-vae_amd/build/variants/$v.so; [ "$v" = base ] && lib=$PWD/disentangled-vae_amd/libdvae_hip.so
  DVAE_LIB=$lib python bench.py --no-extras --steps 300 --warmup 30 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()}, flush=True)"
done; done
