#!/bin/bash
# usage (GPU box): tools/r05/stft_lib_ab.sh <variant> ...  -- tools/bench_stft.py on the product library and prebuilt variants, alternating, 3 rounds
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in base "$@"; do
  lib=$PWD/disentangled-vae_amd/build/variants/$v.so; [ "$v" = base ] && lib=$PWD/disentangled-vae_amd/libdvae_hip.so
  DVAE_LIB=$lib python tools/bench_stft.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for c in ('5s_float64','600s_float64','600s_float32'):
    print('$v', c, {k:round(v,1) for k,v in d[c].items() if k.endswith('_us')}, flush=True)"
done; done
