#!/bin/bash
# GPU box: the float32-arithmetic STFT at 4 / 3 / 2 waves per SIMD (prebuilt variants), alternating
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in base stft_occ3 stft_occ2; do
  lib=$PWD/disentangled-vae_amd/build/variants/$v.so; [ "$v" = base ] && lib=$PWD/disentangled-vae_amd/libdvae_hip.so
  DVAE_LIB=$lib python tools/bench_stft.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['600s_float32']
print('$v', {k:round(v,1) for k,v in d.items() if 'f32arith' in k or k in ('stft_us','stft_power_us','istft_us')}, flush=True)"
done; done
