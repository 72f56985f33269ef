#!/bin/bash
# GPU box: the float32-arithmetic inverse transform (dvae_istft_f32) at 2 / 3 / 4 waves per SIMD (build/variants/if32_occ<n>.so from
# tools/r05/mkvariant.sh if32_occ<n> stft.hip -DISTFT_F32_OCC=<n>) next to the double-arithmetic walk, ten minutes of audio, alternating
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in 2 3 4; do
  DVAE_LIB=$PWD/disentangled-vae_amd/build/variants/if32_occ$v.so python tools/bench_stft.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['600s_float32']
print('occ $v', {k: round(d[k],1) for k in ('istft_us','istft_bin_major_us','istft_f32arith_us','istft_f32arith_bin_major_us','stft_f32arith_us')}, flush=True)"
done; done
