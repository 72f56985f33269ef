cd $GRAFT_REPO_ROOT
for v in 2 3 4; do for sl in 2048 3072 4096 6144 8192; do
  DVAE_ISTFT_SLOTS=$sl DVAE_LIB=$PWD/disentangled-vae_amd/build/variants/if32_occ$v.so python tools/bench_stft.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['600s_float32']
print('occ $v slots $sl', {k: round(d[k],1) for k in ('istft_us','istft_f32arith_us')}, flush=True)"
done; done
