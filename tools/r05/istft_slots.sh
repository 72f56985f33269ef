#!/bin/bash
# GPU box: ISTFT walk kernel with the frames dealt to 1024 ... 4096 waves (DVAE_ISTFT_SLOTS; 2048 = one round at two waves per SIMD): halo work against latency hiding
cd $GRAFT_REPO_ROOT
for r in 1 2; do for sl in 2048 1024 1536 3072 4096; do
  DVAE_ISTFT_SLOTS=$sl python tools/bench_stft.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('slots $sl', {c: round(d[c]['istft_us'],1) for c in ('5s_float64','600s_float64','600s_float32')}, flush=True)"
done; done
