#!/bin/bash
# usage (GPU box): tools/r05/mcem_lib_ab.sh <variant> ...  -- MCEM bench (one utterance through MCEM_M2.run(); 8 / 25 utterances side by side) on the product
# library and on prebuilt variants (build/variants/<name>.so), alternating, 2 rounds
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in base "$@"; do
  lib=$PWD/disentangled-vae_amd/build/variants/$v.so; [ "$v" = base ] && lib=$PWD/disentangled-vae_amd/libdvae_hip.so
  DVAE_LIB=$lib python tools/bench_mcem.py --no-cpu --batch 8 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', {p: dict(ms_per_utt=round(d[p]['seconds_per_utterance']*1e3,1), mh_us=round(d[p]['mh_iteration_us'],2)) for p in ('fp32','bf16x3','bf16')},
      {k: round(v['utterances_per_s'],1) for k,v in d['batched'].items()}, flush=True)"
done; done
