#!/bin/bash
# GPU box: the exact-fp32 MCEM chain on 4-frame tiles (default for short chains) against 16-frame tiles (DVAE_MCEM_TILE=16), alternating
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for tile in 0 16; do
  DVAE_MCEM_TILE=$tile python tools/bench_mcem.py --no-cpu --batch 2 4 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('tile', '$tile' if '$tile' != '0' else 'auto', {p: dict(ms_per_utt=round(d[p]['seconds_per_utterance']*1e3,1), mh_us=round(d[p]['mh_iteration_us'],2)) for p in ('fp32','bf16x3')},
      {k: round(v['utterances_per_s'],1) for k,v in d['batched'].items() if k.startswith('fp32')}, flush=True)"
done; done
