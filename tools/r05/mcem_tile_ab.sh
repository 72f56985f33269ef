#!/bin/bash
# GPU box: the MCEM chain on 16-frame tiles (default for chains whose 16-frame tiles fit the chip in one round) against 32-frame tiles
# (DVAE_MCEM_TILE=32), alternating: one utterance of 300 frames through the drop-in MCEM_M2.run(), and 8 / 25 utterances side by side
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for tile in 0 32; do
  DVAE_MCEM_TILE=$tile python tools/bench_mcem.py --no-cpu --batch 8 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('tile', '$tile' if '$tile' != '0' else 'auto', {p: dict(ms_per_utt=round(d[p]['seconds_per_utterance']*1e3,1), e_step_us=round(d[p]['e_step_us'],1), mh_us=round(d[p]['mh_iteration_us'],2), m_step_us=round(d[p]['m_step_us'],1)) for p in ('fp32','bf16x3','bf16')},
      {k: round(v['utterances_per_s'],1) for k,v in d['batched'].items()}, flush=True)"
done; done
