#!/bin/bash
# GPU box: MCEM bench (single utterance through the drop-in, 25 utterances batched) with / without the XCD-paired M-step frames kernel, and the
# drop-in's stepwise loop (DVAE_MCEM_RUN=steps) beside the one-call-per-iteration loop
cd $GRAFT_REPO_ROOT
summ='import json,sys
d=json.loads(sys.stdin.read())
print(sys.argv[1], {p:(round(d[p]["seconds_per_utterance"]*1e3,1), round(d[p]["e_step_us"]), round(d[p]["m_step_us"],1)) for p in ("fp32","bf16x3","bf16")}, {k:round(v["utterances_per_s"],1) for k,v in d["batched"].items()}, flush=True)'
for r in 1 2; do
  python tools/bench_mcem.py --no-cpu --batch 25 2>/dev/null | python -c "$summ" base
  DVAE_LIB=$PWD/disentangled-vae_amd/build/variants/mstep_nopairs.so python tools/bench_mcem.py --no-cpu --batch 25 2>/dev/null | python -c "$summ" nopairs
  DVAE_MCEM_RUN=steps python tools/bench_mcem.py --no-cpu --batch 25 2>/dev/null | python -c "$summ" steps
done
