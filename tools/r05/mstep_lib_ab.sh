#!/bin/bash
# usage (GPU box): tools/r05/mstep_lib_ab.sh <variant> ...  -- the M-step launches at 25 utterances x 300 frames (tools/exp_mstep.py) on the product library and variants
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in base "$@"; do
  lib=$PWD/disentangled-vae_amd/build/variants/$v.so; [ "$v" = base ] && lib=$PWD/disentangled-vae_amd/libdvae_hip.so
  echo -n "$v: "; DVAE_LIB=$lib python tools/exp_mstep.py 2>/dev/null | tail -1
done; done
