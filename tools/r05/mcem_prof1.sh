#!/bin/bash
# GPU box: rocprofv3 kernel stats of ONE utterance (300 frames, 100 EM iterations, bf16x3) through the drop-in MCEM_M2.run() -> gpurun_out/r05/mcem_prof1
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r05/mcem_prof1; mkdir -p $OUT
cat > /tmp/mcem_one1.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import torch
import bench_mcem as bm
from packages.models import mcem
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
m, X, S, y = bm.make("M2", 1, 300, "cuda", prec)
em = mcem.MCEM_M2(niter=100, nsamples_E_step=10, burnin_E_step=30, nsamples_WF=25, burnin_WF=75)
em.precision = prec
bm.run(em, m, X, S, y, "cuda", 3)
t, cost = bm.run(em, m, X, S, y, "cuda", 100)
print("seconds per utterance", t)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 /tmp/mcem_one1.py ${1:-bf16x3} > $OUT/run.log 2>&1
tail -2 $OUT/run.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-150
cp $f $OUT/../mcem_single_kernel_stats.csv
find $OUT \( -name "*kernel_trace.csv" -o -name "*.db" \) -delete
