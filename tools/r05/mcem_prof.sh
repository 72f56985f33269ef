#!/bin/bash
# GPU box: rocprofv3 kernel stats of the batched MCEM run (25 utterances x 300 frames, 100 EM iterations, bf16x3) -> gpurun_out/r05/mcem_prof
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r05/mcem_prof; mkdir -p $OUT
cat > /tmp/mcem_one.py <<'PY'
import sys, os, time, importlib
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import torch
import bench_mcem as bm
dev = importlib.import_module("disentangled-vae_amd.mcem")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
m, X, S, y = bm.make("M2", 1, 300, "cuda", prec)
mb = dev.McemBatch(m, niter=2, precision=prec)
mb.init_parameters([X] * 25, [y] * 25); mb.run()
mb.niter = 100
mb.init_parameters([X] * 25, [y] * 25)
torch.cuda.synchronize(); t0 = time.perf_counter()
mb.run()
torch.cuda.synchronize(); print("seconds", time.perf_counter() - t0, "utt/s", 25 / (time.perf_counter() - t0))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 /tmp/mcem_one.py bf16x3 > $OUT/run.log 2>&1
cat $OUT/run.log | tail -2
f=$(find $OUT -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
