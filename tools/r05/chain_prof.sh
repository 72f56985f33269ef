#!/bin/bash
# GPU box: rocprofv3 kernel trace of tools/r05/chain_time.py <utterances>: per chain kernel the durations of the 30 + 10 and the 75 + 25 launches
# (their difference over the step counts = time per pass; the rest = the launch's fixed part: weights, X2 / Vb tiles, label terms, copies)
U=${1:-25}
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r05/chain_prof; mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/r05/chain_time.py $U > $OUT/run.log 2>&1
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "mcem_resident" in n: d[n.split("(")[0][-60:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in d.items():
    v.sort()
    # launches: 3 warm-up + 20 timed of each chain length (+ the McemBatch.run() of the set-up)
    lo = [x for x in v if x < (v[0] + v[-1]) / 2]; hi = [x for x in v if x >= (v[0] + v[-1]) / 2]
    med = lambda a: a[len(a) // 2]
    print(n, "launches", len(v), "30+10 median us %.1f" % med(lo), "75+25 median us %.1f" % med(hi))
PY
find $OUT \( -name "*kernel_trace.csv" -o -name "*.db" \) -delete
