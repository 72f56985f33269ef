# round 4: rocprofv3 kernel stats of the MCEM bench (25 utterances side by side + single utterance)
O=$PWD/gpurun_out/r04; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $O/mcem_prof2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mcem_prof2 -- python3 $R/tools/bench_mcem.py --no-cpu --batch 25 > $O/mcem_prof2.json 2> $O/mcem_prof2.err
cd $R
f=$(find $O/mcem_prof2 -name "*kernel_stats.csv" | head -1)
cp $f $O/mcem_kernel_stats2.csv
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/mcem_kernel_stats2.csv")))[:8]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
rm -rf $O/mcem_prof2
