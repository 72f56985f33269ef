# HBM traffic of the step kernels at a large batch (rocprofv3 FETCH_SIZE / WRITE_SIZE, own passes)
O=$PWD/gpurun_out/r03/bigb; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --batch ${1:-262144} --steps 6 --warmup 2 --prewarm-ms 0 --pool-gb 2 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B > $O/write.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
O="gpurun_out/r03/bigb"
for tag,key in (("fetch","FETCH_SIZE"),("write","WRITE_SIZE")):
    acc=collections.defaultdict(list)
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==key: acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        if any(s in k for s in ("rows","wgrad","apply")):
            print(tag, k, "launches", len(v), "MB/launch", round((2 if key=="FETCH_SIZE" else 1)*1024*sum(v)/len(v)/1e6,1))
dur=collections.defaultdict(list)
for f in glob.glob(f"{O}/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-3)
for k,v in dur.items():
    if any(s in k for s in ("rows","wgrad","apply")): print("us", k, len(v), round(sum(v)/len(v),1))
PY
