# round 5: SQ counters of the STFT / ISTFT walk kernels (own passes, --kernel-trace only)
O=$PWD/gpurun_out/r05; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $O/stft_sq $O/stft_sq2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/stft_sq -- python3 $R/tools/bench_stft.py > $O/stft_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/stft_sq2 -- python3 $R/tools/bench_stft.py > $O/stft_sq2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.getcwd(), "gpurun_out/r05")
for d in ("stft_sq", "stft_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void dvae::", "")
            if "stft" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(d, k[:60], {n: round(max(v)) for n, v in c.items()})
PY
find $O/stft_sq $O/stft_sq2 \( -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*.db" \) -delete
