#!/bin/bash
# Round-2 measurement set (run on the GPU box): bench lines for the SURVEY 8d configurations + rocprofv3 kernel stats / counters
# for the headline configuration under both bf16 policies.  Outputs land in gpurun_out/r02/ and are copied into profiles/ by hand.
set -e
O=gpurun_out/r02; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --model M1 --no-extras > $O/bench_M1.json 2>/dev/null
python bench.py --model M2 --y-dim 1 --no-extras > $O/bench_M2_y1.json 2>/dev/null
python bench.py --model M2_info --precision bf16x3 --no-extras > $O/bench_M2_info.json 2>/dev/null
python bench.py --batch 1048576 --steps 20 --warmup 5 --pool-gb 8 --no-extras --no-cpu-baseline > $O/bench_B1048576.json 2>/dev/null
python bench.py --batch 65536 --steps 50 --warmup 10 --pool-gb 4 --no-extras --no-cpu-baseline > $O/bench_B65536.json 2>/dev/null
python bench.py --precision bf16 --batch 1048576 --steps 20 --warmup 5 --pool-gb 8 --no-extras --no-cpu-baseline > $O/bench_bf16_B1048576.json 2>/dev/null
python bench.py --impl modules --steps 2000 --warmup 300 --no-extras --no-cpu-baseline > $O/bench_modules_B8192.json 2>/dev/null
python bench.py --impl modules --batch 128 --steps 2000 --warmup 300 --no-extras --no-cpu-baseline > $O/bench_modules_B128.json 2>/dev/null
DVAE_MODULE_PATH=layers python bench.py --impl modules --batch 128 --steps 500 --warmup 100 --no-extras --no-cpu-baseline > $O/bench_modules_layers_B128.json 2>/dev/null
tools/pmc_collect.sh x3 --precision bf16x3 > $O/pmc_x3.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_x3 $O/pmc_M2_y513_B8192_bf16x3.json M2 513 8192 bf16x3 > $O/pmc_x3_summary.txt
tools/pmc_collect.sh bf --precision bf16 > $O/pmc_bf.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_bf $O/pmc_M2_y513_B8192_bf16.json M2 513 8192 bf16 > $O/pmc_bf_summary.txt
cp gpurun_out/pmc_x3/stats/*/*kernel_stats.csv $O/kernel_stats_M2_y513_B8192_bf16x3.csv
cp gpurun_out/pmc_bf/stats/*/*kernel_stats.csv $O/kernel_stats_M2_y513_B8192_bf16.csv
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/mod_stats -- python3 $OLDPWD/bench.py --impl modules --batch 128 --steps 200 --warmup 50 --no-extras --no-cpu-baseline > $OLDPWD/$O/mod_stats.log 2>&1
cd $OLDPWD
cp $O/mod_stats/*/*kernel_stats.csv $O/kernel_stats_modules_path_B128.csv
python tools/stamp_rows.py bf16x3 8192 > $O/stamps_bf16x3.txt 2>/dev/null
python tools/stamp_rows.py bf16 8192 > $O/stamps_bf16.txt 2>/dev/null
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], round(d["ms_per_step"]*1e3,1), "us/step", round(d["value"]/1e6,2), "Mf/s", d["dtype"], (d.get("roofline") or {}).get("avg_us"))
PY
done
