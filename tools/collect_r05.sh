#!/bin/bash
# Round-5 measurement set (run on the GPU box, final code of the round).  Outputs land in gpurun_out/r05c/ and are copied into profiles/r05_*.
#   part 1 (no argument): bench lines of every SURVEY 8d configuration (incl. the module path, the self-launched 2-rank rehearsal), STFT / MCEM side benches, phase stamps
#   part 2 (argument "pmc"): rocprofv3 kernel stats + counter passes (incl. the per-class instruction split) of the headline configuration and of M2_info, side kernels
O=gpurun_out/r05c; mkdir -p $O
if [ "$1" != "pmc" ]; then
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --model M1 --no-extras > $O/bench_M1.json 2>/dev/null
python bench.py --model M2 --y-dim 1 --no-extras > $O/bench_M2_y1.json 2>/dev/null
python bench.py --model M2_info --no-extras > $O/bench_M2_info.json 2>/dev/null
python bench.py --precision fp32 --no-extras --no-cpu-baseline > $O/bench_fp32.json 2>/dev/null
python bench.py --batch 1048576 --steps 20 --warmup 5 --pool-gb 8 --no-extras --no-cpu-baseline > $O/bench_B1048576.json 2>/dev/null
python bench.py --batch 65536 --steps 50 --warmup 10 --pool-gb 4 --no-extras --no-cpu-baseline > $O/bench_B65536.json 2>/dev/null
python bench.py --impl modules --steps 1000 --warmup 200 --no-extras --no-cpu-baseline > $O/bench_modules_B8192.json 2>/dev/null
python bench.py --impl modules --batch 128 --steps 1000 --warmup 200 --no-extras --no-cpu-baseline > $O/bench_modules_B128.json 2>/dev/null
python bench.py --impl modules --model M2_info --steps 300 --warmup 50 --no-extras --no-cpu-baseline > $O/bench_modules_M2_info.json 2>/dev/null
python tools/bench_stft.py > $O/bench_stft.json 2>/dev/null
python tools/bench_mcem.py --batch 8 25 > $O/bench_mcem.json 2>/dev/null
# the N > 1 line as the driver would ask for it (no launcher in the environment): bench.py starts the ranks itself; gloo lets the two ranks share the one GPU
DVAE_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --batch 4096 --no-extras --no-cpu-baseline > $O/bench_2rank_selflaunch_gloo_rehearsal.json 2> $O/bench_2rank_selflaunch.err
timeout -k 10 120 python bench.py --gpus 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_2rank_rccl_on_one_gpu.out 2> $O/bench_2rank_rccl_on_one_gpu.err; echo "exit code $?" >> $O/bench_2rank_rccl_on_one_gpu.err
python tools/stamp_rows.py bf16x3 8192 > $O/stamps_bf16x3.txt 2>/dev/null
else
tools/pmc_collect.sh x3 --precision bf16x3 > $O/pmc_x3.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_x3 $O/pmc_M2_y513_B8192_bf16x3.json M2 513 8192 bf16x3 > $O/pmc_x3_summary.txt
cp gpurun_out/pmc_x3/stats/*/*kernel_stats.csv $O/kernel_stats_M2_y513_B8192_bf16x3.csv
tools/pmc_collect.sh info --model M2_info > $O/pmc_info.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_info $O/pmc_M2_info_B8192_bf16x3.json M2_info 1 8192 bf16x3 > $O/pmc_info_summary.txt
cp gpurun_out/pmc_info/stats/*/*kernel_stats.csv $O/kernel_stats_M2_info_B8192_bf16x3.csv
bash tools/r05/collect_side.sh > $O/side.log 2>&1
rm -rf gpurun_out/pmc_x3 gpurun_out/pmc_info
fi
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d["ms_per_step"]*1e3,1), "us/step", round(d["value"]/1e6,2), "Mf/s", d["dtype"], (d.get("roofline") or {}).get("avg_us"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
