#!/bin/bash
# Round-4 measurement set (run on the GPU box, final code of the round).  Outputs land in gpurun_out/r04/ and are copied into profiles/r04_*.
#   part 1 (no argument): bench lines of every SURVEY 8d configuration (incl. the module path, 2-rank rehearsals), STFT / MCEM side benches, phase stamps
#   part 2 (argument "pmc"): rocprofv3 kernel stats + counter passes of the headline configuration and of M2_info, side kernels
O=gpurun_out/r04; mkdir -p $O
if [ "$1" != "pmc" ]; then
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --model M1 --no-extras > $O/bench_M1.json 2>/dev/null
python bench.py --model M2 --y-dim 1 --no-extras > $O/bench_M2_y1.json 2>/dev/null
python bench.py --model M2_info --no-extras > $O/bench_M2_info.json 2>/dev/null
python bench.py --batch 1048576 --steps 20 --warmup 5 --pool-gb 8 --no-extras --no-cpu-baseline > $O/bench_B1048576.json 2>/dev/null
python bench.py --batch 65536 --steps 50 --warmup 10 --pool-gb 4 --no-extras --no-cpu-baseline > $O/bench_B65536.json 2>/dev/null
python bench.py --impl modules --steps 1000 --warmup 200 --no-extras --no-cpu-baseline > $O/bench_modules_B8192.json 2>/dev/null
python bench.py --impl modules --batch 128 --steps 1000 --warmup 200 --no-extras --no-cpu-baseline > $O/bench_modules_B128.json 2>/dev/null
python bench.py --impl modules --model M2_info --steps 300 --warmup 50 --no-extras --no-cpu-baseline > $O/bench_modules_M2_info.json 2>/dev/null
python tools/bench_stft.py > $O/bench_stft.json 2>/dev/null
python tools/bench_mcem.py --batch 25 > $O/bench_mcem.json 2>/dev/null
for ex in gloo direct; do
  DVAE_DIST_BACKEND=gloo DVAE_ALLREDUCE=$([ $ex = direct ] && echo direct || echo rccl) timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 20 --warmup 5 --batch 4096 --no-extras --no-cpu-baseline > $O/bench_2rank_${ex}_rehearsal.json 2> $O/bench_2rank_${ex}.err
done
python tools/stamp_rows.py bf16x3 8192 > $O/stamps_bf16x3.txt 2>/dev/null
DVAE_COLD=1 DVAE_HSTAMPS=1 python tools/stamp_rows.py bf16x3 8192 > $O/stamps_bf16x3_cold.txt 2>/dev/null
else
tools/pmc_collect.sh x3 --precision bf16x3 > $O/pmc_x3.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_x3 $O/pmc_M2_y513_B8192_bf16x3.json M2 513 8192 bf16x3 > $O/pmc_x3_summary.txt
cp gpurun_out/pmc_x3/stats/*/*kernel_stats.csv $O/kernel_stats_M2_y513_B8192_bf16x3.csv
tools/pmc_collect.sh info --model M2_info > $O/pmc_info.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_info $O/pmc_M2_info_B8192_bf16x3.json M2_info 1 8192 bf16x3 > $O/pmc_info_summary.txt
cp gpurun_out/pmc_info/stats/*/*kernel_stats.csv $O/kernel_stats_M2_info_B8192_bf16x3.csv
bash tools/r04/collect_side.sh > $O/side.log 2>&1
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
