# round 4: weight-stationary MH chain (mcem_resident.hip) beside the streaming kernel
set -x
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_mcem.py -q -x -m gpu > $O/t_mcem3.log 2>&1; tail -15 $O/t_mcem3.log
DVAE_MCEM_CHAIN=stream timeout -k 10 300 python tools/bench_mcem.py --no-cpu --batch 25 > $O/mcem_stream.json 2>$O/mcem_stream.err
timeout -k 10 300 python tools/bench_mcem.py --no-cpu --batch 25 > $O/mcem_res.json 2>$O/mcem_res.err
python - <<'PY'
import json
for n in ("stream","res"):
    try:
        d=json.load(open(f"gpurun_out/r04/mcem_{n}.json"))
        print(n, {k:(round(v["utterances_per_s"],1), round(v["ms_per_em_iteration"],3)) for k,v in d["batched"].items()}, "single x3:", round(d["bf16x3"]["e_step_us"],1), round(d["bf16x3"]["m_step_us"],1), round(d["bf16x3"]["utterances_per_s"],2))
    except Exception as e: print(n, "unreadable", e)
PY
