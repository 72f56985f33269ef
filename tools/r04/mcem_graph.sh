# round 4: EM iterations replayed from a HIP graph (McemBatch.run) beside the eager loop
set -x
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_gpu_mcem.py -q -x -m gpu > $O/t_mcem2.log 2>&1; tail -5 $O/t_mcem2.log
DVAE_MCEM_GRAPH=0 python tools/bench_mcem.py --no-cpu --batch 25 > $O/mcem_eager.json 2>$O/mcem_eager.err
python tools/bench_mcem.py --no-cpu --batch 25 > $O/mcem_graph.json 2>$O/mcem_graph.err
python - <<'PY'
import json
for n in ("eager","graph"):
    try:
        d=json.load(open(f"gpurun_out/r04/mcem_{n}.json"))
        print(n, {k:(round(v["utterances_per_s"],1), round(v["ms_per_em_iteration"],3)) for k,v in d["batched"].items()})
    except Exception as e: print(n, "unreadable", e)
PY
