cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_fused.py -x -q -k "direct_exchange or two_rank" > gpurun_out/tg.log 2>&1; tail -6 gpurun_out/tg.log
# 2-rank rehearsal of bench.py on the one GPU over gloo, both exchanges (NOT a scaling measurement)
for ex in rccl direct; do
DVAE_ALLREDUCE=$ex DVAE_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 50 --warmup 10 --batch 4096 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$ex', round(d['ms_per_step']*1e3,1), 'us/step', d['multi_gpu'])"
done
