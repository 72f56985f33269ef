cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py -x -q -k "info or M2_info or M2info" > gpurun_out/tf.log 2>&1; tail -5 gpurun_out/tf.log
for r in 1 2; do for rows in 2 1; do DVAE_ROWS=$rows timeout -k 10 300 python bench.py --model M2_info --no-extras --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('M2_info rows=$rows', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()}, d['config']['final_elbo'])"; done; done
