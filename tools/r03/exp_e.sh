cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fused.py tests/test_gpu_module_path.py -x -q > gpurun_out/te.log 2>&1; tail -3 gpurun_out/te.log
for r in 1 2 3; do python bench.py --no-extras --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('cur', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()}, 'ev', round(d['roofline']['event_pair_overhead_us'],2))"; done
DVAE_HSTAMPS=1 DVAE_COLD=1 python tools/stamp_rows.py bf16x3 8192 2>/dev/null | grep -v amdgpu
python bench.py > gpurun_out/bench_default_e.json 2> gpurun_out/bench_default_e.err; python -c "
import json; d=json.load(open('gpurun_out/bench_default_e.json')); print(d['ms_per_step'], d['value'], d['roofline']['avg_us'], d['roofline']['frac'], d.get('side_kernels'))"
