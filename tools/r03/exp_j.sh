cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_module_path.py tests/test_gpu_fused.py tests/test_losszoo.py tests/test_gpu_stft.py -m gpu -x -q > gpurun_out/tj.log 2>&1; tail -3 gpurun_out/tj.log
for r in 1 2 3; do python bench.py --no-extras --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('cur', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()})"; done
V=disentangled-vae_amd/build/variants; mkdir -p $V
for occ in 3 2; do DVAE_CFLAGS="-DMCEM_OCC=$occ" python disentangled-vae_amd/build.py --force > /dev/null 2>&1; cp disentangled-vae_amd/libdvae_hip.so $V/m$occ.so; done
for r in 1 2; do for occ in 3 2; do DVAE_LIB=$PWD/$V/m$occ.so python tools/bench_mcem.py --no-cpu --batch 25 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('occ $occ', 'single fp32 utt/s', round(d['fp32']['utterances_per_s'],2), 'e-step us', round(d['fp32']['e_step_us'],1), 'batched', {k:round(v['utterances_per_s'],1) for k,v in d['batched'].items()})"; done; done
