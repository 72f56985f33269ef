cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fused.py tests/test_gpu_module_path.py tests/test_gpu_models.py tests/test_gpu_frames.py -x -q > gpurun_out/tc.log 2>&1; tail -5 gpurun_out/tc.log
ABN_ARGS="" bash tools/abn.sh "" "-DR2_HELPY=0" "-DR2_YLOSEG=0" "-DR2_PDO_X3=8" "-DR2_PDO_X3=10" "-DR2_HELPY=0 -DR2_PDO_X3=10"
V=disentangled-vae_amd/build/variants
for j in 0 1; do echo "stamps v$j"; DVAE_LIB=$PWD/$V/v$j.so DVAE_HSTAMPS=1 DVAE_COLD=1 python tools/stamp_rows.py bf16x3 8192 2>/dev/null | grep -v amdgpu; done
for mode in return deposit; do DVAE_MODULE_GRADS=$mode python bench.py --impl modules --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('modules $mode', d['ms_per_step'], d['spread'])"; done
for mode in return deposit; do DVAE_MODULE_GRADS=$mode python bench.py --impl modules --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('modules $mode', d['ms_per_step'], d['spread'])"; done
