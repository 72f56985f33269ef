# round 3, experiment B: the restructured rows kernel -- parity tests first, then flag variants alternated on one box, then stamps
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fused.py tests/test_gpu_module_path.py tests/test_gpu_models.py tests/test_gpu_frames.py -x -q > gpurun_out/tb.log 2>&1; tail -5 gpurun_out/tb.log
ABN_ARGS="" bash tools/abn.sh "" "-DR2_HELPY=0" "-DR2_EARLY_Y=0" "-DR2_HPRIO=1" "-DR2_PDO_X3=8" "-DR2_PDO_X3=10" "-DR2_YSPREAD=17"
V=disentangled-vae_amd/build/variants
for j in 0 1 3; do echo "stamps v$j"; DVAE_LIB=$PWD/$V/v$j.so DVAE_HSTAMPS=1 DVAE_COLD=1 python tools/stamp_rows.py bf16x3 8192 2>/dev/null | grep -v amdgpu; done
