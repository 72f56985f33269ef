cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/td.log 2>&1; tail -4 gpurun_out/td.log
bash tools/r03/collect_side.sh > gpurun_out/side.log 2>&1; tail -5 gpurun_out/side.log
DVAE_CFLAGS="-DR2_FINE" python disentangled-vae_amd/build.py --force > /dev/null 2>&1
DVAE_FSTAMPS=1 DVAE_COLD=1 python tools/stamp_rows.py bf16x3 8192 2>/dev/null | grep -v amdgpu
DVAE_FSTAMPS=1 python tools/stamp_rows.py bf16x3 8192 2>/dev/null | grep -v amdgpu
