cd $GRAFT_REPO_ROOT
DVAE_CFLAGS="-DR2_STASH_SC=1" python disentangled-vae_amd/build.py --force > /dev/null 2>&1
echo "== sc1 + s_nop"; timeout -k 10 300 python tests/diag/sc_check.py 2>&1 | grep -v amdgpu | tail -10
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -3
