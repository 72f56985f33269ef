cd $GRAFT_REPO_ROOT
DVAE_CFLAGS="-DR2_STASH_AUX=16" python disentangled-vae_amd/build.py --force > /dev/null 2>&1
echo "== aux 16 (sc1, buffer store)"; timeout -k 10 300 python tests/diag/sc_check.py 2>&1 | grep -v amdgpu | tail -10
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -3
ABN_ARGS="" bash tools/abn.sh "" "-DR2_STASH_AUX=16" "-DR2_STASH_AUX=17"
