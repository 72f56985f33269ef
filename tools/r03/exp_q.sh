cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout -k 10 300 python -m pytest tests/test_gpu_fused.py -x -q -k "fp32_matches_reference_vectors" 2>&1 | tail -2; done
DVAE_CFLAGS="-DW4_SLAB_SC=0 -DR2_STASH_SC=0" python disentangled-vae_amd/build.py --force > /dev/null 2>&1
for i in 1 2; do timeout -k 10 300 python -m pytest tests/test_gpu_fused.py -x -q -k "fp32_matches_reference_vectors" 2>&1 | tail -2; done
