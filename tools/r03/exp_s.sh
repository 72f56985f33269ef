cd $GRAFT_REPO_ROOT
for fl in "" "-DR2_STASH_SC=3" "-DR2_STASH_SC=0"; do
  DVAE_CFLAGS="$fl" python disentangled-vae_amd/build.py --force > /dev/null 2>&1
  echo "== flags '$fl'"; timeout -k 10 300 python tests/diag/sc_check.py 2>&1 | grep -v amdgpu | tail -12
done
