# usage: exp_stamp.sh "<cflags>" ...  -- rebuild with the flags, print the rows-kernel phase stamps (bf16x3, 8192 frames)
set -e
cd $GRAFT_REPO_ROOT
for fl in "$@"; do
  DVAE_CFLAGS="$fl" python disentangled-vae_amd/build.py --force > /dev/null 2>&1
  echo "flags: $fl"; python tools/stamp_rows.py bf16x3 8192 2>/dev/null | grep -v amdgpu
done
