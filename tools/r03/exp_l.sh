cd $GRAFT_REPO_ROOT
ABN_ARGS="" bash tools/abn.sh "" "-DR2_STAGGER=2" "-DR2_STAGGER=6" "-DR2_STAGGER=16"
V=disentangled-vae_amd/build/variants
for j in 0 1 2 3; do echo "stamps v$j"; DVAE_LIB=$PWD/$V/v$j.so python tools/stamp_rows.py bf16x3 8192 2>/dev/null | grep -v amdgpu; done
