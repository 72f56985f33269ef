cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tests/diag/sc_check.py 2>&1 | grep -v amdgpu | tail -9
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/tw.log 2>&1; tail -3 gpurun_out/tw.log
ABN_ARGS="" bash tools/abn.sh "" "-DW4_SLAB_AUX=0" "-DR2_STASH_AUX=0 -DW4_SLAB_AUX=0"
