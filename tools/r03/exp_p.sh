cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py -x -q > gpurun_out/tp.log 2>&1; tail -3 gpurun_out/tp.log
ABN_ARGS="" bash tools/abn.sh "" "-DAPPLY_SC=1"
ABN_ARGS="--model M2_info" bash tools/abn.sh "" "-DR2_STASH_SC=0"
