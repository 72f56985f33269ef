# placement of the x-stash re-read (R2_XSP) and the split-fp16 image itself (R2_XF16=0: round-3 form): us per step, same box, alternating
mkdir -p gpurun_out/r04
for v in "" "-DR2_XSP=1" "-DR2_XSP=2" "-DR2_XF16=0"; do
  DVAE_CFLAGS="$v" python disentangled-vae_amd/build.py --force > /dev/null 2>&1
  for r in 1 2; do echo "[$v]" $(python tools/bench_short.py --no-extras); done
  echo "[$v] M2_info" $(python tools/bench_short.py --no-extras --model M2_info)
done 2>&1 | tee gpurun_out/r04/exp_xsp.txt
python disentangled-vae_amd/build.py --force > /dev/null 2>&1
