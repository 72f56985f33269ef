# round 4: the twelve-wave rows kernel (diagnostic build, DVAE_ROWS=3) beside the product kernel: phase stamps, step time, parity tests
set -x
O=gpurun_out/r04; mkdir -p $O
export DVAE_LIB=$PWD/disentangled-vae_amd/libdvae_hip_diag.so
{
for R in 2 3; do
  echo "== DVAE_ROWS=$R (diagnostic library), M2 y513 B=8192 bf16x3"
  DVAE_ROWS=$R python tools/stamp_rows.py bf16x3 8192
  DVAE_ROWS=$R python tools/bench_short.py --no-extras
done
} > $O/rows3_stamps.txt 2>&1
cat $O/rows3_stamps.txt
python -m pytest tests/test_gpu_fused.py -q -x -m gpu -k "twelve_wave" > $O/t_rows3.log 2>&1; tail -5 $O/t_rows3.log
unset DVAE_LIB
python tools/bench_short.py --no-extras > $O/base2.txt 2>&1; cat $O/base2.txt
