cd $GRAFT_REPO_ROOT
V=disentangled-vae_amd/build/variants; mkdir -p $V
i=0
for fl in "" "-DR2_ROLL=0"; do
  DVAE_CFLAGS="$fl" python disentangled-vae_amd/build.py --force > /dev/null 2>&1; cp disentangled-vae_amd/libdvae_hip.so $V/r$i.so; echo "r$i = '$fl'"; i=$((i+1))
done
python disentangled-vae_amd/build.py --force > /dev/null 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -3
for r in 1 2 3; do for j in 0 1; do for m in "" "--model M2_info"; do
  DVAE_LIB=$PWD/$V/r$j.so python bench.py --no-extras --steps 300 --warmup 30 --no-cpu-baseline $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('r$j $m', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()})"
done; done; done
DVAE_LIB=$PWD/$V/r0.so python tools/stamp_rows.py bf16x3 8192 2>/dev/null | head -3
DVAE_LIB=$PWD/$V/r1.so python tools/stamp_rows.py bf16x3 8192 2>/dev/null | head -3
