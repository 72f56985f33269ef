# round 3, experiment A: EARLY_Y x RAW_INPUTS on the headline configuration (alternating on one box) + helper stamps
cd $GRAFT_REPO_ROOT
V=disentangled-vae_amd/build/variants; mkdir -p $V
DVAE_CFLAGS="" python disentangled-vae_amd/build.py --force > /dev/null 2>&1; cp disentangled-vae_amd/libdvae_hip.so $V/a0.so
DVAE_CFLAGS="-DR2_EARLY_Y=1" python disentangled-vae_amd/build.py --force > /dev/null 2>&1; cp disentangled-vae_amd/libdvae_hip.so $V/a1.so
run() {  # lib, raw
  if [ "$2" = "1" ]; then export DVAE_RAW_INPUTS=1; else unset DVAE_RAW_INPUTS; fi
  DVAE_LIB=$PWD/$V/$1.so python bench.py --no-extras --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 raw=$2', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()})"
}
for r in 1 2 3; do run a0 0; run a0 1; run a1 0; run a1 1; done
unset DVAE_RAW_INPUTS
for lib in a0 a1; do for raw in 0 1; do
  if [ "$raw" = "1" ]; then export DVAE_RAW_INPUTS=1; else unset DVAE_RAW_INPUTS; fi
  echo "stamps $lib raw=$raw"; DVAE_LIB=$PWD/$V/$lib.so DVAE_HSTAMPS=1 DVAE_COLD=1 python tools/stamp_rows.py bf16x3 8192 2>/dev/null | grep -v amdgpu
done; done
