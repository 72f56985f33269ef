cd $GRAFT_REPO_ROOT
V=disentangled-vae_amd/build/variants; mkdir -p $V
i=0
for fl in "" "-DW4_FOLD=0"; do
  DVAE_CFLAGS="$fl" python disentangled-vae_amd/build.py --force > /dev/null 2>&1; cp disentangled-vae_amd/libdvae_hip.so $V/w$i.so; echo "w$i = '$fl'"; i=$((i+1))
done
for r in 1 2 3; do for j in 0 1; do
  DVAE_LIB=$PWD/$V/w$j.so python bench.py --no-extras --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('w$j', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()})"
done; done
