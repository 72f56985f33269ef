#!/bin/bash
# Build rows2-kernel variants with different weight-ring depths into gpurun_out/variants/*.so (run locally),
# then time them on the GPU box with:  tools/exp_rows2.sh run
# A variant = "name:-DR2_...=.. -DR2_...=.."
set -e
cd "$(dirname "$0")/.."
V=disentangled-vae_amd/build/variants
if [ "$1" = "run" ]; then
    for so in $V/*.so; do
        n=$(basename $so .so)
        for p in bf16x3 bf16; do
            DVAE_LIB=$PWD/$so python bench.py --precision $p --steps 200 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$n $p', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()})"
        done
    done
    exit 0
fi
mkdir -p $V; rm -f $V/*.so
C=disentangled-vae_amd/csrc
OBJS=$(ls disentangled-vae_amd/build/*.o | grep -v train_rows2)
while IFS=: read -r name flags; do
    [ -z "$name" ] && continue
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $C/train_rows2.hip -o /tmp/r2_$name.o $flags &
done <<'LIST'
a:-DR2_PD_X3=4 -DR2_PRE_X3=2 -DR2_DBIG_X3=4 -DR2_PBIG_X3=2 -DR2_D128_X3=4 -DR2_P128_X3=2 -DR2_DBIG_BF=8 -DR2_PBIG_BF=4
b:-DR2_PD_X3=4 -DR2_PRE_X3=2 -DR2_DBIG_X3=8 -DR2_PBIG_X3=4 -DR2_D128_X3=8 -DR2_P128_X3=4 -DR2_DBIG_BF=16 -DR2_PBIG_BF=8
c:-DR2_PD_X3=4 -DR2_PRE_X3=2 -DR2_DBIG_X3=8 -DR2_PBIG_X3=2 -DR2_D128_X3=8 -DR2_P128_X3=2 -DR2_DBIG_BF=16 -DR2_PBIG_BF=4
d:-DR2_PD_X3=4 -DR2_PRE_X3=2 -DR2_DBIG_X3=6 -DR2_PBIG_X3=2 -DR2_D128_X3=8 -DR2_P128_X3=4 -DR2_DBIG_BF=12 -DR2_PBIG_BF=6
e:-DR2_PD_X3=6 -DR2_PRE_X3=2 -DR2_DBIG_X3=8 -DR2_PBIG_X3=4 -DR2_D128_X3=8 -DR2_P128_X3=4 -DR2_PD_BF=12 -DR2_DBIG_BF=16 -DR2_PBIG_BF=8
f:-DR2_PD_X3=4 -DR2_PRE_X3=2 -DR2_DBIG_X3=12 -DR2_PBIG_X3=4 -DR2_D128_X3=8 -DR2_P128_X3=8 -DR2_DBIG_BF=16 -DR2_PBIG_BF=12
LIST
wait
for o in /tmp/r2_*.o; do
    n=$(basename $o .o); n=${n#r2_}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$n.so $OBJS $o
done
ls -la $V
