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
            DVAE_ROWS=2 DVAE_LIB=$PWD/$so python bench.py --no-extras --precision $p --steps 200 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
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
e3_6:-DR2_PD_X3=3 -DR2_PD_BF=6
e4_10:-DR2_PD_X3=4 -DR2_PD_BF=10
e5_12:-DR2_PD_X3=5 -DR2_PD_BF=12
e6_14:-DR2_PD_X3=6 -DR2_PD_BF=14
e4b4:-DR2_PD_X3=4 -DR2_BD_X3=4 -DR2_PD_BF=10
LIST
wait
for o in /tmp/r2_*.o; do
    n=$(basename $o .o); n=${n#r2_}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$n.so $OBJS $o
done
ls -la $V
