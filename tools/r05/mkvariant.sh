#!/bin/bash
# usage: tools/r05/mkvariant.sh <name> <file.hip> "<cflags>"  -- in the BUILD container: compile ONE source of the library with extra flags, link it with
# the product objects of the other sources into disentangled-vae_amd/build/variants/<name>.so (travels to the GPU box with the snapshot; DVAE_LIB=<path> loads it)
set -e
cd "$(dirname "$0")/../.."
P=disentangled-vae_amd
[ -n "$MKV_NOBUILD" ] || python $P/build.py > /dev/null
mkdir -p $P/build/variants
name=$1; src=$2; shift 2
extra=""
case "$src" in mcem_mstep.hip|train_rows3.hip) extra="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $P/csrc/$src -o $P/build/variants/$name.$src.o -Wall -Wno-unused-function $extra $@
objs=""
for o in $P/build/*.hip.o; do b=$(basename $o); if [ "$b" = "$src.o" ]; then objs="$objs $P/build/variants/$name.$src.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/build/variants/$name.so $objs
echo $P/build/variants/$name.so
