# usage: ab.sh <lib A> <lib B> [bench args]  -- alternate two builds of the library on the same box (3 rounds), print us/step and kernel times
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift; shift
for r in 1 2 3; do for so in $A $B; do
  DVAE_LIB=$PWD/$so python bench.py --no-extras --steps 300 --warmup 30 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$so', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()})"
done; done
