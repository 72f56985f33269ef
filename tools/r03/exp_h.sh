cd $GRAFT_REPO_ROOT
run() {
  if [ "$1" = "0" ]; then unset DVAE_RAW_INPUTS; else export DVAE_RAW_INPUTS=$1; fi
  python bench.py --no-extras --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('raw=$1', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()})"
}
for r in 1 2 3; do run 0; run x; run xy; done
unset DVAE_RAW_INPUTS
timeout -k 10 300 python -m pytest tests/test_gpu_fused.py -x -q -k "raw_inputs" 2>&1 | tail -2
