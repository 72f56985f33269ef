cd $GRAFT_REPO_ROOT
ABN_ARGS="" bash tools/abn.sh "" "-DDVAE_W4_WIDE=0"
python bench.py --batch 1048576 --steps 20 --warmup 5 --pool-gb 8 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=2^20', round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e6,1), 'Mf/s', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()})"
python bench.py --batch 65536 --steps 50 --warmup 10 --pool-gb 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=65536', round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e6,1), 'Mf/s', {k:round(v,1) for k,v in d['roofline']['avg_us'].items()})"
