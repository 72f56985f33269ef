cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/ti.log 2>&1; tail -4 gpurun_out/ti.log
python tools/bench_stft.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print({k:(round(v['stft_us'],1),round(v['stft_power_us'],1),round(v['istft_us'],1)) for k,v in d.items() if isinstance(v,dict)})"
