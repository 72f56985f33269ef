# sweep: wgrad waves per workgroup (env DVAE_GPW) x frame slices (plan); prints per-kernel hipEvent times
set -e
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  gpw="${spec%%@*}"; ks="${spec##*@}"
  DVAE_GPW=$gpw python - <<PY
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "300", "--warmup", "30", "--ksplit", "$ks"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
r = json.loads(out)
print("GPW $gpw ksplit $ks", "us/step", round(r["ms_per_step"]*1e3, 2), {k: round(v, 2) for k, v in r["roofline"]["avg_us"].items()})
PY
done
