# usage: exp_batch.sh "<cflags>" ... : bench at several per-GPU batch sizes for each flag set
set -e
cd $GRAFT_REPO_ROOT
for fl in "$@"; do
  DVAE_CFLAGS="$fl" python disentangled-vae_amd/build.py --force > /dev/null 2>&1
  for spec in "8192 300 30" "65536 60 10" "1048576 12 3"; do
    set -- $spec
    python - <<PY
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--batch", "$1", "--steps", "$2", "--warmup", "$3"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
r = json.loads(out)
print(repr("$fl"), "B=$1", "us/step", round(r["ms_per_step"]*1e3, 1), "M f/s", round(r["value"]/1e6, 1), {k: round(v, 1) for k, v in r["roofline"]["avg_us"].items()})
PY
  done
done
