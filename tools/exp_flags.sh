# usage: exp_flags.sh "<cflags A>" "<cflags B>" ...   -- rebuild with each flag set and print the bench line's us/step and kernel times
set -e
cd $GRAFT_REPO_ROOT
for fl in "$@"; do
  DVAE_CFLAGS="$fl" python disentangled-vae_amd/build.py --force > /dev/null 2>&1
  python - <<PY
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "300", "--warmup", "30"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
r = json.loads(out)
print(repr("$fl"), "us/step", round(r["ms_per_step"]*1e3, 2), {k: round(v, 2) for k, v in r["roofline"]["avg_us"].items()})
PY
done
