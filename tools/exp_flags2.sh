# usage: exp_flags2.sh "<cflags>@<bench args>" ...  -- rebuild with the flags, run the bench with extra args, print kernel times
set -e
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  fl="${spec%%@*}"; args="${spec#*@}"
  DVAE_CFLAGS="$fl" python disentangled-vae_amd/build.py --force > /dev/null 2>&1
  python - <<PY
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-extras", "--steps", "300", "--warmup", "30"] + "$args".split(), capture_output=True, text=True).stdout.strip().splitlines()[-1]
r = json.loads(out)
print(repr("$fl"), "$args", "us/step", round(r["ms_per_step"]*1e3, 2), {k: round(v, 2) for k, v in r["roofline"]["avg_us"].items()}, flush=True)
PY
done
