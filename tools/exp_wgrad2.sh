# usage: exp_wgrad2.sh "<cflags>@<ksplit>[@precision]" ...  -- rebuild with the flags, run the bench with that many frame slices
set -e
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  fl="${spec%%@*}"; rest="${spec#*@}"; ks="${rest%%@*}"; pr="bf16x3"; [ "$rest" != "$ks" ] && pr="${rest#*@}"
  DVAE_CFLAGS="$fl" python disentangled-vae_amd/build.py --force > /dev/null 2>&1
  python - <<PY
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-extras", "--steps", "300", "--warmup", "30", "--ksplit", "$ks", "--precision", "$pr"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
r = json.loads(out)
print(repr("$fl"), "ks $ks $pr", "us/step", round(r["ms_per_step"]*1e3, 2), {k: round(v, 2) for k, v in r["roofline"]["avg_us"].items()}, flush=True)
PY
done
