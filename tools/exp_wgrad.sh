set -e
cd $GRAFT_REPO_ROOT
run() { # cflags ksplit label
  DVAE_CFLAGS="$1" python disentangled-vae_amd/build.py --force > /dev/null 2>&1
  python - <<PY
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "200", "--warmup", "20", "--ksplit", "$2"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
r = json.loads(out)
print("$3", "us/step", round(r["ms_per_step"]*1e3, 2), "kernels", {k: round(v, 2) for k, v in r.get("kernel_us", {}).items()} if "kernel_us" in r else r.get("roofline", {}).get("kernel"))
PY
}
run "" 8 "base RD8 ks8"
run "-DDVAE_WRING_BF16=12" 8 "RD12 ks8"
run "-DDVAE_WGRAD_OCC=2" 8 "occ2 RD8 ks8"
run "-DDVAE_WGRAD_OCC=2" 16 "occ2 RD8 ks16"
run "-DDVAE_WGRAD_OCC=2 -DDVAE_WRING_BF16=6" 16 "occ2 RD6 ks16"
run "-DDVAE_WGRAD_OCC=2 -DDVAE_WRING_BF16=4" 16 "occ2 RD4 ks16"
run "" 16 "base RD8 ks16"
