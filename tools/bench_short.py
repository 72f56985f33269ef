"""Run bench.py with the given args and print a one-line summary (diagnostic helper)."""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True).stdout.strip().splitlines()
d = json.loads(out[-1])
r = d.get("roofline") or {}
print(" ".join(sys.argv[1:]), "|", round(d["value"] / 1e6, 1), "Mframes/s", round(d["ms_per_step"] * 1e3, 1), "us/step",
      {k: round(v, 1) for k, v in (r.get("avg_us") or {}).items()})
