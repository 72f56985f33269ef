"""GPU box: the scripts' training loop on the drop-in modules with the Linear + activation stacks as ONE autograd node each (default) against one
node per layer (DVAE_LINEAR_STACK=0), alternating in one process.  usage: stack_ab.py [model] [B] [steps]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
model = sys.argv[1] if len(sys.argv) > 1 else "M2_info"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
synth = importlib.import_module("disentangled-vae_amd.synth")
dims = dict(x_dim=513, y_dim=513 if model == "M2" else 1, z_dim=16, h_dim=[128, 128])
dev = torch.device("cuda", 0)
impl = bench.ModulesImpl(model, dims, dev, 1)
batches = synth.device_batches(dims, B, 4, 1234, dev)
def loop(n):
    for i in range(n):
        x, y, e = batches[i % 4]
        impl.step(x, y, e).item()
for v in ("1", "0"):
    os.environ["DVAE_LINEAR_STACK"] = v; loop(100)
res = {"1": [], "0": []}
for r in range(9):
    for v in ("1", "0"):
        os.environ["DVAE_LINEAR_STACK"] = v
        torch.cuda.synchronize(); t0 = time.perf_counter(); loop(steps); torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / steps * 1e6)
import statistics as st
print(model, "B", B, "one node per stack: median %.1f min %.1f us / step;  one node per layer: median %.1f min %.1f" % (st.median(res["1"]), min(res["1"]), st.median(res["0"]), min(res["0"])), [round(t) for t in res["1"]], [round(t) for t in res["0"]])
