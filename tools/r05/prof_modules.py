"""GPU box: where the host time of the scripts-unchanged training loop goes (packages.models modules + autograd + stock torch.optim.Adam,
scripts/training_M2.py:132-147) -- cProfile over the loop at the scripts' batch size, next to a plain nn.Module of the same architecture."""
import cProfile, importlib, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
model = sys.argv[3] if len(sys.argv) > 3 else "M2"
synth = importlib.import_module("disentangled-vae_amd.synth")
dims = dict(x_dim=513, y_dim=513 if model == "M2" else 1, z_dim=16, h_dim=[128, 128])
dev = torch.device("cuda", 0)
impl = bench.ModulesImpl(model, dims, dev, 1)
batches = synth.device_batches(dims, B, 4, 1234, dev)

def loop(n, item):
    for i in range(n):
        x, y, e = batches[i % 4]
        loss = impl.step(x, y, e)
        if item:
            loss.item()                      # the scripts read the loss every step (training_M2.py:150)

for item in (False, True):
    loop(100, item); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(steps, item); torch.cuda.synchronize()
    print("drop-in modules, B %d, loss.item() every step: %s -> %.1f us / step" % (B, item, (time.perf_counter() - t0) / steps * 1e6), flush=True)

if model != "M2":
    pr = cProfile.Profile(); pr.enable(); loop(steps, True); pr.disable(); torch.cuda.synchronize()
    for key, n in (("tottime", 45), ("cumulative", 60)):
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
        print("\n".join(l[:200] for l in s.getvalue().splitlines()))
    sys.exit(0)
# plain torch modules of the same shapes, same loop (PyTorch's own floor on this host)
import torch.nn as nn
class Plain(nn.Module):
    def __init__(s):
        super().__init__()
        s.e1, s.e2, s.mu, s.lv = nn.Linear(1026, 128), nn.Linear(128, 128), nn.Linear(128, 16), nn.Linear(128, 16)
        s.d1, s.d2, s.out = nn.Linear(529, 128), nn.Linear(128, 128), nn.Linear(128, 513)
    def forward(s, x, y, e):
        h = torch.tanh(s.e2(torch.tanh(s.e1(torch.cat([x, y], 1)))))
        mu, lv = s.mu(h), s.lv(h)
        z = torch.addcmul(mu, torch.exp(0.5 * lv), e)
        d = torch.tanh(s.d2(torch.tanh(s.d1(torch.cat([z, y], 1)))))
        return torch.exp(s.out(d)), mu, lv
from packages.models.utils import elbo
pm = Plain().to(dev); popt = torch.optim.Adam(pm.parameters(), lr=1e-4)
def ploop(n, item):
    for i in range(n):
        x, y, e = batches[i % 4]
        r, mu, lv = pm(x, y, e)
        loss = torch.mean(torch.sum(x / r - torch.log(x + 1e-8) + torch.log(r) - 1, 1)) - 0.5 * torch.mean(torch.sum(lv - mu.pow(2) - lv.exp(), 1))
        loss.backward(); popt.step(); popt.zero_grad()
        if item:
            loss.item()
for item in (False, True):
    ploop(100, item); torch.cuda.synchronize()
    t0 = time.perf_counter(); ploop(steps, item); torch.cuda.synchronize()
    print("plain torch modules, B %d, loss.item() every step: %s -> %.1f us / step" % (B, item, (time.perf_counter() - t0) / steps * 1e6), flush=True)

pr = cProfile.Profile(); pr.enable(); loop(steps, True); pr.disable(); torch.cuda.synchronize()
for key, n in (("tottime", 45), ("cumulative", 45)):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
    print("\n".join(l[:200] for l in s.getvalue().splitlines()))
