"""Experiment: Metropolis-Hastings chain time (40 iterations + 10 decodes) vs number of 32-frame tiles, i.e. workgroups
per CU (256 CUs): shows what co-resident workgroups buy for the kernel variant compiled in."""
import importlib, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import golden_util as gu
from impl_modules import build_model
dev = importlib.import_module("disentangled-vae_amd.mcem")
dims = dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128))
m = build_model("M2", dims); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in gu.make_params("M2", dims, 3).items()}); m.cuda()
out = {}
for prec in ("fp32", "bf16"):
    pack = dev.DecoderPack(m.decoder, 1, prec)
    for tiles in (256, 512, 768, 1024):
        n = tiles * 32
        g = torch.ones(n, device="cuda"); Vb = torch.rand(513, n, device="cuda") + 0.1; X2 = torch.rand(513, n, device="cuda") + 0.01
        Z = torch.randn(16, n, device="cuda"); y = (torch.rand(1, n, device="cuda") > 0.5).float()
        noise = torch.randn(40, 16, n, device="cuda"); logu = torch.log(torch.rand(40, n, device="cuda"))
        for _ in range(2): pack.sample(Z, y, g, Vb, X2, noise, logu, 30)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): pack.sample(Z, y, g, Vb, X2, noise, logu, 30)
        e1.record(); torch.cuda.synchronize()
        out[f"{prec}_{tiles}tiles_us"] = round(e0.elapsed_time(e1) * 200, 1)
print(json.dumps(out))
