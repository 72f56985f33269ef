"""GPU box: time of ONE MCEM chain launch (E-step chain 30 + 10, Wiener chain 75 + 25) over U utterances of 300 frames, hipEvent mean of 20 launches.
usage: chain_time.py [utterances]   (DVAE_LIB selects the library, DVAE_MCEM_TILE the kernel)"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch
import golden_util as gu, mcem_cases as mc
from impl_modules import build_model
M = importlib.import_module("disentangled-vae_amd.mcem")
U = int(sys.argv[1]) if len(sys.argv) > 1 else 25
dims = dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128))
m = build_model("M2", dims); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in gu.make_params("M2", dims, 3).items()}); m.eval().cuda()
for p in m.parameters(): p.requires_grad = False
mc.DIMS["bench"] = dims
X, S, y = mc.make_utterance(dict(seed=5, N=300, model="bench"))
out = {}
for prec in ("fp32", "bf16x3", "bf16"):
    mb = M.McemBatch(m, niter=3, nsamples_E_step=10, burnin_E_step=30, nsamples_WF=25, burnin_WF=75, precision=prec)
    mb.init_parameters([X] * U, [torch.from_numpy(y).cuda()] * U); mb.run()
    for (R, B) in ((10, 30), (25, 75)):
        for _ in range(3): mb._chain(R, B)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): mb._chain(R, B)
        e1.record(); torch.cuda.synchronize()
        out[f"{prec}_{B}+{R}"] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
print(f"U={U} chain us:", out, flush=True)
