"""Diagnostic: where a chain step of the weight-stationary MCEM kernel spends its shader clocks (per wave, mean over workgroups and steps).
usage: stamp_mcem.py [fp32|bf16x3|bf16] [utterances]"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch
import golden_util as gu, mcem_cases as mc
from impl_modules import build_model
N = importlib.import_module("disentangled-vae_amd.native"); M = importlib.import_module("disentangled-vae_amd.mcem")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"; U = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dims = dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128))
m = build_model("M2", dims); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in gu.make_params("M2", dims, 3).items()}); m.eval().cuda()
for p in m.parameters(): p.requires_grad = False
mc.DIMS["bench"] = dims
X, S, y = mc.make_utterance(dict(seed=5, N=300, model="bench"))
mb = M.McemBatch(m, niter=3, nsamples_E_step=10, burnin_E_step=30, nsamples_WF=25, burnin_WF=75, precision=prec)
mb.init_parameters([X] * U, [torch.from_numpy(y).cuda()] * U); mb.run()
ntiles = mb.ntot // 4                                    # room for 4-frame tiles (csrc/mcem_resident4.hip); 16- / 32-frame launches fill the first quarter / eighth
buf = torch.zeros(ntiles * 4 * 16, dtype=torch.int64, device="cuda")
N.load().dvae_mcem_debug_stamps(N.ptr(buf))
Zs, Vs = mb._chain(10, 30)
torch.cuda.synchronize(); N.load().dvae_mcem_debug_stamps(None)
r = buf.cpu().numpy().reshape(ntiles, 4, 16).astype(np.float64)
r = r[r[:, 0, 9] > 0]; ntiles = len(r)
steps = r[:, :, 9:10]
per = r[:, :, :9] / np.maximum(steps, 1)
names = ["P4+P0 (accept, next proposal)", "wait B0", "L1 + tanh + put", "wait B1", "L2 + tanh + put + bin-512 terms", "wait B2", "output layer + likelihood", "reduce + red write", "wait B3"]
print(f"{prec}, {U} utterances ({ntiles} tiles): shader clocks per chain step, mean over tiles; total {per.sum(axis=2).mean():.0f}")
if r[:, :, 10].max() > 0:      # the 32- and 4-frame kernels also stamp the launch's parts (the 4-frame kernel: prologue and chain)
    print("  launch, shader clocks (mean over tiles and waves): prologue (weights, label terms, X2 / Vb tiles) %.0f  chain %.0f  tail (decoder passes / slot copies, stores) %.0f"
          % (r[:, :, 10].mean(), r[:, :, 11].mean(), r[:, :, 12].mean()))
    if r[:, :, 13].max() > 0:
        print("    of the prologue: resident fragments + bias tables landed after %.0f" % r[:, :, 13].mean())
for w in range(4):
    print(f"  wave {w}: " + "  ".join(f"{n}={per[:, w, i].mean():.0f}" for i, n in enumerate(names)))
