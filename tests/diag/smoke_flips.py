"""Diagnostic: how many updated parameters of the smoke step differ from the oracle's by more than 1.05 lr (an Adam sign flip), and how large
the oracle gradient is there relative to the tensor's maximum."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import golden_util as gu
from oracle import vae_oracle as vo
T = importlib.import_module("disentangled-vae_amd.trainer")
dims = dict(x_dim=513, y_dim=513, z_dim=16, h_dim=(128, 128))
for B in (64, 8192):
    params = gu.make_params("M2", dims, seed=7)
    x, y, e = gu.make_batch(dims, B, seed=8)
    p_ref = {k: v.copy() for k, v in params.items()}
    out_ref, g_ref = vo.train_step_vae("M2", p_ref, vo.AdamState(list(p_ref)), x, y, e)
    for prec in ("fp32", "bf16x3"):
        tr = T.Trainer("M2", dims, params, batch=B, device="cuda:0", precision=prec)
        tr.step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(e).cuda())
        p = tr.state_dict_numpy(); g = tr.grads_numpy()
        tot = flips = 0; worst = 0.0
        for k in p_ref:
            d = np.abs(p[k] - p_ref[k]); gr = np.abs(np.asarray(g_ref[k], np.float64).reshape(d.shape))
            f = d > 1.05e-4
            tot += d.size; flips += int(f.sum())
            if f.any(): worst = max(worst, float((gr[f] / gr.max()).max()))
        print(f"B {B} {prec}: {flips} of {tot} parameters differ by more than 1.05 lr; largest |oracle gradient| / tensor max among them {worst:.2e}")
