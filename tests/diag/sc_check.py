"""Diagnostic: are stash stores with cache-policy bits safe?  Several Trainers one after another (workspace addresses get reused by the
allocator), several steps each; gradients of every step against the float64 oracle, for the fp32 and bf16x3 policies, small and big batch."""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import golden_util as gu
from oracle import vae_oracle as vo
T = importlib.import_module("disentangled-vae_amd.trainer")
t = lambda a: None if a is None else torch.from_numpy(a).cuda()
worst = {}
for model, y_dim, B in (("M1", 0, 32), ("M2", 513, 32), ("M2", 1, 300), ("M2", 513, 2048)):
    dims = dict(x_dim=513, y_dim=y_dim, z_dim=16, h_dim=(128, 128))
    params = gu.make_params(model, dims, 5)
    batches = [gu.make_batch(dims, B, 70 + s) for s in range(3)]
    refs = []
    p = {k: v.astype(np.float64) for k, v in params.items()}
    opt = vo.AdamState(list(p))
    for (x, y, e) in batches:
        out, g = vo.train_step_vae(model, p, opt, x.astype(np.float64), None if y is None else y.astype(np.float64), e.astype(np.float64))
        refs.append({k: np.asarray(v, np.float64).copy() for k, v in g.items()})
    for prec in ("fp32", "bf16x3"):
        for inst in range(4):
            tr = T.Trainer(model, dims, params, batch=B, precision=prec)
            for s, (x, y, e) in enumerate(batches):
                tr.step(t(x), t(y), t(e))
                g = tr.grads_numpy()
                w = max(float(np.abs(g[k] - refs[s][k].reshape(g[k].shape)).max() / (np.abs(refs[s][k]).max() + 1e-30)) for k in g)
                key = (model, y_dim, B, prec)
                worst[key] = max(worst.get(key, 0.0), w)
                if w > (1e-3 if prec == "bf16x3" else 1e-4) and s == 0:
                    print("BAD", key, "instance", inst, "step", s, "worst relmax", w)
for k, v in worst.items():
    print(k, "worst relmax over 4 trainers x 3 steps: %.2e" % v)
