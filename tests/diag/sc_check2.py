"""Diagnostic: after one train step, dump the stash region of the workspace (D2H copy) and the gradient slabs; run under two builds
(DVAE_LIB) and compare offline: is the stash in MEMORY right under sc1 stores (then the reader sees stale lines) or wrong (then the stores are)?"""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import golden_util as gu
T = importlib.import_module("disentangled-vae_amd.trainer")
t = lambda a: None if a is None else torch.from_numpy(a).cuda()
out = sys.argv[1]
dims = dict(x_dim=513, y_dim=513, z_dim=16, h_dim=(128, 128))
params = gu.make_params("M2", dims, 5)
x, y, e = gu.make_batch(dims, 2048, 70)
res = {}
for inst in range(2):
    tr = T.Trainer("M2", dims, params, batch=2048, precision="bf16x3")
    tr.grads_only(t(x), t(y), t(e))
    torch.cuda.synchronize()
    ws = tr.ws.cpu().numpy().copy()
    go = tr.plan.grad_offset_bytes
    res[f"ws{inst}"] = ws[:go]                 # tables, weight copies, stash
    res[f"slab{inst}"] = ws[go:go + 4 * tr.plan.n_params * tr.plan.ksplit].view(np.float32)
    print("instance", inst, "slab finite:", np.isfinite(res[f"slab{inst}"]).all(), "abs max", float(np.nanmax(np.abs(res[f"slab{inst}"]))))
np.savez(out, **res)
