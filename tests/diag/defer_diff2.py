"""Deferred optimizer step vs the three-launch step: which parameters differ after the second in-kernel update?  (diagnostic; GPU)"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
T = importlib.import_module("disentangled-vae_amd.trainer")
dims = dict(x_dim=513, y_dim=513, z_dim=16, h_dim=(128, 128))
params = gu.make_params("M2", dims, 71)
B = 8192
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
batches = [gu.make_batch(dims, B, 80 + i) for i in range(4)]
out = {}
for defer in ("1", "0"):
    os.environ["DVAE_DEFER_APPLY"] = defer
    tr = T.Trainer("M2", dims, params, batch=B, precision="bf16x3")
    snaps = []
    for i, (x, y, e) in enumerate(batches):
        tr.step(t(x), t(y), t(e))
        torch.cuda.synchronize()
        snaps.append((tr._params.cpu().numpy().copy(), tr._m.cpu().numpy().copy(), tr._v.cpu().numpy().copy()))
    out[defer] = (snaps, tr)
sa, tr = out["1"]; sb, _ = out["0"]
for i in range(1, 4):
    for nm, k in (("p", 0), ("m", 1), ("v", 2)):
        a, b = sa[i][k], sb[i - 1][k]
        d = np.flatnonzero(a != b)
        print(f"after deferred step {i} vs three-launch step {i-1}: {nm} differing {d.size}")
        if d.size:
            for ti, name in enumerate(tr.names):
                o, r, c = tr.plan.tensor_offset[ti], tr.plan.tensor_rows[ti], tr.plan.tensor_cols[ti]
                dd = d[(d >= o) & (d < o + r * c)] - o
                if dd.size:
                    rows, cols = dd // c, dd % c
                    print(f"     {name}: {dd.size} elements; rows {np.unique(rows)[:12]} cols {np.unique(cols)[:24]}  max |diff| {np.abs(a[o + dd] - b[o + dd]).max():.3e}")
