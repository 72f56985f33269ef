"""Where does the bf16x3 M2_info gradient error sit?  (diagnostic; GPU)"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from oracle import vae_oracle as vo
trainer = importlib.import_module("disentangled-vae_amd.trainer")
dims = dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128))
B = 8192
for seed in (31, 131, 231):
    params = gu.make_params("M2_info", dims, seed)
    x, y, e = gu.make_batch(dims, B, seed + 1)
    p32 = {k: v.astype(np.float32) for k, v in params.items()}
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    o32, a1, a2 = vo.m2info_losses_and_grads(p32, x, y, e, 0.5, 10.0, 1.0)
    o64, b1, b2 = vo.m2info_losses_and_grads(p64, x.astype(np.float64), y.astype(np.float64), e.astype(np.float64), 0.5, 10.0, 1.0)
    tot = lambda g1, g2, k: np.asarray(g1[k], np.float64) + (np.asarray(g2[k], np.float64) if k in g2 else 0.0)
    for prec in ("fp32", "bf16x3"):
        tr = trainer.Trainer("M2_info", dims, params, batch=B, precision=prec, alpha=0.5, beta=10.0, gamma=1.0)
        t = lambda a: torch.from_numpy(a).cuda()
        tr.step(t(x), t(y), t(e))
        g = tr.grads_numpy()
        for k in params:
            if not (k.startswith("auxiliary") or "classifier" in k):
                continue
            r32, r64 = tot(a1, a2, k).reshape(g[k].shape), tot(b1, b2, k).reshape(g[k].shape)
            e32 = np.abs(g[k] - r32) / np.abs(r32).max(); e64 = np.abs(g[k] - r64) / np.abs(r64).max()
            oo = np.abs(r32 - r64).max() / np.abs(r64).max()
            if g[k].ndim == 2:
                rows = np.sort(e64.max(axis=1))[::-1][:4]
                print(f"seed {seed} {prec:7s} {k:44s} vs f32 {e32.max():.1e} vs f64 {e64.max():.1e} (f32 oracle vs f64 {oo:.1e}) worst rows {np.array2string(rows, precision=1)}")
            else:
                print(f"seed {seed} {prec:7s} {k:44s} vs f32 {e32.max():.1e} vs f64 {e64.max():.1e} (f32 oracle vs f64 {oo:.1e})")
