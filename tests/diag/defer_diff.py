"""Deferred optimizer step vs the three-launch step: where do the workspaces differ?  (diagnostic; GPU)"""
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
batches = [gu.make_batch(dims, B, 80 + i) for i in range(3)]
out = {}
for defer in ("1", "0"):
    os.environ["DVAE_DEFER_APPLY"] = defer
    tr = T.Trainer("M2", dims, params, batch=B, precision="bf16x3")
    snaps = []
    for i, (x, y, e) in enumerate(batches):
        l = tr.step(t(x), t(y), t(e)).cpu().numpy().copy()
        torch.cuda.synchronize()
        snaps.append((l, tr.ws.cpu().numpy().copy(), tr._params.cpu().numpy().copy(), tr._m.cpu().numpy().copy()))
    out[defer] = (snaps, tr.plan.grad_offset_bytes, tr.plan.workspace_bytes, tr.plan.n_params)
(sa, go, wb, npar), (sb, _, _, _) = out["1"], out["0"]
print("grad offset", go, "workspace", wb, "n_params", npar)
for i in range(3):
    la, wa, pa, ma = sa[i]; lb, wb_, pb, mb = sb[i]
    print(f"step {i}: losses equal {np.array_equal(la, lb)} {la} {lb}")
    # deferred: after step i the workspace copies hold the update of step i-1 (three-launch: of step i).  Compare deferred step i with
    # three-launch step i - 1 for the weight copies, same step for everything written by rows / wgrad
    d = np.flatnonzero(wa != wb_)
    print(f"   workspace bytes differing (same step): {d.size}; first {d[:3]} last {d[-3:] if d.size else []}")
    if i > 0:
        d2 = np.flatnonzero(wa[:go] != sb[i - 1][1][:go])
        print(f"   below the slabs vs three-launch step {i-1}: {d2.size}; first {d2[:5]} last {d2[-5:] if d2.size else []}")
        if d2.size:
            # histogram by 64 KB regions
            reg, cnt = np.unique(d2 >> 16, return_counts=True)
            print("   64KB regions:", list(zip(reg.tolist(), cnt.tolist()))[:40])
        print("   params equal to three-launch previous step:", np.array_equal(pa, sb[i - 1][2]), " m:", np.array_equal(ma, sb[i - 1][3]))
    gs = np.flatnonzero(wa[go:] != wb_[go:])
    print(f"   slab bytes differing: {gs.size}")
