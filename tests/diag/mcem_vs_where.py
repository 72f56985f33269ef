"""Diagnostic: where do the decoder variances of the MH kernel differ from the oracle (bins / frames / samples)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import test_gpu_mcem as T
for model, y_dim, N in (("M1", 0, 45), ("M2", 1, 70)):
    params, prefix, pack, X2, y, Z, g, W, H, rng = T.setup(model, y_dim, N, 5, precision="bf16x3")
    nit, burnin = 12, 5
    noise = rng.standard_normal((nit, 16, N)).astype(np.float32)
    logu = np.log(rng.random((nit, N)).astype(np.float32))
    Vb = (W @ H).astype(np.float32)
    t = T.t
    Zs, Vs, accp, accd = pack.sample(t(Z), t(y), t(g), t(Vb), t(X2), t(noise), t(logu), burnin, trace=True)
    Zs, Vs = Zs.cpu().numpy(), Vs.cpu().numpy()
    Vs_o = T.mo.compute_vs(params, prefix, Zs, y)
    rel = np.abs(Vs - Vs_o) / np.abs(Vs_o)
    bad = np.argwhere(rel > 1e-4)
    print(model, "violations", len(bad), "bins", np.unique(bad[:, 1])[:40], "frames", np.unique(bad[:, 2])[:70], "samples", np.unique(bad[:, 0]))
    print("  worst rel per bin (top 8):", sorted(((float(rel[:, b, :].max()), b) for b in range(513)), reverse=True)[:8])
