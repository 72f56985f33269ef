"""Diagnostic: phase shares of the rows kernel from in-kernel wall-clock stamps (100 MHz).
usage: stamp_rows.py [bf16|fp32] [B] [noapply]"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch
import golden_util as gu
T = importlib.import_module("disentangled-vae_amd.trainer"); N = importlib.import_module("disentangled-vae_amd.native")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"; B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
noapply = len(sys.argv) > 3 and sys.argv[3] == "noapply"
dims = dict(x_dim=513, y_dim=513, z_dim=16, h_dim=(128, 128))
tr = T.Trainer("M2", dims, None, batch=B, precision=prec, seed=0)
x, y, e = (torch.from_numpy(a).cuda() for a in gu.make_batch(dims, B, 1))
run = (lambda: tr.grads_only(x, y, e)) if noapply else (lambda: tr.step(x, y, e))
for _ in range(5): run()
if os.environ.get("DVAE_COLD"):      # inputs cold in HBM, as in bench.py: cycle through a > 1 GB pool before the stamped step
    pool = [(torch.rand_like(x) * 3 + 1e-3, (torch.rand_like(y) > 0.5).float(), torch.randn_like(e)) for _ in range(36)]
    for (px, py, pe) in pool: tr.step(px, py, pe)
    x, y, e = pool[0]
    run = (lambda: tr.grads_only(x, y, e)) if noapply else (lambda: tr.step(x, y, e))
buf = torch.zeros(tr.plan.rows_grid * 32, dtype=torch.int64, device="cuda")
N.load().dvae_train_debug_stamps(N.ptr(buf)); run(); torch.cuda.synchronize(); N.load().dvae_train_debug_stamps(None)
s = buf.cpu().numpy().reshape(-1, 32)[:, :16].astype(np.float64) * 0.01      # us
names = ["x load+commit+stash", "L1 x gemm", "y load+stash+L1 y gemm", "h1 epilogue", "L2", "heads", "dec L1", "dec L2", "out layer", "bwd d2", "bwd d1", "bwd z", "bwd h2", "bwd h1", "loss sums"]
d = np.diff(s, axis=1)
print(f"{prec} B={B} {'grads only' if noapply else 'full step'}: kernel span {s[:,15].max()-s[:,0].min():.1f} us; per-WG median {np.median(s[:,15]-s[:,0]):.1f} us; start skew {s[:,0].max()-s[:,0].min():.1f} us")
print("  " + "  ".join(f"{n}={np.median(d[:, i]):.2f}" for i, n in enumerate(names)))

if os.environ.get("DVAE_FINE"):
    raw = buf.cpu().numpy().reshape(-1, 32).astype(np.float64) * 0.01
    f = raw[:, [4, 16, 17, 18, 19, 20, 5]]
    f2 = raw[:, [21, 22, 23, 24, 25, 26]]
    print("  out tile 4 (wave 0, us): gemm=%.2f prefetch=%.2f x+bias lds reads=%.2f epilogue math=%.2f put=%.2f" % tuple(np.median(np.diff(f2, axis=1), axis=0)))
    print("  L2 fine (us, median): ring filled=%.2f stash issued=%.2f gemm done=%.2f tanh done=%.2f lds put=%.2f barrier=%.2f" % tuple(np.median(np.diff(f, axis=1), axis=0)))

if os.environ.get("DVAE_HSTAMPS"):
    r = buf.cpu().numpy().reshape(-1, 32).astype(np.float64) * 0.01
    t0 = r[:, 0:1]
    names_h = ["helper: y issued", "x stash issued", "past BL1X", "y committed", "lo-plane check done (at BY)"]
    print("  helper (us since tile start, median): " + "  ".join(f"{n}={np.median(r[:, 16 + i] - t0[:, 0]):.2f}" for i, n in enumerate(names_h)) +
          "\n  out layer, helper (us since BD2): " + "  ".join(f"{'arrive' if i % 2 == 0 else 'pass'} RB{i // 2}={np.median(r[:, 21 + i] - r[:, 8]):.2f}" for i in range(8)) +
          f" BDA={np.median(r[:, 9] - r[:, 8]):.2f}" +
          f"  | chain: BX={np.median(r[:,1]-t0[:,0]):.2f} L1x done={np.median(r[:,2]-t0[:,0]):.2f} L1y done={np.median(r[:,3]-t0[:,0]):.2f}")
if os.environ.get("DVAE_FSTAMPS"):       # -DR2_FINE build: chain-side sub-phase stamps of the output layer (slots 16 + 3 I: round start, GEMM done, a -> U done)
    r = buf.cpu().numpy().reshape(-1, 32).astype(np.float64) * 0.01
    med = lambda a: float(np.median(a))
    parts = []
    for I in range(4):
        parts.append(f"round {I}: start@{med(r[:, 16 + 3 * I] - r[:, 8]):.2f} gemm={med(r[:, 17 + 3 * I] - r[:, 16 + 3 * I]):.2f} put={med(r[:, 18 + 3 * I] - r[:, 17 + 3 * I]):.2f}")
    print("  out layer, chain (us; start since BD2): " + "  ".join(parts) + f"  tail(17th tile)={med(r[:, 28] - r[:, 27]):.2f} BDA wait={med(r[:, 9] - r[:, 28]):.2f}")
raw = buf.cpu().numpy().reshape(-1, 32).astype(np.float64)
cyc = raw[:, 31] - raw[:, 30]; us = (raw[:, 15] - raw[:, 0]) * 0.01
print("  shader clock during the kernel: %.0f MHz (median over workgroups)" % np.median(cyc / us))
