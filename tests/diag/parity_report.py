"""Measured deviation of the benchmarked operand policy (bf16x3) -- and of the fp32 / bf16 policies beside it -- per configuration:
(a) one 8192-frame train step against the float64 numpy oracle (losses: relative; gradients: worst |err| / max|ref| per tensor);
(b) the reference-captured golden vectors (tests/golden/vae_golden.npz, step 1) through Trainer(precision=...): the same two figures.
Writes profiles-style JSON to the path given (default gpurun_out/r03_parity.json).  Test infrastructure: uses oracle/ as the checker."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import golden_util as gu
from oracle import vae_oracle as vo
T = importlib.import_module("disentangled-vae_amd.trainer")

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03_parity.json")
t = lambda a: None if a is None else torch.from_numpy(a).cuda()
relmax = lambda a, b: float(np.max(np.abs(np.asarray(a, np.float64) - b)) / (np.max(np.abs(b)) + 1e-30))
res = {"oracle_8192": {}, "golden_step1": {}}
for model, y_dim in (("M2", 513), ("M2", 1), ("M1", 0)):
    dims = dict(x_dim=513, y_dim=y_dim, z_dim=16, h_dim=(128, 128))
    params = gu.make_params(model, dims, 11)
    x, y, e = gu.make_batch(dims, 8192, 12)
    p = {k: v.copy() for k, v in params.items()}
    outo, grads = vo.train_step_vae(model, p, vo.AdamState(list(p)), x.astype(np.float64), None if y is None else y.astype(np.float64), e.astype(np.float64))
    ref = np.array([outo["loss"], outo["recon"], outo["kl"]])
    for prec in ("fp32", "bf16x3", "bf16"):
        tr = T.Trainer(model, dims, params, batch=8192, precision=prec)
        losses = tr.step(t(x), t(y), t(e)).cpu().numpy()
        g = tr.grads_numpy()
        per = {k: relmax(g[k], np.asarray(grads[k], np.float64).reshape(g[k].shape)) for k in grads}
        res["oracle_8192"][f"{model}_y{y_dim}_{prec}"] = {"loss_rel": float(np.max(np.abs(losses - ref) / np.abs(ref))), "grad_relmax_worst": max(per.values()),
                                                         "grad_relmax_worst_tensor": max(per, key=per.get)}
# M2_info (alpha 0.5, beta 10, gamma 1) against the float32 oracle (the classifier saturates: see tests/test_gpu_fused.py)
dims = dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128))
params = gu.make_params("M2_info", dims, 31)
x, y, e = gu.make_batch(dims, 8192, 32)
p32 = {k: v.astype(np.float32) for k, v in params.items()}
outo, g1, g2 = vo.m2info_losses_and_grads(p32, x, y, e, 0.5, 10.0, 1.0)
ref = np.array([outo["ELBO"], outo["recon"], outo["kl"], outo["enc_loss"], outo["classif_loss"], outo["aux_loss"], outo["aux_enc_loss"]])
for prec in ("fp32", "bf16x3"):
    tr = T.Trainer("M2_info", dims, params, batch=8192, precision=prec, alpha=0.5, beta=10.0, gamma=1.0)
    losses = tr.step(t(x), t(y), t(e)).cpu().numpy()
    g = tr.grads_numpy()
    per, per_clean = {}, {}
    for k in params:
        gr = (np.asarray(g1[k], np.float64) + (np.asarray(g2[k], np.float64) if k in g2 else 0.0)).reshape(g[k].shape)
        err = np.abs(g[k].astype(np.float64) - gr) / (np.abs(gr).max() + 1e-30)
        per[k] = float(err.max())
        rows = np.sort(err.reshape(err.shape[0], -1).max(axis=1))[::-1]
        per_clean[k] = float(rows[2]) if rows.size > 2 else float(rows[-1])            # third-worst row: outside <= 2 ReLU-tie rows
    res["oracle_8192"][f"M2_info_y1_{prec}"] = {"loss_rel": float(np.max(np.abs(losses[:7] - ref) / (np.abs(ref) + 1e-5))), "grad_relmax_worst": max(per.values()),
                                               "grad_relmax_worst_tensor": max(per, key=per.get), "grad_relmax_worst_outside_2_rows_per_tensor": max(per_clean.values())}
fix = np.load(os.path.join(ROOT, "tests", "golden", "vae_golden.npz"))
for case in gu.CASES:
    name, model, dims, B, wscale = case
    if "full" not in name:
        continue
    seed = 100 + [c[0] for c in gu.CASES].index(name)
    params = gu.make_params(model, dims, seed, wscale)
    x, y, e = gu.make_batch(dims, B, seed * 1000 + 1)
    for prec in ("fp32", "bf16x3"):
        tr = T.Trainer(model, dims, params, batch=B, precision=prec)
        losses = tr.step(t(x), t(y), t(e)).cpu().numpy().astype(np.float64)
        g = tr.grads_numpy()
        refl = fix[f"{name}/step1/losses"]
        worst, wt = 0.0, None
        for k in params:
            for sub in ("grad", "grad_enc"):
                key = f"{name}/step1/{sub}/{k}"
                if model == "M2_info" and k.startswith("auxiliary."):
                    key = f"{name}/step1/grad_aux_total/{k}"
                flat = g[k].ravel()
                if key + "/full" in fix: r = fix[key + "/full"]
                elif key + "/sample" in fix: r = fix[key + "/sample"]; flat = flat[::gu.SAMPLE_STRIDE]
                else: continue
                d = float(np.max(np.abs(flat.astype(np.float64) - r)) / (np.max(np.abs(r)) + 1e-30))
                if d > worst: worst, wt = d, k
                break
        res["golden_step1"][f"{name}_{prec}"] = {"loss_rel": float(np.max(np.abs(losses[:len(refl)] - refl) / (np.abs(refl) + 1e-6))), "grad_relmax_worst": worst, "grad_relmax_worst_tensor": wt}
res["note"] = ("loss_rel = worst relative deviation of the step's loss scalars; grad_relmax_worst = worst over tensors of max|got - ref| / max|ref|. "
               "oracle_8192: one 8192-frame step vs oracle/vae_oracle.py in float64; golden_step1: step 1 of the vectors captured from the reference (fp32 torch CPU).")
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
