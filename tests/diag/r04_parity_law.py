"""Round-4 parity diagnostics of the benchmarked operand policy (bf16x3).  GPU; test infrastructure (uses oracle/ as the checker).

(a) M2_info: is the bf16x3 gradient residue outside the "two worst rows" a ReLU-mask effect or operand precision?
    The float64 oracle gives every hidden pre-activation of the two ReLU nets (classifier on x, auxiliary net on z) and its
    *margin* |pre| / (sum_k |w_k in_k| + |b|): a frame whose smallest margin is below delta can take the other ReLU branch under
    any arithmetic whose relative product error is ~delta.  A flip in layer 2 moves EVERY row of the layer-1 weight gradient of
    that net (d pre1 = W2^T d pre2 * mask1), so "at most two rows per tensor" only holds for the layer a flip happens in.
    Report: (i) risky frames per delta, (ii) the raw batch, (iii) the same batch with the risky frames replaced by safe ones.
(b) M2 y513: worst gradient deviation against the float64 oracle per batch size, binary labels and full-mantissa labels.

    python tests/diag/r04_parity_law.py [out.json]
"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import golden_util as gu
from oracle import vae_oracle as vo
T = importlib.import_module("disentangled-vae_amd.trainer")

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r04_parity_law.json")
which = os.environ.get("PARITY_LAW", "ab")
t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
res = {}


def relu_margins(params, x, e):
    """float64: per frame, the smallest margin over all hidden units of the four ReLU layers (classifier L1, L2; auxiliary L1, L2)."""
    p = {k: v.astype(np.float64) for k, v in params.items()}
    x = x.astype(np.float64)
    enc = vo.encoder_fwd(p, "enc_dec_clf.encoder.", x, e.astype(np.float64))
    worst = np.full(x.shape[0], np.inf)
    per_layer = {}
    for prefix, inp in (("enc_dec_clf.classifier.", x), ("auxiliary.", enc["z"])):
        h = inp
        for name in vo._hidden_names(p, prefix):
            W, b = p[name + ".weight"], p[name + ".bias"]
            pre = h @ W.T + b
            mag = np.abs(h) @ np.abs(W).T + np.abs(b) + 1e-300
            m = (np.abs(pre) / mag).min(axis=1)
            per_layer[name] = m
            worst = np.minimum(worst, m)
            h = np.maximum(pre, 0)
    return worst, per_layer


def info_report(params, x, y, e, prec, tag):
    dims = dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128))
    B = x.shape[0]
    p32 = {k: v.astype(np.float32) for k, v in params.items()}
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    o32, a1, a2 = vo.m2info_losses_and_grads(p32, x, y, e, 0.5, 10.0, 1.0)
    o64, b1, b2 = vo.m2info_losses_and_grads(p64, x.astype(np.float64), y.astype(np.float64), e.astype(np.float64), 0.5, 10.0, 1.0)
    tot = lambda g1, g2, k: np.asarray(g1[k], np.float64) + (np.asarray(g2[k], np.float64) if k in g2 else 0.0)
    tr = T.Trainer("M2_info", dims, params, batch=B, precision=prec, alpha=0.5, beta=10.0, gamma=1.0)
    tr.step(t(x), t(y), t(e))
    g = tr.grads_numpy()
    rep = {}
    for k in params:
        r32, r64 = tot(a1, a2, k).reshape(g[k].shape), tot(b1, b2, k).reshape(g[k].shape)
        e32 = np.abs(g[k] - r32) / np.abs(r32).max(); e64 = np.abs(g[k] - r64) / np.abs(r64).max()
        rows = np.sort(e32.reshape(e32.shape[0], -1).max(axis=1))[::-1]
        rep[k] = dict(vs_f32=float(e32.max()), vs_f64=float(e64.max()), oracle_f32_vs_f64=float(np.abs(r32 - r64).max() / np.abs(r64).max()),
                      rows_over_2e4=int((rows >= 2e-4).sum()), rows_over_1e4=int((rows >= 1e-4).sum()), third_worst_row=float(rows[min(2, rows.size - 1)]))
    worst = max(rep, key=lambda k: rep[k]["vs_f32"])
    print(f"[{tag}] {prec:7s} worst tensor {worst} {rep[worst]['vs_f32']:.2e}; tensors over 2e-4: "
          f"{[(k.split('.')[-3] + '.' + k.split('.')[-2] + '.' + k.split('.')[-1], round(v['vs_f32'], 6), v['rows_over_2e4']) for k, v in rep.items() if v['vs_f32'] >= 2e-4]}", flush=True)
    return dict(worst=rep[worst]["vs_f32"], worst_tensor=worst, per_tensor=rep)


if "a" in which:
    dims = dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128))
    B = 8192
    res["m2info_ties"] = {}
    for seed in (31, 131):
        params = gu.make_params("M2_info", dims, seed)
        x, y, e = gu.make_batch(dims, B, seed + 1)
        worst, per_layer = relu_margins(params, x, e)
        entry = {"risky_frames": {f"{d:g}": int((worst < d).sum()) for d in (1e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3)},
                 "risky_by_layer_1e-4": {k: int((m < 1e-4).sum()) for k, m in per_layer.items()}}
        print(f"seed {seed}: risky frames by delta {entry['risky_frames']}; by layer at 1e-4 {entry['risky_by_layer_1e-4']}", flush=True)
        for prec in ("fp32", "bf16x3"):
            entry[f"raw_{prec}"] = info_report(params, x, y, e, prec, f"seed {seed} raw")
        for delta in (1e-4, 1e-3):
            xs, ys, es = x.copy(), y.copy(), e.copy()
            for it in range(4):                                   # z of a replaced frame changes with its noise: iterate to a fixed point
                worst, _ = relu_margins(params, xs, es)
                risky = np.flatnonzero(worst < delta)
                if risky.size == 0:
                    break
                safe = np.flatnonzero(worst >= 4 * delta)
                src = safe[(np.arange(risky.size) * 7919) % safe.size]
                xs[risky], ys[risky], es[risky] = xs[src], ys[src], es[src]
            entry[f"clean_{delta:g}_replaced"] = int((np.abs(xs - x).max(axis=1) > 0).sum())
            for prec in ("fp32", "bf16x3"):
                entry[f"clean_{delta:g}_{prec}"] = info_report(params, xs, ys, es, prec, f"seed {seed} tie-free(delta {delta:g}, {entry[f'clean_{delta:g}_replaced']} frames replaced)")
        res["m2info_ties"][f"seed{seed}"] = entry

if "b" in which:
    dims = dict(x_dim=513, y_dim=513, z_dim=16, h_dim=(128, 128))
    params = gu.make_params("M2", dims, 11)
    res["m2_y513_by_batch"] = {}
    relmax = lambda a, b: float(np.max(np.abs(np.asarray(a, np.float64) - b)) / (np.max(np.abs(b)) + 1e-30))
    # PARITY_LAW_BIG=1: also the bench lines' large batches (the float64 oracle of 2^20 frames takes ~40 s and ~60 GB of host memory)
    for B in (1000, 8192, 20000, 65536) + ((262144, 1048576) if os.environ.get("PARITY_LAW_BIG") == "1" else ()):
        x, y, e = gu.make_batch(dims, B, 12)
        soft = np.random.default_rng(3).random(y.shape).astype(np.float32)
        for labels, yy in (("binary", y), ("soft", soft)):
            t0 = time.time()
            p = {k: v.copy() for k, v in params.items()}
            outo, grads = vo.train_step_vae("M2", p, vo.AdamState(list(p)), x.astype(np.float64), yy.astype(np.float64), e.astype(np.float64))
            ref = np.array([outo["loss"], outo["recon"], outo["kl"]])
            t_or = time.time() - t0
            for prec in ("fp32", "bf16x3"):
                tr = T.Trainer("M2", dims, params, batch=B, precision=prec)
                losses = tr.step(t(x), t(yy), t(e)).cpu().numpy()[:3]
                g = tr.grads_numpy()
                per = {k: relmax(g[k], np.asarray(grads[k], np.float64).reshape(g[k].shape)) for k in grads}
                # error relative to the rms of the tensor as well: the maximum of a tensor is one element, the rms the whole
                per_rms = {k: float(np.sqrt(np.mean((g[k].astype(np.float64) - np.asarray(grads[k], np.float64).reshape(g[k].shape)) ** 2)) /
                                    (np.sqrt(np.mean(np.asarray(grads[k], np.float64) ** 2)) + 1e-30)) for k in grads}
                wk = max(per, key=per.get)
                res["m2_y513_by_batch"][f"B{B}_{labels}_{prec}"] = dict(loss_rel=float(np.max(np.abs(losses - ref) / np.abs(ref))), grad_relmax_worst=per[wk],
                                                                        grad_relmax_worst_tensor=wk, grad_relrms_worst=max(per_rms.values()),
                                                                        grad_max_of_worst=float(np.abs(np.asarray(grads[wk])).max()), per_tensor=per)
                print(f"B {B:6d} {labels:6s} {prec:7s} loss {res['m2_y513_by_batch'][f'B{B}_{labels}_{prec}']['loss_rel']:.1e} grad worst {per[wk]:.2e} ({wk}, max {np.abs(np.asarray(grads[wk])).max():.3e}) "
                      f"rms-rel worst {max(per_rms.values()):.2e}  [oracle {t_or:.0f}s]", flush=True)
                del tr
            torch.cuda.empty_cache()

os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print("wrote", out_path)
