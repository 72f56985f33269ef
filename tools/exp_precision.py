#!/usr/bin/env python3
"""Which MFMA operands need a bf16 hi/lo split to hold the parity bar?  (design experiment, CPU only)

Simulates the M2 / M1 train step in float64 with bf16 rounding applied to chosen GEMM operands
(accumulation stays wide, as the fp32 MFMA accumulators are ~1e-7) and reports, against the unrounded
float64 step: relative error of the losses and max|g - g_ref| / max|g_ref| per gradient tensor -- the
two quantities tests/test_gpu_fused.py asserts.

A GEMM "mode" is a pair (activation operand, weight operand), each one of
    'r'  rounded to bf16 once           (1 MFMA)
    's'  hi + lo bf16 pair              (the product then takes 2 MFMAs; 's','s' takes 3: hi*hi + lo*hi + hi*lo)
    'x'  exact
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu  # noqa: E402


def rb(a):
    """round-to-nearest-even to bf16, returned as float64"""
    f = np.ascontiguousarray(a, dtype=np.float32)
    u = f.view(np.uint32)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16).astype(np.uint32)
    return r.view(np.float32).astype(np.float64)


def op(a, mode):
    """operand as the list of bf16 terms it is represented by"""
    if mode == "x":
        return [np.asarray(a, np.float64)]
    hi = rb(a)
    if mode == "r":
        return [hi]
    return [hi, rb(np.asarray(a, np.float64) - hi)]


def mm(a, b, ma, mb):
    """a @ b with operand modes; 's','s' drops the lo*lo term like the 3-MFMA form"""
    A, Bm = op(a, ma), op(b, mb)
    out = A[0] @ Bm[0]
    if len(A) > 1:
        out = out + A[1] @ Bm[0]
    if len(Bm) > 1:
        out = out + A[0] @ Bm[1]
    return out


def step(model, P, x, y, e, cfg):
    """cfg: dict of modes.  fwd_x: (x, W1x); fwd_y: (y, W); fwd: (act, W) hidden layers; bwd: (dpre, W^T);
    wg_x: (dpre, x); wg_y: (dpre, y); wg: (dpre, act)."""
    B = x.shape[0]
    pre = "encoder."
    W1 = P[pre + "hidden.0.weight"].astype(np.float64)
    xd = x.shape[1]
    g = {}
    f = lambda k: P[k].astype(np.float64)
    pre1 = mm(x, W1[:, :xd].T, *cfg["fwd_x"]) + f(pre + "hidden.0.bias")
    if model == "M2":
        pre1 = pre1 + mm(y, W1[:, xd:].T, *cfg["fwd_y"])
    h1 = np.tanh(pre1)
    h2 = np.tanh(mm(h1, f(pre + "hidden.1.weight").T, *cfg["fwd"]) + f(pre + "hidden.1.bias"))
    mu = mm(h2, f(pre + "sample.mu.weight").T, *cfg["fwd"]) + f(pre + "sample.mu.bias")
    lv = mm(h2, f(pre + "sample.log_var.weight").T, *cfg["fwd"]) + f(pre + "sample.log_var.bias")
    sd = np.exp(0.5 * lv)
    z = mu + sd * e
    W3 = f("decoder.hidden.0.weight")
    pre3 = mm(z, W3[:, :16].T, *cfg["fwd"]) + f("decoder.hidden.0.bias")
    if model == "M2":
        pre3 = pre3 + mm(y, W3[:, 16:].T, *cfg["fwd_y"])
    d1 = np.tanh(pre3)
    d2 = np.tanh(mm(d1, f("decoder.hidden.1.weight").T, *cfg["fwd"]) + f("decoder.hidden.1.bias"))
    a = mm(d2, f("decoder.reconstruction.weight").T, *cfg["fwd"]) + f("decoder.reconstruction.bias")
    xe = x * np.exp(-a)
    recon = np.mean(np.sum(xe - np.log(x + 1e-8) + a - 1, axis=1))
    kl = -0.5 * np.mean(np.sum(lv - mu ** 2 - np.exp(lv), axis=1))
    # backward
    da = (1 - xe) / B
    g["decoder.reconstruction.weight"] = mm(da.T, d2, *cfg["wg"]); g["decoder.reconstruction.bias"] = da.sum(0)
    dd2 = mm(da, f("decoder.reconstruction.weight"), *cfg["bwd"]) * (1 - d2 * d2)
    g["decoder.hidden.1.weight"] = mm(dd2.T, d1, *cfg["wg"]); g["decoder.hidden.1.bias"] = dd2.sum(0)
    dd1 = mm(dd2, f("decoder.hidden.1.weight"), *cfg["bwd"]) * (1 - d1 * d1)
    gw3 = mm(dd1.T, z, *cfg["wg"])
    if model == "M2":
        gw3 = np.concatenate([gw3, mm(dd1.T, y, *cfg["wg_y"])], axis=1)
    g["decoder.hidden.0.weight"] = gw3; g["decoder.hidden.0.bias"] = dd1.sum(0)
    dz = mm(dd1, W3[:, :16], *cfg["bwd"])
    dmu = dz + mu / B
    dlv = dz * e * sd * 0.5 - 0.5 * (1 - np.exp(lv)) / B
    g[pre + "sample.mu.weight"] = mm(dmu.T, h2, *cfg["wg"]); g[pre + "sample.mu.bias"] = dmu.sum(0)
    g[pre + "sample.log_var.weight"] = mm(dlv.T, h2, *cfg["wg"]); g[pre + "sample.log_var.bias"] = dlv.sum(0)
    dh2 = (mm(dmu, f(pre + "sample.mu.weight"), *cfg["bwd"]) + mm(dlv, f(pre + "sample.log_var.weight"), *cfg["bwd"])) * (1 - h2 * h2)
    g[pre + "hidden.1.weight"] = mm(dh2.T, h1, *cfg["wg"]); g[pre + "hidden.1.bias"] = dh2.sum(0)
    dh1 = mm(dh2, f(pre + "hidden.1.weight"), *cfg["bwd"]) * (1 - h1 * h1)
    gw1 = mm(dh1.T, x, *cfg["wg_x"])
    if model == "M2":
        gw1 = np.concatenate([gw1, mm(dh1.T, y, *cfg["wg_y"])], axis=1)
    g[pre + "hidden.0.weight"] = gw1; g[pre + "hidden.0.bias"] = dh1.sum(0)
    return np.array([recon + kl, recon, kl]), g


EX = ("x", "x")
CONFIGS = {
    # name: fwd_x, fwd_y, fwd, bwd, wg_x, wg_y, wg     (activation-side operand, weight/other operand)
    "plain bf16 (round 1)":            dict(fwd_x=("r", "r"), fwd_y=("r", "r"), fwd=("r", "r"), bwd=("r", "r"), wg_x=("r", "r"), wg_y=("r", "r"), wg=("r", "r")),
    "x split in L1 + wgrad only":      dict(fwd_x=("s", "s"), fwd_y=("r", "s"), fwd=("r", "r"), bwd=("r", "r"), wg_x=("r", "s"), wg_y=("r", "r"), wg=("r", "r")),
    "weights split fwd, x split":      dict(fwd_x=("s", "s"), fwd_y=("r", "s"), fwd=("r", "s"), bwd=("r", "r"), wg_x=("r", "s"), wg_y=("r", "r"), wg=("r", "r")),
    "weights split fwd+bwd, x split":  dict(fwd_x=("s", "s"), fwd_y=("r", "s"), fwd=("r", "s"), bwd=("r", "s"), wg_x=("r", "s"), wg_y=("r", "r"), wg=("r", "r")),
    "  + wgrad dpre split":            dict(fwd_x=("s", "s"), fwd_y=("r", "s"), fwd=("r", "s"), bwd=("r", "s"), wg_x=("s", "s"), wg_y=("s", "r"), wg=("s", "r")),
    "  + wgrad both split":            dict(fwd_x=("s", "s"), fwd_y=("r", "s"), fwd=("r", "s"), bwd=("r", "s"), wg_x=("s", "s"), wg_y=("s", "r"), wg=("s", "s")),
    "everything split (3 MFMA)":       dict(fwd_x=("s", "s"), fwd_y=("r", "s"), fwd=("s", "s"), bwd=("s", "s"), wg_x=("s", "s"), wg_y=("s", "r"), wg=("s", "s")),
    "acts split, weights rounded":     dict(fwd_x=("s", "r"), fwd_y=("r", "r"), fwd=("s", "r"), bwd=("s", "r"), wg_x=("s", "s"), wg_y=("s", "r"), wg=("s", "s")),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="M2")
    ap.add_argument("--y-dim", type=int, default=513)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--seeds", type=int, nargs=2, default=[11, 12])
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    dims = dict(x_dim=513, y_dim=a.y_dim if a.model == "M2" else 0, z_dim=16, h_dim=(128, 128))
    P = gu.make_params(a.model, dims, a.seeds[0])
    x, y, e = gu.make_batch(dims, a.batch, a.seeds[1])
    x64, e64 = x.astype(np.float64), e.astype(np.float64)
    y64 = None if y is None else y.astype(np.float64)
    exact = dict(fwd_x=EX, fwd_y=EX, fwd=EX, bwd=EX, wg_x=EX, wg_y=EX, wg=EX)
    l0, g0 = step(a.model, P, x64, y64, e64, exact)
    print(f"{a.model} y{dims['y_dim']} B={a.batch}: reference losses {l0}")
    for name, cfg in CONFIGS.items():
        if a.only and a.only not in name:
            continue
        l, g = step(a.model, P, x64, y64, e64, cfg)
        lrel = np.abs(l - l0) / np.abs(l0)
        rel = {k: float(np.max(np.abs(g[k] - g0[k])) / np.max(np.abs(g0[k]))) for k in g0}
        worst = max(rel, key=rel.get)
        print(f"{name:34s} loss rel {lrel.max():.2e}   grad relmax worst {rel[worst]:.2e} ({worst})   median {np.median(list(rel.values())):.2e}")
        if a.only:
            for k, v in sorted(rel.items(), key=lambda kv: -kv[1]):
                print(f"      {k:36s} {v:.2e}")


if __name__ == "__main__":
    main()
