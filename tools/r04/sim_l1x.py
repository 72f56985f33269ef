#!/usr/bin/env python3
"""Which operand role carries the bf16x3 gradient residue, and what fp16 planes would buy in the one GEMM that carries it (CPU only).

tools/exp_precision.py simulates the M2 train step in float64 with the split-bf16 rounding applied per operand role.  Here:
  (1) every role split (= the kernel today) against one role at a time left exact / one role at a time split;
  (2) the L1 x GEMM with split-FP16 operands instead (x scaled per frame by a power of two so that the frame maximum lands in
      [2^13, 2^14), weights by 2^8; 11 + 11 mantissa bits per operand; products hi*hi + lo*hi + hi*lo as today), with and without
      flushing fp16 subnormals (v_mfma_f32_32x32x16_f16 on gfx950 keeps them: tools/r04/mfma_f16_denorm.hip, measured).

    python tools/r04/sim_l1x.py [B] [param_seed batch_seed]       (default 20000 11 12: the batch behind the 2.2e-4 of round 3)
"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import exp_precision as E
import golden_util as gu


def rh(a, ftz=False):
    """round to fp16 (nearest even), as float64; optionally flush subnormals"""
    with np.errstate(over="ignore"):
        out = np.asarray(a, np.float64).astype(np.float16).astype(np.float64)
    return np.where(np.abs(out) < 2.0 ** -14, 0.0, out) if ftz else out


def mm_f16(x, Wt, ftz, wscale=2.0 ** 8, top=14):
    mx = np.abs(x).max(axis=1, keepdims=True)
    s = 2.0 ** (top - np.ceil(np.log2(np.maximum(mx, 1e-300))))
    xs = x * s
    xh = rh(xs, ftz); xl = rh(xs - xh, ftz)
    ws = Wt * wscale
    wh = rh(ws, ftz); wl = rh(ws - wh, ftz)
    return (xh @ wh + xl @ wh + xh @ wl) / s / wscale


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    seeds = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (11, 12)
    dims = dict(x_dim=513, y_dim=513, z_dim=16, h_dim=(128, 128))
    P = gu.make_params("M2", dims, seeds[0])
    x, y, e = gu.make_batch(dims, B, seeds[1])
    x64, y64, e64 = x.astype(np.float64), y.astype(np.float64), e.astype(np.float64)
    EX, S = ("x", "x"), ("s", "s")
    exact = dict(fwd_x=EX, fwd_y=EX, fwd=EX, bwd=EX, wg_x=EX, wg_y=EX, wg=EX)
    allsplit = dict(fwd_x=S, fwd_y=("r", "s"), fwd=S, bwd=S, wg_x=S, wg_y=("s", "r"), wg=S)
    l0, g0 = E.step("M2", P, x64, y64, e64, exact)

    def rep(name, cfg):
        l, g = E.step("M2", P, x64, y64, e64, cfg)
        rel = {k: float(np.max(np.abs(g[k] - g0[k])) / np.max(np.abs(g0[k]))) for k in g0}
        top3 = sorted(rel.items(), key=lambda kv: -kv[1])[:3]
        print(f"{name:46s} loss {np.max(np.abs(l - l0) / np.abs(l0)):.1e}  " + "  ".join(f"{k} {v:.1e}" for k, v in top3), flush=True)

    print(f"M2 y513, B = {B}, seeds {seeds}: worst |g - g_exact| / max|g_exact| per tensor (three worst)")
    rep("every operand split bf16 (the kernel today)", allsplit)
    for role in ("fwd_x", "fwd_y", "fwd", "bwd", "wg_x", "wg_y", "wg"):
        c = dict(allsplit); c[role] = EX
        rep("  all split, EXACT " + role, c)
    for role in ("fwd_x", "fwd", "bwd", "wg_x"):
        c = dict(exact); c[role] = S
        rep("  all exact, SPLIT only " + role, c)
    orig = E.mm
    for ftz in (False, True):
        E.mm = lambda a, b, ma, mb, ftz=ftz: mm_f16(a, b, ftz) if ma == "h" else orig(a, b, ma, mb)
        c = dict(allsplit); c["fwd_x"] = ("h", "h")
        rep(f"L1 x GEMM in split fp16, subnormals {'flushed' if ftz else 'kept'}", c)
    E.mm = orig


if __name__ == "__main__":
    main()
