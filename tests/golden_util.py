"""Deterministic input/parameter generators shared by tests/golden/make_golden.py
(which runs the REFERENCE on them, in the build container) and by the parity
tests (which run the oracle / the HIP path on the same arrays).

numpy Generator streams (PCG64 + ziggurat normals) are used so the arrays can
be regenerated from a seed instead of being committed; every fixture also
stores a float64 checksum of what was generated, asserted by the tests.
"""
import math
import numpy as np

# (name, model, dims dict, batch, wscale)
CASES = [
    ("M1_small", "M1", dict(x_dim=37, y_dim=0, z_dim=5, h_dim=(24, 16)), 7, 1.0),
    ("M2_small_y1", "M2", dict(x_dim=37, y_dim=1, z_dim=5, h_dim=(24, 16)), 7, 1.0),
    ("M2_small_y37", "M2", dict(x_dim=37, y_dim=37, z_dim=5, h_dim=(24, 16)), 5, 2.0),
    ("M2info_small", "M2_info", dict(x_dim=37, y_dim=1, z_dim=5, h_dim=(24, 16)), 7, 1.5),
    ("M1_full", "M1", dict(x_dim=513, y_dim=0, z_dim=16, h_dim=(128, 128)), 32, 1.0),
    ("M2_full_y1", "M2", dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128)), 32, 1.0),
    ("M2_full_y513", "M2", dict(x_dim=513, y_dim=513, z_dim=16, h_dim=(128, 128)), 32, 1.0),
    ("M2_full_y513_hot", "M2", dict(x_dim=513, y_dim=513, z_dim=16, h_dim=(128, 128)), 5, 2.5),
    ("M2info_full", "M2_info", dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128)), 32, 1.0),
    ("M2info_full_b1", "M2_info", dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128)), 1, 1.0),
]
NSTEPS = 3
SAMPLE_STRIDE = 61      # strided sample kept from big gradient / parameter tensors
FULL_KEEP = 4096        # tensors up to this many elements are stored whole

# The BENCHMARKED batch (BASELINE.json configs 1-3: 8192 frames per step) captured from the reference itself, tests/golden/vae_golden_b8192.npz
# (make_golden.py --big): loop bodies scripts/training_M1.py:125-139, training_M2.py:132-147, training_M2_info_vad.py:153-198.  Outputs are
# stored as strided samples + moments like the gradients (r alone would be 16 MB).
_D513 = dict(x_dim=513, z_dim=16, h_dim=(128, 128))
BIG_CASES = [
    ("M2_full_y513_B8192", "M2", dict(_D513, y_dim=513), 8192, 1.0),
    ("M1_full_B8192", "M1", dict(_D513, y_dim=0), 8192, 1.0),
    ("M2info_full_B8192", "M2_info", dict(_D513, y_dim=1), 8192, 1.0),
]
OUT_STRIDE = 997        # strided sample kept from r [8192, 513] of the big cases


def case_seed(name):
    """Seed of a case's parameter / batch streams: 100 + index in CASES, 200 + index in BIG_CASES."""
    names = [c[0] for c in CASES]
    if name in names:
        return 100 + names.index(name)
    return 200 + [c[0] for c in BIG_CASES].index(name)


def layer_dims(model, x_dim, y_dim, z_dim, h_dim):
    """state_dict name -> shape in the reference's registration order
    (verified against the reference in make_golden.py)."""
    h = list(h_dim)
    rh = list(reversed(h))
    out = []

    def stack(prefix, d, hs):
        for i, n in enumerate(hs):
            out.append((f"{prefix}hidden.{i}.weight", (n, d)))
            out.append((f"{prefix}hidden.{i}.bias", (n,)))
            d = n
        return d

    def enc(prefix, inp):
        d = stack(prefix, inp, h)
        for nm in ("mu", "log_var"):
            out.append((f"{prefix}sample.{nm}.weight", (z_dim, d)))
            out.append((f"{prefix}sample.{nm}.bias", (z_dim,)))

    def dec(prefix, inp):
        d = stack(prefix, inp, rh)
        out.append((f"{prefix}reconstruction.weight", (x_dim, d)))
        out.append((f"{prefix}reconstruction.bias", (x_dim,)))

    def clf(prefix, inp):
        d = stack(prefix, inp, h)
        out.append((f"{prefix}output_layer.weight", (y_dim, d)))
        out.append((f"{prefix}output_layer.bias", (y_dim,)))

    if model == "M1":
        enc("encoder.", x_dim); dec("decoder.", z_dim)
    elif model == "M2":
        enc("encoder.", x_dim + y_dim); dec("decoder.", z_dim + y_dim)
    elif model == "M2_info":
        enc("enc_dec_clf.encoder.", x_dim); dec("enc_dec_clf.decoder.", z_dim + y_dim)
        clf("enc_dec_clf.classifier.", x_dim); clf("auxiliary.", z_dim)
    else:
        raise ValueError(model)
    return out


def make_params(model, dims, seed, wscale=1.0):
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in layer_dims(model, **dims):
        if name.endswith("weight"):
            std = wscale * math.sqrt(2.0 / (shape[0] + shape[1]))
            p[name] = (rng.standard_normal(shape) * std).astype(np.float32)
        else:
            p[name] = (rng.standard_normal(shape) * 0.05).astype(np.float32)
    return p


import importlib as _importlib
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
make_batch = _importlib.import_module("disentangled-vae_amd.synth").make_batch    # the benchmark's generator: one definition


def checksum(arrs):
    s = 0.0
    for a in arrs:
        if a is None:
            continue
        a64 = np.asarray(a, dtype=np.float64).ravel()
        s += float(np.sum(a64 * (1.0 + (np.arange(a64.size) % 7))))
    return s


def summarize(a, stride=SAMPLE_STRIDE):
    """Whole tensor if small, else strided sample + moments (float64)."""
    a = np.asarray(a)
    flat = a.ravel()
    if flat.size <= FULL_KEEP:
        return dict(full=flat.astype(np.float32))
    f64 = flat.astype(np.float64)
    return dict(sample=flat[::stride].astype(np.float32),
                moments=np.array([f64.sum(), np.abs(f64).sum(), (f64 * f64).sum()], dtype=np.float64))


def compare_params(name, got, fix, key, rtol, atol, max_bad_frac, hard_atol):
    """Post-Adam parameters.  Adam's first steps are sign-like (lr*g/(|g|+1e-8)), so an element
    whose gradient is of order 1e-8 or is a saturated-tanh few-ulp quantity moves by a different
    fraction of one lr step in any two correct fp32 implementations.  Require: every element
    within `hard_atol` (steps * lr), and all but `max_bad_frac` of them within rtol/atol."""
    flat = np.asarray(got).ravel()
    if key + "/full" in fix:
        ref = fix[key + "/full"]
    else:
        ref = fix[key + "/sample"]
        flat = flat[::SAMPLE_STRIDE]
    err = np.abs(flat.astype(np.float64) - ref.astype(np.float64))
    assert err.max() <= hard_atol, (name, err.max())
    bad = float(np.mean(err > atol + rtol * np.abs(ref)))
    assert bad <= max_bad_frac, (name, bad, err.max())


def compare_summary(name, got, fix, key, rtol, atol, atol_rel=0.0, stride=SAMPLE_STRIDE):
    """Assert `got` matches the stored summary fix[key + ...].
    Tolerance per element: rtol*|ref| + atol + atol_rel*max|ref| (the last term
    covers elements that are small only through cancellation in a batch sum)."""
    got = np.asarray(got)
    flat = got.ravel()
    if key + "/full" in fix:
        ref = fix[key + "/full"]
        np.testing.assert_allclose(flat, ref, rtol=rtol, atol=atol + atol_rel * float(np.max(np.abs(ref))), err_msg=name)
        return
    ref = fix[key + "/sample"]
    mtol = 10 * max(rtol, atol_rel if rtol == 0.0 else 0.0)          # moments: ten times the elementwise relative bound
    atol = atol + atol_rel * float(np.max(np.abs(ref)))
    np.testing.assert_allclose(flat[::stride], ref, rtol=rtol, atol=atol, err_msg=name)
    f64 = flat.astype(np.float64)
    mom = np.array([f64.sum(), np.abs(f64).sum(), (f64 * f64).sum()])
    ref = fix[key + "/moments"]
    # moments: |sum| is cancellation-prone, compare it against the abs-sum scale
    assert abs(mom[0] - ref[0]) <= mtol * ref[1] + atol, (name, mom, ref)
    np.testing.assert_allclose(mom[1:], ref[1:], rtol=mtol, err_msg=name)
