"""Seeded inputs and the call list for the loss-zoo parity test (shared by the golden generator, which runs the
REFERENCE's packages/models/utils.py, and the tests, which run the drop-in's)."""
import numpy as np
import torch

CASES = [("small", 7, 37, 5, 1), ("full", 33, 513, 16, 2)]
EPS = 1e-8


def make(B, F, L, seed):
    rng = np.random.default_rng(seed)
    f32 = lambda a: a.astype(np.float32)
    d = dict(
        x=f32(rng.standard_normal((B, F)) ** 2), r=f32(np.exp(rng.standard_normal((B, F)) * 0.5)),
        mu=f32(rng.standard_normal((B, L))), logvar=f32(rng.standard_normal((B, L)) * 0.3),
        p=f32(rng.random((B, F)) * 0.98 + 0.01), p2=f32(rng.random((B, F)) * 0.98 + 0.01), t=f32(rng.random((B, F)) > 0.5),
        y_soft=f32(rng.random((B, 1)) * 0.9 + 0.05), mask=f32(rng.random((B, F))), mask_hat=f32(rng.random((B, F))),
        s_c=(rng.standard_normal((B, F)) + 1j * rng.standard_normal((B, F))).astype(np.complex64),
        x_c=(rng.standard_normal((B, F)) + 1j * rng.standard_normal((B, F))).astype(np.complex64),
        hard=(rng.random(B * 9) > 0.5).astype(np.float32), truth=(rng.random(B * 9) > 0.4).astype(np.float32),
        lse=f32(rng.standard_normal((B, F)) * 3),
    )
    return d


def checksum(d):
    return float(sum(np.abs(v).astype(np.float64).sum() for v in d.values()))


def evaluate(U, d):
    """Every public loss of utils.py; values as numpy (scalars or per-row vectors)."""
    n = lambda t: t.detach().cpu().numpy()
    out = {}
    out["bce"] = n(U.binary_cross_entropy(d["p"], d["t"], EPS))
    out["bce_v2"] = n(U.binary_cross_entropy_v2(d["p"], EPS))
    out["bce_v3"] = n(U.binary_cross_entropy_v3(d["p"], EPS))
    out["bce_2c"] = n(U.binary_cross_entropy_2classes(d["p"], d["p2"], d["t"], EPS))
    out["isd"] = n(U.ikatura_saito_divergence(d["r"], d["x"], EPS))
    for k, v in zip(("elbo", "elbo_recon", "elbo_kl"), U.elbo(d["x"], d["r"], d["mu"], d["logvar"], EPS)):
        out[k] = n(v)
    for k, v in zip(("L", "L_recon", "L_kl"), U.L_loss(d["x"], d["r"], d["mu"], d["logvar"], EPS)):
        out[k] = n(v)
    for k, v in zip(("U", "U_L", "U_recon", "U_kl"), U.U_loss(d["x"], d["r"], d["mu"], d["logvar"], d["y_soft"], EPS)):
        out[k] = n(v)
    out["mse_signal"] = n(U.mean_square_error_signal(d["x"], d["mask"], d["mask_hat"]))
    out["mse_mask"] = n(U.mean_square_error_mask(d["mask"], d["mask_hat"]))
    out["msa"] = n(U.magnitude_spectrum_approxiamation_loss(d["x_c"], d["s_c"], d["mask_hat"]))
    for k, v in zip(("f1_acc", "f1_prec", "f1_rec", "f1"), U.f1_loss(d["hard"], d["truth"])):
        out[k] = n(v)
    out["lse"] = n(U.log_sum_exp(d["lse"]))
    return out
