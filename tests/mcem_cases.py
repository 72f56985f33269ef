"""MCEM parity cases shared by tests/golden/make_mcem_golden.py (runs the REFERENCE) and the parity tests.
Inputs are regenerated from seeds; the fixture stores a checksum of them."""
import numpy as np

EPS = np.finfo(float).eps          # scripts/evaluate_ntcd_M2.py: eps = np.finfo(float).eps

DIMS = {
    "M1": dict(x_dim=513, y_dim=0, z_dim=16, h_dim=(128, 128)),
    "M2": dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128)),
    "M2_info": dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128)),
}

CASES = [
    dict(name="M1", model="M1", N=45, K=10, niter=2, n_e=3, b_e=4, n_wf=4, b_wf=3, seed=11, wscale=1.0),
    dict(name="M2_y1", model="M2", N=45, K=10, niter=2, n_e=3, b_e=4, n_wf=4, b_wf=3, seed=12, wscale=1.0),
    dict(name="M2v3_y1", model="M2_info", N=70, K=10, niter=2, n_e=3, b_e=4, n_wf=4, b_wf=3, seed=13, wscale=1.5),
]


def effective_counts(case):
    """(nsamples, burnin) actually run per E-step and for the Wiener filter.  MCEM_M1 passes
    (Z, nsamples, burnin) into sample_posterior(Z, y, nsamples=10, burnin=30) (mcem.py:207,297-298,314-315):
    nsamples <- its burnin argument, burnin <- the default 30."""
    if case["model"] == "M1":
        return case["b_e"], 30, case["b_wf"], 30
    return case["n_e"], case["b_e"], case["n_wf"], case["b_wf"]


def make_utterance(case):
    """Synthetic complex STFTs X (mixture), S (clean), (F, N) complex64, and labels y (y_dim, N) float32."""
    rng = np.random.default_rng(case["seed"] + 500)
    F, N = 513, case["N"]
    env = np.exp(rng.standard_normal((F, 1)) * 0.7 - 1.0) * np.exp(rng.standard_normal((1, N)) * 0.5)
    S = (np.sqrt(env / 2) * (rng.standard_normal((F, N)) + 1j * rng.standard_normal((F, N)))).astype(np.complex64)
    noise = (0.3 * (rng.standard_normal((F, N)) + 1j * rng.standard_normal((F, N)))).astype(np.complex64)
    X = (S + noise).astype(np.complex64)
    ydim = DIMS[case["model"]]["y_dim"]
    y = (rng.random((max(ydim, 1), N)) > 0.4).astype(np.float32)
    return X, S, (y if ydim else None)


def checksum(X, S, y):
    tot = float(np.abs(X).astype(np.float64).sum() + np.abs(S).astype(np.float64).sum())
    return tot + (float(y.sum()) if y is not None else 0.0)
