"""Every loss of packages/models/utils.py against golden values captured from the reference's own utils.py
(tests/golden/make_losszoo_golden.py): host tensors here, CUDA tensors (HIP kernels for elbo / BCE) in the gpu test."""
import os

import numpy as np
import pytest
import torch

import losszoo_inputs as li
from packages.models import utils as U

FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "losszoo_golden.npz"))


def run(device, rtol):
    for name, B, F, L, seed in li.CASES:
        d = li.make(B, F, L, seed)
        assert abs(li.checksum(d) - float(FIX[f"{name}/checksum"])) < 1e-6 * float(FIX[f"{name}/checksum"])
        dt = {k: torch.from_numpy(v).to(device) for k, v in d.items()}
        res = li.evaluate(U, dt)
        for k, v in res.items():
            np.testing.assert_allclose(v, FIX[f"{name}/{k}"], rtol=rtol, atol=1e-6, err_msg=f"{name}/{k}")


def test_loss_zoo_matches_reference_on_host():
    run("cpu", 1e-6)


@pytest.mark.gpu
def test_loss_zoo_matches_reference_on_gpu():
    run("cuda", 1e-4)


@pytest.mark.gpu
def test_loss_zoo_gradients_on_gpu_match_host_autograd():
    """Backward of every native loss (HIP kernels with hand-written gradients) against torch autograd of the reference's own
    expressions on the host, same inputs: L_loss (per-frame rows, weighted so that every row gradient differs), U_loss,
    binary_cross_entropy_2classes, the squared-error mask / signal losses and the magnitude-spectrum approximation."""
    name, B, F, L, seed = li.CASES[1]
    d = li.make(B, F, L, seed)
    wrow = torch.linspace(0.5, 1.5, B)

    def losses(dev):
        t = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
        leaf = {k: t[k].clone().requires_grad_(True) for k in ("r", "mu", "logvar", "p", "p2", "y_soft", "mask", "mask_hat")}
        w = wrow.to(dev)
        out = {}
        Lr, rec, kl = U.L_loss(t["x"], leaf["r"], leaf["mu"], leaf["logvar"], li.EPS)
        out["L"] = ((Lr * w).sum() + 0.3 * (rec * w).sum() - 0.2 * kl.sum(), ("r", "mu", "logvar"))
        out["isd"] = ((U.ikatura_saito_divergence(leaf["r"], t["x"], li.EPS) * w).sum(), ("r",))
        Uv, Lm, recm, klm = U.U_loss(t["x"], leaf["r"], leaf["mu"], leaf["logvar"], leaf["y_soft"], li.EPS)
        out["U"] = (Uv + 0.5 * Lm + 0.25 * recm - 0.1 * klm, ("r", "mu", "logvar", "y_soft"))
        out["bce_2c"] = (U.binary_cross_entropy_2classes(leaf["p"], leaf["p2"], t["t"], li.EPS), ("p", "p2"))
        out["mse_signal"] = (U.mean_square_error_signal(t["x"], leaf["mask"], leaf["mask_hat"]), ("mask", "mask_hat"))
        out["mse_mask"] = (U.mean_square_error_mask(leaf["mask"], leaf["mask_hat"]), ("mask", "mask_hat"))
        out["msa"] = (U.magnitude_spectrum_approxiamation_loss(t["x_c"], t["s_c"], leaf["mask_hat"]), ("mask_hat",))
        res = {}
        for k, (val, wrt) in out.items():
            gs = torch.autograd.grad(val, [leaf[n] for n in wrt], retain_graph=True)
            res[k] = (float(val), {n: g.detach().cpu().numpy() for n, g in zip(wrt, gs)})
        return res
    host, dev = losses("cpu"), losses("cuda")
    for k in host:
        np.testing.assert_allclose(dev[k][0], host[k][0], rtol=2e-5, err_msg=k)
        for n in host[k][1]:
            ref = host[k][1][n]
            np.testing.assert_allclose(dev[k][1][n], ref, rtol=2e-4, atol=2e-6 * float(np.abs(ref).max()), err_msg=f"{k}/{n}")


@pytest.mark.gpu
def test_loss_zoo_accepts_what_the_reference_expressions_accept():
    """the secondary losses' HIP kernels cover one dtype and one shape; a [B, 1] target broadcast against [B, F], float64 inputs
    and complex spectra that require grad are legal inputs of the reference's expressions (packages/models/utils.py:65-118) and must give
    the reference's values and gradients, not an exception -- they run as the ATen expression on the tensors' own device."""
    g = torch.Generator().manual_seed(5)
    B, F = 12, 33
    y = torch.rand(B, F, generator=g); yh = torch.rand(B, F, generator=g)
    y1 = torch.rand(B, 1, generator=g)
    r1 = torch.rand(B, F, generator=g) * 0.8 + 0.1; r2 = torch.rand(B, F, generator=g) * 0.8 + 0.1
    xc = torch.complex(torch.randn(B, F, generator=g), torch.randn(B, F, generator=g))
    sc = torch.complex(torch.randn(B, F, generator=g), torch.randn(B, F, generator=g))

    def run(dev):
        out = {}
        a = yh.to(dev).requires_grad_(True)
        v = U.mean_square_error_mask(y1.to(dev), a)                                  # broadcast target
        out["mask_b"] = (float(v), torch.autograd.grad(v, a)[0].cpu().numpy())
        a = yh.double().to(dev).requires_grad_(True)
        v = U.mean_square_error_mask(y.double().to(dev), a)                          # float64
        out["mask_f64"] = (float(v), torch.autograd.grad(v, a)[0].cpu().numpy())
        a = r1.to(dev).requires_grad_(True)
        v = U.binary_cross_entropy_2classes(a, r2.to(dev), y1.to(dev), 1e-8)        # broadcast target
        out["bce2_b"] = (float(v), torch.autograd.grad(v, a)[0].cpu().numpy())
        a = xc.to(dev).requires_grad_(True)
        v = U.magnitude_spectrum_approxiamation_loss(a, sc.to(dev), yh.to(dev))      # gradient with respect to the complex spectrum
        out["msa_cgrad"] = (float(v), torch.view_as_real(torch.autograd.grad(v, a)[0]).cpu().numpy())
        return out
    host, dev = run("cpu"), run("cuda")
    for k in host:
        np.testing.assert_allclose(dev[k][0], host[k][0], rtol=1e-5, err_msg=k)
        np.testing.assert_allclose(dev[k][1], host[k][1], rtol=1e-4, atol=1e-7, err_msg=k)
