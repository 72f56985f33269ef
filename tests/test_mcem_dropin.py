"""packages/models/mcem.py (drop-in for the reference's MCEM classes).

CPU: a seeded run reproduces the golden vectors captured from the reference with the same seed (same
draw order).  GPU: the device path, fed the reference's recorded draws, reproduces the reference's state
after every EM iteration; with its own device draws it converges to the same cost level."""
import os

import numpy as np
import pytest
import torch

import golden_util as gu
import mcem_cases as mc
from impl_modules import build_model
from packages.models import mcem

FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "mcem_golden.npz"))
CLS = {"M1": mcem.MCEM_M1, "M2": mcem.MCEM_M2, "M2_info": mcem.MCEM_M2v3}


def case_fix(name):
    return {k.split("/", 1)[1]: FIX[k] for k in FIX.files if k.startswith(name + "/")}


def make_em(case, device):
    dims = mc.DIMS[case["model"]]
    params = gu.make_params(case["model"], dims, case["seed"], case["wscale"])
    m = build_model(case["model"], dims)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    m.eval().to(device)
    for p in m.parameters():
        p.requires_grad = False
    vae = m.enc_dec_clf if case["model"] == "M2_info" else m
    X, S, y = mc.make_utterance(case)
    em = CLS[case["model"]](niter=case["niter"], nsamples_E_step=case["n_e"], burnin_E_step=case["b_e"],
                            nsamples_WF=case["n_wf"], burnin_WF=case["b_wf"], var_RW=0.01)
    kw = dict(X=X, S=S, vae=vae, nmf_rank=case["K"], eps=mc.EPS, device=device)
    if case["model"] != "M1":
        kw["y"] = torch.from_numpy(y).to(device)
    return em, kw, X


@pytest.mark.parametrize("case", mc.CASES, ids=[c["name"] for c in mc.CASES])
def test_cpu_run_reproduces_reference_with_same_seed(case):
    fix = case_fix(case["name"])
    em, kw, X = make_em(case, "cpu")
    torch.manual_seed(case["seed"] + 1000)
    em.init_parameters(**kw)
    np.testing.assert_allclose(em.Z.numpy(), fix["Z0"], rtol=1e-5, atol=5e-6)      # host BLAS differs by a few ulp between CPU models (1.7e-6 seen on a GPU box)
    cost = em.run()
    np.testing.assert_allclose(cost, fix["cost"], rtol=1e-5)
    np.testing.assert_allclose(em.W.numpy(), fix["W"][-1], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(em.H.numpy(), fix["H"][-1], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(em.g.numpy(), fix["g"][-1], rtol=1e-4)
    np.testing.assert_allclose(em.S_hat, fix["S_hat"], rtol=1e-3, atol=1e-5)
    assert em.S_hat.shape == X.shape and em.N_hat.shape == X.shape and em.S_hat.dtype == X.dtype


def test_importing_seeds_the_generators():
    import importlib
    torch.manual_seed(123); np.random.seed(123)
    importlib.reload(mcem)
    a, b = torch.rand(1).item(), np.random.rand()
    torch.manual_seed(0); np.random.seed(0)
    assert a == torch.rand(1).item() and b == np.random.rand()


def test_operand_policy_of_unchanged_scripts_comes_from_the_environment():
    """The reference's evaluate scripts construct MCEM_M2(...) and never set `.precision`: DVAE_MCEM_PRECISION picks the device path's operand
    policy for them (default: exact fp32 products)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "from packages.models import mcem; print(mcem.MCEM_M2(niter=1).precision)"
    for env_val, want in ((None, "fp32"), ("bf16x3", "bf16x3")):
        env = {k: v for k, v in os.environ.items() if k != "DVAE_MCEM_PRECISION"}
        if env_val is not None:
            env["DVAE_MCEM_PRECISION"] = env_val
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip().splitlines()[-1] == want


def test_m1_quirk_counts():
    """MCEM_M1 hands (Z, nsamples, burnin) to sample_posterior(Z, y, nsamples, burnin=30): kept samples = burn-in argument."""
    case = mc.CASES[0]
    em, kw, _ = make_em(case, "cpu")
    em.init_parameters(**kw)
    em.E_step()
    assert em.Vs.shape == (case["b_e"], 513, case["N"])


class FeedDraws:
    """Replaces torch.rand / torch.randn by a queue of recorded draws (moved to the requested device)."""

    def __init__(self, queue):
        self.q = list(queue)
        self._rand, self._randn = torch.rand, torch.randn

    def _next(self, kind, shape, device):
        k, arr = self.q[0]
        if k == kind and tuple(arr.shape) == tuple(shape):
            self.q.pop(0)
        else:
            # the drop-in draws for up to sixteen E-step chains in one generator call ((B, nit, L, N) normals, then (B, nit, N) uniforms):
            # chain i of the block takes the i-th recorded draw of that kind still in the queue, as one call per chain would have
            idx = [i for i, (kk, _) in enumerate(self.q) if kk == kind][:shape[0]]
            parts = [self.q[i][1] for i in idx]
            assert len(parts) == shape[0] and all(tuple(p.shape) == tuple(shape[1:]) for p in parts), (kind, shape, [p.shape for p in parts])
            for i in reversed(idx):
                self.q.pop(i)
            arr = np.stack(parts)
        return torch.from_numpy(np.ascontiguousarray(arr)).to(device if device is not None else "cpu")

    def __enter__(self):
        def shape_of(a):
            return tuple(a[0]) if len(a) == 1 and not isinstance(a[0], int) else tuple(a)
        torch.rand = lambda *a, device=None, **k: self._next("rand", shape_of(a), device)
        torch.randn = lambda *a, device=None, **k: self._next("randn", shape_of(a), device)
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randn = self._rand, self._randn


@pytest.mark.gpu
@pytest.mark.parametrize("case", mc.CASES, ids=[c["name"] for c in mc.CASES])
def test_gpu_run_on_recorded_draws_matches_reference(case):
    fix = case_fix(case["name"])
    em, kw, X = make_em(case, "cuda")
    queue = [("rand", fix["rand_W"]), ("rand", fix["rand_H"]), ("randn", fix["eps_X"]), ("randn", fix["eps_S"])]
    for i in range(case["niter"] + 1):          # one (noise, uniforms) pair per chain in the device path
        queue += [("randn", fix[f"noise{i}"]), ("rand", np.exp(fix[f"logu{i}"].astype(np.float64)).astype(np.float32))]
    with FeedDraws(queue) as feed:
        em.init_parameters(**kw)
        np.testing.assert_allclose(em.Z.cpu().numpy(), fix["Z0"], rtol=1e-4, atol=1e-5)
        cost = em.run()
        assert not feed.q
    np.testing.assert_allclose(cost, fix["cost"], rtol=2e-3)
    np.testing.assert_allclose(em.W.cpu().numpy(), fix["W"][-1], rtol=1e-2, atol=1e-6)
    bad = np.abs(em.S_hat - fix["S_hat"]) > 5e-3 * np.abs(X).max()
    assert bad.mean() < 0.05, bad.mean()
    assert em.S_hat.shape == X.shape and em.S_hat.dtype == X.dtype


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_gpu_run_with_device_draws_converges_like_reference(precision):
    case = dict(mc.CASES[1], niter=6)
    fix = case_fix(case["name"])
    em, kw, X = make_em(case, "cuda")
    em.precision = precision
    torch.manual_seed(0)
    em.init_parameters(**kw)
    cost = em.run()
    assert np.all(np.diff(cost) < 0.02), cost                       # EM: the expected negative log-likelihood goes down
    assert cost[1] < fix["cost"][1] * 1.1 and cost[-1] < fix["cost"][-1]
    assert np.isfinite(em.S_hat).all() and np.abs(em.S_hat + em.N_hat - X).max() < 1e-4 * np.abs(X).max()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", mc.CASES[:2], ids=[c["name"] for c in mc.CASES[:2]])
def test_gpu_fused_run_equals_the_stepwise_loop(case, precision, monkeypatch):
    """_MCEM.run on the device is one library call per EM iteration (dvae_mcem_em_iteration); DVAE_MCEM_RUN=steps keeps the reference's
    loop structure (E_step / M_step / cost through Python, mcem.py:156-160).  Same generator seed -> the same draws in the same order
    -> the same kernels on the same operands: costs, factors and estimates are equal bit for bit."""
    res = {}
    # "fused": two M-step launches per iteration (dvae_mcem_em_iteration_lazy: W normalised by the frames kernel, the cost formed one
    # iteration later and flushed after the loop); "fused3": the three-launch iteration (DVAE_MCEM_LAZY=0)
    for mode in ("fused", "fused3", "steps"):
        monkeypatch.setenv("DVAE_MCEM_RUN", "steps" if mode == "steps" else "fused")
        monkeypatch.setenv("DVAE_MCEM_LAZY", "0" if mode == "fused3" else "1")
        em, kw, X = make_em(dict(case, niter=5), "cuda")
        em.precision = precision
        torch.manual_seed(7)
        em.init_parameters(**kw)
        cost = em.run()
        res[mode] = (np.asarray(cost, np.float64), em.W.cpu().numpy(), em.H.cpu().numpy(), em.g.cpu().numpy(), em.Z.cpu().numpy(), em.S_hat, em.N_hat)
    for other in ("steps", "fused3"):
        for a, b in zip(res["fused"], res[other]):
            assert a.shape == b.shape and np.array_equal(a, b), other
    assert res["fused"][0].shape == (5,) and np.all(np.isfinite(res["fused"][0]))
