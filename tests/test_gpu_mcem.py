"""MCEM kernels (include/dvae_mcem.h) against the numpy oracle on the same draws, and against the golden
vectors captured from the reference's own packages/models/mcem.py."""
import importlib
import os

import numpy as np
import pytest
import torch

import golden_util as gu
import mcem_cases as mc
from impl_modules import build_model
from oracle import mcem_oracle as mo

pytestmark = pytest.mark.gpu
mcem_dev = importlib.import_module("disentangled-vae_amd.mcem")

FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "mcem_golden.npz"))


def case_fix(name):
    return {k.split("/", 1)[1]: FIX[k] for k in FIX.files if k.startswith(name + "/")}


def t(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def setup(model, y_dim, N, seed, wscale=1.0, precision="fp32"):
    dims = dict(x_dim=513, y_dim=y_dim, z_dim=16, h_dim=(128, 128))
    params = gu.make_params(model, dims, seed, wscale)
    m = build_model(model, dims)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    m.cuda()
    vae = m.enc_dec_clf if model == "M2_info" else m
    pack = mcem_dev.DecoderPack(vae.decoder, y_dim, precision)
    prefix = "enc_dec_clf.decoder." if model == "M2_info" else "decoder."
    rng = np.random.default_rng(seed + 77)
    X2 = (rng.standard_normal((513, N)) ** 2 * np.exp(rng.standard_normal((513, 1)) - 1)).astype(np.float32) + 1e-4
    y = (rng.random((y_dim, N)) > 0.5).astype(np.float32) if y_dim else None
    Z = rng.standard_normal((16, N)).astype(np.float32)
    g = np.exp(0.2 * rng.standard_normal(N)).astype(np.float32)
    W = np.maximum(rng.random((513, 10)), 1e-6).astype(np.float32)
    H = np.maximum(rng.random((10, N)), 1e-6).astype(np.float32)
    return params, prefix, pack, X2, y, Z, g, W, H, rng


def pick_tile(monkeypatch, tile, precision):
    """Frames per workgroup of the weight-stationary chain: 4 (csrc/mcem_resident4.hip: exact fp32 only, what short fp32 chains take by
    default), 16 (csrc/mcem_resident16.hip, what chains take by default while their 16-frame tiles fit the chip in one round) or 32
    (csrc/mcem_resident.hip)."""
    if tile == "4" and precision != "fp32":
        pytest.skip("the 4-frame chain kernel exists for the exact-fp32 policy")
    monkeypatch.setenv("DVAE_MCEM_TILE", tile)


def chains_agree(accd, trace_a):
    """per frame: index of the first iteration whose accept decision differs (nit if none)."""
    diff = accd != trace_a
    first = np.where(diff.any(axis=0), diff.argmax(axis=0), diff.shape[0])
    return first


@pytest.mark.parametrize("tile", ["4", "16", "32"])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("model,y_dim,N", [("M1", 0, 45), ("M2", 1, 70), ("M2", 513, 33), ("M2_info", 1, 100)])
def test_sample_posterior_matches_oracle(model, y_dim, N, precision, tile, monkeypatch):
    """fp32 = exact fp32 products; bf16x3 = split-bf16 operands (16 mantissa bits, three MFMAs per product): the same bounds."""
    pick_tile(monkeypatch, tile, precision)
    params, prefix, pack, X2, y, Z, g, W, H, rng = setup(model, y_dim, N, 5, precision=precision)
    nit, burnin = 12, 5
    noise = rng.standard_normal((nit, 16, N)).astype(np.float32)
    logu = np.log(rng.random((nit, N)).astype(np.float32))
    Vb = (W @ H).astype(np.float32)
    Zs_o, tp, ta = mo.sample_posterior(params, prefix, Z, y, g, Vb, X2, noise, logu, burnin, return_trace=True)
    Zs, Vs, accp, accd = pack.sample(t(Z), t(y), t(g), t(Vb), t(X2), t(noise), t(logu), burnin, trace=True)
    Zs, Vs, accp, accd = Zs.cpu().numpy(), Vs.cpu().numpy(), accp.cpu().numpy(), accd.cpu().numpy().astype(bool)
    first = chains_agree(accd, ta)
    # log acceptance ratios agree as long as the two chains are in the same state (sums of 513 terms of size ~5:
    # the reference's own float32 summation noise is ~1e-4 absolute)
    for n in range(N):
        k = min(first[n] + 1, nit)
        np.testing.assert_allclose(accp[:k, n], tp[:k, n], rtol=2e-4, atol=2e-3)
    same = first == nit
    assert same.mean() >= 0.97, same.mean()
    np.testing.assert_allclose(Zs[same], Zs_o[same], rtol=1e-5, atol=1e-6)
    assert 0.02 < accd.mean() < 0.98            # the chain actually moves and actually rejects
    # variances of the kept samples (compute_Vs) in the same launch
    Vs_o = mo.compute_vs(params, prefix, Zs, y)
    np.testing.assert_allclose(Vs, Vs_o, rtol=1e-4, atol=1e-9)
    # and alone
    Vs2 = pack.decode(t(Zs), t(y)).cpu().numpy()
    np.testing.assert_array_equal(Vs2, Vs)


@pytest.mark.parametrize("tile", ["4", "16", "32"])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("decisions", ["mixed", "accept", "reject"])
@pytest.mark.parametrize("nit,burnin", [(1, 0), (4, 0), (5, 4), (12, 5), (70, 2)])
def test_kept_variances_are_those_of_a_decoder_pass(nit, burnin, decisions, precision, tile, monkeypatch):
    """The chain kernels take the kept samples' decoder variances from their own passes (4- and 16-frame tiles: the state's copy on chip;
    32-frame tiles under exact fp32: the proposal's row stored by the kept step, rejected rows copied behind the chain, one pass for the
    state at the end of the burn-in; more than 64 kept samples: decoder passes).  Whatever the chain did -- no burn-in, one kept sample,
    every proposal accepted, every one rejected -- Vs is bit for bit what `dvae_mcem_decode` returns for the stored samples."""
    pick_tile(monkeypatch, tile, precision)
    N = 70
    params, prefix, pack, X2, y, Z, g, W, H, rng = setup("M2", 1, N, 21, precision=precision)
    noise = rng.standard_normal((nit, 16, N)).astype(np.float32)
    logu = np.log(rng.random((nit, N)).astype(np.float32))
    if decisions == "accept":
        logu[:] = -1e30
    elif decisions == "reject":
        logu[:] = 1e30
    Vb = (W @ H).astype(np.float32)
    Zs, Vs, accp, accd = pack.sample(t(Z), t(y), t(g), t(Vb), t(X2), t(noise), t(logu), burnin, trace=True)
    accd = accd.cpu().numpy().astype(bool)
    if decisions != "mixed":
        assert accd.all() == (decisions == "accept") and accd.any() == (decisions == "accept")
    if decisions == "reject":                              # the chain never left its initial state
        np.testing.assert_array_equal(Zs.cpu().numpy(), np.broadcast_to(Z.T[:, None, :], (N, nit - burnin, 16)))
    Vs2 = pack.decode(Zs, t(y))
    assert Vs.shape == (nit - burnin, 513, N)
    np.testing.assert_array_equal(Vs.cpu().numpy(), Vs2.cpu().numpy())


@pytest.mark.parametrize("N,R,K", [(45, 3, 10), (300, 10, 10), (32, 1, 4), (1, 2, 16)])
def test_m_step_and_wiener_match_oracle(N, R, K):
    rng = np.random.default_rng(N + R)
    X2 = (rng.standard_normal((513, N)) ** 2).astype(np.float32) + 1e-4
    Vs = np.exp(rng.standard_normal((R, 513, N)) - 0.5).astype(np.float32)
    W = np.maximum(rng.random((513, K)), 1e-6).astype(np.float32)
    H = np.maximum(rng.random((K, N)), 1e-6).astype(np.float32)
    g = np.exp(0.3 * rng.standard_normal(N)).astype(np.float32)
    Vb = (W @ H).astype(np.float32)
    Wo, Ho, go, Vbo, Vxo, cost_o = mo.m_step(X2, Vs, W, H, g, Vb, dtype=np.float64)
    dW, dH, dg, dVb = t(W), t(H), t(g), t(Vb)
    cost = mcem_dev.m_step_(t(X2), t(Vs), dW, dH, dg, dVb)
    np.testing.assert_allclose(dW.cpu().numpy(), Wo, rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(dH.cpu().numpy(), Ho, rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(dg.cpu().numpy(), go, rtol=1e-4)
    np.testing.assert_allclose(dVb.cpu().numpy(), Vbo, rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(cost.item(), cost_o, rtol=1e-5)
    WFs_o, WFn_o = mo.wiener(Vs.astype(np.float64), go, Vbo)
    WFs, WFn = mcem_dev.wiener(t(Vs), dg, dVb)
    np.testing.assert_allclose(WFs.cpu().numpy(), WFs_o, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(WFn.cpu().numpy(), WFn_o, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose((WFs + WFn).cpu().numpy(), 1.0, rtol=1e-5)        # the two gains partition the mixture


@pytest.mark.parametrize("tile", ["4", "16", "32"])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("N", [1, 15, 16, 17, 31, 33])
def test_chain_on_tiny_and_ragged_frame_counts(N, precision, tile, monkeypatch):
    """One frame, one short of a tile, exactly a tile, one more (both tile sizes): log acceptance ratios of the first chain step (every chain
    still in its initial state) against the oracle, kept samples finite, nothing written outside the N frames."""
    pick_tile(monkeypatch, tile, precision)
    params, prefix, pack, X2, y, Z, g, W, H, rng = setup("M2", 1, N, 11, precision=precision)
    nit, burnin = 6, 2
    noise = rng.standard_normal((nit, 16, N)).astype(np.float32)
    logu = np.log(rng.random((nit, N)).astype(np.float32))
    Vb = (W @ H).astype(np.float32)
    Zs_o, tp, ta = mo.sample_posterior(params, prefix, Z, y, g, Vb, X2, noise, logu, burnin, return_trace=True)
    Zs, Vs, accp, accd = pack.sample(t(Z), t(y), t(g), t(Vb), t(X2), t(noise), t(logu), burnin, trace=True)
    assert Zs.shape == (N, nit - burnin, 16) and Vs.shape == (nit - burnin, 513, N)
    np.testing.assert_allclose(accp.cpu().numpy()[0], tp[0], rtol=2e-4, atol=2e-3)
    assert torch.isfinite(Zs).all() and torch.isfinite(Vs).all() and (Vs > 0).all()
    first = chains_agree(accd.cpu().numpy().astype(bool), ta)
    same = first == nit
    np.testing.assert_allclose(Zs.cpu().numpy()[same], Zs_o[same], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tile", ["4", "16", "32"])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", mc.CASES, ids=[c["name"] for c in mc.CASES])
def test_full_run_matches_reference_golden(case, precision, tile, monkeypatch):
    """EM.run on the draws recorded from the reference: state after every iteration vs the reference's (exact-fp32 and split-bf16
    chain policies, 16- and 32-frame chain kernels, against the same bounds)."""
    pick_tile(monkeypatch, tile, precision)
    fix = case_fix(case["name"])
    dims = mc.DIMS[case["model"]]
    params, prefix, pack, *_ = setup(case["model"], dims["y_dim"], case["N"], case["seed"], case["wscale"], precision=precision)
    X, S, y = mc.make_utterance(case)
    X2 = t((np.abs(X) ** 2).astype(np.float32))
    yd = t(y) if case["model"] != "M1" else None
    W, H, g = mo.init_nmf(fix["rand_W"], fix["rand_H"], mc.EPS)
    W, H, g = t(W), t(H), t(g)
    Vb = (W @ H).contiguous()
    Z = t(fix["Z0"])
    n_e, b_e, n_wf, b_wf = mc.effective_counts(case)
    for it in range(case["niter"]):
        Zs, Vs = pack.sample(Z, yd, g, Vb, X2, t(fix[f"noise{it}"]), t(fix[f"logu{it}"]), b_e)
        Z = Zs[:, -1, :].t().contiguous()
        cost = mcem_dev.m_step_(X2, Vs, W, H, g, Vb)
        dz = np.abs(Z.cpu().numpy() - fix["Z"][it]).max(axis=0)
        assert (dz > 1e-3).mean() <= 0.03, (it, dz.max())
        ok = dz <= 1e-3
        np.testing.assert_allclose(g.cpu().numpy()[ok], fix["g"][it][ok], rtol=2e-3)
        np.testing.assert_allclose(H.cpu().numpy()[:, ok], fix["H"][it][:, ok], rtol=2e-3, atol=1e-6)
        np.testing.assert_allclose(W.cpu().numpy(), fix["W"][it], rtol=5e-3, atol=1e-6)
        np.testing.assert_allclose(cost.item(), fix["cost"][it], rtol=1e-3)
    it = case["niter"]
    Zs, Vs = pack.sample(Z, yd, g, Vb, X2, t(fix[f"noise{it}"]), t(fix[f"logu{it}"]), b_wf)
    WFs, WFn = mcem_dev.wiener(Vs, g, Vb)
    assert (np.abs(WFs.cpu().numpy() - fix["WFs"]) > 5e-3).mean() < 0.05
    assert (np.abs(WFn.cpu().numpy() - fix["WFn"]) > 5e-3).mean() < 0.05


@pytest.mark.parametrize("tile", ["4", "16", "32"])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16"])
@pytest.mark.parametrize("model,y_dim,N", [("M2", 1, 300), ("M2", 513, 200), ("M1", 0, 257)])
def test_chain_launches_are_bit_identical(model, y_dim, N, precision, tile, monkeypatch):
    """The weight-stationary chain feeds its output-layer MFMAs from AGPR-pinned fragments through inline asm, i.e. outside the compiler's
    hazard bookkeeping (csrc/mcem_resident.hip): a write-after-read slip there shows up as bf16-level differences that change from launch
    to launch (that is how the first version was caught).  Four launches on the same draws return the same bits, every policy and label
    variant, both tile sizes, full-length E-step chain."""
    pick_tile(monkeypatch, tile, precision)
    params, prefix, pack, X2, y, Z, g, W, H, rng = setup(model, y_dim, N, 5, precision=precision)
    nit, burnin = 40, 30
    noise = rng.standard_normal((nit, 16, N)).astype(np.float32)
    logu = np.log(rng.random((nit, N)).astype(np.float32))
    Vb = (W @ H).astype(np.float32)
    outs = []
    for _ in range(4):
        Zs, Vs = pack.sample(t(Z), t(y), t(g), t(Vb), t(X2), t(noise), t(logu), burnin)
        outs.append((Zs.cpu().numpy().copy(), Vs.cpu().numpy().copy()))
    assert np.isfinite(outs[0][1]).all() and (outs[0][1] > 0).all()
    for Zs, Vs in outs[1:]:
        np.testing.assert_array_equal(Zs, outs[0][0])
        np.testing.assert_array_equal(Vs, outs[0][1])


def test_bf16_chain_is_statistically_close():
    """Throughput mode (bf16 matrix-core operands): first-iteration log ratios within bf16 noise of the oracle,
    same acceptance rate to a few percent."""
    params, prefix, pack, X2, y, Z, g, W, H, rng = setup("M2", 1, 256, 9, precision="bf16")
    nit, burnin = 20, 10
    noise = rng.standard_normal((nit, 16, 256)).astype(np.float32)
    logu = np.log(rng.random((nit, 256)).astype(np.float32))
    Vb = (W @ H).astype(np.float32)
    _, tp, ta = mo.sample_posterior(params, prefix, Z, y, g, Vb, X2, noise, logu, burnin, return_trace=True)
    Zs, Vs, accp, accd = pack.sample(t(Z), t(y), t(g), t(Vb), t(X2), t(noise), t(logu), burnin, trace=True)
    accp, accd = accp.cpu().numpy(), accd.cpu().numpy().astype(bool)
    err = np.abs(accp[0] - tp[0])
    assert np.median(err) < 0.15 and err.max() < 2.0, (np.median(err), err.max())
    assert abs(accd.mean() - ta.mean()) < 0.05
    assert torch.isfinite(Vs).all()


def test_bad_arguments_fail_loudly():
    params, prefix, pack, X2, y, Z, g, W, H, rng = setup("M2", 1, 8, 3)
    noise = torch.zeros((4, 16, 8), device="cuda"); logu = torch.zeros((4, 8), device="cuda")
    with pytest.raises(RuntimeError):
        pack.sample(t(Z), None, t(g), t(W @ H), t(X2), noise, logu, 1)        # y missing although y_dim == 1
    with pytest.raises(RuntimeError):
        pack.sample(t(Z), t(y), t(g), t(W @ H), t(X2), noise, logu, 4)        # burnin == nit: nothing kept
    with pytest.raises(RuntimeError):
        mcem_dev.m_step_(t(X2), torch.ones((2, 513, 8), device="cuda"), torch.ones((513, 17), device="cuda"),
                         torch.ones((17, 8), device="cuda"), t(g), t((W @ H)))   # K > 16
    # dvae_mcem_em_iteration (one EM iteration per call): missing scratch / bad chain lengths are codes with a message, never a launch
    import ctypes
    N = importlib.import_module("disentangled-vae_amd.native")
    lib = N.load()
    Zd, gd, Vb, X2d, yd = t(Z), t(g), t(W @ H).contiguous(), t(X2), t(y)
    Wd, Hd = t(W).contiguous(), t(H).contiguous()
    Zs = torch.empty((8, 3, 16), device="cuda"); Vs = torch.empty((3, 513, 8), device="cuda"); cost = torch.empty(1, device="cuda")
    ws = torch.empty(lib.dvae_mcem_m_step_workspace_bytes(8, W.shape[1], 1), dtype=torch.uint8, device="cuda")
    call = lambda zs, nit, burnin: lib.dvae_mcem_em_iteration(ctypes.byref(pack.plan), N.ptr(pack.weights), N.ptr(Zd), N.ptr(yd), N.ptr(gd), N.ptr(Vb),
                                                               N.ptr(X2d), N.ptr(noise), N.ptr(logu), nit, burnin, 0.01, 8, W.shape[1], 1, None, None, None,
                                                               N.ptr(Wd), N.ptr(Hd), zs, N.ptr(Vs), N.ptr(cost), N.ptr(ws), N.stream())
    assert call(None, 4, 1) != 0 and "Zs" in lib.dvae_last_error().decode()
    assert call(N.ptr(Zs), 4, 4) != 0                                          # nothing kept
    assert call(N.ptr(Zs), 4, 1) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(cost).all() and torch.isfinite(Zd).all()


def test_batched_run_equals_per_utterance_runs():
    """McemBatch (utterances side by side, padded to 32 frames) == each utterance alone on the same draws."""
    counts = [45, 64, 7]
    dims = dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128))
    params = gu.make_params("M2", dims, 21)
    m = build_model("M2", dims)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    m.cuda().eval()
    utts = [mc.make_utterance(dict(seed=40 + i, N=n, model="M2")) for i, n in enumerate(counts)]
    mb = mcem_dev.McemBatch(m, niter=3, nsamples_E_step=3, burnin_E_step=4, nsamples_WF=4, burnin_WF=3)
    torch.manual_seed(1)
    mb.init_parameters([u[0] for u in utts], [u[2] for u in utts])
    W0, H0, Z0, Vb0 = mb.W.clone(), mb.H.clone(), mb.Z.clone(), mb.Vb.clone()
    gen = torch.Generator(device="cuda"); gen.manual_seed(2)
    draws = []
    for it in range(4):
        nit = 7
        draws.append((torch.randn(nit, 16, mb.ntot, device="cuda", generator=gen),
                      torch.log(torch.rand(nit, mb.ntot, device="cuda", generator=gen))))
    cost = mb.run(draws)
    assert cost.shape == (3, 3) and np.isfinite(cost).all()
    for u, n in enumerate(counts):
        s = mb.starts[u]
        sl = slice(s, s + n)
        X2 = mb.X2[:, sl].contiguous(); y = mb.y[:, sl].contiguous()
        W, H, g = W0[u].clone(), H0[:, sl].contiguous(), torch.ones(n, device="cuda")
        Vb, Z = Vb0[:, sl].contiguous(), Z0[:, sl].contiguous()
        for it in range(3):
            Zs, Vs = mb._pack.sample(Z, y, g, Vb, X2, draws[it][0][:, :, sl].contiguous(), draws[it][1][:, sl].contiguous(), 4)
            Z = Zs[:, -1, :].t().contiguous()
            c = mcem_dev.m_step_(X2, Vs, W, H, g, Vb)
            np.testing.assert_allclose(c.item(), cost[it, u], rtol=1e-6)
        torch.testing.assert_close(W, mb.W[u], rtol=1e-6, atol=0)
        torch.testing.assert_close(H, mb.H[:, sl], rtol=1e-6, atol=0)
        torch.testing.assert_close(g, mb.g[sl], rtol=1e-6, atol=0)
        Zs, Vs = mb._pack.sample(Z, y, g, Vb, X2, draws[3][0][:, :, sl].contiguous(), draws[3][1][:, sl].contiguous(), 3)
        WFs, WFn = mcem_dev.wiener(Vs, g, Vb)
        np.testing.assert_allclose(WFs.cpu().numpy() * utts[u][0], mb.S_hat[u], rtol=1e-5, atol=1e-7)
        assert mb.S_hat[u].shape == utts[u][0].shape


def test_graph_replayed_iterations_equal_the_eager_loop():
    """McemBatch.run(graph=True) replays one captured EM iteration (HIP graph; opt-in, measured no faster); on recorded draws (copied into the graph's
    static draw buffers before each replay) costs, NMF factors, gains and the Wiener estimates equal the eager loop's bit for bit.
    With the generator inside the graph (no recorded draws) the run is a different random sequence: finite, and the cost falls."""
    counts = [70, 33]
    dims = dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128))
    params = gu.make_params("M2", dims, 21)
    m = build_model("M2", dims)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    m.cuda().eval()
    utts = [mc.make_utterance(dict(seed=60 + i, N=n, model="M2")) for i, n in enumerate(counts)]
    niter, res = 9, {}
    for graph in (True, False):
        mb = mcem_dev.McemBatch(m, niter=niter, nsamples_E_step=3, burnin_E_step=4, nsamples_WF=4, burnin_WF=3, precision="bf16x3")
        torch.manual_seed(1)
        mb.init_parameters([u[0] for u in utts], [u[2] for u in utts])
        gen = torch.Generator(device="cuda"); gen.manual_seed(2)
        draws = [(torch.randn(7, 16, mb.ntot, device="cuda", generator=gen), torch.log(torch.rand(7, mb.ntot, device="cuda", generator=gen)))
                 for _ in range(niter + 1)]
        cost = mb.run(draws, graph=graph)
        res[graph] = (cost, mb.W.cpu().numpy(), mb.H.cpu().numpy(), mb.g.cpu().numpy(), mb.Z.cpu().numpy(), [a.copy() for a in mb.S_hat])
    for a, b in zip(res[True][:5], res[False][:5]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(res[True][5], res[False][5]):
        np.testing.assert_array_equal(a, b)
    assert np.isfinite(res[True][0]).all()
    mb = mcem_dev.McemBatch(m, niter=12, nsamples_E_step=3, burnin_E_step=4, nsamples_WF=4, burnin_WF=3, precision="bf16x3")
    torch.manual_seed(1)
    mb.init_parameters([u[0] for u in utts], [u[2] for u in utts])
    cost = mb.run(graph=True)                                    # generator draws inside the graph
    assert np.isfinite(cost).all() and np.all(cost[-1] < cost[0])
    assert len({float(c) for c in cost[:, 0]}) == 12             # every replay drew fresh numbers (no frozen generator state)


def test_enhancement_example_end_to_end(tmp_path):
    """wav -> STFT -> VAD labels -> batched MCEM -> Wiener -> ISTFT -> wav: the two estimates add up to the mixture."""
    import subprocess, sys
    from scipy.io import wavfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "enh")
    rng = np.random.default_rng(0)
    paths = []
    for i, n in enumerate((16000, 21000)):
        # no digital silence: an all-zero frame drives g and H to 0 and the reference's own updates to 0/0
        w = (0.3 * rng.standard_normal(n) * (np.arange(n) % 4000 < 2000) + 0.02 * rng.standard_normal(n)).astype(np.float32)
        p = str(tmp_path / f"mix{i}.wav"); wavfile.write(p, 16000, (w * 32767).astype(np.int16)); paths.append((p, n))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "enhance_mcem.py"), "--wav"] + [p for p, _ in paths] +
                       ["--niter", "5", "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for i, (p, n) in enumerate(paths):
        _, x = wavfile.read(p)
        x = x.astype(np.float64) / 32768.0
        _, s = wavfile.read(os.path.join(out, f"mix{i}_s_est.wav"))
        _, nz = wavfile.read(os.path.join(out, f"mix{i}_n_est.wav"))
        assert s.shape == (n,) and nz.shape == (n,) and np.isfinite(s).all() and np.isfinite(nz).all()
        inner = slice(1024, n - 1280)                         # interior: full window overlap (edges are ill-conditioned)
        assert np.abs(s[inner] + nz[inner] - x[inner]).max() < 1e-3 * np.abs(x).max() + 1e-5
        assert np.abs(s).max() > 0 and np.abs(nz).max() > 0
