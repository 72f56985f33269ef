"""GPU parity of the layer-level C-ABI entry points against the numpy oracle
(float64 truth; tolerance = fp32 rounding of an fp32 fma chain, far inside the 1e-4
relative bar of BASELINE.json's north_star)."""
import importlib

import numpy as np
import pytest
import torch

from oracle import vae_oracle as vo

pytestmark = pytest.mark.gpu

native = importlib.import_module("disentangled-vae_amd.native")
ops = importlib.import_module("disentangled-vae_amd.ops")

ACTS = {0: lambda v: v, 1: np.tanh, 2: lambda v: np.maximum(v, 0), 3: lambda v: 1 / (1 + np.exp(-v)), 4: np.exp}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel_err(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(np.asarray(got, dtype=np.float64) - ref)) / (np.max(np.abs(ref)) + 1e-30))


# (B, k0, k1, N): ragged sizes, tile edges (63/64/65), the model's own shapes, B=1, multi K-slab
SHAPES = [(1, 5, 0, 3), (7, 37, 1, 24), (64, 64, 0, 64), (65, 63, 2, 65), (130, 513, 513, 128), (33, 16, 513, 128),
          (257, 128, 0, 513), (96, 128, 0, 16), (50, 128, 0, 1), (1000, 513, 1, 128)]


@pytest.mark.parametrize("B,k0,k1,N", SHAPES)
@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
def test_linear_act_forward(B, k0, k1, N, act):
    rng = np.random.default_rng(B * 131 + k0 + 7 * k1 + N + act)
    x0 = rng.standard_normal((B, k0)).astype(np.float32)
    x1 = (rng.random((B, k1)) < 0.4).astype(np.float32) if k1 else None
    W = (rng.standard_normal((N, k0 + k1)) / np.sqrt(k0 + k1)).astype(np.float32)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    out = ops.linear_act(dev(x0), dev(W), dev(b), act, None if x1 is None else dev(x1)).cpu().numpy()
    xin = x0 if x1 is None else np.concatenate([x0, x1], 1)
    ref = ACTS[act](xin.astype(np.float64) @ W.astype(np.float64).T + b)
    assert out.shape == (B, N)
    assert rel_err(out, ref) < 5e-6


@pytest.mark.parametrize("B,k0,k1,N", SHAPES)
def test_linear_backward_all_grads(B, k0, k1, N):
    rng = np.random.default_rng(B + k0 + k1 + N)
    act = 1
    x0 = rng.standard_normal((B, k0)).astype(np.float32)
    x1 = rng.standard_normal((B, k1)).astype(np.float32) if k1 else None
    W = (rng.standard_normal((N, k0 + k1)) / np.sqrt(k0 + k1)).astype(np.float32)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    g = rng.standard_normal((B, N)).astype(np.float32)
    tx0 = dev(x0).requires_grad_()
    tx1 = None if x1 is None else dev(x1).requires_grad_()
    tW, tb = dev(W).requires_grad_(), dev(b).requires_grad_()
    out = ops.linear_act(tx0, tW, tb, act, tx1)
    out.backward(dev(g))
    xin = (x0 if x1 is None else np.concatenate([x0, x1], 1)).astype(np.float64)
    o = np.tanh(xin @ W.astype(np.float64).T + b)
    dpre = g * (1 - o * o)
    assert rel_err(tW.grad.cpu().numpy(), dpre.T @ xin) < 1e-5
    assert rel_err(tb.grad.cpu().numpy(), dpre.sum(0)) < 1e-5
    dx = dpre @ W.astype(np.float64)
    assert rel_err(tx0.grad.cpu().numpy(), dx[:, :k0]) < 1e-5
    if x1 is not None:
        assert rel_err(tx1.grad.cpu().numpy(), dx[:, k0:]) < 1e-5


def test_bwd_weight_single_slice_is_deterministic_and_matches_split():
    lib = native.load()
    rng = np.random.default_rng(0)
    B, K, N = 4096, 513, 128
    dpre, x = dev(rng.standard_normal((B, N)).astype(np.float32)), dev(rng.standard_normal((B, K)).astype(np.float32))
    outs = []
    for ks in (1, 1, 8, 0):
        dW = torch.full((N, K), 7.0, device="cuda")
        db = torch.full((N,), 7.0, device="cuda")
        native.check(lib.dvae_linear_bwd_weight(native.ptr(dpre), N, native.ptr(x), K, K, None, 0, 0, native.ptr(dW), K,
                                                native.ptr(db), B, N, ks, native.stream()), "bwd_weight")
        outs.append((dW.cpu().numpy(), db.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    ref = dpre.double().T @ x.double()
    for dW, db in outs:
        assert rel_err(dW, ref.cpu().numpy()) < 1e-5
        assert rel_err(db, dpre.double().sum(0).cpu().numpy()) < 1e-5


@pytest.mark.parametrize("B,k0,k1,N", [(8192, 513, 0, 128), (8192, 16, 1, 128), (5000, 128, 0, 1), (300, 513, 513, 128), (64, 128, 0, 128)])
def test_bwd_weight_of_the_modules_is_deterministic(B, k0, k1, N):
    """dvae_linear_bwd_weight_det (what the drop-in modules' backward calls): split over frame slices like the atomic form, the slices
    combined by an ordered sum -- two runs return the same bits (the reference's GPU GEMMs do), padded strides and concatenated inputs
    included, values against float64."""
    lib = native.load()
    rng = np.random.default_rng(B + k0)
    ldp, ld0, ld1, ldw = N + 3, k0 + 5, k1 + 2, k0 + k1 + 7
    dpre = dev(rng.standard_normal((B, ldp)).astype(np.float32)); x0 = dev(rng.standard_normal((B, ld0)).astype(np.float32))
    x1 = dev(rng.standard_normal((B, ld1)).astype(np.float32)) if k1 else None
    nws = lib.dvae_linear_bwd_weight_workspace_bytes(B, N, k0 + k1, 0)
    assert nws == 0 if B <= 256 else nws > 0                            # one slice needs no workspace
    outs = []
    for _ in range(3):
        ws = torch.empty(max(nws, 1), dtype=torch.uint8, device="cuda")
        dW = torch.full((N, ldw), 7.0, device="cuda"); db = torch.full((N,), 7.0, device="cuda")
        native.check(lib.dvae_linear_bwd_weight_det(native.ptr(dpre), ldp, native.ptr(x0), k0, ld0, native.ptr(x1), k1, ld1 if k1 else 0,
                                                    native.ptr(dW), ldw, native.ptr(db), B, N, 0, native.ptr(ws), native.stream()), "bwd_weight_det")
        outs.append((dW.cpu().numpy(), db.cpu().numpy()))
    for dW, db in outs[1:]:
        assert np.array_equal(dW, outs[0][0]) and np.array_equal(db, outs[0][1])
    xin = x0[:, :k0].double() if not k1 else torch.cat([x0[:, :k0], x1[:, :k1]], dim=1).double()
    ref = (dpre[:, :N].double().T @ xin).cpu().numpy()
    assert rel_err(outs[0][0][:, :k0 + k1], ref) < 1e-5
    assert np.all(outs[0][0][:, k0 + k1:] == 7.0)                     # the padding of dW is not touched
    assert rel_err(outs[0][1], dpre[:, :N].double().sum(0).cpu().numpy()) < 1e-5


def test_bwd_data_accumulate_sums_the_two_heads():
    lib = native.load()
    rng = np.random.default_rng(1)
    B, Z, H = 77, 16, 128
    dmu, dlv = (dev(rng.standard_normal((B, Z)).astype(np.float32)) for _ in range(2))
    Wm, Wv = (dev(rng.standard_normal((Z, H)).astype(np.float32)) for _ in range(2))
    dh = torch.empty((B, H), device="cuda")
    s = native.stream()
    native.check(lib.dvae_linear_bwd_data(native.ptr(dmu), Z, native.ptr(Wm), H, 0, native.ptr(dh), H, B, Z, H, 0, s), "bwd_data")
    native.check(lib.dvae_linear_bwd_data(native.ptr(dlv), Z, native.ptr(Wv), H, 0, native.ptr(dh), H, B, Z, H, 1, s), "bwd_data")
    ref = (dmu.double() @ Wm.double() + dlv.double() @ Wv.double()).cpu().numpy()
    assert rel_err(dh.cpu().numpy(), ref) < 5e-6


def test_reparam_elbo_bce_against_oracle():
    rng = np.random.default_rng(5)
    B, F, Z = 300, 513, 16
    x = np.clip(np.exp(4 * rng.standard_normal((B, F)) - 8) * rng.chisquare(2, (B, F)) / 2, 1e-12, 1e4).astype(np.float32)
    a = rng.standard_normal((B, F)).astype(np.float32) * 2
    r = np.exp(a)
    mu, lv, e = (rng.standard_normal((B, Z)).astype(np.float32) for _ in range(3))
    tmu, tlv = dev(mu).requires_grad_(), dev(lv).requires_grad_()
    z = ops.Reparam.apply(tmu, tlv, dev(e))
    np.testing.assert_allclose(z.detach().cpu().numpy(), mu + np.exp(0.5 * lv) * e, rtol=2e-6, atol=1e-6)
    tr = dev(r).requires_grad_()
    out = ops.Elbo.apply(dev(x), tr, tmu, tlv, 1e-8)
    ref = vo.elbo(x.astype(np.float64), r.astype(np.float64), mu.astype(np.float64), lv.astype(np.float64), 1e-8)
    assert len(out) == 3 and all(o.dim() == 0 for o in out)                     # (recon + KL, recon, KL): three 0-dim tensors
    np.testing.assert_allclose([o.item() for o in out], ref, rtol=2e-6)
    (out[0] * 1.5 + out[2] * 0.25 + z.sum() * 0.01).backward()
    da, dmu, dlv = vo.elbo_bwd(x.astype(np.float64), a.astype(np.float64), mu.astype(np.float64), lv.astype(np.float64))
    # dr = da / r ; loss weight 1.5 on recon, 1.75 on KL; z path adds 0.01 and 0.01*eps*0.5*std
    assert rel_err(tr.grad.cpu().numpy() * r, 1.5 * da) < 2e-5
    np.testing.assert_allclose(tmu.grad.cpu().numpy(), 1.75 * dmu + 0.01, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(tlv.grad.cpu().numpy(), 1.75 * dlv + 0.01 * e * 0.5 * np.exp(0.5 * lv), rtol=1e-5, atol=1e-7)

    for Y in (1, 513):
        p = (1 / (1 + np.exp(-rng.standard_normal((B, Y))))).astype(np.float32)
        t = (rng.random((B, Y)) < 0.5).astype(np.float32)
        p[0, 0], t[0, 0] = 1.0, 0.0        # log(1 - 1 + eps): the eps-inside-the-log corner
        for variant, fn in ((0, lambda: vo.binary_cross_entropy(p.astype(np.float64), t, 1e-8)),
                            (1, lambda: vo.binary_cross_entropy_v2(p.astype(np.float64), 1e-8)),
                            (2, lambda: vo.binary_cross_entropy_v3(p.astype(np.float64), 1e-8))):
            tp = dev(p).requires_grad_()
            val = ops.Bce.apply(tp, dev(t) if variant == 0 else None, 1e-8, variant)
            np.testing.assert_allclose(val.item(), fn(), rtol=3e-6)
            (val * 2.0).backward()
            if variant == 0:
                ref_g = vo.bce_bwd(p.astype(np.float64), t.astype(np.float64), 1e-8, 2.0)
                m = np.abs(ref_g) < 1e6      # the forced corner has a 1e8-scale derivative in fp32
                np.testing.assert_allclose(tp.grad.cpu().numpy()[m], ref_g[m], rtol=2e-5, atol=1e-7)


def test_adam_kernel_matches_torch_adam():
    rng = np.random.default_rng(9)
    n = 100003
    p0 = rng.standard_normal(n).astype(np.float32)
    pt = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([pt], lr=1e-4, betas=(0.9, 0.999))
    p, m, v = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 6):
        g = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 2, n)).astype(np.float32)
        if step == 3:
            g[:100] = 0.0
        pt.grad = torch.from_numpy(g.copy())
        opt.step()
        ops.adam_step_(p, dev(g), m, v, step)
        np.testing.assert_allclose(p.cpu().numpy(), pt.detach().numpy(), rtol=2e-7, atol=3e-8 * step)
    # oracle too
    po, mo, vo_ = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    po, mo, vo_ = vo.adam_step(po, g, mo, vo_, 1)
    assert np.all(np.isfinite(po))


def test_missing_grad_inputs_and_frozen_params():
    x = torch.randn(10, 8, device="cuda")
    W = torch.randn(4, 8, device="cuda")                       # frozen (reconstruct scripts)
    b = torch.zeros(4, device="cuda")
    out = ops.linear_act(x, W, b, 1)
    assert not out.requires_grad
    with pytest.raises(TypeError):
        ops.linear_act(x.double(), W, b, 1)
    with pytest.raises(RuntimeError):
        ops.linear_act(x, torch.randn(4, 9, device="cuda"), b, 1)
