"""Out-of-bounds writes would land in a neighbouring allocation and go unnoticed: every hand-indexed kernel family is
called through the C ABI with guard bands around its OUTPUT buffers (ragged sizes), and the bands must survive."""
import ctypes
import importlib

import numpy as np
import pytest
import torch

import golden_util as gu
from impl_modules import build_model

pytestmark = pytest.mark.gpu
N = importlib.import_module("disentangled-vae_amd.native")
H = importlib.import_module("disentangled-vae_amd.stft")
mcem_dev = importlib.import_module("disentangled-vae_amd.mcem")
G = 1 << 16


class Guarded:
    def __init__(self):
        self.raws = []

    def buf(self, shape, dtype):
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        raw = torch.full((n + 2 * G,), 0xA5, dtype=torch.uint8, device="cuda")
        self.raws.append(raw)
        return raw[G:G + n].view(dtype).view(*shape)

    def check(self):
        torch.cuda.synchronize()
        for i, raw in enumerate(self.raws):
            assert bool((raw[:G] == 0xA5).all()) and bool((raw[-G:] == 0xA5).all()), f"buffer {i}: guard band overwritten"


@pytest.mark.parametrize("n,layout", [(1024 + 256 * 36 + 5, 0), (1024 + 256 * 36 + 5, 1), (16000 * 40 + 3, 1), (16000 * 40 + 3, 0), (1024, 0)])
def test_stft_istft_outputs(n, layout):
    lib = N.load()
    g = Guarded()
    x = torch.randn(n, dtype=torch.float64, device="cuda")
    w = H.window_f64("hann", 1024, x.device)
    T = 1 + (n - 1024) // 256
    out = g.buf((513, T), torch.complex64) if layout == 0 else g.buf((T, 513), torch.float32)
    N.check(lib.dvae_stft(N.ptr(x), 1, n, N.ptr(w), 1024, 256, T, N.ptr(out), layout, N.stream()), "dvae_stft")
    if layout == 0:
        out_len = n - 7
        y = g.buf((out_len,), torch.float32)
        ws = g.buf((lib.dvae_istft_workspace_bytes(T, 1024),), torch.uint8)
        N.check(lib.dvae_istft(N.ptr(out), T, T, N.ptr(w), 1024, 256, 0, N.ptr(y), out_len, N.ptr(ws), N.stream()), "dvae_istft")
        assert torch.isfinite(y).all()
    g.check()


@pytest.mark.parametrize("model,y_dim,n,precision", [("M2", 1, 45, "fp32"), ("M2", 513, 33, "bf16"), ("M1", 0, 1, "fp32"), ("M2", 1, 300, "bf16")])
def test_mcem_outputs(model, y_dim, n, precision):
    lib = N.load()
    dims = dict(x_dim=513, y_dim=y_dim, z_dim=16, h_dim=(128, 128))
    m = build_model(model, dims)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in gu.make_params(model, dims, 2).items()})
    m.cuda()
    pack = mcem_dev.DecoderPack(m.decoder, y_dim, precision)
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    rnd = lambda *s: torch.rand(*s, device="cuda", generator=gen)
    nit, burnin, K = 7, 3, 10
    R = nit - burnin
    g = Guarded()
    X2, Vb, gg = rnd(513, n) + 0.01, rnd(513, n) + 0.1, rnd(n) + 0.5
    Z = torch.randn(16, n, device="cuda", generator=gen)
    y = (rnd(y_dim, n) > 0.5).float() if y_dim else None
    noise = torch.randn(nit, 16, n, device="cuda", generator=gen); logu = torch.log(rnd(nit, n))
    Zs, Vs = g.buf((n, R, 16), torch.float32), g.buf((R, 513, n), torch.float32)
    accp, accd = g.buf((nit, n), torch.float32), g.buf((nit, n), torch.uint8)
    N.check(lib.dvae_mcem_sample(ctypes.byref(pack.plan), N.ptr(pack.weights), N.ptr(Z), N.ptr(y), N.ptr(gg), N.ptr(Vb), N.ptr(X2), N.ptr(noise),
                                 N.ptr(logu), nit, burnin, 0.01, n, N.ptr(Zs), N.ptr(Vs), N.ptr(accp), N.ptr(accd), N.stream()), "sample")
    Vs2 = g.buf((R, 513, n), torch.float32)
    N.check(lib.dvae_mcem_decode(ctypes.byref(pack.plan), N.ptr(pack.weights), N.ptr(Zs), N.ptr(y), R, n, N.ptr(Vs2), N.stream()), "decode")
    W, Hm, g2, Vb2, cost = g.buf((513, K), torch.float32), g.buf((K, n), torch.float32), g.buf((n,), torch.float32), g.buf((513, n), torch.float32), g.buf((1,), torch.float32)
    W.copy_(rnd(513, K) + 0.01); Hm.copy_(rnd(K, n) + 0.01); g2.copy_(gg); Vb2.copy_(Vb)
    ws = g.buf((lib.dvae_mcem_m_step_workspace_bytes(n, K, 1),), torch.uint8)
    N.check(lib.dvae_mcem_m_step(N.ptr(X2), N.ptr(Vs), R, n, K, N.ptr(W), N.ptr(Hm), N.ptr(g2), N.ptr(Vb2), N.ptr(cost), N.ptr(ws), N.stream()), "m_step")
    WFs, WFn = g.buf((513, n), torch.float32), g.buf((513, n), torch.float32)
    N.check(lib.dvae_mcem_wiener(N.ptr(Vs), R, n, N.ptr(g2), N.ptr(Vb2), N.ptr(WFs), N.ptr(WFn), N.stream()), "wiener")
    assert torch.isfinite(WFs).all() and torch.isfinite(cost).all()
    g.check()


@pytest.mark.parametrize("n,cols", [(1, 513), (1000, 513), (4097, 1)])
def test_frames_and_labels_outputs(n, cols):
    lib = N.load()
    g = Guarded()
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    src = torch.rand(cols, n, device="cuda", generator=gen)
    rows = g.buf((n, cols), torch.float32)
    N.check(lib.dvae_transpose(N.ptr(src), cols, n, n, N.ptr(rows), cols, N.stream()), "transpose")
    idx = torch.randperm(n, device="cuda", generator=gen)
    dst = g.buf((n, cols), torch.float32)
    N.check(lib.dvae_gather_rows(N.ptr(rows), cols, n, N.ptr(idx), n, cols, N.ptr(dst), cols, None, N.stream()), "gather")
    wav = torch.randn(1024 + 256 * (n % 50 + 3) + 9, dtype=torch.float64, device="cuda", generator=gen)
    T = 1 + (wav.numel() + 256 - 1024) // 256
    vad = g.buf((T,), torch.float32)
    ws = g.buf((lib.dvae_vad_workspace_bytes(T),), torch.uint8)
    N.check(lib.dvae_vad_labels(N.ptr(wav), 1, wav.numel(), 1024, 256, T, 1.7, N.ptr(vad), N.ptr(ws), N.stream()), "vad")
    S = torch.randn(513, T, dtype=torch.complex64, device="cuda")
    mask = g.buf((513, T), torch.float32)
    ws2 = g.buf((lib.dvae_ibm_workspace_bytes(),), torch.uint8)
    N.check(lib.dvae_ibm_labels(N.ptr(torch.view_as_real(S)), 513, T, 1e-8, 50.0, None, N.ptr(mask), N.ptr(ws2), N.stream()), "ibm")
    g.check()
    assert torch.equal(dst, rows[idx])
