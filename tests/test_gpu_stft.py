"""GPU parity of STFT / ISTFT (packages.processing.stft on the HIP path) against the numpy
oracle, the reference's own HDF5 power frames, and round-trip properties."""
import os
import numpy as np
import pytest
import torch

from oracle import stft_oracle as so
from packages.processing import stft as ps

pytestmark = pytest.mark.gpu

KW = dict(fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=False)


def _c64_close(got, ref, tol):
    scale = np.abs(ref).max()
    assert got.shape == ref.shape and got.dtype == ref.dtype
    assert np.abs(got - ref).max() <= tol * scale


@pytest.mark.parametrize("n", [1024, 1279, 1280, 11008, 16000, 32000, 73045, 82944])
def test_stft_matches_oracle_bitwise_indexing(n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) * np.exp(rng.standard_normal(n))
    ref = so.stft(x, **KW)
    got = ps.stft(x, **KW)
    assert got.shape == ref.shape                       # frame count / pad rule: exact
    _c64_close(got, ref, 2e-7)
    assert np.mean(got == ref) > 0.95                   # float64 transform, cast once: mostly bit-identical


@pytest.mark.parametrize("utt", ["01M_sa1", "08F_si519"])
def test_stft_reproduces_reference_h5_power_frames(stft_golden, utt):
    f = stft_golden
    x = f[utt + "_wav_head_i16"].astype(np.float64) / 32768.0
    x = x / (float(f[utt + "_peak_i16"]) / 32768.0)
    P = np.abs(ps.stft(x, **KW)) ** 2
    X = f[utt + "_X"]
    np.testing.assert_allclose(P[:, :X.shape[1]].astype(np.float32), X, rtol=4e-7)


def test_power_frame_layout_and_float32_input():
    import importlib
    H = importlib.import_module("disentangled-vae_amd.stft")
    rng = np.random.default_rng(0)
    x = rng.standard_normal(20000).astype(np.float32)
    S = ps.stft(x, **KW)
    P = H.stft_numpy(x, 16000, 64e-3, "hann", 0.25, False, "reflect", True, "complex64", layout=1)
    assert P.shape == (S.shape[1], 513) and P.dtype == np.float32
    np.testing.assert_allclose(P, (np.abs(S) ** 2).T, rtol=1e-6, atol=1e-12)
    _c64_close(S, so.stft(x, **KW), 2e-7)


def test_center_reflect_and_default_window_length():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(9000)
    kw = dict(fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=True, pad_mode="reflect")
    _c64_close(ps.stft(x, **kw), so.stft(x, **kw), 2e-7)
    got = ps.stft(x)                                      # defaults: fs=16e3, 50 ms -> nfft 800 (generic DFT path)
    ref = so.stft(x)
    assert got.shape == ref.shape == (401, ref.shape[1])
    _c64_close(got, ref, 5e-7)
    with pytest.raises(ValueError, match="not an integer"):
        ps.stft(x, fs=16000, wlen_sec=50.01e-3)


@pytest.mark.parametrize("n,center", [(16000, False), (73045, False), (20000, True)])
def test_istft_matches_oracle_and_round_trips(n, center):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n)
    kw = dict(fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=center)
    S = so.stft(x, **kw)
    ref = so.istft(S, max_len=n, **kw)
    got = ps.istft(S, max_len=n, **kw)
    assert got.shape == ref.shape == (n,) and got.dtype == np.float32
    # ISTFT: parity unpinned by the reference; compare with the restatement.  Interior: 1e-4 rel
    # (north_star); the first/last hops divide by a tiny window sum: absolute bound.
    lo, hi = 1024, n - 1024
    np.testing.assert_allclose(got[lo:hi], ref[lo:hi], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got[lo:hi], x[lo:hi], atol=3e-5)
    edge = np.r_[0:lo, hi:n]
    assert np.all(np.isfinite(got))
    assert np.max(np.abs(got[edge] - ref[edge]) / (np.abs(ref[edge]) + 1e-2)) < 1e-3
    nat = ps.istft(S, **kw)                               # max_len=None: natural length
    assert nat.shape == so.istft(S, **kw).shape


def test_istft_zero_pads_and_trims():
    rng = np.random.default_rng(2)
    x = rng.standard_normal(8000)
    S = so.stft(x, **KW)
    long = ps.istft(S, max_len=12000, **KW)
    assert long.shape == (12000,) and np.all(long[9000:] == 0)
    short = ps.istft(S, max_len=3000, **KW)
    np.testing.assert_allclose(short, so.istft(S, max_len=3000, **KW), rtol=1e-4, atol=1e-5)


def test_pytorch_variants_legacy_layout():
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal(30000).astype(np.float32))
    out = ps.stft_pytorch(x, fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=False)
    npad = 30000 + (256 if so.pad_decision(30000, 16000, 64e-3, 0.25) else 0)
    ref = torch.stft(torch.nn.functional.pad(x, (0, npad - 30000)), 1024, 256, window=torch.hann_window(1024),
                     center=False, return_complex=True)
    assert out.shape == (513, ref.shape[1], 2) and out.dtype == torch.float32 and not out.is_cuda
    assert float((torch.view_as_complex(out) - ref).abs().max()) <= 5e-6 * float(ref.abs().max())
    outc = ps.stft_pytorch(x.cuda(), fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=True)
    refc = torch.stft(torch.nn.functional.pad(x, (0, npad - 30000)), 1024, 256, window=torch.hann_window(1024),
                      center=True, pad_mode="reflect", return_complex=True)
    assert outc.is_cuda and outc.shape[:2] == refc.shape
    assert float((torch.view_as_complex(outc).cpu() - refc).abs().max()) <= 5e-6 * float(refc.abs().max())
    y = ps.istft_pytorch(outc, fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=True)
    yref = torch.istft(refc, 1024, 256, window=torch.hann_window(1024), center=True)
    assert y.shape == yref.shape
    assert float((y.cpu() - yref).abs().max()) < 1e-4


@pytest.mark.parametrize("n", [1024, 1279, 30000, 16000 * 95 + 37, 16000 * 300 + 5])
def test_float32_arithmetic_transform_matches_torch_stft(n):
    """dvae_stft_f32 -- the transform behind stft_pytorch for float32 signals (packages/processing/stft.py:123-152: torch.stft of a
    float32 tensor with torch.hann_window; window product, FFT and result in float32) -- against torch.stft itself on the host at the
    bound the double-arithmetic path was held to (5e-6 of the spectrogram's maximum), one frame per wave up to 18 frames per wave, the
    tail frame included; the power layout is the caller's x_tf[..., 0] ** 2 + x_tf[..., 1] ** 2 (packages/data_handling.py:136) of the
    same complex values; and the double-arithmetic path (DVAE_STFT_F32=0) agrees within float32 rounding of the sum."""
    import importlib
    H = importlib.import_module("disentangled-vae_amd.stft")
    rng = np.random.default_rng(n)
    x = torch.from_numpy((rng.standard_normal(n) * np.exp(0.5 * rng.standard_normal(n))).astype(np.float32))
    T = H.frame_count(n, 1024, 256)
    ref = torch.stft(x, 1024, 256, window=torch.hann_window(1024), center=False, return_complex=True)       # [513, T]
    got = H.stft_device_f32(x.cuda(), 1024, 256, T, 2)
    assert got.shape == (T, 513) and got.dtype == torch.complex64
    scale = float(ref.abs().max())
    assert float((got.T.cpu() - ref).abs().max()) <= 5e-6 * scale
    pw = H.stft_device_f32(x.cuda(), 1024, 256, T, 1)
    gr = torch.view_as_real(got)
    # (the two instantiations of the kernel are compiled separately: their transforms may differ in the last bit where the compiler
    # contracted different multiply-add pairs, so the comparison is float32 rounding of the sum, not bits)
    torch.testing.assert_close(pw, gr[..., 0] ** 2 + gr[..., 1] ** 2, rtol=2e-6, atol=2e-7 * scale * scale)
    dbl = H.stft_device(x.cuda(), H.window_f64("hann", 1024, "cuda"), 1024, 256, T, 2)
    assert float((got - dbl).abs().max()) <= 5e-6 * scale
    # stft_pytorch takes it (shape, dtype, values of the legacy [513, T, 2] real view) and can be switched back
    out = ps.stft_pytorch(x.cuda(), fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=False, pad_at_end=False)
    assert out.shape == (513, T, 2) and out.dtype == torch.float32
    assert torch.equal(torch.view_as_complex(out.contiguous()), got.T)


@pytest.mark.parametrize("T,center", [(1, False), (3, False), (4, True), (5, True), (117, False), (1024, True), (2049, True), (18760, True)])
def test_float32_arithmetic_inverse_matches_torch_istft(T, center):
    """dvae_istft_f32 -- the transform behind istft_pytorch for complex64 spectrograms (packages/processing/stft.py:154-190: torch.istft
    with torch.hann_window; inverse FFT, window product, overlap-add and envelope division in float32) -- against torch.istft itself on
    the host at 5e-6 of the signal's maximum (one frame up to ten frames per wave; centre trimming on and off), against the
    double-arithmetic path (DVAE_ISTFT_F32=0) at the same bound, frame-major memory read in place = row-contiguous input transposed on
    the device (bits), and istft_pytorch takes it for both layouts."""
    import importlib
    H = importlib.import_module("disentangled-vae_amd.stft")
    rng = np.random.default_rng(T)
    S = torch.from_numpy((rng.standard_normal((513, T)) + 1j * rng.standard_normal((513, T))).astype(np.complex64))
    S[0].imag = 0; S[512].imag = 0
    ntot = 1024 + 256 * (T - 1)
    start = 512 if center else 0
    out_len = ntot - 1024 if center else ntot
    if out_len <= 0:
        pytest.skip("nothing left after the centre trim")
    y_bin = H.istft_device_f32(S.cuda(), 1024, 256, T, start, out_len)                      # row-contiguous [513, T]
    rows = S.T.contiguous().cuda()                                                         # [T, 513]
    y_fr = H.istft_device_f32(rows.T, 1024, 256, T, start, out_len)                        # frame-major memory, in place
    assert y_bin.shape == (out_len,) and y_bin.dtype == torch.float32 and torch.equal(y_bin, y_fr)
    dbl = H.istft_device(rows.T, H.window_f64("hann", 1024, "cuda"), 1024, 256, T, start, out_len)
    # where the window envelope is small (the first and last hop of an uncentred signal) the division amplifies the float32 rounding of
    # the overlap-add by 1 / envelope, in torch.istft as here: the bound is held where the envelope is above 1 % of its plateau (1.5)
    w2 = torch.hann_window(1024, dtype=torch.float64).numpy() ** 2
    env = np.zeros(ntot)
    for tt in range(T):
        env[256 * tt:256 * tt + 1024] += w2
    ok = torch.from_numpy(env[start:start + out_len] > 0.015).cuda()
    assert bool(ok.any())
    scale = float(dbl[ok].abs().max())
    assert float((y_fr - dbl)[ok].abs().max()) <= 5e-6 * scale
    assert bool(torch.isfinite(y_fr).all())
    if ntot > 20:
        # an odd first sample (the paired stores fall back to single ones) and a length past the signal (zeros behind it)
        y_odd = H.istft_device_f32(rows.T, 1024, 256, T, 3, ntot + 5)
        full = H.istft_device_f32(rows.T, 1024, 256, T, 0, ntot)
        assert torch.equal(y_odd[:ntot - 3], full[3:]) and bool((y_odd[ntot - 3:] == 0).all())
        # rows with padding behind the 513 bins (leading dimension 520)
        wide = torch.zeros((T, 520), dtype=torch.complex64, device="cuda")
        wide[:, :513] = rows
        assert torch.equal(H.istft_device_f32(wide[:, :513].T, 1024, 256, T, 0, ntot), full)
        # a sliced spectrogram (frames 1 .. T - 1 of the same memory: leading dimension 513, first row one frame in)
        if T > 2:
            part = H.istft_device_f32(rows[1:].T, 1024, 256, T - 1, 0, 1024 + 256 * (T - 2))
            ref_part = H.istft_device_f32(rows[1:].contiguous().T, 1024, 256, T - 1, 0, 1024 + 256 * (T - 2))
            assert torch.equal(part, ref_part)
    if center or T >= 4:
        # torch.istft refuses envelopes that touch zero (the first and last hop of an uncentred signal): compare the interior
        if center:
            ref = torch.istft(S, 1024, 256, window=torch.hann_window(1024), center=True)
            assert ref.shape == (out_len,)
            assert float((y_fr.cpu() - ref).abs().max()) <= 5e-6 * scale
        got = ps.istft_pytorch(torch.view_as_real(rows.T), fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=center)
        assert torch.equal(got, y_fr)
        got_c = ps.istft_pytorch(torch.view_as_real(S).cuda(), fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=center)
        assert torch.equal(got_c, y_fr)


def test_float32_arithmetic_round_trip_at_the_benchmarked_length():
    """Ten minutes of audio (37 497 frames, the length bench.py and tools/bench_stft.py time): dvae_stft_f32 followed by dvae_istft_f32
    on the frame-major spectrogram returns the signal wherever four frames overlap (the Hann window's squares at hop 256 sum to 1.5, the
    envelope division undoes them): float32 round-off of two transforms, 2e-5 of the signal's maximum; the first and last three hops
    (fewer frames) are held to the same bound relative to their smaller envelope."""
    import importlib
    H = importlib.import_module("disentangled-vae_amd.stft")
    n = 16000 * 600
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    x = torch.randn(n, device="cuda", generator=g) * torch.exp(0.5 * torch.randn(n, device="cuda", generator=g))
    T = H.frame_count(n, 1024, 256)
    S = H.stft_device_f32(x, 1024, 256, T, 2)                                             # [T, 513]
    covered = 1024 + 256 * (T - 1)
    y = H.istft_device_f32(S.T, 1024, 256, T, 0, covered)
    scale = float(x.abs().max())
    inner = slice(768, covered - 768)
    assert float((y[inner] - x[:covered][inner]).abs().max()) <= 2e-5 * scale
    assert bool(torch.isfinite(y).all())
    # the power layout of the same transform carries the same energy (Parseval per frame is not exact under a window: compare the layouts)
    P = H.stft_device_f32(x, 1024, 256, T, 1)
    torch.testing.assert_close(P.sum(dtype=torch.float64), (torch.view_as_real(S) ** 2).sum(dtype=torch.float64), rtol=1e-6, atol=0.0)


def test_float32_arithmetic_inverse_can_be_switched_off_and_rejects_other_sizes(monkeypatch):
    import importlib
    H = importlib.import_module("disentangled-vae_amd.stft")
    N = importlib.import_module("disentangled-vae_amd.native")
    lib = N.load()
    rng = np.random.default_rng(9)
    S = torch.from_numpy((rng.standard_normal((513, 40)) + 1j * rng.standard_normal((513, 40))).astype(np.complex64)).cuda()
    a = ps.istft_pytorch(S, fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=True)
    monkeypatch.setenv("DVAE_ISTFT_F32", "0")
    b = ps.istft_pytorch(S, fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=True)          # the double-arithmetic walk
    assert a.shape == b.shape and not torch.equal(a, b) and float((a - b).abs().max()) <= 5e-6 * float(b.abs().max())
    w = H.window_f32(1024, S.device)
    y = torch.empty(1024 + 256 * 39, dtype=torch.float32, device="cuda")
    ws = torch.empty((40, 513), dtype=torch.complex64, device="cuda")
    call = lambda nfft, hop, ld, frames, wsp: lib.dvae_istft_f32(N.ptr(S), 40, ld, frames, N.ptr(w), nfft, hop, 0, N.ptr(y), y.numel(), wsp, N.stream())
    assert call(1024, 256, 40, 0, N.ptr(ws)) == 0
    assert call(1024, 128, 40, 0, N.ptr(ws)) != 0 and "hop" in lib.dvae_last_error().decode()       # hop
    assert call(512, 128, 40, 0, N.ptr(ws)) != 0                                                     # window length
    assert call(1024, 256, 39, 0, N.ptr(ws)) != 0                                                    # leading dimension
    assert call(1024, 256, 40, 0, None) != 0                                                         # bin-major input without the workspace


def test_float32_arithmetic_transform_rejects_what_it_does_not_cover():
    import importlib
    H = importlib.import_module("disentangled-vae_amd.stft")
    N = importlib.import_module("disentangled-vae_amd.native")
    lib = N.load()
    x = torch.zeros(4096, dtype=torch.float32, device="cuda")
    w = H.window_f32(1024, x.device)
    out = torch.empty((13, 513), dtype=torch.complex64, device="cuda")
    assert lib.dvae_stft_f32(N.ptr(x), 4096, N.ptr(w), 1024, 128, 13, N.ptr(out), 2, N.stream()) != 0        # hop
    assert lib.dvae_stft_f32(N.ptr(x), 4096, N.ptr(w), 512, 256, 13, N.ptr(out), 2, N.stream()) != 0         # nfft
    assert lib.dvae_stft_f32(N.ptr(x), 4096, N.ptr(w), 1024, 256, 13, N.ptr(out), 0, N.stream()) != 0        # bin-major layout
    assert lib.dvae_stft_f32(N.ptr(x), 4096, N.ptr(w), 1024, 256, 14, N.ptr(out), 2, N.stream()) != 0        # 14 frames do not fit
    assert lib.dvae_stft_f32(N.ptr(x), 4096, N.ptr(w), 1024, 256, 13, N.ptr(out), 2, N.stream()) == 0
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0


def test_large_batch_of_frames_linearity():
    """~10 minutes of audio (37.5k frames): linearity and Parseval-style energy check."""
    rng = np.random.default_rng(4)
    n = 16000 * 600
    a, b = rng.standard_normal(n), rng.standard_normal(n)
    Sa, Sb, Sab = ps.stft(a, **KW), ps.stft(b, **KW), ps.stft(2 * a - 3 * b, **KW)
    assert Sa.shape[1] == 1 + (n + (256 if so.pad_decision(n, 16000, 64e-3, 0.25) else 0) - 1024) // 256
    err = np.abs(Sab - (2 * Sa - 3 * Sb)).max() / np.abs(Sab).max()
    assert err < 1e-6
    t = 1234
    fr = a[t * 256:t * 256 + 1024] * np.hanning(1025)[:1024]
    np.testing.assert_allclose(Sa[:, t], np.fft.rfft(fr).astype(np.complex64), rtol=0, atol=2e-6 * np.abs(Sa[:, t]).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_long_audio_many_frames_per_wave(dtype):
    """>2048 frames: each wave walks several consecutive frames and carries 6 of its 8 sample pairs over from one
    frame to the next (power layout), and the complex layout runs many staged blocks; the last frame straddles the
    end pad.  Also ISTFT of the same length (staged blocks of 8 frames) back to the signal."""
    import importlib
    H = importlib.import_module("disentangled-vae_amd.stft")
    n = 16000 * 95 + 37                                     # 5 936 frames -> 3 frames per wave
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(n) * np.exp(0.5 * rng.standard_normal(n))).astype(dtype)
    ref = so.stft(x.astype(np.float64), **KW)
    got = ps.stft(x, **KW)
    assert got.shape == ref.shape and got.shape[1] > 2 * 2048
    _c64_close(got, ref, 2e-7)
    P = H.stft_numpy(x, 16000, 64e-3, "hann", 0.25, False, "reflect", True, "complex64", layout=1)
    np.testing.assert_allclose(P, (np.abs(ref) ** 2).T, rtol=2e-6, atol=1e-10)
    y = ps.istft(got, max_len=n, **KW)
    lo, hi = 1024, n - 1280
    np.testing.assert_allclose(y[lo:hi], x[lo:hi].astype(np.float32), atol=2e-5 * np.abs(x).max())


@pytest.mark.parametrize("n,center", [(300, False), (16000, False), (73045, True), (16000 * 90, False)])
def test_istft_fused_overlap_add_equals_two_pass_bitwise(n, center):
    """The one-kernel ISTFT (inverse FFT + overlap-add in LDS, frame order, one float rounding per addition) gives the very
    bits of the frames-to-scratch + gather form (DVAE_ISTFT_2PASS), chunk borders, halo frames and the tail included."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal(max(n, 1024))
    kw = dict(fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=center)
    S = so.stft(x, **kw)
    for max_len in (None, len(x), len(x) + 5000, max(len(x) - 700, 1)):
        fused = ps.istft(S, max_len=max_len, **kw)
        os.environ["DVAE_ISTFT_2PASS"] = "1"
        try:
            two = ps.istft(S, max_len=max_len, **kw)
        finally:
            del os.environ["DVAE_ISTFT_2PASS"]
        assert fused.shape == two.shape and fused.dtype == two.dtype == np.float32
        assert np.array_equal(fused.view(np.uint32), two.view(np.uint32))


@pytest.mark.parametrize("n,dtype,wlen", [(1024, np.float64, 64e-3), (73045, np.float32, 64e-3), (16000 * 140 + 11, np.float64, 64e-3),
                                          (20000, np.float64, 32e-3), (20000, np.float32, 50e-3)])
def test_frame_major_complex_layout_is_the_bin_major_one_transposed(n, dtype, wlen):
    """dvae_stft layout 2 ([T, F] complex64, what stft() now computes) holds the very values of layout 0 ([F, T]); stft() returns
    its transpose view: the reference's shape with the memory order of librosa's Fortran-ordered result
    (packages/processing/stft.py:50-57).  Covers the one-wave-per-frame kernel (1024) with one and several frames per wave and
    the generic power-of-two (512) and plain-DFT (800) kernels."""
    import importlib
    H = importlib.import_module("disentangled-vae_amd.stft")
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(dtype)
    nfft, hop = H.sizes(16000, wlen, 0.25)
    pad = hop if H.needs_end_pad(n, 16000, wlen, 0.25) else 0
    xd = torch.from_numpy(np.pad(x, (0, pad))).cuda()
    T = H.frame_count(n + pad, nfft, hop)
    w = H.window_f64("hann", nfft, xd.device)
    a = H.stft_device(xd, w, nfft, hop, T, 0)
    b = H.stft_device(xd, w, nfft, hop, T, 2)
    assert a.shape == (nfft // 2 + 1, T) and b.shape == (T, nfft // 2 + 1)
    assert torch.equal(torch.view_as_real(a), torch.view_as_real(b.T.contiguous()))
    got = ps.stft(x, fs=16000, wlen_sec=wlen, hop_percent=0.25, center=False)
    assert got.shape == a.shape and got.dtype == np.complex64 and got.flags.f_contiguous
    assert np.array_equal(np.ascontiguousarray(got).view(np.float32), a.cpu().numpy().view(np.float32))


@pytest.mark.parametrize("T", [1, 5, 300, 2560, 2561, 6656, 6657, 9000])
def test_istft_of_frame_major_input_equals_bin_major_bitwise(T):
    """dvae_istft_frames (S as [T, ld] rows, whole frames staged per load) against dvae_istft (S as [513, T]) on the same values:
    identical bits for every chunk size the host picks (T <= 2560: 8-frame passes, <= 6656: 16, above: 2 x 16), a padded row
    stride, and through the numpy wrapper for C- and Fortran-ordered spectrograms."""
    import importlib
    H = importlib.import_module("disentangled-vae_amd.stft")
    rng = np.random.default_rng(T)
    S = (rng.standard_normal((513, T)) + 1j * rng.standard_normal((513, T))).astype(np.complex64)
    w = H.window_f64("hann", 1024, torch.device("cuda", 0))
    ntot = 1024 + 256 * (T - 1)
    S_ft = torch.from_numpy(S).cuda()                                       # [513, T] row-contiguous: the bin-major kernel
    assert S_ft.stride(1) == 1 or T == 1
    ref = H.istft_device(S_ft, w, 1024, 256, T, 0, ntot + 300)
    rows = torch.from_numpy(np.ascontiguousarray(S.T)).cuda()              # [T, 513]
    got = H.istft_device(rows.T, w, 1024, 256, T, 0, ntot + 300)                  # register-resident walk (one wave per run of frames)
    N_ = importlib.import_module("disentangled-vae_amd.native")
    diag = bool(N_.load().dvae_build_has_diag())      # the staged kernel on frame-major rows exists in the diagnostic library only
    os.environ["DVAE_ISTFT_STAGED"] = "1"
    try:
        got_staged = H.istft_device(rows.T, w, 1024, 256, T, 0, ntot + 300) if diag else got   # the LDS-staged kernel reading frame-major rows
        ref_staged = H.istft_device(S_ft, w, 1024, 256, T, 0, ntot + 300)         # ... and bin-major rows (T >= 1024 otherwise transposes + walks)
    finally:
        del os.environ["DVAE_ISTFT_STAGED"]
    assert torch.equal(ref.view(torch.int32), ref_staged.view(torch.int32))
    wide = torch.zeros((T, 520), dtype=torch.complex64, device="cuda")
    wide[:, :513] = rows
    got_wide = H.istft_device(wide[:, :513].T, w, 1024, 256, T, 0, ntot + 300)
    if T > 1:
        assert rows.T.stride(0) == 1 and wide[:, :513].T.stride(1) == 520
        assert torch.equal(ref.view(torch.int32), got.view(torch.int32))
        assert torch.equal(ref.view(torch.int32), got_wide.view(torch.int32))
        assert torch.equal(ref.view(torch.int32), got_staged.view(torch.int32))
        for st, ln in ((512, ntot - 1024), (512, 777), (0, 1), (300, ntot)):     # centre trim, odd lengths, a start inside the first hop
            r2 = H.istft_device(S_ft, w, 1024, 256, T, st, ln)
            g2 = H.istft_device(rows.T, w, 1024, 256, T, st, ln)
            assert torch.equal(r2.view(torch.int32), g2.view(torch.int32))
    kw = dict(fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=False)
    y_c = ps.istft(S, **kw)
    y_f = ps.istft(np.asfortranarray(S), **kw)
    assert np.array_equal(y_c.view(np.uint32), y_f.view(np.uint32))
    assert np.array_equal(y_c.view(np.uint32), ref[:ntot].cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("T", [300, 1500])
def test_istft_c_abi_with_a_padded_leading_dimension(T):
    """dvae_istft on a [513, ldT] buffer with ldT > T (the first T columns are the frames): the staged kernel (T < 1024) and the
    transposing pass + walk (T >= 1024) both honour the row stride -- same bits as the tightly packed spectrogram."""
    import importlib
    H = importlib.import_module("disentangled-vae_amd.stft")
    N = importlib.import_module("disentangled-vae_amd.native")
    lib = N.load()
    rng = np.random.default_rng(T)
    S = (rng.standard_normal((513, T)) + 1j * rng.standard_normal((513, T))).astype(np.complex64)
    w = H.window_f64("hann", 1024, torch.device("cuda", 0))
    ntot = 1024 + 256 * (T - 1)
    tight = torch.from_numpy(S).cuda()
    ref = H.istft_device(tight, w, 1024, 256, T, 0, ntot)
    ld = T + 7
    wide = torch.full((513, ld), complex(7.0, -3.0), dtype=torch.complex64, device="cuda")      # the padding must never be read as a frame
    wide[:, :T] = tight
    y = torch.empty(ntot, dtype=torch.float32, device="cuda")
    ws = torch.empty(max(lib.dvae_istft_workspace_bytes_hop(T, 1024, 256), 16), dtype=torch.uint8, device="cuda")
    N.check(lib.dvae_istft(N.ptr(wide), T, ld, N.ptr(w), 1024, 256, 0, N.ptr(y), ntot, N.ptr(ws), N.stream()), "dvae_istft")
    assert torch.equal(ref.view(torch.int32), y.view(torch.int32))
