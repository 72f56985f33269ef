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
