"""Host side of STFT / ISTFT: integer / indexing decisions in Python doubles exactly
as the reference makes them (packages/processing/stft.py:34-50: window length, hop,
the floating-point end-pad rule, quirk Q6), then the device kernels of
csrc/stft.hip through the C ABI.  No CPU transform exists here: without the
library or a GPU these functions raise.
"""
import math
import os

import numpy as np
import torch

from . import native as N


def sizes(fs, wlen_sec, hop_percent, what="STFT"):
    """nfft / hop exactly as packages/processing/stft.py:34-37 (raises on non-integer window)."""
    if wlen_sec * fs != int(wlen_sec * fs):
        raise ValueError("wlen_sample of %s is not an integer." % what)
    nfft = int(wlen_sec * fs)
    hopsamp = int(hop_percent * nfft)
    return nfft, hopsamp


def needs_end_pad(n, fs, wlen_sec, hop_percent):
    """packages/processing/stft.py:45-47, same operation order in Python doubles (bit-exact
    frame indexing depends on it: some exact multiples of the hop ARE padded)."""
    utt_len = n / fs
    return math.ceil(utt_len / wlen_sec / hop_percent) != int(utt_len / wlen_sec / hop_percent)


def frame_count(n_padded, nfft, hop):
    """librosa.util.frame: 1 + (len - n_fft) // hop."""
    if n_padded < nfft:
        raise ValueError("Input signal length=%d is too small to analyze with n_fft=%d" % (n_padded, nfft))
    return 1 + (n_padded - nfft) // hop


_window_cache = {}


def window_f64(win, nfft, device):
    """Periodic window as float64 on `device` (librosa: scipy.signal.get_window(win, n, fftbins=True))."""
    key = (str(win), int(nfft), str(device))
    w = _window_cache.get(key)
    if w is None:
        from scipy.signal import get_window
        w = torch.from_numpy(np.ascontiguousarray(get_window(win, nfft, fftbins=True), dtype=np.float64)).to(device)
        _window_cache[key] = w
    return w


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("STFT/ISTFT run on the MI355X HIP path only: no GPU is visible (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def stft_device(x_dev, window, nfft, hop, T, layout=0):
    """x_dev: 1-D float32/float64 CUDA tensor already padded.  layout 0 -> complex64 [F, T];
    layout 1 -> float32 power frames [T, F]; layout 2 -> complex64 [T, F] (frame-major: the values of layout 0 in the memory
    order of librosa's Fortran-ordered result; `.T` of it is the reference's array, strides included)."""
    lib = N.load()
    if not x_dev.is_cuda or x_dev.dim() != 1 or x_dev.dtype not in (torch.float32, torch.float64):
        raise TypeError("stft_device: 1-D float32/float64 CUDA tensor required")
    x_dev = x_dev.contiguous()
    F = nfft // 2 + 1
    if layout == 0:
        out = torch.empty((F, T), dtype=torch.complex64, device=x_dev.device)
    elif layout == 2:
        out = torch.empty((T, F), dtype=torch.complex64, device=x_dev.device)
    else:
        out = torch.empty((T, F), dtype=torch.float32, device=x_dev.device)
    N.check(lib.dvae_stft(N.ptr(x_dev), 1 if x_dev.dtype == torch.float64 else 0, x_dev.numel(), N.ptr(window), nfft, hop, T,
                          N.ptr(out), layout, N.stream()), "dvae_stft")
    return out


def window_f32(nfft, device):
    """torch.hann_window(nfft) (periodic, float32) on `device`: the window of the reference's stft_pytorch (stft.py:141)."""
    key = ("hann_f32", int(nfft), str(device))
    w = _window_cache.get(key)
    if w is None:
        w = torch.hann_window(window_length=nfft).to(device)
        _window_cache[key] = w
    return w


def f32_transform_covers(x_dev, nfft, hop, T):
    """dvae_stft_f32: nfft 1024 / hop 256, float32 signal, 32-bit byte offsets."""
    return (nfft == 1024 and hop == 256 and x_dev.dtype == torch.float32 and x_dev.numel() * 4 < 2 ** 31 and T * 513 * 8 < 2 ** 31
            and os.environ.get("DVAE_STFT_F32", "1") != "0")


def stft_device_f32(x_dev, nfft, hop, T, layout=2):
    """The float32-ARITHMETIC transform (torch.stft's on a float32 tensor: packages/processing/stft.py:123-152): x_dev 1-D float32 CUDA
    tensor already padded -> layout 2: complex64 [T, F] (frame-major; `.T` is the [F, T] result), layout 1: float32 [T, F] re^2 + im^2."""
    lib = N.load()
    if not x_dev.is_cuda or x_dev.dim() != 1 or x_dev.dtype != torch.float32:
        raise TypeError("stft_device_f32: 1-D float32 CUDA tensor required")
    x_dev = x_dev.contiguous()
    F = nfft // 2 + 1
    out = torch.empty((T, F), dtype=torch.complex64 if layout == 2 else torch.float32, device=x_dev.device)
    N.check(lib.dvae_stft_f32(N.ptr(x_dev), x_dev.numel(), N.ptr(window_f32(nfft, x_dev.device)), nfft, hop, T, N.ptr(out), layout, N.stream()),
            "dvae_stft_f32")
    return out


def istft_device(S_dev, window, nfft, hop, n_frames, start, out_len):
    """S_dev: complex64 [F, >= n_frames] CUDA tensor -> float32 [out_len].  A tensor whose memory is frame-major (the `.T` view of
    a contiguous [T, F] tensor, e.g. of stft_device(..., layout=2)) is read in place by the frame-major kernel; anything else is
    made row-contiguous first."""
    lib = N.load()
    if not S_dev.is_cuda or S_dev.dtype != torch.complex64 or S_dev.dim() != 2 or S_dev.shape[0] != nfft // 2 + 1:
        raise TypeError("istft_device: complex64 [nfft/2+1, T] CUDA tensor required")
    y = torch.empty((out_len,), dtype=torch.float32, device=S_dev.device)
    if S_dev.shape[1] > 1 and S_dev.stride(0) == 1 and S_dev.stride(1) >= S_dev.shape[0] and nfft == 1024 and hop == 256:
        # the frame-major walk needs no scratch; the A/B switches that route it to the two-pass kernels do (frames in double)
        two_pass = os.environ.get("DVAE_STFT_LEGACY") is not None or os.environ.get("DVAE_ISTFT_2PASS") is not None
        ws = torch.empty(max(lib.dvae_istft_workspace_bytes(n_frames, nfft), 16) if two_pass else 16, dtype=torch.uint8, device=S_dev.device)
        N.check(lib.dvae_istft_frames(N.ptr(S_dev), n_frames, S_dev.stride(1), N.ptr(window), nfft, hop, start, N.ptr(y), out_len,
                                      N.ptr(ws), N.stream()), "dvae_istft_frames")
        return y
    ws = torch.empty(max(lib.dvae_istft_workspace_bytes_hop(n_frames, nfft, hop), 16), dtype=torch.uint8, device=S_dev.device)
    S_dev = S_dev.contiguous()
    N.check(lib.dvae_istft(N.ptr(S_dev), n_frames, S_dev.shape[1], N.ptr(window), nfft, hop, start, N.ptr(y), out_len,
                           N.ptr(ws), N.stream()), "dvae_istft")
    return y


def f32_inverse_covers(S_dev, nfft, hop):
    """dvae_istft_f32: nfft 1024 / hop 256, complex64 [513, T], 32-bit byte offsets."""
    return (nfft == 1024 and hop == 256 and S_dev.dtype == torch.complex64 and S_dev.dim() == 2 and S_dev.shape[0] == 513
            and max(S_dev.shape[1] * 513, S_dev.shape[1] * S_dev.stride(1) if S_dev.stride(0) == 1 else 0) * 8 < 2 ** 31
            and os.environ.get("DVAE_ISTFT_F32", "1") != "0")


def istft_device_f32(S_dev, nfft, hop, n_frames, start, out_len):
    """The float32-ARITHMETIC inverse transform (torch.istft's on a complex64 tensor: packages/processing/stft.py:154-190): S_dev
    complex64 [513, >= n_frames] CUDA tensor -> float32 [out_len].  Frame-major memory (the `.T` view of a contiguous [T, 513] tensor,
    what stft_pytorch returns) is read in place; a row-contiguous tensor is transposed on the device first."""
    lib = N.load()
    if not S_dev.is_cuda or S_dev.dtype != torch.complex64 or S_dev.dim() != 2 or S_dev.shape[0] != nfft // 2 + 1:
        raise TypeError("istft_device_f32: complex64 [nfft/2+1, T] CUDA tensor required")
    y = torch.empty((out_len,), dtype=torch.float32, device=S_dev.device)
    w = window_f32(nfft, S_dev.device)
    if S_dev.shape[1] > 1 and S_dev.stride(0) == 1 and S_dev.stride(1) >= S_dev.shape[0]:
        N.check(lib.dvae_istft_f32(N.ptr(S_dev), n_frames, S_dev.stride(1), 1, N.ptr(w), nfft, hop, start, N.ptr(y), out_len, None, N.stream()),
                "dvae_istft_f32")
        return y
    S_dev = S_dev.contiguous()
    ws = torch.empty((n_frames, nfft // 2 + 1), dtype=torch.complex64, device=S_dev.device)
    N.check(lib.dvae_istft_f32(N.ptr(S_dev), n_frames, S_dev.shape[1], 0, N.ptr(w), nfft, hop, start, N.ptr(y), out_len, N.ptr(ws), N.stream()),
            "dvae_istft_f32")
    return y


def stft_numpy(x, fs, wlen_sec, win, hop_percent, center, pad_mode, pad_at_end, dtype, layout=0):
    """numpy in / numpy out body of packages.processing.stft.stft."""
    nfft, hop = sizes(fs, wlen_sec, hop_percent, "STFT")
    x = np.asarray(x)
    if not np.issubdtype(x.dtype, np.floating):
        raise TypeError("stft: audio must be floating point (as librosa requires)")
    x_ = x
    if pad_at_end and needs_end_pad(len(x), fs, wlen_sec, hop_percent):
        x_ = np.pad(x, (0, hop), mode="constant")
    # pad_at_end=False: the reference leaves x_ undefined (quirk Q7); here it means "no end pad"
    if center:
        x_ = np.pad(x_, int(nfft // 2), mode=pad_mode)
    T = frame_count(len(x_), nfft, hop)
    dev = _device()
    xin = np.ascontiguousarray(x_, dtype=np.float64 if x_.dtype != np.float32 else np.float32)
    # the complex result is computed frame-major (whole frames leave the kernel as contiguous rows) and returned as the
    # transpose view: a Fortran-ordered [F, T] array, which is also what librosa.stft hands the reference
    out = stft_device(torch.from_numpy(xin).to(dev), window_f64(win, nfft, dev), nfft, hop, T, 2 if layout == 0 else layout)
    res = out.cpu().numpy()
    if layout == 0:
        res = res.T
        if np.dtype(dtype) != res.dtype:
            res = res.astype(dtype)
    return res


def istft_numpy(Sxx, fs, wlen_sec, win, hop_percent, center, dtype, max_len):
    """numpy in / numpy out body of packages.processing.stft.istft (librosa.istft semantics)."""
    nfft, hop = sizes(fs, wlen_sec, hop_percent, "iSTFT")
    S = np.asarray(Sxx)
    if S.ndim != 2 or S.shape[0] != 1 + nfft // 2:
        raise ValueError("istft: expected a [%d, T] spectrogram" % (1 + nfft // 2))
    n_frames = S.shape[1]
    if max_len:
        padded = max_len + nfft if center else max_len
        n_frames = min(n_frames, int(np.ceil(padded / hop)))
    ntot = nfft + hop * (n_frames - 1)
    start = nfft // 2 if center else 0
    if max_len is None:
        out_len = ntot - 2 * (nfft // 2) if center else ntot
    else:
        out_len = int(max_len)
    dev = _device()
    # frame-major on the device (each frame one contiguous row): free for a Fortran-ordered S (what stft() returns), one host
    # transpose for a C-ordered one
    S_dev = torch.from_numpy(np.ascontiguousarray(S.T, dtype=np.complex64)).to(dev).T
    y = istft_device(S_dev, window_f64(win, nfft, dev), nfft, hop, n_frames, start, out_len).cpu().numpy()
    if np.dtype(dtype) != y.dtype:
        y = y.astype(dtype)
    if max_len:
        y = y[:int(max_len * fs)]      # quirk Q8: max_len is already in samples, so this is a no-op
    return y
