"""numpy restatement of the reference STFT / ISTFT path (TEST ORACLE, not product code).

The wrapper logic follows packages/processing/stft.py:13-60 (stft) and :63-99
(istft).  The transform itself lives in librosa (unpinned third-party
dependency of the reference, not vendored, not installable here; the
reference's comments cite the librosa 0.7.2 docs, scripts/reconstruct_M2.py:46);
its published algorithm is restated here:

  librosa.core.stft : periodic window scipy.signal.get_window(win, n_fft,
      fftbins=True) in float64; optional np.pad(y, n_fft//2, mode=pad_mode)
      when center; frames y[t*hop : t*hop+n_fft], T = 1 + (len-n_fft)//hop;
      rfft(window * frame) in the input precision (float64), cast to `dtype`.
  librosa.core.istft: y = zeros(n_fft + hop*(T-1), dtype); per frame
      y[t*hop:+n_fft] += window * irfft(S[:, t]); window sum-square built the
      same way in `dtype`; y[wss > tiny] /= wss[wss > tiny]; start = n_fft//2
      if center else 0; fix_length(y[start:], length).

Pin: forward STFT is pinned against the reference's own HDF5 fixtures
(tests/golden/stft_ref_fixture.npz, made by tests/golden/make_stft_golden.py)
and the pad-rule known answers of SURVEY.md 8c.  ISTFT: PARITY UNPINNED by the
reference (no fixture, librosa absent); cross-checked with torch.istft only.
"""
import math
import numpy as np
from scipy.signal import get_window


def stft_sizes(fs, wlen_sec, hop_percent, what="STFT"):
    """packages/processing/stft.py:34-37."""
    if wlen_sec * fs != int(wlen_sec * fs):
        raise ValueError("wlen_sample of %s is not an integer." % what)
    nfft = int(wlen_sec * fs)
    hopsamp = int(hop_percent * nfft)
    return nfft, hopsamp


def pad_decision(n, fs, wlen_sec, hop_percent):
    """End-pad rule, packages/processing/stft.py:45-50, evaluated in Python
    doubles with the reference's operation order (quirk Q6: exact multiples of
    the hop are sometimes padded because of rounding)."""
    utt_len = n / fs
    return math.ceil(utt_len / wlen_sec / hop_percent) != int(utt_len / wlen_sec / hop_percent)


def frame_count(n_padded, nfft, hop):
    """librosa.util.frame: 1 + (len - frame_length) // hop."""
    if n_padded < nfft:
        raise ValueError("Input signal length=%d is too small for n_fft=%d" % (n_padded, nfft))
    return 1 + (n_padded - nfft) // hop


def stft(x, fs=16e3, wlen_sec=50e-3, win="hann", hop_percent=0.25, center=True,
         pad_mode="reflect", pad_at_end=True, dtype="complex64"):
    """packages/processing/stft.py:13-60 + librosa.core.stft semantics."""
    nfft, hop = stft_sizes(fs, wlen_sec, hop_percent, "STFT")
    x = np.asarray(x)
    if pad_at_end:
        x_ = np.pad(x, (0, hop), mode="constant") if pad_decision(len(x), fs, wlen_sec, hop_percent) else x
    else:
        raise NameError("name 'x_' is not defined")  # quirk Q7 (stft.py:45-52)
    window = get_window(win, nfft, fftbins=True)
    if center:
        x_ = np.pad(x_, nfft // 2, mode=pad_mode)
    T = frame_count(len(x_), nfft, hop)
    idx = np.arange(nfft)[:, None] + hop * np.arange(T)[None, :]
    frames = x_[idx]
    return np.fft.rfft(window[:, None] * frames, axis=0).astype(dtype)


def window_sumsquare(win, n_frames, nfft, hop, dtype):
    """librosa.filters.window_sumsquare (norm=None)."""
    n = nfft + hop * (n_frames - 1)
    out = np.zeros(n, dtype=dtype)
    win_sq = get_window(win, nfft, fftbins=True) ** 2
    for i in range(n_frames):
        s = i * hop
        out[s:min(n, s + nfft)] += win_sq[:max(0, min(nfft, n - s))]
    return out


def istft(Sxx, fs=16000, wlen_sec=50e-3, win="hann", hop_percent=0.25, center=True,
          dtype="float32", max_len=None):
    """packages/processing/stft.py:63-99 + librosa.core.istft semantics."""
    nfft, hop = stft_sizes(fs, wlen_sec, hop_percent, "iSTFT")
    Sxx = np.asarray(Sxx)
    assert Sxx.shape[0] == 1 + nfft // 2
    window = get_window(win, nfft, fftbins=True)
    n_frames = Sxx.shape[1]
    if max_len:
        padded = max_len + nfft if center else max_len
        n_frames = min(n_frames, int(np.ceil(padded / hop)))
    n = nfft + hop * (n_frames - 1)
    y = np.zeros(n, dtype=dtype)
    # reference-era numpy computes irfft in double whatever the input precision
    ytmp = window[:, None] * np.fft.irfft(Sxx[:, :n_frames].astype(np.complex128), n=nfft, axis=0)
    for t in range(n_frames):
        y[t * hop:t * hop + nfft] += ytmp[:, t]
    wss = window_sumsquare(win, n_frames, nfft, hop, dtype)
    nz = wss > np.finfo(wss.dtype).tiny
    y[nz] /= wss[nz]
    if max_len is None:
        if center:
            y = y[nfft // 2:-(nfft // 2)]
    else:
        start = nfft // 2 if center else 0
        y = y[start:]
        if len(y) > max_len:
            y = y[:max_len]
        elif len(y) < max_len:
            y = np.pad(y, (0, max_len - len(y)), mode="constant")
    if max_len:
        y = y[:int(max_len * fs)]  # quirk Q8: no-op slice (max_len is already in samples)
    return y


def power_spectrogram(x, **kw):
    """scripts/create_train_set.py:138-152 / scripts/reconstruct_M2.py:143-153: |STFT|^2 as float32."""
    return (np.abs(stft(x, **kw)) ** 2).astype(np.float32)
