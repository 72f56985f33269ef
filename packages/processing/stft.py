"""Drop-in replacement for the reference's packages/processing/stft.py.

Same four functions and keyword signatures: `stft` / `istft` (numpy in, numpy
out; librosa semantics, reference lines 13-60 and 63-99) and `stft_pytorch` /
`istft_pytorch` (tensor in, tensor out; reference lines 102-193).  The window
length / hop / end-pad decisions are made on the host in Python doubles exactly
as the reference makes them; the transforms are hand-written HIP kernels
(batched real FFT in LDS, windowed overlap-add) reached through libdvae_hip.so.
librosa is not imported and there is no CPU transform: without a GPU these
functions raise.
"""
import torch

from packages import _native


def stft(x,
         fs=16e3,
         wlen_sec=50e-3,
         win='hann',
         hop_percent=0.25,
         center=True,
         pad_mode='reflect',
         pad_at_end=True,
         dtype='complex64'):
    """x: time series (float).  Returns Sxx [1 + nfft/2, T] complex (null frequency included)."""
    return _native.stft_host().stft_numpy(x, fs, wlen_sec, win, hop_percent, center, pad_mode, pad_at_end, dtype)


def istft(Sxx,
          fs=16000,
          wlen_sec=50e-3,
          win='hann',
          hop_percent=0.25,
          center=True,
          dtype='float32',
          max_len=None):
    """Sxx: [1 + nfft/2, T] complex.  Returns the time signal (trimmed / zero padded to max_len samples)."""
    return _native.stft_host().istft_numpy(Sxx, fs, wlen_sec, win, hop_percent, center, dtype, max_len)


def stft_pytorch(x,
                 fs=16e3,
                 wlen_sec=50e-3,
                 win='hann',
                 hop_percent=0.25,
                 center=True,
                 pad_mode='reflect',
                 pad_at_end=True):
    """x: 1-D float tensor.  Returns the legacy real view [1 + nfft/2, T, 2] (float32) that the
    reference's pre-1.8 torch.stft call produced (quirk Q9: callers index [..., 0] / [..., 1])."""
    H = _native.stft_host()
    nfft, hop = H.sizes(fs, wlen_sec, hop_percent, "STFT")
    x_ = x
    if pad_at_end and H.needs_end_pad(len(x), fs, wlen_sec, hop_percent):
        x_ = torch.nn.functional.pad(x, (0, hop), mode='constant')
    if center:
        x_ = torch.nn.functional.pad(x_[None, None], (nfft // 2, nfft // 2), mode=pad_mode)[0, 0]
    T = H.frame_count(x_.numel(), nfft, hop)
    dev = x_.device if x_.is_cuda else H._device()
    if win != 'hann':
        raise ValueError("stft_pytorch: only win='hann' is defined (as in the reference)")
    xin = x_.to(dev)
    if xin.dtype not in (torch.float32, torch.float64):
        xin = xin.to(torch.float32)
    if H.f32_transform_covers(xin, nfft, hop, T):
        # torch.stft's own arithmetic for a float32 signal (window product, FFT, result: float32), computed frame-major; the [F, T, 2]
        # result is the transpose view of that memory -- as the legacy torch.stft's was (it transposed its [T, F, 2] transform in place)
        out = torch.view_as_real(H.stft_device_f32(xin, nfft, hop, T, 2).T)
    else:
        window = torch.hann_window(window_length=nfft).to(torch.float64).to(dev)
        out = torch.view_as_real(H.stft_device(xin, window, nfft, hop, T, 0))
    return out if x.is_cuda else out.cpu()


def istft_pytorch(Sxx,
                  fs=16000,
                  wlen_sec=50e-3,
                  win='hann',
                  hop_percent=0.25,
                  center=True,
                  max_len=None):
    """Sxx: [1 + nfft/2, T, 2] real view (or complex) tensor -> 1-D float32 tensor."""
    H = _native.stft_host()
    nfft, hop = H.sizes(fs, wlen_sec, hop_percent, "iSTFT")
    if win != 'hann':
        raise ValueError("istft_pytorch: only win='hann' is defined (as in the reference)")
    if torch.is_complex(Sxx):
        S = Sxx
    elif Sxx.dim() == 3 and Sxx.shape[2] == 2 and Sxx.stride(2) == 1 and Sxx.stride(0) % 2 == 0 and Sxx.stride(1) % 2 == 0:
        S = torch.view_as_complex(Sxx)                     # e.g. the real view stft_pytorch returned: frame-major memory, read in place
    else:
        S = torch.view_as_complex(Sxx.contiguous())
    dev = S.device if S.is_cuda else H._device()
    T = S.shape[1]
    ntot = nfft + hop * (T - 1)
    start = nfft // 2 if center else 0
    out_len = ntot - 2 * (nfft // 2) if center else ntot
    S = S.to(dev).to(torch.complex64)
    if H.f32_inverse_covers(S, nfft, hop):
        # torch.istft's own arithmetic for a complex64 spectrogram (inverse FFT, window, overlap-add, envelope division: float32)
        y = H.istft_device_f32(S, nfft, hop, T, start, out_len)
    else:
        window = torch.hann_window(window_length=nfft).to(torch.float64).to(dev)
        y = H.istft_device(S, window, nfft, hop, T, start, out_len)
    if max_len:
        y = y[:int(max_len * fs)]
    return y if Sxx.is_cuda else y.cpu()
