"""Device side of the label makers (include/dvae.h: dvae_vad_labels, dvae_ibm_labels) and the fused front end
wav -> (power frames [T,513], labels [T,y_dim]) in the training layout (reference scripts/create_train_set.py:133-194)."""
import numpy as np
import torch

from . import native as N
from . import stft as H


def _dev(t):
    if t.is_cuda:
        return t
    if not torch.cuda.is_available():
        raise RuntimeError("label kernels need the MI355X HIP path (no CPU fallback)")
    return t.cuda()


def vad_labels(y, nfft, hop, frames, vad_threshold=1.70):
    """y: 1-D float32/float64 samples (center padding, if any, already applied; the zero end-pad is implied).  -> (frames) float32."""
    lib = N.load()
    y = _dev(y)
    if y.dtype not in (torch.float32, torch.float64):
        y = y.to(torch.float64)
    y = y.contiguous()
    vad = torch.empty(frames, dtype=torch.float32, device=y.device)
    ws = torch.empty(lib.dvae_vad_workspace_bytes(frames), dtype=torch.uint8, device=y.device)
    N.check(lib.dvae_vad_labels(N.ptr(y), 1 if y.dtype == torch.float64 else 0, y.numel(), nfft, hop, frames, float(vad_threshold),
                                N.ptr(vad), N.ptr(ws), N.stream()), "dvae_vad_labels")
    return vad


def ibm_labels(S, eps=1e-8, ibm_threshold=50, vad_gate=None):
    """S: complex64 (rows, cols).  -> float32 mask of the same shape; vad_gate (cols) multiplies each column."""
    lib = N.load()
    S = _dev(S)
    if S.dtype != torch.complex64:
        raise TypeError(f"ibm_labels: complex64 expected, got {S.dtype}")
    S = S.contiguous()
    rows, cols = S.shape
    mask = torch.empty((rows, cols), dtype=torch.float32, device=S.device)
    ws = torch.empty(lib.dvae_ibm_workspace_bytes(), dtype=torch.uint8, device=S.device)
    gate = None if vad_gate is None else _dev(vad_gate).to(torch.float32).contiguous()
    N.check(lib.dvae_ibm_labels(N.ptr(torch.view_as_real(S)), rows, cols, float(eps), float(ibm_threshold), N.ptr(gate), N.ptr(mask),
                                N.ptr(ws), N.stream()), "dvae_ibm_labels")
    return mask


def utterance_to_frames(speech, labels="vad_labels", fs=16000, wlen_sec=64e-3, hop_percent=0.25, vad_threshold=1.70, eps=1e-8,
                        ibm_threshold=50, device="cuda:0"):
    """One utterance of the training-set builder (scripts/create_train_set.py:133-170) without leaving the GPU:
    speech (float64 samples as soundfile returns them) -> peak-normalise -> STFT (hann, center=False, end-pad rule)
    -> X = |S|^2 as [T, 513] float32 rows, Y = VAD [T, 1] or IBM [T, 513] rows (the layout the train step reads)."""
    lib = N.load()
    speech = np.asarray(speech, dtype=np.float64)
    speech = speech / np.max(np.abs(speech))                              # create_train_set.py:137
    nfft, hop = H.sizes(fs, wlen_sec, hop_percent, "STFT")
    n = len(speech)
    pad = hop if H.needs_end_pad(n, fs, wlen_sec, hop_percent) else 0
    x = torch.from_numpy(speech).to(device)
    if pad:
        x = torch.nn.functional.pad(x, (0, pad))
    T = H.frame_count(n + pad, nfft, hop)
    w = H.window_f64("hann", nfft, x.device)
    X = H.stft_device(x, w, nfft, hop, T, 1)                              # power frames, training layout
    if labels == "vad_labels":
        Y = vad_labels(x, nfft, hop, T, vad_threshold)[:, None]
    elif labels == "ibm_labels":
        # the mask is elementwise against the global peak: computed on the frame-major complex frames it IS the [T, 513] label rows
        Y = ibm_labels(H.stft_device(x, w, nfft, hop, T, 2), eps, ibm_threshold)
    else:
        raise ValueError(labels)
    return X, Y
