"""Drop-in replacement for the label makers of the reference's packages/processing/target.py that the scripts use
(create_train_set.py, create_video_train_files.py, reconstruct_*.py, run_metrics.py): `clean_speech_VAD`,
`clean_speech_IBM`, `noise_robust_clean_speech_IBM` -- same keyword signatures, numpy in, numpy float32 out.

The frame bookkeeping (window / hop sizes, the end-pad rule in Python doubles, quirk Q6) is done on the host as the
reference does it; the per-frame energies, the global extrema and the thresholds run in the HIP kernels of
csrc/target.hip (labels agree bit for bit with the label files the reference wrote for data/subset).  librosa is
not imported and there is no CPU path: without a GPU these functions raise.  The reference's never-called
`noise_aware_IBM` / `threshold_IBM` experiments are not carried over.
"""
import math

import numpy as np
import torch

from packages import _native


def _frames_for(n_samples, fs, wlen_sec, hop_percent, center, pad_at_end):
    nfft = int(wlen_sec * fs)
    hopsamp = int(hop_percent * nfft)
    padded = n_samples
    if pad_at_end:
        utt_len = n_samples / fs
        if math.ceil(utt_len / wlen_sec / hop_percent) != int(utt_len / wlen_sec / hop_percent):
            padded += hopsamp
    if center:
        padded += 2 * int(nfft // 2)
    if padded < nfft:
        raise ValueError("Input signal length=%d is too small for frame_length=%d" % (padded, nfft))
    return nfft, hopsamp, padded, 1 + (padded - nfft) // hopsamp


def clean_speech_VAD(speech_t,
                     fs=16e3,
                     wlen_sec=50e-3,
                     hop_percent=0.25,
                     center=True,
                     pad_mode='reflect',
                     pad_at_end=True,
                     vad_threshold=1.70):
    """Time-domain VAD: frame energy > 10**vad_threshold * (energy of the quietest frame).  Returns (1, T) float32."""
    T = _native.target_dev()
    y = np.asarray(speech_t)
    nfft, hopsamp, padded, frames = _frames_for(len(y), fs, wlen_sec, hop_percent, center, pad_at_end)
    if center:                                  # the end-pad zeros sit inside the reflect padding: materialise both
        if padded - 2 * int(nfft // 2) > len(y):
            y = np.pad(y, (0, hopsamp), mode='constant')
        y = np.pad(y, int(nfft // 2), mode=pad_mode)
    vad = T.vad_labels(torch.from_numpy(np.ascontiguousarray(y)), nfft, hopsamp, frames, vad_threshold)
    return vad.cpu().numpy()[None]


def clean_speech_IBM(speech_tf,
                     eps=1e-8,
                     ibm_threshold=50):
    """Ideal binary mask: bins within `ibm_threshold` dB of the loudest bin of the utterance.  float32, same shape."""
    T = _native.target_dev()
    S = np.asarray(speech_tf)
    if S.dtype != np.complex64:
        raise TypeError("clean_speech_IBM: the HIP path reproduces the float32 arithmetic of complex64 input (got %s)" % S.dtype)
    return T.ibm_labels(torch.from_numpy(np.ascontiguousarray(S)), eps, ibm_threshold).cpu().numpy()


def noise_robust_clean_speech_IBM(speech_t,
                                  speech_tf,
                                  fs=16e3,
                                  wlen_sec=50e-3,
                                  hop_percent=0.25,
                                  center=True,
                                  pad_mode='reflect',
                                  pad_at_end=True,
                                  vad_threshold=1.70,
                                  eps=1e-8,
                                  ibm_threshold=50):
    """IBM gated by the time-domain VAD (labels robust to noise before / after the speech)."""
    vad = clean_speech_VAD(speech_t, fs=fs, wlen_sec=wlen_sec, hop_percent=hop_percent, center=center,
                           pad_mode=pad_mode, pad_at_end=pad_at_end, vad_threshold=vad_threshold)
    return clean_speech_IBM(speech_tf, eps=eps, ibm_threshold=ibm_threshold) * vad
