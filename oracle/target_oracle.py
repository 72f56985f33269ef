"""numpy restatement of the reference's label makers (TEST ORACLE, not product code):
packages/processing/target.py:5-56 (clean_speech_VAD), :58-70 (clean_speech_IBM), :72-104
(noise_robust_clean_speech_IBM).

The reference frames the signal with librosa.util.frame (third-party, not importable here; published
behaviour: frames y[t*hop : t*hop + frame_length], T = 1 + (len - frame_length) // hop, shape
(frame_length, T)); restated with a strided view.  Pinned by tests/golden/labels_fixture.npz: the
per-utterance *_vad_labels.h5 / *_ibm_labels.h5 files the reference's own pipeline wrote for the wavs of
data/subset (tests/golden/make_labels_golden.py) -- bit-exact label agreement is the bar.
"""
import math

import numpy as np

from . import stft_oracle as so


def frame(y, frame_length, hop_length):
    n = 1 + (len(y) - frame_length) // hop_length
    s = y.strides[0]
    return np.lib.stride_tricks.as_strided(y, shape=(frame_length, n), strides=(s, hop_length * s), writeable=False)


def clean_speech_VAD(speech_t, fs=16e3, wlen_sec=50e-3, hop_percent=0.25, center=True, pad_mode="reflect",
                     pad_at_end=True, vad_threshold=1.70):
    """target.py:5-56: frame energy > 10^vad_threshold * min frame energy -> (1, T) float32."""
    nfft = int(wlen_sec * fs)
    hopsamp = int(hop_percent * nfft)
    if pad_at_end:
        utt_len = len(speech_t) / fs
        if math.ceil(utt_len / wlen_sec / hop_percent) != int(utt_len / wlen_sec / hop_percent):
            y = np.pad(speech_t, (0, hopsamp), mode="constant")
        else:
            y = speech_t.copy()
    else:
        y = speech_t.copy()
    if center:
        y = np.pad(y, int(nfft // 2), mode=pad_mode)
    power = np.power(frame(np.ascontiguousarray(y), nfft, hopsamp), 2).sum(axis=0)
    vad = power > np.power(10, vad_threshold) * np.min(power)
    return np.float32(vad)[None]


def clean_speech_IBM(speech_tf, eps=1e-8, ibm_threshold=50):
    """target.py:58-70: 20 log10(|S| + eps) > max - ibm_threshold -> float32 mask of S's shape."""
    mag = abs(speech_tf)
    power_db = 20 * np.log10(mag + eps)
    return np.float32(power_db > np.max(power_db) - ibm_threshold)


def noise_robust_clean_speech_IBM(speech_t, speech_tf, fs=16e3, wlen_sec=50e-3, hop_percent=0.25, center=True,
                                  pad_mode="reflect", pad_at_end=True, vad_threshold=1.70, eps=1e-8, ibm_threshold=50):
    """target.py:72-104: IBM gated by the time-domain VAD."""
    vad = clean_speech_VAD(speech_t, fs, wlen_sec, hop_percent, center, pad_mode, pad_at_end, vad_threshold)
    return clean_speech_IBM(speech_tf, eps, ibm_threshold) * vad


def reference_front_end(wav_i16, fs=16000, wlen_sec=64e-3, hop_percent=0.25):
    """scripts/create_train_set.py:133-152: peak-normalise, STFT (complex64), power spectrogram."""
    speech = wav_i16.astype(np.float64) / 32768.0                 # soundfile's int16 -> float64 scaling
    speech = speech / np.max(np.abs(speech))
    S = so.stft(speech, fs=fs, wlen_sec=wlen_sec, win="hann", hop_percent=hop_percent, center=False, pad_mode="reflect",
                pad_at_end=True, dtype="complex64")
    return speech, S, np.power(abs(S), 2)
