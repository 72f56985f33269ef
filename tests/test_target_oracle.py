"""Label-maker oracle (oracle/target_oracle.py) against the label files the reference's own pipeline wrote
(tests/golden/labels_fixture.npz): every label bit must agree."""
import os

import numpy as np
import pytest

from oracle import target_oracle as to

FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "labels_fixture.npz"))
UTTS = ["08F_sa2", "01M_sa1", "08F_si519"]
KW = dict(fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=False, pad_mode="reflect", pad_at_end=True)


def unpack(key):
    shape = tuple(FIX[key + "_shape"])
    return np.unpackbits(FIX[key + "_bits"])[:int(np.prod(shape))].reshape(shape).astype(np.float32)


@pytest.mark.parametrize("utt", UTTS)
def test_vad_and_ibm_labels_bit_exact(utt):
    speech, S, _ = to.reference_front_end(FIX[utt + "_wav_i16"])
    vad = to.clean_speech_VAD(speech, vad_threshold=1.70, **KW)
    ref_vad = unpack(utt + "_vad")
    assert vad.shape == ref_vad.shape and vad.dtype == np.float32
    assert np.array_equal(vad, ref_vad)
    ibm = to.clean_speech_IBM(S, eps=1e-8, ibm_threshold=50)
    ref_ibm = unpack(utt + "_ibm")
    assert ibm.shape == ref_ibm.shape
    assert np.array_equal(ibm, ref_ibm)
    nr = to.noise_robust_clean_speech_IBM(speech, S, vad_threshold=1.70, eps=1e-8, ibm_threshold=50, **KW)
    assert np.array_equal(nr, ref_ibm * ref_vad)
