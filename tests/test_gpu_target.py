"""Label makers on the GPU (packages/processing/target.py drop-in over csrc/target.hip) and the fused front end:
bit-exact against the label files the reference's own pipeline wrote (tests/golden/labels_fixture.npz) and against
the numpy oracle on synthetic signals."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import target_oracle as to
from oracle import stft_oracle as so
from packages.processing import target as T
from packages.processing.stft import stft

pytestmark = pytest.mark.gpu
tdev = importlib.import_module("disentangled-vae_amd.target")
FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "labels_fixture.npz"))
UTTS = ["08F_sa2", "01M_sa1", "08F_si519"]
KW = dict(fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=False, pad_mode="reflect", pad_at_end=True)


def unpack(key):
    shape = tuple(FIX[key + "_shape"])
    return np.unpackbits(FIX[key + "_bits"])[:int(np.prod(shape))].reshape(shape).astype(np.float32)


@pytest.mark.parametrize("utt", UTTS)
def test_reference_label_files_reproduced_bit_exact(utt):
    speech = FIX[utt + "_wav_i16"].astype(np.float64) / 32768.0
    speech = speech / np.max(np.abs(speech))
    vad = T.clean_speech_VAD(speech, vad_threshold=1.70, **KW)
    assert vad.dtype == np.float32 and np.array_equal(vad, unpack(utt + "_vad"))
    S = stft(speech, win="hann", dtype="complex64", **KW)                 # the drop-in HIP STFT
    ibm = T.clean_speech_IBM(S, eps=1e-8, ibm_threshold=50)
    ref = unpack(utt + "_ibm")
    assert ibm.shape == ref.shape and ibm.dtype == np.float32
    # the mask is a threshold on float32 dB values of OUR STFT (1-ulp differences from the reference's are possible):
    # demand bit-exact labels except bins whose level sits within 1e-4 dB of the threshold
    mism = ibm != ref
    if mism.any():
        db = 20 * np.log10(np.abs(S) + np.float32(1e-8))
        assert np.all(np.abs(db[mism] - (db.max() - 50)) < 1e-4), int(mism.sum())
    assert mism.mean() < 1e-5
    nr = T.noise_robust_clean_speech_IBM(speech, S, vad_threshold=1.70, eps=1e-8, ibm_threshold=50, **KW)
    assert np.array_equal(nr, ibm * vad)


@pytest.mark.parametrize("utt", UTTS[:2])
def test_ibm_kernel_is_bit_exact_on_the_oracle_spectrogram(utt):
    """Same complex64 input as the reference had (oracle STFT reproduces its X to the last bit on 99.9 % of the bins)."""
    _, S, _ = to.reference_front_end(FIX[utt + "_wav_i16"])
    ibm = T.clean_speech_IBM(S, eps=1e-8, ibm_threshold=50)
    assert np.array_equal(ibm, to.clean_speech_IBM(S))
    assert np.array_equal(ibm, unpack(utt + "_ibm"))


@pytest.mark.parametrize("n,center,dtype", [(16000, False, np.float64), (40000, True, np.float64), (16384, False, np.float32), (1030, False, np.float64)])
def test_vad_matches_oracle_on_synthetic_signals(n, center, dtype):
    rng = np.random.default_rng(n)
    env = np.repeat(rng.random(n // 500 + 1) ** 3, 500)[:n]
    x = (rng.standard_normal(n) * env).astype(dtype)
    kw = dict(KW, center=center)
    got = T.clean_speech_VAD(x, vad_threshold=1.2, **kw)
    want = to.clean_speech_VAD(x.astype(np.float64) if dtype == np.float32 else x, vad_threshold=1.2, **kw)
    assert got.shape == want.shape
    if dtype == np.float64:
        assert np.array_equal(got, want)
    else:       # float32 input: numpy squares in float32, the kernel in double: only near-ties may differ
        assert (got != want).mean() < 0.01
    if n > 4000:
        assert 0 < got.mean() < 1


def test_ibm_matches_oracle_on_random_spectrogram_and_gate():
    rng = np.random.default_rng(5)
    S = ((rng.standard_normal((513, 200)) + 1j * rng.standard_normal((513, 200))) * np.exp(3 * rng.standard_normal((513, 200)))).astype(np.complex64)
    got = T.clean_speech_IBM(S, eps=1e-8, ibm_threshold=40)
    want = to.clean_speech_IBM(S, eps=1e-8, ibm_threshold=40)
    assert np.array_equal(got, want) and 0 < got.mean() < 1
    gate = (rng.random(200) > 0.5).astype(np.float32)
    g = tdev.ibm_labels(torch.from_numpy(S), 1e-8, 40, torch.from_numpy(gate)).cpu().numpy()
    assert np.array_equal(g, want * gate[None])
    with pytest.raises(TypeError):
        T.clean_speech_IBM(S.astype(np.complex128))


@pytest.mark.parametrize("labels", ["vad_labels", "ibm_labels"])
def test_fused_front_end_writes_training_layout(labels):
    utt = "08F_sa2"
    raw = FIX[utt + "_wav_i16"].astype(np.float64) / 32768.0
    X, Y = tdev.utterance_to_frames(raw, labels)
    _, S, P = to.reference_front_end(FIX[utt + "_wav_i16"])
    assert X.shape == (P.shape[1], 513) and X.is_contiguous()
    np.testing.assert_allclose(X.cpu().numpy(), P.T, rtol=2e-5, atol=1e-9)
    if labels == "vad_labels":
        assert np.array_equal(Y.cpu().numpy(), unpack(utt + "_vad").T)
    else:
        ref = unpack(utt + "_ibm").T
        assert Y.shape == ref.shape and (Y.cpu().numpy() != ref).mean() < 1e-5


def test_train_set_builder_example(tmp_path):
    """examples/build_train_set.py: reference dataset names / orientation, labels and statistics as the oracle computes them."""
    import subprocess, sys
    from scipy.io import wavfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lst, paths = tmp_path / "list.txt", []
    for utt in UTTS[:2]:
        p = str(tmp_path / f"{utt}.wav"); wavfile.write(p, 16000, FIX[utt + "_wav_i16"]); paths.append(p)
    lst.write_text("\n".join(paths))
    out = str(tmp_path / "set.npz")
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "build_train_set.py"), "--wav-list", str(lst), "--labels", "ibm_labels",
                        "--split", "train", "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = np.load(out)
    P = [to.reference_front_end(FIX[u + "_wav_i16"])[2] for u in UTTS[:2]]
    Xref = np.concatenate(P, axis=1).astype(np.float32)
    assert d["X_train"].shape == Xref.shape and d["Y_train"].shape == Xref.shape
    np.testing.assert_allclose(d["X_train"], Xref, rtol=2e-5, atol=1e-9)
    Yref = np.concatenate([unpack(u + "_ibm") for u in UTTS[:2]], axis=1)
    assert (d["Y_train"] != Yref).mean() < 1e-5
    n = Xref.shape[1]
    mean = Xref.astype(np.float64).sum(1) / n
    std = np.sqrt((np.sum(Xref.astype(np.float64) ** 2, 1) - n * mean ** 2) / (n - 1))
    np.testing.assert_allclose(d["X_train_mean"][:, 0], mean, rtol=1e-4)
    np.testing.assert_allclose(d["X_train_std"][:, 0], std, rtol=1e-4)
