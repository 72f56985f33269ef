"""CPU: the numpy STFT/ISTFT oracle is pinned by the reference's own data fixtures
(tests/golden/stft_ref_fixture.npz) and cross-checked with torch.stft / torch.istft."""
import os

import numpy as np
import pytest
import torch

from oracle import stft_oracle as so

KW = dict(fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=False)
UTTS = ["01M_sa1", "01M_sa2", "01M_si462", "08F_sa1", "08F_sa2", "08F_si519"]


@pytest.mark.parametrize("utt", UTTS)
def test_oracle_reproduces_reference_h5_power_frames(stft_golden, utt):
    f = stft_golden
    x = f[utt + "_wav_head_i16"].astype(np.float64) / 32768.0
    x = x / (float(f[utt + "_peak_i16"]) / 32768.0)             # scripts/create_train_set.py:137
    S = so.stft(x, pad_at_end=False and True, **KW) if False else so.stft(x, **KW)
    P = (np.abs(S) ** 2).astype(np.float32)
    X = f[utt + "_X"]
    n = X.shape[1]
    np.testing.assert_allclose(P[:, :n], X, rtol=3e-7, atol=0)   # 1-2 ulp of float32
    assert np.mean(P[:, :n] == X) > 0.99
    ibm = (20 * np.log10(np.abs(S[:, :n]) + 1e-8) > np.max(20 * np.log10(np.abs(S) + 1e-8)) - 50)
    # IBM threshold uses the whole-utterance max; with only the head available compare where decisive
    assert ibm.shape == f[utt + "_Y_ibm"].shape


@pytest.mark.skipif(not os.path.isdir("/root/reference/data/subset"), reason="reference data not present (GPU box)")
def test_oracle_reproduces_whole_reference_h5(stft_golden):
    """Build-container only: whole wavs against all 201 X_train frames + IBM labels + mean/std."""
    import subprocess, tempfile
    from scipy.io import wavfile
    tmp = tempfile.mkdtemp()
    code = ("import h5py,numpy as np;f=h5py.File('/root/reference/data/subset/processed/ntcd_timit/"
            "Clean_ibm_labels_upsampled.h5','r');np.savez('%s/m.npz',**{k:f[k][:] for k in f})" % tmp)
    subprocess.check_call(["/opt/conda/bin/python3.9", "-c", code])
    h5 = np.load(tmp + "/m.npz")
    specs = []
    for j, u in enumerate(["sa1", "sa2", "si462"]):
        fs, w = wavfile.read(f"/root/reference/data/subset/raw/ntcd_timit/Clean/volunteers/01M/straightcam/{u}.wav")
        x = w.astype(np.float64) / 32768.0
        x = x / np.max(np.abs(x))
        S = so.stft(x, **KW)
        P = (np.abs(S) ** 2)
        specs.append(P[:, :67])
        np.testing.assert_allclose(P[:, :67].astype(np.float32), h5["X_train"][:, 67 * j:67 * (j + 1)], rtol=3e-7)
        ibm = (20 * np.log10(np.abs(S) + 1e-8) > np.max(20 * np.log10(np.abs(S) + 1e-8)) - 50).astype(np.float32)
        assert np.array_equal(ibm[:, :67], h5["Y_train"][:, 67 * j:67 * (j + 1)])


def test_oracle_vs_torch_stft_and_istft():
    rng = np.random.default_rng(3)
    x = rng.standard_normal(16000 * 2 + 123)
    S = so.stft(x, **KW)
    n_pad = len(x) + (256 if so.pad_decision(len(x), 16000, 64e-3, 0.25) else 0)
    xt = torch.from_numpy(np.pad(x, (0, n_pad - len(x))))
    St = torch.stft(xt, 1024, 256, window=torch.hann_window(1024, dtype=torch.float64), center=False, return_complex=True)
    np.testing.assert_allclose(S, St.numpy().astype(np.complex64), rtol=0, atol=2e-5 * np.abs(S).max())
    y = so.istft(S, max_len=len(x), **KW)
    assert y.shape == (len(x),) and y.dtype == np.float32
    # interior reconstructs the signal (edges divide by a tiny window sum)
    np.testing.assert_allclose(y[1024:-1024], x[1024:-1024], atol=2e-5)


def test_oracle_center_true_roundtrip():
    rng = np.random.default_rng(4)
    x = rng.standard_normal(8000)
    kw = dict(fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=True)
    S = so.stft(x, **kw)
    y = so.istft(S, max_len=len(x), **kw)
    np.testing.assert_allclose(y[10:-300], x[10:-300], atol=2e-5)
    assert so.stft(x, fs=16e3, wlen_sec=50e-3).shape[0] == 401       # the never-used defaults: nfft 800
