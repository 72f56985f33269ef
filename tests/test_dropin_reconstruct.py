"""Build-container only: the reference's OWN scripts/reconstruct_M2.py runs unchanged against this repo's `packages`
on the reference's data/subset test utterances: wav -> stft -> |.|^2 -> DeepGenerativeModel(x, y) -> figures
(reference scripts/reconstruct_M2.py:41-49, 105-113, 137-320).

What is real: the script, this repo's packages.processing.stft / packages.models.models / packages.dataset.ntcd_timit /
packages.visualization / packages.processing.target, the wav and directory layout of data/subset.
What is stood in (absent third-party modules and files, nothing of the hot path):
  * soundfile -> scipy.io.wavfile (float64 in [-1, 1), like sf.read);  librosa -> empty module (imported, never called);
  * h5py -> per-path arrays: lip video "X" (67, 67, T) zeros, labels "Y" = the VAD oracle on the clean wav;
  * torch.load -> a seeded state_dict (checkpoints are git-ignored in the reference);
  * this container has no GPU and the STFT has no CPU mode, so the DEVICE calls of disentangled-vae_amd/stft.py are replaced by
    the float64 oracle transform (the HIP kernels themselves are tested against that oracle in tests/test_gpu_stft.py); the
    wrapper's own logic (window length, end-pad rule, frame count, dtypes, numpy in / numpy out) runs as shipped.
"""
import importlib
import os
import runpy
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(REF + "/scripts"), reason="reference scripts not present (GPU box)")


def _wav_f64(path):
    from scipy.io import wavfile
    fs, a = wavfile.read(path)
    assert a.dtype == np.int16
    return a.astype(np.float64) / 32768.0, fs


def test_reconstruct_M2_runs_unchanged(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    from oracle import stft_oracle as so
    from oracle import target_oracle as to

    # ---- working directory: this repo's packages, the reference's data (read only), a writable output tree
    os.symlink(os.path.join(ROOT, "packages"), tmp_path / "packages")
    os.symlink(os.path.join(ROOT, "disentangled-vae_amd"), tmp_path / "disentangled-vae_amd")
    sub = tmp_path / "data" / "subset"
    sub.mkdir(parents=True)
    os.symlink(REF + "/data/subset/raw", sub / "raw")
    os.symlink(REF + "/data/subset/processed", sub / "processed")
    (sub / "models").mkdir()
    monkeypatch.chdir(tmp_path)
    for k in [k for k in sys.modules if k == "packages" or k.startswith("packages.")]:
        monkeypatch.delitem(sys.modules, k)
    monkeypatch.setattr(sys, "path", [str(tmp_path)] + [p for p in sys.path if os.path.abspath(p or ".") != ROOT])

    # ---- stand-ins for absent third-party modules
    reads = []
    sf = types.ModuleType("soundfile")

    def sf_read(path):
        a, fs = _wav_f64(path)
        reads.append((path, len(a)))
        return a, fs
    sf.read = sf_read

    class H5:
        def __init__(self, path, mode="r", **kw):
            self.path = path

        def __getitem__(self, k):
            if k == "X":                                    # upsampled lip video: longer than any utterance, so nothing is cropped
                return np.zeros((67, 67, 4000), dtype=np.float32)
            assert k == "Y" and self.path.endswith("_vad_labels.h5"), (k, self.path)
            wav = self.path.replace("data/subset/processed/ntcd_timit/Clean/", "data/subset/processed/ntcd_timit/qutnoise_databases/ntcd_timit/Clean/")
            s_t, _ = _wav_f64(wav.replace("_vad_labels.h5", "_s.wav"))
            return to.clean_speech_VAD(s_t, fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=False, pad_mode="reflect", pad_at_end=True, vad_threshold=1.70)

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False
    h5 = types.ModuleType("h5py")
    h5.File = H5
    for name, mod in (("soundfile", sf), ("h5py", h5), ("librosa", types.ModuleType("librosa"))):
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)

    # ---- checkpoint: none ships with the reference
    def fake_load(path, map_location=None, **kw):
        assert path == os.path.join("models", "ntcd_M2_VAD_nonorm_hdim_128_128_zdim_016_end_epoch_500/M2_epoch_118_vloss_407.90.pt")
        from packages.models.models import DeepGenerativeModel
        torch.manual_seed(0)
        return DeepGenerativeModel([513, 1, 16, [128, 128]], None).state_dict()
    monkeypatch.setattr(torch, "load", fake_load)

    # ---- device transform -> float64 oracle (no GPU in this container; see the module docstring)
    H = importlib.import_module("disentangled-vae_amd.stft")
    monkeypatch.setattr(H, "_device", lambda: torch.device("cpu"))
    calls = {"stft": 0}

    def stft_cpu(x_dev, window, nfft, hop, T, layout=0):
        calls["stft"] += 1
        x = x_dev.numpy().astype(np.float64)
        assert x.ndim == 1 and layout in (0, 2) and window.dtype == torch.float64 and window.numel() == nfft
        idx = np.arange(nfft)[:, None] + hop * np.arange(T)[None, :]
        S = np.fft.rfft(window.numpy()[:, None] * x[idx], axis=0)
        assert 1 + (x.size - nfft) // hop == T
        S = S.astype(np.complex64)
        return torch.from_numpy(np.ascontiguousarray(S.T if layout == 2 else S))   # layout 2: frame-major [T, F]
    monkeypatch.setattr(H, "stft_device", stft_cpu)

    def istft_cpu(S_dev, window, nfft, hop, n_frames, start, out_len):
        y = so.istft(S_dev.numpy()[:, :n_frames], fs=16000, wlen_sec=nfft / 16000, hop_percent=hop / nfft, center=False, dtype="float32", max_len=out_len)
        return torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32))
    monkeypatch.setattr(H, "istft_device", istft_cpu)

    # ---- observe the model calls and the figures
    import packages.models.models as M
    seen = []
    real_forward = M.DeepGenerativeModel.forward

    def forward(self, x, y):
        out = real_forward(self, x, y)
        seen.append((tuple(x.shape), tuple(y.shape), x.dtype, tuple(out[0].shape), out[0].dtype, bool(torch.isfinite(out[0]).all())))
        return out
    monkeypatch.setattr(M.DeepGenerativeModel, "forward", forward)
    import packages.visualization as V
    figs = []

    class Fig:
        def savefig(self, path):
            figs.append(path)

    def display(signal_list, **kw):
        for wave, tf, mask in signal_list:
            if tf is not None:
                assert tf.shape[0] == 513
            if mask is not None and tf is not None:
                assert mask.shape[-1] == tf.shape[-1]
        return Fig()
    monkeypatch.setattr(V, "display_multiple_signals", display)

    runpy.run_path(os.path.join(REF, "scripts", "reconstruct_M2.py"), run_name="__main__")

    mods = sys.modules["packages.models.models"]
    assert os.path.realpath(mods.__file__).startswith(os.path.realpath(ROOT)), "script imported the reference's packages, not ours"
    # 3 test utterances (34M: sa1, sa2, si494) x 4 reconstructions each (clean, noisy, ones, zeros)
    assert len(seen) == 12 and len(figs) == 12 and calls["stft"] == 6
    frames = {n: 1 + (n + (256 if np.ceil(n / 16000 / 64e-3 / 0.25) != int(n / 16000 / 64e-3 / 0.25) else 0) - 1024) // 256 for _, n in reads}
    assert sorted(set(s[0][0] for s in seen)) == sorted(set(frames.values()))        # T of every forward = the end-pad rule's frame count
    for xs, ys, xd, rs, rd, ok in seen:
        assert xs[1] == 513 and ys == (xs[0], 1) and xd == torch.float32 and rs == xs and rd == torch.float32 and ok
    assert all(p.startswith("data/subset/models/ntcd_M2_VAD") and p.endswith(".png") for p in figs)
    assert not any(os.path.realpath(p).startswith(REF) for p in figs)                 # nothing is written into the reference tree

    # the same spectrogram goes back to the time domain through the drop-in istft (the evaluate scripts' last step,
    # scripts/evaluate_ntcd_M2.py:211-225): length = the original sample count
    from packages.processing.stft import stft, istft
    wav = REF + "/data/subset/processed/ntcd_timit/qutnoise_databases/ntcd_timit/Clean/test/34M/sa1_x.wav"
    x_t, _ = _wav_f64(wav)
    kw = dict(fs=16000, wlen_sec=64e-3, win="hann", hop_percent=0.25, center=False)
    X = stft(x_t, pad_mode="reflect", pad_at_end=True, dtype="complex64", **kw)
    assert X.dtype == np.complex64 and X.shape == (513, frames[len(x_t)])
    x_hat = istft(X, dtype="float32", max_len=len(x_t), **kw)
    assert x_hat.dtype == np.float32 and x_hat.shape == (len(x_t),)
    inner = slice(1024, len(x_t) - 1024)
    assert np.abs(x_hat[inner] - x_t[inner]).max() < 1e-5


def test_listing_helpers_equal_the_reference_on_data_subset():
    """packages/dataset/ntcd_timit.py (own pathlib implementation) returns exactly what the reference's functions return on
    the reference's data/subset tree, for every split / size / label name the scripts use."""
    import importlib.util

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    mine = load(os.path.join(ROOT, "packages", "dataset", "ntcd_timit.py"), "_mine_ntcd")
    ref = load(REF + "/packages/dataset/ntcd_timit.py", "_ref_ntcd")
    raw, proc = REF + "/data/subset/raw/", REF + "/data/subset/processed/"
    for split in ("train", "validation", "test", "other"):
        assert mine.speech_list(raw, split) == ref.speech_list(raw, split)
        assert mine.video_list(raw, split) == ref.video_list(raw, split)
        for size in ("subset", "complete"):
            for up in (False, True):
                for lab in ("vad_labels", "ibm_labels"):
                    a, b = (m.proc_noisy_clean_pair_dict(proc, split, size, lab, up) for m in (mine, ref))
                    assert a == b and list(a) == list(b)
    assert len(mine.speech_list(raw, "test")[0]) == 3


def test_display_multiple_signals_draws_a_figure(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("_own_visualization", os.path.join(ROOT, "packages", "visualization.py"))
    V = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(V)
    display_multiple_signals = V.display_multiple_signals
    rng = np.random.default_rng(0)
    wave = rng.standard_normal(16000)
    tf = rng.standard_normal((513, 60)) + 1j * rng.standard_normal((513, 60))
    fig = display_multiple_signals([[wave, tf, None], [None, np.abs(tf), (rng.random((1, 60)) > 0.5).astype(np.float32)]],
                                   fs=16000, vmin=-40, vmax=20, wlen_sec=64e-3, hop_percent=0.25, xticks_sec=2.0, fontsize=30)
    out = tmp_path / "f.png"
    fig.savefig(str(out))
    assert out.stat().st_size > 10000
    import matplotlib.pyplot as plt
    plt.close(fig)
