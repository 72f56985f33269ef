"""Training-set builder on the MI355X-native path: what scripts/create_train_set.py:129-219 does per split (read each
clean utterance, peak-normalise, STFT, power spectrogram, VAD or IBM labels, append; channel mean / std of the
training split), with every transform on the GPU (disentangled-vae_amd/target.py: utterance_to_frames) and the frames
kept in HBM for the trainer.

    python examples/build_train_set.py --wav-list train.txt --labels ibm_labels --out train_set.npz
    python examples/build_train_set.py --synthetic 64 --labels vad_labels --out /tmp/set.npz

Output datasets use the reference's names and orientation -- X_<split> (513, N) float32, Y_<split> (y_dim, N) float32,
X_<split>_mean / _std (513, 1) -- in an .npz (and in the reference's HDF5 layout when h5py is importable and --out
ends in .h5).  The video-length crop of the reference (create_train_set.py:180-186) needs its video files and is left out.
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np
import torch
from scipy.io import wavfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tdev = importlib.import_module("disentangled-vae_amd.target")
DeviceFrames = importlib.import_module("disentangled-vae_amd.frames").DeviceFrames


def channel_stats(X):
    """create_train_set.py:205-207: mean and the empirical std (n - 1) per frequency bin, from sums in float64."""
    n = X.shape[0]
    s1 = X.double().sum(0)
    s2 = (X.double() ** 2).sum(0)
    mean = s1 / n
    std = torch.sqrt((s2 - n * mean ** 2) / (n - 1))
    return mean.float()[:, None], std.float()[:, None]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wav-list", default=None, help="text file with one 16 kHz wav path per line")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--labels", choices=["vad_labels", "ibm_labels"], default="vad_labels")
    ap.add_argument("--split", default="train")
    ap.add_argument("--out", default="train_set.npz")
    a = ap.parse_args()
    waves = []
    if a.wav_list:
        for p in open(a.wav_list).read().split():
            fs, w = wavfile.read(p)
            assert fs == 16000, f"{p}: 16 kHz expected"
            waves.append(w.astype(np.float64) / (32768.0 if w.dtype == np.int16 else 1.0))
    for i in range(a.synthetic if not waves else 0):
        rng = np.random.default_rng(i)
        n = 16000 * 3 + 977 * (i % 7)
        env = np.repeat((rng.random(n // 1600 + 1) > 0.4).astype(np.float64), 1600)[:n]
        waves.append(env * rng.standard_normal(n) * 0.3 + 0.003 * rng.standard_normal(n))
    if not waves:
        ap.error("give --wav-list or --synthetic N")
    t0 = time.perf_counter()
    Xs, Ys = [], []
    for w in waves:
        X, Y = tdev.utterance_to_frames(w, a.labels)           # [T, 513], [T, y_dim] on the GPU
        Xs.append(X); Ys.append(Y)
    X = torch.cat(Xs); Y = torch.cat(Ys)
    mean, std = channel_stats(X)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{len(waves)} utterances -> {X.shape[0]} frames, labels {a.labels} ({float(Y.mean()):.3f} active) in {dt * 1e3:.1f} ms")
    data = {f"X_{a.split}": X.t().cpu().numpy(), f"Y_{a.split}": Y.t().cpu().numpy(),
            f"X_{a.split}_mean": mean.cpu().numpy(), f"X_{a.split}_std": std.cpu().numpy()}
    if a.out.endswith(".h5"):
        import h5py
        with h5py.File(a.out, "w") as f:
            for k, v in data.items():
                f.create_dataset(k, data=v, chunks=(v.shape[0], 1), compression="lzf")     # create_train_set.py:70-72,116
    else:
        np.savez(a.out, **data)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
