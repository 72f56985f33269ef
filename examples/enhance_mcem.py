"""Speech enhancement of a batch of utterances on the MI355X-native path: the per-utterance flow of the reference's
scripts/evaluate_ntcd_M2.py:139-230 (STFT -> MCEM with a VAE speech prior and an NMF noise model -> Wiener
filtering -> ISTFT), for many utterances at once (disentangled-vae_amd/mcem.py: McemBatch).

    python examples/enhance_mcem.py --wav a.wav b.wav --checkpoint models/M2_epoch_118_vloss_407.90.pt --out enhanced/
    python examples/enhance_mcem.py --synthetic 8                       # no data at hand: modulated-noise "speech" + noise

The labels y fed to the M2 decoder are the time-domain VAD of the mixture (packages/processing/target.py); the
reference's evaluate script takes them from a video classifier or from the clean signal (oracle), neither of which
ships with it.  Writes <name>_s_est.wav and <name>_n_est.wav like the reference (evaluate_ntcd_M2.py:232-245).
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
from packages.models.models import DeepGenerativeModel
from packages.processing.stft import stft, istft
from packages.processing.target import clean_speech_VAD

McemBatch = importlib.import_module("disentangled-vae_amd.mcem").McemBatch
STFT = dict(fs=16000, wlen_sec=64e-3, win="hann", hop_percent=0.25, center=False)      # evaluate_ntcd_M2.py:37-45


def synthetic_mixture(seconds, seed):
    rng = np.random.default_rng(seed)
    n = int(16000 * seconds)
    env = np.repeat((rng.random(n // 800 + 1) > 0.5).astype(np.float64), 800)[:n]      # 50 ms on/off "speech"
    s = env * rng.standard_normal(n) * np.sin(2 * np.pi * 220 * np.arange(n) / 16000 + rng.random())
    return s + 0.2 * rng.standard_normal(n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wav", nargs="*", default=[])
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic 3-4 s mixtures instead of --wav")
    ap.add_argument("--checkpoint", default=None, help="M2 (y_dim 1) state_dict; random weights when absent")
    ap.add_argument("--niter", type=int, default=100)
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32")
    ap.add_argument("--out", default="enhanced")
    a = ap.parse_args()
    names, waves = [], []
    for p in a.wav:
        fs, w = wavfile.read(p)
        assert fs == 16000, f"{p}: 16 kHz expected"
        waves.append(w.astype(np.float64) / (32768.0 if w.dtype == np.int16 else 1.0)); names.append(os.path.splitext(os.path.basename(p))[0])
    for i in range(a.synthetic if not a.wav else 0):
        waves.append(synthetic_mixture(3.0 + 0.25 * (i % 5), i)); names.append(f"synthetic_{i:02d}")
    if not waves:
        ap.error("give --wav files or --synthetic N")
    torch.manual_seed(0)
    vae = DeepGenerativeModel([513, 1, 16, [128, 128]], None)
    if a.checkpoint:
        vae.load_state_dict(torch.load(a.checkpoint, map_location="cpu", weights_only=True))
    vae = vae.cuda().eval()
    for p in vae.parameters():
        p.requires_grad = False

    t0 = time.perf_counter()
    X = [stft(w, pad_mode="reflect", pad_at_end=True, dtype="complex64", **STFT) for w in waves]          # (513, N_u) complex64
    Y = [clean_speech_VAD(w, fs=16000, wlen_sec=64e-3, hop_percent=0.25, center=False, pad_mode="reflect", pad_at_end=True,
                          vad_threshold=1.70) for w in waves]                                              # (1, N_u)
    mb = McemBatch(vae, niter=a.niter, nsamples_E_step=10, burnin_E_step=30, nsamples_WF=25, burnin_WF=75, var_RW=0.01,
                   nmf_rank=10, precision=a.precision)                                                      # evaluate_ntcd_M2.py:92-99
    mb.init_parameters(X, Y)
    cost = mb.run()
    os.makedirs(a.out, exist_ok=True)
    for name, w, S_hat, N_hat in zip(names, waves, mb.S_hat, mb.N_hat):
        s_hat = istft(S_hat, max_len=len(w), **STFT)
        n_hat = istft(N_hat, max_len=len(w), **STFT)
        wavfile.write(os.path.join(a.out, name + "_s_est.wav"), 16000, s_hat.astype(np.float32))
        wavfile.write(os.path.join(a.out, name + "_n_est.wav"), 16000, n_hat.astype(np.float32))
    dt = time.perf_counter() - t0
    frames = sum(x.shape[1] for x in X)
    print(f"{len(waves)} utterances, {frames} frames, {a.niter} EM iterations: {dt:.2f} s wall ({len(waves) / dt:.1f} utterances/s); "
          f"cost {cost[0].mean():.3f} -> {cost[-1].mean():.3f}")


if __name__ == "__main__":
    main()
