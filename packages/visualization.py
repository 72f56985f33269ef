"""Figures the reconstruct / evaluate scripts save (reference packages/visualization.py:201-326 draws them with
librosa.display, which is not installable here).  `display_multiple_signals` keeps the reference's signature and returns a
matplotlib Figure with, per signal, a waveform panel, a dB spectrogram panel and a mask panel -- drawn with matplotlib alone.
Plotting is outside the hot path: this exists so that `scripts/reconstruct_*.py` import and run unchanged."""
import numpy as np


def _db(tf, floor=1e-10):
    return 20.0 * np.log10(np.maximum(np.abs(np.asarray(tf)), floor))


def display_multiple_signals(signal_list, fs=16e3, vmin=-60, vmax=10, wlen_sec=50e-3, hop_percent=0.5, xticks_sec=1.0, fontsize=50):
    """signal_list: [[waveform or None, tf_signal or None, mask or None], ...] -> Figure (one column per signal)."""
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    n = len(signal_list)
    hop_sec = wlen_sec * hop_percent
    fig, axes = plt.subplots(3, n, figsize=(25 * n, 25), squeeze=False)
    for j, (wave, tf, mask) in enumerate(signal_list):
        ax_w, ax_s, ax_m = axes[0][j], axes[1][j], axes[2][j]
        if wave is not None:
            wave = np.asarray(wave)
            ax_w.plot(np.arange(wave.size) / fs, wave, linewidth=0.5)
            ax_w.set_xlim(0, wave.size / fs)
        else:
            ax_w.axis("off")
        if tf is not None:
            S = _db(tf)
            ax_s.imshow(S, origin="lower", aspect="auto", vmin=vmin, vmax=vmax, cmap="magma",
                        extent=(0, S.shape[1] * hop_sec, 0, fs / 2e3))
            ax_s.set_ylabel("kHz", fontsize=fontsize)
        else:
            ax_s.axis("off")
        if mask is not None:
            M = np.asarray(mask)
            M = M.reshape(1, -1) if M.ndim == 1 else M
            ax_m.imshow(M, origin="lower", aspect="auto", vmin=0, vmax=1, cmap="gray", extent=(0, M.shape[1] * hop_sec, 0, max(M.shape[0], 1)))
        else:
            ax_m.axis("off")
        for ax in (ax_w, ax_s, ax_m):
            ax.tick_params(labelsize=fontsize * 0.6)
            if ax.axison and xticks_sec:
                lo, hi = ax.get_xlim()
                ax.set_xticks(np.arange(0, hi, xticks_sec))
        ax_m.set_xlabel("s", fontsize=fontsize)
    return fig
