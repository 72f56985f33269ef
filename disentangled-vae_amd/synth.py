"""Synthetic workload of the benchmark and the parity tests: heavy-tailed positive power-spectrogram
frames, Bernoulli labels, N(0,1) reparametrisation noise; and the reference's model constructors by name.

`make_batch` (numpy Generator streams: the arrays the golden fixtures were generated from, regenerated from a seed) is
shared with tests/golden_util.py; `device_batches` draws the same distribution on the GPU for bench.py's input pool."""
import numpy as np


def make_batch(dims, B, seed):
    rng = np.random.default_rng(seed)
    xd, yd, zd = dims["x_dim"], dims["y_dim"], dims["z_dim"]
    n1, n2, n3 = (rng.standard_normal((B, xd)) for _ in range(3))
    x = np.exp(4 * n1 - 8) * (n2 ** 2 + n3 ** 2) / 2
    x = np.clip(x, 1e-12, 1e4).astype(np.float32)
    if yd == 0:
        y = None
    else:
        prob = 0.6 if yd == 1 else 0.3
        y = (rng.random((B, yd)) < prob).astype(np.float32)
    eps = rng.standard_normal((B, zd)).astype(np.float32)
    return x, y, eps


def device_batches(dims, B, nb, seed, device):
    """`nb` batches (x, y, eps) of the same distribution generated on `device` (plumbing, outside any timed region)."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    xd, yd, zd = dims["x_dim"], dims["y_dim"], dims["z_dim"]
    out = []
    for _ in range(nb):
        n1 = torch.randn((B, xd), generator=g, device=device)
        n2 = torch.randn((B, xd), generator=g, device=device)
        n3 = torch.randn((B, xd), generator=g, device=device)
        x = (torch.exp(4 * n1 - 8) * (n2 * n2 + n3 * n3) / 2).clamp_(1e-12, 1e4)
        y = None
        if yd:
            y = (torch.rand((B, yd), generator=g, device=device) < (0.6 if yd == 1 else 0.3)).float()
        e = torch.randn((B, zd), generator=g, device=device)
        out.append((x, y, e))
    return out


def build_model(model, dims):
    """The drop-in module for a model name: VariationalAutoencoder (M1), DeepGenerativeModel (M2), DeepGenerativeModel_v5 (M2_info)
    with the constructor arguments of scripts/training_M1.py:93, training_M2.py:100, training_M2_info_vad.py:118."""
    from packages.models import models as M
    h = list(dims["h_dim"])
    if model == "M1":
        return M.VariationalAutoencoder([dims["x_dim"], dims["z_dim"], h])
    if model == "M2":
        return M.DeepGenerativeModel([dims["x_dim"], dims["y_dim"], dims["z_dim"], h], None)
    return M.DeepGenerativeModel_v5([dims["x_dim"], dims["y_dim"], dims["z_dim"], h])
