"""disentangled-vae_amd: MI355X-native hot path of sp-uhh/disentangled-vae.

Holds only what the hot path needs:
  csrc/       hand-written HIP kernels for gfx950 + the C ABI (include/dvae.h)
  native.py   ctypes binding of libdvae_hip.so (no fallback)
  ops.py      torch.autograd.Function shells over the layer-level kernels
  stft.py     host side of STFT / ISTFT (pad rule in double, device FFT)
  trainer.py  fused train-step harness (mirrors the scripts' loop bodies)

The directory name carries a hyphen, so import it through
``packages._native`` (importlib) or ``importlib.import_module("disentangled-vae_amd")``.
"""
from . import native  # noqa: F401
