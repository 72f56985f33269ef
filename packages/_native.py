"""Resolve the native package (its directory name, `disentangled-vae_amd`, is not a
Python identifier) for the drop-in modules under packages/."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_cache = {}


def _mod(name):
    """The named submodule of the native package, imported once (importlib.import_module costs ~10 us per call even when the module
    is loaded, and the drop-in modules come through here three times per training step)."""
    m = _cache.get(name)
    if m is None:
        m = _cache[name] = importlib.import_module("disentangled-vae_amd" + ("." + name if name else ""))
    return m


def pkg():
    return _mod("")


def ops():
    return _mod("ops")


def native():
    return _mod("native")


def module_path():
    return _mod("module_path")


def stft_host():
    return _mod("stft")


def mcem_dev():
    return _mod("mcem")


def target_dev():
    return _mod("target")
