"""Resolve the native package (its directory name, `disentangled-vae_amd`, is not a
Python identifier) for the drop-in modules under packages/."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_pkg = None


def pkg():
    global _pkg
    if _pkg is None:
        _pkg = importlib.import_module("disentangled-vae_amd")
    return _pkg


def ops():
    return importlib.import_module("disentangled-vae_amd.ops")


def native():
    return importlib.import_module("disentangled-vae_amd.native")


def module_path():
    return importlib.import_module("disentangled-vae_amd.module_path")


def stft_host():
    return importlib.import_module("disentangled-vae_amd.stft")


def mcem_dev():
    return importlib.import_module("disentangled-vae_amd.mcem")


def target_dev():
    return importlib.import_module("disentangled-vae_amd.target")
