"""Build-container only: the reference's OWN training scripts run unchanged against this repo's
`packages`: import surface, constructor/forward/loss signatures, state_dict."""
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


import fake_h5


class _Done(Exception):
    pass


def _run(script, y_dim, tmp_path, monkeypatch):
    rng = np.random.default_rng(0)
    n = 64
    arrays = {}
    for split in ("train", "validation"):
        arrays["X_" + split] = (rng.random((513, n)) ** 2).astype(np.float32)
        arrays["Y_" + split] = (rng.random((y_dim, n)) < 0.5).astype(np.float32)
    fake_h5.install(monkeypatch, arrays)
    for name, mod in (("torchaudio", types.ModuleType("torchaudio")), ("librosa", types.ModuleType("librosa"))):
        monkeypatch.setitem(sys.modules, name, mod)
    # the script does sys.path.append('.') and imports packages.* : cwd must hold THIS repo's packages
    os.symlink(os.path.join(ROOT, "packages"), tmp_path / "packages")
    os.symlink(os.path.join(ROOT, "disentangled-vae_amd"), tmp_path / "disentangled-vae_amd")
    monkeypatch.chdir(tmp_path)
    for k in [k for k in sys.modules if k == "packages" or k.startswith("packages.")]:
        monkeypatch.delitem(sys.modules, k)
    monkeypatch.setattr(sys, "path", [str(tmp_path)] + [p for p in sys.path if os.path.abspath(p or ".") != ROOT])
    saved = {}

    def fake_save(obj, path, *a, **k):
        saved["sd"] = obj
        saved["path"] = path
        raise _Done()
    monkeypatch.setattr(torch, "save", fake_save)
    real_loader = torch.utils.data.DataLoader

    def loader(ds, *a, **k):
        k["num_workers"] = 0
        k["pin_memory"] = False
        return real_loader(ds, *a, **k)
    monkeypatch.setattr(torch.utils.data, "DataLoader", loader)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    with pytest.raises(_Done):
        runpy.run_path(os.path.join(REF, "scripts", script), run_name="__main__")
    mods = sys.modules["packages.models.models"]
    assert os.path.realpath(mods.__file__).startswith(os.path.realpath(ROOT)), "script imported the reference's packages, not ours"
    return saved


def test_training_M1_runs_unchanged(tmp_path, monkeypatch):
    saved = _run("training_M1.py", 1, tmp_path, monkeypatch)
    assert len(saved["sd"]) == 14 and "M1_epoch_001_vloss_" in saved["path"]
    assert sum(v.numel() for v in saved["sd"].values()) == 171297


def test_training_M2_runs_unchanged(tmp_path, monkeypatch):
    saved = _run("training_M2.py", 513, tmp_path, monkeypatch)          # script default: ibm_labels, y_dim 513
    assert len(saved["sd"]) == 14 and sum(v.numel() for v in saved["sd"].values()) == 302625


def test_training_M2_info_runs_unchanged(tmp_path, monkeypatch):
    saved = _run("training_M2_info_vad.py", 1, tmp_path, monkeypatch)
    assert len(saved["sd"]) == 26 and sum(v.numel() for v in saved["sd"].values()) == 272675
    log = open(tmp_path / "models" / os.listdir(tmp_path / "models")[0] / "output_batch.log").read()
    assert "Classif.: 0.000" in log                                      # alpha = 0 (quirk Q4)


def test_training_M2_info_pretrain_runs_unchanged(tmp_path, monkeypatch):
    """The pretrain variant loads a checkpoint, keeps its classifier tensors (key filter, training_M2_info_vad_pretrain.py:102-112)
    and trains on: the filtered tensors must land in the model that is saved."""
    from packages.models.models import DeepGenerativeModel_v5
    torch.manual_seed(3)
    donor = DeepGenerativeModel_v5([513, 1, 16, [128, 128]]).state_dict()
    donor = {k: v.clone() + 1.0 for k, v in donor.items()}                  # recognisable values
    monkeypatch.setattr(torch, "load", lambda *a, **k: donor)
    saved = _run("training_M2_info_vad_pretrain.py", 1, tmp_path, monkeypatch)
    assert len(saved["sd"]) == 26
    k = "enc_dec_clf.classifier.hidden.1.weight"
    # alpha = 0: the classifier receives exactly zero gradients, so Adam leaves the loaded tensors untouched (quirks Q4, Q5)
    assert torch.equal(saved["sd"][k], donor[k])
    assert not torch.equal(saved["sd"]["enc_dec_clf.encoder.hidden.0.weight"], donor["enc_dec_clf.encoder.hidden.0.weight"])
