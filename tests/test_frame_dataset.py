"""The frame dataset / frame-store ingest against an extraction of the reference's own training file
(data/subset/processed/ntcd_timit/Clean_ibm_labels_upsampled.h5: datasets X_<split>, Y_<split>, one frame per column,
reference packages/data_handling.py:42-59, scripts/create_train_set.py:91-219).  h5py is not installable here: the
file's arrays come from tests/golden/stft_ref_fixture.npz (the first 8 frames of each of its 6 utterances, extracted
by tests/golden/make_stft_golden.py) behind a stand-in h5py module."""
import importlib
import os
import pickle

import numpy as np
import pytest
import torch

import fake_h5

HERE = os.path.dirname(os.path.abspath(__file__))
LAYOUT = {"train": ("01M", ["sa1", "sa2", "si462"]), "validation": ("08F", ["sa1", "sa2", "si519"])}


def reference_file_arrays():
    fx = np.load(os.path.join(HERE, "golden", "stft_ref_fixture.npz"))
    out = {}
    for split, (spk, utts) in LAYOUT.items():
        out["X_" + split] = np.concatenate([fx[f"{spk}_{u}_X"] for u in utts], axis=1)         # (513, 24): the file's orientation
        out["Y_" + split] = np.concatenate([fx[f"{spk}_{u}_Y_ibm"] for u in utts], axis=1)
    return out


def test_dataset_contract_on_the_reference_file_layout(monkeypatch):
    arrays = reference_file_arrays()
    FF = fake_h5.install(monkeypatch, arrays)
    from packages.data_handling import HDF5CleanSpectrogramLabeledFrames, frame_file_path
    assert frame_file_path("data/x", "ntcd_timit", "ibm_labels", True) == os.path.join("data/x", "ntcd_timit", "Clean_ibm_labels_upsampled.h5")
    assert frame_file_path("d", "n", "vad_labels", False) == os.path.join("d", "n", "Clean_vad_labels.h5")
    for split in ("train", "validation"):
        before = FF.opened
        ds = HDF5CleanSpectrogramLabeledFrames("data/complete/processed", "ntcd_timit", split, "complete", labels="ibm_labels", upsampled=True)
        assert FF.opened == before                               # nothing stays open after construction (DataLoader workers open their own)
        assert len(ds) == arrays["X_" + split].shape[1] == 24
        for i in (0, 7, 23):
            x, y = ds[i]
            assert x.dtype == torch.float32 and y.dtype == torch.float32 and x.shape == (513,) and y.shape == (513,)
            assert np.array_equal(x.numpy(), arrays["X_" + split][:, i]) and np.array_equal(y.numpy(), arrays["Y_" + split][:, i])
        assert FF.opened == before + 1
        clone = pickle.loads(pickle.dumps(ds))                   # what a spawned worker receives: closed, opens on first use
        assert clone.frames._h is None
        assert torch.equal(clone[3][0], ds[3][0])
        loader = torch.utils.data.DataLoader(ds, batch_size=8, shuffle=False, num_workers=0)
        xb, yb = next(iter(loader))
        assert xb.shape == (8, 513) and np.array_equal(xb.numpy(), arrays["X_" + split][:, :8].T)
        del loader, clone
        ds.frames.close()
        assert FF.opened == before
    assert set(np.unique(arrays["Y_train"])) <= {0.0, 1.0}       # IBM labels of the reference file are binary


@pytest.mark.gpu
def test_device_frames_from_hdf5_matches_the_reference_file(monkeypatch):
    arrays = reference_file_arrays()
    fake_h5.install(monkeypatch, arrays)
    frames = importlib.import_module("disentangled-vae_amd.frames")
    for split in ("train", "validation"):
        df = frames.DeviceFrames.from_hdf5("data/subset/processed/ntcd_timit/Clean_ibm_labels_upsampled.h5", split)
        assert len(df) == 24 and df.x.shape == (24, 513) and df.y.shape == (24, 513)
        assert np.array_equal(df.x.cpu().numpy(), arrays["X_" + split].T)          # frames-major rows, bit for bit
        assert np.array_equal(df.y.cpu().numpy(), arrays["Y_" + split].T)
    from packages.data_handling import HDF5CleanSpectrogramLabeledFrames
    ds = HDF5CleanSpectrogramLabeledFrames("data/subset/processed", "ntcd_timit", "train", "subset", labels="ibm_labels", upsampled=True)
    dev = ds.to_device()
    assert torch.equal(dev.x.cpu(), torch.stack([ds[i][0] for i in range(len(ds))]))
    # the store feeds the fused trainer through the in-kernel gather
    trainer = importlib.import_module("disentangled-vae_amd.trainer")
    tr = trainer.Trainer("M2", dict(x_dim=513, y_dim=513, z_dim=16, h_dim=(128, 128)), batch=16, precision="bf16x3", seed=0)
    for idx in dev.index_batches(16, shuffle=True, drop_last=True, generator=torch.Generator(device="cuda").manual_seed(0)):
        losses = tr.step(dev.x, dev.y, rows=idx)
    assert torch.isfinite(losses).all() and tr.bad_row_count() == 0
    bad = torch.tensor([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 24, -1], device="cuda")   # two indices outside the 24-frame store
    tr.step(dev.x, dev.y, rows=bad)
    assert tr.bad_row_count() == 2 and tr.bad_row_count() == 0       # counted, clamped to row 0, never dereferenced
