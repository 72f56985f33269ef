"""Frame dataset the training scripts construct at import time (reference packages/data_handling.py:19-67).

`HDF5CleanSpectrogramLabeledFrames` keeps the reference's constructor signature and item contract --
`ds[i] -> (x[513], y[y_dim])` float32 tensors, `len(ds)` = frames of the split -- because
scripts/training_M1.py:79-84, training_M2.py:77-82 and training_M2_info_vad.py build it at import time.
It is written on this repo's own terms: a split of the frame file is a `FrameFile` (below) that knows the
on-disk layout and opens lazily (once per DataLoader worker); the dataset is a thin Dataset view over it,
and `to_device()` hands the whole split to the GPU-resident frame store (disentangled-vae_amd/frames.py),
which is how the fused trainer consumes it (examples/train_fused.py).

On-disk format (scripts/create_train_set.py:91-219): datasets `X_<split>` (513, N) float32 and
`Y_<split>` (y_dim, N) float32, one frame per COLUMN, lzf-compressed with one-frame chunks.
h5py is imported only when a file is opened, so this module imports on machines without it.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset


class FrameFile:
    """One split of a labelled-frame HDF5 file.  Nothing is held open until the first read; a pickled copy
    (DataLoader worker) starts closed and opens its own handle."""

    def __init__(self, path, split, cache_bytes=40 * 1024 ** 2, cache_slots=1e4):
        self.path, self.split = path, split
        self.cache_bytes, self.cache_slots = cache_bytes, cache_slots
        self._h = self._x = self._y = None

    def _open(self, **kw):
        import h5py
        return h5py.File(self.path, 'r', **kw)

    def frames(self):
        """Number of frames in the split (opens the file briefly, keeps nothing open)."""
        with self._open() as f:
            return f['X_' + self.split].shape[-1]

    def _ensure(self):
        if self._h is None:
            # chunk cache sized by the caller: every read is one (513, 1) chunk
            self._h = self._open(rdcc_nbytes=self.cache_bytes, rdcc_nslots=self.cache_slots)
            self._x, self._y = self._h['X_' + self.split], self._h['Y_' + self.split]

    def column(self, i):
        """Frame i: (x (513,), y (y_dim,)) as float32 numpy vectors."""
        self._ensure()
        return np.asarray(self._x[..., i], dtype=np.float32), np.asarray(self._y[..., i], dtype=np.float32)

    def arrays(self):
        """The open X / Y datasets, on-disk orientation (features, N)."""
        self._ensure()
        return self._x, self._y

    def close(self):
        if self._h is not None:
            self._h.close()
        self._h = self._x = self._y = None

    def __getstate__(self):
        st = dict(self.__dict__)
        st['_h'] = st['_x'] = st['_y'] = None
        return st


def frame_file_path(root, dataset_name, labels, upsampled):
    """<root>/<dataset_name>/Clean_<labels>[_upsampled].h5 (scripts/training_M2.py:72-75)."""
    return os.path.join(root, dataset_name, 'Clean_{}{}.h5'.format(labels, '_upsampled' if upsampled else ''))


class HDF5CleanSpectrogramLabeledFrames(Dataset):
    def __init__(self,
                 input_video_dir, dataset_name, dataset_type,
                 dataset_size, labels='vad_labels', upsampled=False,
                 rdcc_nbytes=1024**2*40, rdcc_nslots=1e4):
        self.input_video_dir, self.dataset_name = input_video_dir, dataset_name
        self.dataset_type, self.dataset_size = dataset_type, dataset_size
        self.labels, self.upsampled = labels, upsampled
        self.input_data_file = frame_file_path(input_video_dir, dataset_name, labels, upsampled)
        self.frames = FrameFile(self.input_data_file, dataset_type, rdcc_nbytes, rdcc_nslots)
        self.dataset_len = self.frames.frames()

    def __len__(self):
        return self.dataset_len

    def __getitem__(self, i):
        x, y = self.frames.column(i)
        return torch.from_numpy(x), torch.from_numpy(y)

    def to_device(self, device="cuda:0"):
        """The whole split as a GPU-resident frame store (frames-major rows in HBM)."""
        import importlib
        frames = importlib.import_module("disentangled-vae_amd.frames")
        X, Y = self.frames.arrays()
        return frames.DeviceFrames(X, Y, device)

    def __del__(self):
        fr = self.__dict__.get('frames')
        if fr is not None:
            fr.close()
