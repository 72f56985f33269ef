"""Frame dataset the training scripts construct at import time (reference packages/data_handling.py:19-67).

Only `HDF5CleanSpectrogramLabeledFrames` is provided: it is the one class the hot-path callers
(scripts/training_M1.py:79-84, training_M2.py:77-82, training_M2_info_vad.py) import.  h5py is
imported lazily, so this module imports on machines without it; the whole-utterance datasets of
the reference (for the external audio / video classifier nets) are outside the hot-path scope.

On-disk format honoured: datasets `X_<split>` (513, N) float32 and `Y_<split>` (y_dim, N) float32,
one frame per column; __getitem__(i) -> (x[513], y[y_dim]) float32 tensors.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset


class HDF5CleanSpectrogramLabeledFrames(Dataset):
    def __init__(self,
                 input_video_dir, dataset_name, dataset_type,
                 dataset_size, labels='vad_labels', upsampled=False,
                 rdcc_nbytes=1024**2*40, rdcc_nslots=1e4):
        self.input_video_dir = input_video_dir
        self.dataset_name = dataset_name
        self.dataset_type = dataset_type
        self.dataset_size = dataset_size
        self.labels = labels
        self.upsampled = upsampled
        self.rdcc_nbytes = rdcc_nbytes
        self.rdcc_nslots = rdcc_nslots
        suffix = '_upsampled.h5' if upsampled else '.h5'
        self.input_data_file = os.path.join(input_video_dir, dataset_name, 'Clean' + '_' + labels + suffix)
        import h5py as h5
        # the file is NOT kept open here: DataLoader workers must open their own handle
        with h5.File(self.input_data_file, 'r') as file:
            self.dataset_len = file["X_" + dataset_type].shape[-1]

    def open_hdf5(self):
        import h5py as h5
        self.f = h5.File(self.input_data_file, 'r', rdcc_nbytes=self.rdcc_nbytes, rdcc_nslots=self.rdcc_nslots)
        self.data = self.f['X_' + self.dataset_type]
        self.labels = self.f['Y_' + self.dataset_type]

    def __getitem__(self, i):
        if not hasattr(self, 'f'):
            self.open_hdf5()
        data = np.array(self.data[..., i])
        labels = np.array(self.labels[..., i])
        return torch.Tensor(data), torch.Tensor(labels)

    def __len__(self):
        return self.dataset_len

    def __del__(self):
        if hasattr(self, 'f'):
            self.f.close()
