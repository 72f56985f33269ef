"""GPU-resident frame store: the MI355X-native replacement of the reference's training input pipeline.

The reference reads ONE 513-float column per __getitem__ from an lzf-compressed HDF5 file through 16
DataLoader workers (packages/data_handling.py:45-60, scripts/training_M2.py:77-89).  The fused train
step consumes >100 M frames/s; a whole training set of spectrogram frames (2 KB each) fits the 288 GB
of HBM many times over, so the set is loaded once, kept frames-major ([N][513], the layout the train
step reads), and every epoch is ONE row gather into a second buffer (a random permutation, what
DataLoader(shuffle=True) produces) after which batches are contiguous views: no per-step copy at all.
"""
import numpy as np
import torch

from . import native as N

F_BINS = 513


def _to_device_rows(a, device, chunk=1 << 18):
    """(F, n) host/device matrix, one frame per column (on-disk orientation) -> [n][F] float32 device rows."""
    lib = N.load()
    rows, n = a.shape
    out = torch.empty((n, rows), dtype=torch.float32, device=device)
    for c0 in range(0, n, chunk):               # bounded staging: the file may be larger than pinned host memory
        c1 = min(n, c0 + chunk)
        blk = a[:, c0:c1]
        blk = torch.from_numpy(np.ascontiguousarray(blk, dtype=np.float32)) if not torch.is_tensor(blk) else blk.to(torch.float32)
        blk = blk.to(device).contiguous()
        N.check(lib.dvae_transpose(N.ptr(blk), rows, c1 - c0, c1 - c0, N.ptr(out[c0:c1]), rows, N.stream()), "dvae_transpose")
    return out


class DeviceFrames:
    """Frames X [N][513] and labels Y [N][y_dim] resident in HBM.

    X, Y: arrays / tensors / h5py datasets in the on-disk orientation (513, N), (y_dim, N); Y may be None (M1).
    """

    def __init__(self, X, Y=None, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceFrames needs the MI355X HIP path (no CPU fallback)")
        self.device = torch.device(device)
        with torch.cuda.device(self.device):
            self.x = _to_device_rows(X, self.device)
            self.y = _to_device_rows(Y, self.device) if Y is not None else None
        if self.y is not None and self.y.shape[0] != self.x.shape[0]:
            raise ValueError("X and Y hold different numbers of frames")
        self._xs = self._ys = None
        self._bad = torch.zeros(1, dtype=torch.int32, device=self.device)

    @classmethod
    def from_hdf5(cls, path, split, device="cuda:0"):
        """The reference's training file (scripts/create_train_set.py:91-219): datasets X_<split>, Y_<split>."""
        import h5py
        with h5py.File(path, "r") as f:
            return cls(f["X_" + split], f["Y_" + split] if ("Y_" + split) in f else None, device)

    def __len__(self):
        return self.x.shape[0]

    def _gather(self, src, idx, dst):
        lib = N.load()
        N.check(lib.dvae_gather_rows(N.ptr(src), src.stride(0), src.shape[0], N.ptr(idx), idx.numel(), src.shape[1], N.ptr(dst),
                                     dst.stride(0), N.ptr(self._bad), N.stream()), "dvae_gather_rows")

    def shuffled(self, generator=None):
        """One epoch's order: returns (x, y) permuted copies (buffers are reused from epoch to epoch)."""
        n = len(self)
        with torch.cuda.device(self.device):
            perm = torch.randperm(n, device=self.device, generator=generator)
            if self._xs is None:
                self._xs = torch.empty_like(self.x)
                self._ys = torch.empty_like(self.y) if self.y is not None else None
            self._gather(self.x, perm, self._xs)
            if self.y is not None:
                self._gather(self.y, perm, self._ys)
        self.last_perm = perm
        return self._xs, self._ys

    def index_batches(self, batch, shuffle=True, drop_last=False, generator=None):
        """Yields int64 row-index tensors [b] for `Trainer.step(data.x, data.y, rows=idx)`: the rows kernel gathers the
        frames itself (2 KB contiguous per frame), so an epoch's shuffle costs one randperm instead of a copy of the set."""
        n = len(self)
        with torch.cuda.device(self.device):
            order = torch.randperm(n, device=self.device, generator=generator) if shuffle else torch.arange(n, device=self.device)
        self.last_perm = order
        for s in range(0, n, batch):
            e = min(n, s + batch)
            if e - s < batch and drop_last:
                return
            yield order[s:e]

    def batches(self, batch, shuffle=True, drop_last=False, generator=None):
        """Yields (x [b,513], y [b,y_dim] or None) contiguous device views, b == batch except possibly the last."""
        x, y = self.shuffled(generator) if shuffle else (self.x, self.y)
        n = len(self)
        for s in range(0, n, batch):
            e = min(n, s + batch)
            if e - s < batch and drop_last:
                return
            yield x[s:e], (y[s:e] if y is not None else None)
