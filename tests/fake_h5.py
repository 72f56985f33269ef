"""A stand-in for the absent `h5py` module over in-memory arrays (test helper): File(path, 'r', **kw)[name] -> an object
with .shape and numpy indexing, context manager, `in`, close().  `install(monkeypatch, arrays)` puts it in sys.modules."""
import sys
import types


class FakeDataset:
    def __init__(self, arr):
        self.arr = arr
        self.shape = arr.shape
        self.dtype = arr.dtype

    def __getitem__(self, idx):
        return self.arr[idx]

    def __len__(self):
        return self.shape[0]


class FakeFile:
    data = {}
    opened = 0           # handles currently open (lazy-open / close bookkeeping for the tests)
    opens = 0

    def __init__(self, path, mode="r", **kw):
        self.path, self.kw = path, kw
        FakeFile.opened += 1
        FakeFile.opens += 1

    def __getitem__(self, k):
        return FakeDataset(FakeFile.data[k])

    def __contains__(self, k):
        return k in FakeFile.data

    def __iter__(self):
        return iter(FakeFile.data)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False

    def close(self):
        FakeFile.opened -= 1


def install(monkeypatch, arrays):
    FakeFile.data = dict(arrays)
    FakeFile.opened = FakeFile.opens = 0
    mod = types.ModuleType("h5py")
    mod.File = FakeFile
    monkeypatch.setitem(sys.modules, "h5py", mod)
    return FakeFile
