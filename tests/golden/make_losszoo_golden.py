"""Golden values of every loss in the reference's packages/models/utils.py on seeded inputs (CPU, fp32).
Build-container only (imports /root/reference):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_losszoo_golden.py
Output: tests/golden/losszoo_golden.npz (inputs are regenerated from the seed by tests/losszoo_inputs.py)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import numpy as np
import torch

from packages.models import utils as U
import losszoo_inputs as li


def main():
    out = {}
    for name, B, F, L, seed in li.CASES:
        d = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in li.make(B, F, L, seed).items()}
        res = li.evaluate(U, d)
        for k, v in res.items():
            out[f"{name}/{k}"] = np.asarray(v)
        out[f"{name}/checksum"] = np.array(li.checksum(li.make(B, F, L, seed)))
    path = os.path.join(HERE, "losszoo_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "entries")


if __name__ == "__main__":
    main()
