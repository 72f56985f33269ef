"""Every loss of packages/models/utils.py against golden values captured from the reference's own utils.py
(tests/golden/make_losszoo_golden.py): host tensors here, CUDA tensors (HIP kernels for elbo / BCE) in the gpu test."""
import os

import numpy as np
import pytest
import torch

import losszoo_inputs as li
from packages.models import utils as U

FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "losszoo_golden.npz"))


def run(device, rtol):
    for name, B, F, L, seed in li.CASES:
        d = li.make(B, F, L, seed)
        assert abs(li.checksum(d) - float(FIX[f"{name}/checksum"])) < 1e-6 * float(FIX[f"{name}/checksum"])
        dt = {k: torch.from_numpy(v).to(device) for k, v in d.items()}
        res = li.evaluate(U, dt)
        for k, v in res.items():
            np.testing.assert_allclose(v, FIX[f"{name}/{k}"], rtol=rtol, atol=1e-6, err_msg=f"{name}/{k}")


def test_loss_zoo_matches_reference_on_host():
    run("cpu", 1e-6)


@pytest.mark.gpu
def test_loss_zoo_matches_reference_on_gpu():
    run("cuda", 1e-4)
