"""CPU (no GPU): the C-ABI library loads and exports every symbol include/*.h declares,
the host-side modules behave like the reference on host tensors, and the indexing
logic of the STFT wrappers (pad rule, frame counts) matches the known answers."""
import ctypes
import glob
import importlib
import os
import re
import sys

import numpy as np
import pytest
import torch

import golden_util as gu
from golden_check import check_case
from impl_modules import ModuleImpl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
native = importlib.import_module("disentangled-vae_amd.native")
stft_host = importlib.import_module("disentangled-vae_amd.stft")


def _declared_functions():
    names = []
    for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(dvae_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    if not native.is_built():
        importlib.import_module("disentangled-vae_amd.build").build(verbose=False)
    lib = ctypes.CDLL(native.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
    # and the ctypes binding types every one of them
    assert sorted(native.SIGNATURES) == declared
    typed = native.load()
    assert typed.dvae_abi_version() == native.ABI_VERSION
    assert typed.dvae_last_error() is not None


def test_bad_arguments_are_reported_not_crashed():
    lib = native.load()
    rc = lib.dvae_linear_act_fwd(None, 4, 4, None, 0, 0, None, 4, None, None, 4, 1, 4, 1, None)
    assert rc != 0 and b"linear_act_fwd" in lib.dvae_last_error()
    rc = lib.dvae_stft(None, 1, 0, None, 1024, 256, 1, None, 0, None)
    assert rc != 0


def test_cuda_path_has_no_fallback(monkeypatch):
    """A CUDA tensor must reach the HIP library or raise: hide the library and check the error."""
    ops = importlib.import_module("disentangled-vae_amd.ops")
    monkeypatch.setattr(native, "LIB_PATH", "/nonexistent/libdvae_hip.so")
    monkeypatch.setattr(native, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native.load()
    with pytest.raises(RuntimeError):
        stft_host.stft_numpy(np.zeros(4096), 16000, 64e-3, "hann", 0.25, False, "reflect", True, "complex64") \
            if not torch.cuda.is_available() else native.load()
    assert ops is not None


@pytest.mark.parametrize("case", gu.CASES, ids=[c[0] for c in gu.CASES])
def test_host_modules_match_reference_vectors(vae_golden, case):
    """packages.models on host tensors == the reference's CPU mode (BASELINE config 0)."""
    check_case(ModuleImpl("cpu"), vae_golden, case)


def test_state_dict_keys_and_param_counts():
    from packages.models.models import VariationalAutoencoder, DeepGenerativeModel, DeepGenerativeModel_v5
    from packages.utils import count_parameters
    h = [128, 128]
    assert count_parameters(VariationalAutoencoder([513, 16, h])) == 171297
    assert count_parameters(DeepGenerativeModel([513, 1, 16, h], None)) == 171553
    assert count_parameters(DeepGenerativeModel([513, 513, 16, h], None)) == 302625
    m5 = DeepGenerativeModel_v5([513, 1, 16, h])
    assert count_parameters(m5) == 272675
    assert [(k, tuple(v.shape)) for k, v in m5.state_dict().items()] == \
        gu.layer_dims("M2_info", 513, 1, 16, (128, 128))
    assert m5.enc_dec_clf.z_dim == 16 and m5.enc_dec_clf.y_dim == 1 and m5.enc_dec_clf.flow is None


def test_seeded_init_and_forward_known_answer():
    """Same seed -> same init and same forward as the reference (SURVEY.md 8c KAT)."""
    from packages.models.models import VariationalAutoencoder
    from packages.models.utils import elbo
    torch.manual_seed(0)
    m = VariationalAutoencoder([513, 16, [128, 128]])
    x = torch.rand(32, 513) ** 2
    r, mu, lv = m(x)
    got = [t.item() for t in elbo(x, r, mu, lv, 1e-8)]
    np.testing.assert_allclose(got, [698.5711669921875, 686.7371215820312, 11.834017753601074], rtol=1e-6)
    assert m.kl_divergence.shape == (32,)


def test_decoder_accepts_3d_input():
    from packages.models.models import Decoder
    d = Decoder([5, [8, 8], 11])
    assert d(torch.randn(3, 4, 5)).shape == (3, 4, 11)


# ---- STFT host logic -------------------------------------------------------------------------

PAD_KATS = [  # (n, padded?, T) at fs=16000, wlen 64 ms, hop 25 %, center=False  
    (81920, False, 317), (73045, True, 283), (71680, False, 277), (16000, True, 60), (32000, False, 122),
    (82944, False, 321), (11008, True, 41), (76117, True, 295), (102741, True, 399), (90795, True, 352),
    (70315, True, 272), (94891, True, 368),
]


@pytest.mark.parametrize("n,pad,T", PAD_KATS)
def test_pad_rule_known_answers(n, pad, T):
    from oracle import stft_oracle as so
    for mod in (stft_host.needs_end_pad, so.pad_decision):
        assert mod(n, 16000, 64e-3, 0.25) == pad
    nfft, hop = stft_host.sizes(16000, 64e-3, 0.25)
    assert (nfft, hop) == (1024, 256)
    assert stft_host.frame_count(n + (hop if pad else 0), nfft, hop) == T


def test_pad_rule_fp_quirk_multiples_of_hop():
    """Quirk Q6: some exact multiples of 256 are padded because of double rounding."""
    padded = [k for k in range(4, 150) if stft_host.needs_end_pad(k * 256, 16000, 64e-3, 0.25)]
    for k in (43, 51, 59, 71, 86, 87, 102, 103, 118, 119, 141, 142, 143):
        assert k in padded
    from oracle import stft_oracle as so
    assert padded == [k for k in range(4, 150) if so.pad_decision(k * 256, 16000, 64e-3, 0.25)]


def test_kats_from_reference_label_files(stft_golden):
    for n, T in stft_golden["kat_len_frames"]:
        pad = stft_host.needs_end_pad(int(n), 16000, 64e-3, 0.25)
        assert stft_host.frame_count(int(n) + (256 if pad else 0), 1024, 256) == int(T)


def test_non_integer_window_raises():
    with pytest.raises(ValueError, match="wlen_sample of STFT is not an integer"):
        stft_host.sizes(16000, 50.01e-3, 0.25)
    from packages.processing import stft as ps
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no GPU"):
            ps.stft(np.zeros(4000), fs=16000, wlen_sec=64e-3, center=False)


@pytest.mark.parametrize("model,y_dim", [("M1", 0), ("M2", 1), ("M2", 513), ("M2_info", 1)])
@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
def test_train_plan_is_host_only_and_consistent(model, y_dim, precision):
    """dvae_train_plan is pure host logic (no GPU): tensor table in state_dict order, frame padding, and the slice count of the
    weight-gradient pass -- as many as fill the CUs in one round (the kernel's XCD-aware index map keeps 8 * (ks / 8) of them on
    one XCD each), at most 16 (the apply pass sums that many slabs with independent loads), never more than the
    frames allow; an explicit hint wins."""
    T = importlib.import_module("disentangled-vae_amd.trainer")
    lib = native.load()
    for B in (1, 33, 128, 8192, 1 << 20):
        plan = T.TrainPlan()
        native.check(lib.dvae_train_plan(T.MODEL_CODE[model], y_dim, T.PREC_CODE[precision], B, 0, ctypes.byref(plan)), "dvae_train_plan")
        assert plan.B == B and plan.Bp % 128 == 0 and B <= plan.Bp < B + 128
        assert plan.n_tensors == (26 if model == "M2_info" else 14)
        offs = [plan.tensor_offset[i] for i in range(plan.n_tensors)]
        assert offs == sorted(offs) and all(o % 64 == 0 for o in offs) and plan.n_params >= offs[-1]
        assert 1 <= plan.ksplit <= 16 and plan.ksplit <= max(1, plan.Bp // 128)
        if B == 8192:
            assert plan.ksplit >= 7          # enough (slice, block) workgroups to cover the CUs in one round
        assert plan.workspace_bytes > plan.grad_offset_bytes > 0
        assert plan.workspace_bytes - plan.grad_offset_bytes >= plan.ksplit * plan.n_params * 4
        assert 1 <= plan.rows_grid <= 512
    plan = T.TrainPlan()
    native.check(lib.dvae_train_plan(T.MODEL_CODE[model], y_dim, T.PREC_CODE[precision], 8192, 5, ctypes.byref(plan)), "dvae_train_plan")
    assert plan.ksplit == 5


@pytest.mark.parametrize("model,y_dim", [("M1", 0), ("M2", 1), ("M2", 513), ("M2_info", 1)])
def test_weight_gradient_schedule_of_the_plan(model, y_dim, monkeypatch):
    """The weight-gradient launch reads a host-built item table; the plan says which: reserved0 = 0 -> `ksplit` uniform slices;
    reserved0 > 0 -> class-sliced (every block of tiles cut by its own cost per k-step; reserved0 = workgroups, ksplit = the largest
    slice count = slabs the optimizer launch sums).  Class-sliced is the default under the exact-fp32 policy only (measured: DESIGN 4a),
    DVAE_W4_CLASSES / DVAE_W4_UNIFORM override, an explicit slice hint or a batch of one 128-frame slice is always uniform; one round of
    workgroups (<= 256 items), <= 16 slabs, >= 128 frames per slice."""
    T = importlib.import_module("disentangled-vae_amd.trainer")
    lib = native.load()
    for k in ("DVAE_W4_CLASSES", "DVAE_W4_UNIFORM", "DVAE_WGRAD", "DVAE_FOLD_APPLY", "DVAE_DEFER_APPLY"):
        monkeypatch.delenv(k, raising=False)

    def plan(prec, B, hint=0):
        p = T.TrainPlan()
        native.check(lib.dvae_train_plan(T.MODEL_CODE[model], y_dim, T.PREC_CODE[prec], B, hint, ctypes.byref(p)), "dvae_train_plan")
        return p
    for B in (300, 8192, 262144):
        assert plan("bf16x3", B).reserved0 == 0 and plan("bf16", B).reserved0 == 0
        p = plan("fp32", B)
        assert 0 < p.reserved0 <= 256 and 1 <= p.ksplit <= 16 and p.ksplit <= p.Bp // 128
        assert p.workspace_bytes - p.grad_offset_bytes >= p.ksplit * p.n_params * 4
        assert plan("fp32", B, hint=6).reserved0 == 0 and plan("fp32", B, hint=6).ksplit == 6
    assert plan("fp32", 100).reserved0 == 0                      # one 128-frame slice: nothing to schedule
    monkeypatch.setenv("DVAE_W4_CLASSES", "1")
    p = plan("bf16x3", 8192)
    assert 0 < p.reserved0 <= 256 and p.ksplit <= 16
    monkeypatch.setenv("DVAE_W4_UNIFORM", "1")
    assert plan("bf16x3", 8192).reserved0 == 0 and plan("fp32", 8192).reserved0 == 0


def test_train_plan_m2_dec_exists_for_the_8_wave_kernel_only(monkeypatch):
    """Kernel model M2_DEC (encoder on x alone, decoder on [z | y]: the VAE body of DeepGenerativeModel_v3 / _v5 on the module
    path): same 14 tensors as M2 with a 513-wide first encoder layer; y_dim 1 and the split-bf16 / bf16 policies only -- anything
    else is refused with a message, never planned onto a kernel that does not exist."""
    T = importlib.import_module("disentangled-vae_amd.trainer")
    lib = native.load()
    monkeypatch.delenv("DVAE_ROWS", raising=False)
    plan = T.TrainPlan()
    native.check(lib.dvae_train_plan(T.MODEL_CODE["M2_DEC"], 1, T.PREC_CODE["bf16x3"], 8192, 0, ctypes.byref(plan)), "dvae_train_plan")
    assert plan.n_tensors == 14 and plan.rows_kernel == 2
    assert (plan.tensor_rows[0], plan.tensor_cols[0]) == (128, 513) and (plan.tensor_rows[8], plan.tensor_cols[8]) == (128, 17)
    for y_dim, prec in ((513, "bf16x3"), (0, "bf16x3"), (1, "fp32")):
        assert lib.dvae_train_plan(T.MODEL_CODE["M2_DEC"], y_dim, T.PREC_CODE[prec], 8192, 0, ctypes.byref(plan)) != 0
        assert "M2_DEC" in lib.dvae_last_error().decode("utf-8", "replace")
    monkeypatch.setenv("DVAE_ROWS", "1")
    assert lib.dvae_train_plan(T.MODEL_CODE["M2_DEC"], 1, T.PREC_CODE["bf16x3"], 8192, 0, ctypes.byref(plan)) != 0
