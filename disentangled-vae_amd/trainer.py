"""Fused train-step harness: the loop body of scripts/training_M1.py:134-139 /
scripts/training_M2.py:142-147 (model forward, elbo, backward, Adam step, zero_grad) as
three HIP launches over caller-owned flat buffers (include/dvae_train.h).

PyTorch is plumbing here: it owns the device buffers, the stream and (for N > 1) the
RCCL all-reduce of the flat gradient.  Parameters are exposed under the reference's
state_dict names so checkpoints move both ways.  No fallback: unsupported geometries
raise (the layer-level modules in packages/models cover them).
"""
import ctypes
import os

import numpy as np
import torch

from . import native as N
from . import dp

MAXT = 32


class TrainPlan(ctypes.Structure):
    _fields_ = [("model", ctypes.c_int32), ("y_dim", ctypes.c_int32), ("precision", ctypes.c_int32), ("ksplit", ctypes.c_int32),
                ("B", ctypes.c_int64), ("n_tensors", ctypes.c_int32), ("reserved0", ctypes.c_int32), ("n_params", ctypes.c_int64),
                ("tensor_offset", ctypes.c_int64 * MAXT), ("tensor_rows", ctypes.c_int32 * MAXT), ("tensor_cols", ctypes.c_int32 * MAXT),
                ("workspace_bytes", ctypes.c_int64), ("grad_offset_bytes", ctypes.c_int64), ("Bp", ctypes.c_int64),
                ("rows_grid", ctypes.c_int64), ("flops_per_step", ctypes.c_double), ("min_hbm_bytes_per_step", ctypes.c_double),
                ("info_alpha", ctypes.c_double), ("info_beta", ctypes.c_double), ("info_gamma", ctypes.c_double),
                ("rng_seed", ctypes.c_uint64), ("rng_step", ctypes.c_uint64), ("loss_accum", ctypes.c_uint64),
                ("row_index", ctypes.c_uint64), ("row_count", ctypes.c_int64), ("bad_row_counter", ctypes.c_uint64),
                ("rows_kernel", ctypes.c_int32), ("reserved1", ctypes.c_int32)]


MODEL_CODE = {"M1": 1, "M2": 2, "M2_info": 3, "M2_DEC": 4}    # M2_DEC: encoder on x alone, decoder on [z | y] (the VAE body of _v3 / _v5)
# matrix-core operand policies (include/dvae_train.h): fp32 = exact fp32 MFMA; bf16 = one bf16 per operand (fast, loose);
# bf16x3 = split bf16 (hi + lo planes, three MFMAs per product): the parity-grade throughput mode
PREC_CODE = {"fp32": 0, "bf16": 1, "bf16x3": 2}
TENSOR_NAMES = ["encoder.hidden.0.weight", "encoder.hidden.0.bias", "encoder.hidden.1.weight", "encoder.hidden.1.bias",
                "encoder.sample.mu.weight", "encoder.sample.mu.bias", "encoder.sample.log_var.weight", "encoder.sample.log_var.bias",
                "decoder.hidden.0.weight", "decoder.hidden.0.bias", "decoder.hidden.1.weight", "decoder.hidden.1.bias",
                "decoder.reconstruction.weight", "decoder.reconstruction.bias"]
# M2_info (DeepGenerativeModel_v5): the same 14 under "enc_dec_clf.", then the classifier and the auxiliary net
INFO_NAMES = (["enc_dec_clf." + n for n in TENSOR_NAMES]
              + [f"enc_dec_clf.classifier.{l}.{w}" for l in ("hidden.0", "hidden.1", "output_layer") for w in ("weight", "bias")]
              + [f"auxiliary.{l}.{w}" for l in ("hidden.0", "hidden.1", "output_layer") for w in ("weight", "bias")])


def tensor_names(model):
    return INFO_NAMES if model == "M2_info" else TENSOR_NAMES

def _lib():
    return N.load()


def supported(model, dims):
    return (model in MODEL_CODE and dims["x_dim"] == 513 and dims["z_dim"] == 16 and tuple(dims["h_dim"]) == (128, 128)
            and ((model == "M1" and dims.get("y_dim", 0) in (0, None)) or (model == "M2" and dims["y_dim"] in (1, 513))
                 or (model == "M2_info" and dims["y_dim"] == 1)))


class Trainer:
    """One object = one model replica on one GPU.

    step(x, y, eps_noise) enqueues one full train step and returns a device tensor
    [ELBO, recon, KL] (no host sync).  With `process_group` given (world > 1) the flat
    gradient is summed over ranks with one RCCL all-reduce and scaled by 1 / world."""

    def __init__(self, model, dims, params=None, batch=128, device="cuda:0", precision="fp32", lr=1e-4, betas=(0.9, 0.999),
                 adam_eps=1e-8, elbo_eps=1e-8, process_group=None, world=1, ksplit=0, seed=None, alpha=0.0, beta=10.0, gamma=1.0,
                 share=None, direct_exchange=None):
        if not torch.cuda.is_available():
            raise RuntimeError("Trainer needs the MI355X HIP path (no CPU fallback)")
        if not supported(model, dims):
            raise NotImplementedError(f"fused train step covers M1 / M2 (y 1 or 513) / M2_info (y 1) at x 513, z 16, h [128,128]; got {model} {dims}")
        self.lib = _lib()
        self.model, self.dims, self.B = model, dict(dims), int(batch)
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.precision = precision
        self.lr, self.betas, self.adam_eps, self.elbo_eps = lr, betas, adam_eps, elbo_eps
        self.pg, self.world = process_group, int(world)
        # DVAE_DEFER_APPLY=1 (opt-in): two launches per step, the optimizer update deferred into the next step's rows kernel -- bit-identical,
        # measured NOT faster on the MI355X; single-GPU only: the data-parallel step puts the gradient exchange
        # between the weight-gradient kernel and the update
        self._defer = int(world) == 1 and os.environ.get("DVAE_DEFER_APPLY", "0") == "1"
        self.y_dim = 0 if model == "M1" else int(dims["y_dim"])
        self.plan = TrainPlan()
        N.check(self.lib.dvae_train_plan(MODEL_CODE[model], self.y_dim, PREC_CODE[precision], self.B, ksplit, ctypes.byref(self.plan)),
                "dvae_train_plan")
        self.names = tensor_names(model)
        if model == "M2_info":          # loss weights of scripts/training_M2_info_vad.py:53-55
            self.plan.info_alpha, self.plan.info_beta, self.plan.info_gamma = float(alpha), float(beta), float(gamma)
        P = self.plan.n_params
        with torch.cuda.device(self.device):
            if share is None:
                self._params = torch.zeros(P, dtype=torch.float32, device=self.device)
                self._m = torch.zeros(P, dtype=torch.float32, device=self.device)
                self._v = torch.zeros(P, dtype=torch.float32, device=self.device)
                self._shared = {"step_count": 0, "version": 0}
            else:                       # same parameters / Adam state, another batch size (see fork())
                self._params, self._m, self._v, self._shared = share._params, share._m, share._v, share._shared
            self.ws = torch.empty(self.plan.workspace_bytes, dtype=torch.uint8, device=self.device)
            self.losses = torch.zeros(8 if model == "M2_info" else 3, dtype=torch.float32, device=self.device)
            self.bad_rows = torch.zeros(1, dtype=torch.int32, device=self.device)      # gather indices the kernels refused (see bad_row_count)
        self.plan.bad_row_counter = self.bad_rows.data_ptr()
        go = self.plan.grad_offset_bytes
        self.flat_grad = self.ws[go:go + 4 * P].view(torch.float32)        # slab 0
        # two weight-gradient launches per step (plan made under DVAE_EXCHANGE_GROUPS=2): the parts of the flat gradient they fill
        self._groups, self._group_range = 1, None
        lo, hi, ng = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        N.check(self.lib.dvae_train_group_range(ctypes.byref(self.plan), -1, ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(ng)), "dvae_train_group_range")
        if ng.value == 2:
            self._groups, self._group_range = 2, []
            for grp in (0, 1):
                N.check(self.lib.dvae_train_group_range(ctypes.byref(self.plan), grp, ctypes.byref(lo), ctypes.byref(hi), None), "dvae_train_group_range")
                self._group_range.append((int(lo.value), int(hi.value)))
        # gradient exchange of the data-parallel step: the process group's all-reduce (RCCL over xGMI), or the library's own
        # stream-ordered exchange over peer pointers (DVAE_ALLREDUCE=direct; dp.DirectExchange) -- unmeasured on multi-GPU hardware
        self.direct = share.direct if share is not None else direct_exchange      # direct_exchange: a connected dp.DirectExchange of n_params floats
        if self.direct is not None and self.direct.n != P:
            raise ValueError("direct_exchange was created for another parameter count")
        if self.world > 1 and share is None and self.direct is None and dp.exchange_mode() == "direct":
            with torch.cuda.device(self.device):
                self.direct = dp.DirectExchange(P, process_group)
        self._copy_version = self._shared["version"]
        rank = torch.distributed.get_rank(process_group) if (process_group is not None and torch.distributed.is_initialized()) else 0
        base = int(seed) if seed is not None else (share.plan.rng_seed if share is not None else int(torch.initial_seed()))
        self.plan.rng_seed = (base + (0 if share is not None else rank * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
        if share is None:
            if params is None:
                params = self._reference_init(seed)
            self._write_params(params)
        with torch.cuda.device(self.device):
            N.check(self.lib.dvae_train_init(ctypes.byref(self.plan), N.ptr(self._params), N.ptr(self.ws), N.stream()), "dvae_train_init")

    # ---- deferred optimizer step (include/dvae_train.h: dvae_train_step_deferred): the update of the last step() is applied in the opening
    # of the next step's first kernel; anything that looks at parameters or moments flushes it first (one apply launch)
    def flush(self):
        """Apply this trainer's pending optimizer update, if any (stream-ordered, no host sync)."""
        if self.lib.dvae_train_pending(N.ptr(self.ws)):
            with torch.cuda.device(self.device):
                N.check(self.lib.dvae_train_flush(ctypes.byref(self.plan), N.ptr(self.ws), N.stream()), "dvae_train_flush")
        if self._shared.get("pending") is self:
            self._shared["pending"] = None

    def _flush_shared(self, but=None):
        other = self._shared.get("pending")
        if other is not None and other is not but:
            other.flush()

    @property
    def params(self):
        self._flush_shared()
        return self._params

    @params.setter
    def params(self, t):
        self._flush_shared()
        self._params = t

    @property
    def m(self):
        self._flush_shared()
        return self._m

    @m.setter
    def m(self, t):
        self._flush_shared()
        self._m = t

    @property
    def v(self):
        self._flush_shared()
        return self._v

    @v.setter
    def v(self, t):
        self._flush_shared()
        self._v = t

    @property
    def step_count(self):
        return self._shared["step_count"]

    @step_count.setter
    def step_count(self, v):
        self._shared["step_count"] = v

    def fork(self, batch):
        """A trainer for another batch size over the SAME parameters and Adam moments (e.g. the last, shorter
        batch of an epoch, or the validation batch size).  Each trainer has its own kernel-layout weight copies;
        they are refreshed automatically when the other one has stepped."""
        return Trainer(self.model, self.dims, batch=batch, device=self.device, precision=self.precision, lr=self.lr, betas=self.betas,
                       adam_eps=self.adam_eps, elbo_eps=self.elbo_eps, process_group=self.pg, world=self.world,
                       alpha=self.plan.info_alpha, beta=self.plan.info_beta, gamma=self.plan.info_gamma, share=self)

    def _sync_copies(self):
        self._flush_shared(but=self)                 # a fork's pending update changes the parameters this trainer is about to use
        if self._copy_version != self._shared["version"]:
            self.flush()
            N.check(self.lib.dvae_train_repack(ctypes.byref(self.plan), N.ptr(self._params), N.ptr(self.ws), N.stream()), "dvae_train_repack")
            self._copy_version = self._shared["version"]

    # ---- parameters under the reference's state_dict names ----
    def _reference_init(self, seed):
        """xavier_normal_ weights / zero bias drawn exactly like the reference constructs the model."""
        from packages.models import models as M
        if seed is not None:
            torch.manual_seed(seed)
        if self.model == "M1":
            m = M.VariationalAutoencoder([513, 16, [128, 128]])
        elif self.model == "M2":
            m = M.DeepGenerativeModel([513, self.y_dim, 16, [128, 128]], None)
        else:
            m = M.DeepGenerativeModel_v5([513, self.y_dim, 16, [128, 128]])
        return {k: v.detach() for k, v in m.state_dict().items()}

    def tensor_view(self, i):
        o, r, c = self.plan.tensor_offset[i], self.plan.tensor_rows[i], self.plan.tensor_cols[i]
        v = self.params[o:o + r * c]
        return v.view(r, c) if self.names[i].endswith("weight") else v

    def _write_params(self, sd, strict=True):
        unknown = [k for k in sd if k not in self.names]
        if strict and unknown:
            raise KeyError(f"unexpected keys in state_dict: {unknown[:4]}")
        for i, name in enumerate(self.names):
            if name not in sd:
                if strict:
                    raise KeyError(f"missing key in state_dict: {name}")
                continue
            t = sd[name]
            t = torch.from_numpy(np.ascontiguousarray(t)) if isinstance(t, np.ndarray) else t.detach()
            view = self.tensor_view(i)
            if tuple(t.shape) != tuple(view.shape):
                raise ValueError(f"{name}: shape {tuple(t.shape)} != {tuple(view.shape)}")
            view.copy_(t.to(torch.float32))

    def load_state_dict(self, sd, strict=True):
        """strict=False overwrites only the tensors present (the pretrain flow of
        scripts/training_M2_info_vad_pretrain.py:102-112: `model_dict.update(filtered); load_state_dict(model_dict)`)."""
        self._flush_shared()
        self._write_params(sd, strict)
        with torch.cuda.device(self.device):
            N.check(self.lib.dvae_train_repack(ctypes.byref(self.plan), N.ptr(self._params), N.ptr(self.ws), N.stream()), "dvae_train_repack")
        self._shared["version"] += 1
        self._copy_version = self._shared["version"]

    def check_exchange(self):
        """Multi-GPU with the direct exchange: raises RuntimeError when a bounded wait for a peer ran out in any step so far (synchronises).
        The failure is in-band as well -- the kernel fills the reduced gradient with NaN on every rank, so parameters and losses turn NaN --
        and this check runs wherever the trainer synchronises with the device anyway (state_dict, grads_numpy, losses_host)."""
        if self.direct is not None and self.direct.failed():
            raise RuntimeError("gradient exchange (dvae_allreduce_flat): a bounded wait for a peer ran out (a rank was missing or later than "
                               "the bound, see DirectExchange.set_timeout_ms); the reduced gradient was NaN-filled on every rank -- this "
                               "trainer's parameters are no longer valid")

    def losses_host(self):
        """The last step's loss scalars on the host (synchronises; checks the gradient exchange first)."""
        out = self.losses.cpu().numpy().copy()
        self.check_exchange()
        return out

    def state_dict(self):
        self.check_exchange()
        return {name: self.tensor_view(i).clone() for i, name in enumerate(self.names)}

    def state_dict_numpy(self):
        return {k: v.cpu().numpy() for k, v in self.state_dict().items()}

    def grads_numpy(self):
        """Flat gradient of the last step (sum of the k-split slabs), per tensor, as numpy."""
        self.check_exchange()
        P, ks = self.plan.n_params, self.plan.ksplit
        go = self.plan.grad_offset_bytes
        slabs = self.ws[go:go + 4 * P * ks].view(torch.float32).view(ks, P)
        flat = slabs[:self._used_slabs()].sum(0) if not self._reduced else slabs[0]
        out = {}
        for i, name in enumerate(self.names):
            o, r, c = self.plan.tensor_offset[i], self.plan.tensor_rows[i], self.plan.tensor_cols[i]
            g = flat[o:o + r * c].cpu().numpy()
            out[name] = g.reshape(r, c) if name.endswith("weight") else g
        return out

    def _used_slabs(self):
        if self.plan.reserved0 > 0:          # class-sliced weight-gradient schedule: every block writes its first s_b <= ksplit slabs, the rest hold zeros
            return self.plan.ksplit
        unit = 4 * (16 if self.precision in ("bf16", "bf16x3") else 8)
        per = -(-self.plan.Bp // self.plan.ksplit)
        kper = -(-per // unit) * unit
        return -(-self.plan.Bp // kper)

    _reduced = False

    # ---- one train step ----
    def _check_inputs(self, x, y, rows, eps_noise=None):
        """rows=None: x [B,513], y [B,y_dim] are the batch.  rows = int64 CUDA tensor [B]: x / y are a whole frame store
        ([N,513], [N,y_dim]) and frame b of the step is row rows[b] (gathered inside the kernel: no copy)."""
        B = self.B
        n = B if rows is None else x.shape[0]
        if x.ndim != 2 or x.shape != (n, 513) or x.dtype != torch.float32 or x.device != self.device:
            raise ValueError(f"x must be a float32 tensor [{n if rows is not None else B}, 513] on {self.device}")
        if self.y_dim:
            if y is None or y.shape != (n, self.y_dim) or y.dtype != torch.float32 or y.device != self.device:
                raise ValueError(f"y must be a float32 tensor [{n}, {self.y_dim}] on {self.device}")
        if eps_noise is not None and (eps_noise.shape != (B, 16) or eps_noise.dtype != torch.float32 or eps_noise.device != self.device):
            raise ValueError(f"eps_noise must be a float32 tensor [{B}, 16] on {self.device}")
        if rows is not None:
            if not (rows.device == self.device and rows.dtype == torch.int64 and rows.shape == (B,) and rows.is_contiguous()):
                raise ValueError(f"rows must be a contiguous int64 tensor [{B}] on {self.device}")
        # the kernels clamp an index outside [0, n) to row 0 and count it in self.bad_rows (no out-of-range read)
        self.plan.row_index = 0 if rows is None else rows.data_ptr()
        self.plan.row_count = 0 if rows is None else n

    def bad_row_count(self):
        """Gather indices outside the frame store seen since the last call (synchronises); the affected frames were
        trained on row 0 instead.  step() never reads memory through such an index."""
        n = int(self.bad_rows.item())
        if n:
            self.bad_rows.zero_()
        return n

    # bf16x3 on the 8-wave rows kernel: the x block of encoder layer 1 (and of the M2_info classifier) multiplies split-FP16 planes with FIXED
    # scales, x * 2^-3 and W * 2^6 (csrc/fused_tiles.hpp: struct X16): finite for |x| <= 5.2e5 (twice what a peak-normalised 1024-point Hann
    # frame can hold) and |w| <= 1023.  The reference takes any float32 (e.g. spectra of int16-scaled audio, up to ~1e9): beyond the range the
    # planes overflow and the losses turn NaN.  So the range is CHECKED -- on the first step of a trainer and on request -- and a violation is
    # an error that says what to do, not a NaN three steps later.
    X16_MAX_X, X16_MAX_W = 5.2e5, 1023.0

    def check_operand_range(self, x):
        """Raises ValueError when x (the batch, or the frame store it is gathered from) or the layer-1 weights leave the range of the
        split-fp16 planes of the bf16x3 policy.  Synchronises (two reductions and one read-back); step() calls it once, on the first step."""
        if self.precision != "bf16x3" or self.plan.rows_kernel != 2:
            return
        first = [0] + ([14] if self.model == "M2_info" else [])
        wmax = max(float(self.tensor_view(i)[:, :513].abs().amax().item()) for i in first)
        xmax = float(x.abs().amax().item())
        if not (xmax <= self.X16_MAX_X and wmax <= self.X16_MAX_W):          # (NaN fails the comparison too)
            raise ValueError(f"precision='bf16x3': inputs up to {xmax:.3g} / layer-1 weights up to {wmax:.3g} leave the range of the fixed-scale split-fp16 "
                             f"planes of the layer-1 x block (|x| <= {self.X16_MAX_X:g}, |w| <= {self.X16_MAX_W:g}): rescale the spectra (the reference's "
                             "front end peak-normalises the audio, scripts/create_train_set.py:138-140) or use precision='fp32', which takes any float32")

    _range_checked = False

    def step(self, x, y=None, eps_noise=None, rows=None):
        self._check_inputs(x, y, rows, eps_noise)
        if not self._range_checked:
            self._range_checked = True
            if os.environ.get("DVAE_RANGE_CHECK", "1") != "0":
                self.check_operand_range(x)
        x = x if x.stride(1) == 1 else x.contiguous()
        if eps_noise is not None:           # None: the rows kernel draws the noise itself (Philox, see noise())
            eps_noise = eps_noise.contiguous()
        yp, ldy = (None, 0)
        if self.y_dim:
            y = y if y.stride(1) == 1 else y.contiguous()
            yp, ldy = N.ptr(y), N.ld(y)
        self._sync_copies()
        self.step_count += 1
        self.plan.rng_step = self.step_count
        self._shared["version"] += 1
        self._copy_version = self._shared["version"]
        plan = ctypes.byref(self.plan)
        with torch.cuda.device(self.device):
            return self._launch_step(plan, x, yp, ldy, eps_noise)

    def _launch_step(self, plan, x, yp, ldy, eps_noise):
        s = N.stream()
        if self.world == 1:
            fn = self.lib.dvae_train_step_deferred if self._defer else self.lib.dvae_train_step
            N.check(fn(plan, N.ptr(self._params), N.ptr(self._m), N.ptr(self._v), N.ptr(self.ws), N.ptr(x), N.ld(x),
                                             yp, ldy, N.ptr(eps_noise), self.elbo_eps, self.step_count, self.lr, self.betas[0],
                       self.betas[1], self.adam_eps, N.ptr(self.losses), s), "dvae_train_step")
            if self._defer:
                self._shared["pending"] = self if self.lib.dvae_train_pending(N.ptr(self.ws)) else None
            self._reduced = False
        else:
            direct = self.direct
            if direct is None and self._groups == 2:
                # DVAE_EXCHANGE_GROUPS=2 (opt-in, process-group exchange): rows + the weight gradients of the decoder-side tensors, their part of
                # the flat gradient into the exchange -- asynchronously -- then the encoder's weight gradients on the launch stream beside it,
                # then that part; the optimizer launch waits for both.  UNMEASURED on multi-GPU hardware (exposed-latency model).
                ev = None
                if self._ar_events is not None:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                works = []
                for grp in (0, 1):
                    N.check(self.lib.dvae_train_grads_group(plan, N.ptr(self._params), N.ptr(self.ws), N.ptr(x), N.ld(x), yp, ldy, N.ptr(eps_noise),
                                                            self.elbo_eps, grp, 1, s), "dvae_train_grads_group")
                    if grp == 0 and ev is not None:
                        ev[0].record()
                    lo, hi = self._group_range[grp]
                    works.append(dp.allreduce_flat_async_(self.flat_grad[lo:hi], self.pg))
                for w in works:
                    w.wait()
                if ev is not None:
                    ev[1].record()
                    self._ar_events.append(ev)
                N.check(self.lib.dvae_train_apply(plan, N.ptr(self._params), N.ptr(self._m), N.ptr(self._v), N.ptr(self.ws), 1, self.step_count,
                                                  self.lr, self.betas[0], self.betas[1], self.adam_eps, 1.0 / self.world,
                                                  N.ptr(self.losses), s), "dvae_train_apply")
                self._reduced = True
                return self.losses
            N.check(self.lib.dvae_train_grads(plan, N.ptr(self._params), N.ptr(self.ws), N.ptr(x), N.ld(x), yp, ldy, N.ptr(eps_noise),
                                              self.elbo_eps, 0 if direct is not None else 1, s), "dvae_train_grads")
            ev = None
            if self._ar_events is not None:                           # profiling: device time of the exchange, on the launch stream
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if direct is not None:                                    # slab sum + reduce-scatter + all-gather in one launch on this stream
                direct.allreduce(self.flat_grad, self._used_slabs(), self.plan.n_params, self.flat_grad)
            else:
                dp.allreduce_flat_(self.flat_grad, self.pg)          # RCCL over xGMI: one flat fp32 buffer per step
            if ev is not None:
                ev[1].record()
                self._ar_events.append(ev)
            N.check(self.lib.dvae_train_apply(plan, N.ptr(self._params), N.ptr(self._m), N.ptr(self._v), N.ptr(self.ws), 1, self.step_count,
                                              self.lr, self.betas[0], self.betas[1], self.adam_eps, 1.0 / self.world,
                                              N.ptr(self.losses), s), "dvae_train_apply")
            self._reduced = True
        return self.losses

    def evaluate(self, x, y=None, eps_noise=None, rows=None):
        """Validation pass (scripts/training_M2.py:176-193): forward + elbo on a batch of the trainer's size, no
        backward, no update.  Returns a NEW device tensor [ELBO, recon, KL] (M2_info: 8 entries)."""
        self._check_inputs(x, y, rows, eps_noise)
        x = x if x.stride(1) == 1 else x.contiguous()
        if self.y_dim:
            y = y if y.stride(1) == 1 else y.contiguous()
        self._sync_copies()
        if eps_noise is None:               # drawn in the kernel, from a counter range the training steps never reach
            self._shared["eval_count"] = self._shared.get("eval_count", 0) + 1
            self.plan.rng_step = (1 << 40) + self._shared["eval_count"]
        yp, ldy = (N.ptr(y), N.ld(y)) if self.y_dim else (None, 0)
        out = torch.zeros_like(self.losses)
        with torch.cuda.device(self.device):
            N.check(self.lib.dvae_train_eval(ctypes.byref(self.plan), N.ptr(self._params), N.ptr(self.ws), N.ptr(x), N.ld(x), yp, ldy,
                                             N.ptr(None if eps_noise is None else eps_noise.contiguous()), self.elbo_eps, N.ptr(out), N.stream()),
                    "dvae_train_eval")
        return out

    def accumulate_losses(self, buf):
        """Running sums for epoch logging without a host sync per step: `buf` is a float64 CUDA tensor of 8 elements
        (or None to stop); every step() / evaluate() adds its loss scalars to it on the device."""
        if buf is not None and not (buf.is_cuda and buf.dtype == torch.float64 and buf.numel() >= 8 and buf.is_contiguous()):
            raise ValueError("accumulate_losses: need a contiguous float64 CUDA tensor with >= 8 elements")
        self._accum = buf
        self.plan.loss_accum = 0 if buf is None else buf.data_ptr()

    def noise(self, step):
        """The [B, 16] reparametrisation noise the rows kernel draws for training step `step` (1-based) when no
        eps_noise is passed: Philox4x32-10 keyed by the trainer's seed (and rank), counter (frame, step)."""
        out = torch.empty((self.B, 16), dtype=torch.float32, device=self.device)
        N.check(self.lib.dvae_train_noise(ctypes.byref(self.plan), int(step), N.ptr(out), N.stream()), "dvae_train_noise")
        return out

    def grads_only(self, x, y, eps_noise, reduce=False):
        """rows + wgrad kernels without the optimiser (tests / gradient inspection)."""
        self._check_inputs(x, y, None, eps_noise)
        self._sync_copies()
        x = x if x.stride(1) == 1 else x.contiguous()
        if self.y_dim:
            y = y if y.stride(1) == 1 else y.contiguous()
        yp, ldy = (N.ptr(y), N.ld(y)) if self.y_dim else (None, 0)
        with torch.cuda.device(self.device):
            N.check(self.lib.dvae_train_grads(ctypes.byref(self.plan), N.ptr(self._params), N.ptr(self.ws), N.ptr(x), N.ld(x), yp, ldy,
                                              N.ptr(eps_noise.contiguous()), self.elbo_eps, 1 if reduce else 0, N.stream()), "dvae_train_grads")
        self._reduced = bool(reduce)

    # ---- per-kernel device time (hipEvents on the launch stream) ----
    _ar_events = None

    def profile(self, enable):
        self.lib.dvae_train_profile(1 if enable else 0)
        self._ar_events = [] if (enable and self.world > 1) else None

    def allreduce_us(self):
        """Mean device time (us) of the gradient all-reduce over the steps made since profile(True); None at world 1."""
        if not self._ar_events:
            return None
        torch.cuda.synchronize()
        return 1e3 * sum(a.elapsed_time(b) for a, b in self._ar_events) / len(self._ar_events)

    def profile_read(self):
        ms = (ctypes.c_double * 4)()
        calls = (ctypes.c_int64 * 4)()
        N.check(self.lib.dvae_train_profile_read(ctypes.byref(ms), ctypes.byref(calls)), "dvae_train_profile_read")
        names = ["rows", "wgrad", "reduce", "apply"]
        return {n: (ms[i], calls[i]) for i, n in enumerate(names)}


class BenchImpl:
    """bench.py adapter: the fused path."""

    def __init__(self, model, dims, B, device, world, precision, ksplit=0, direct_exchange=None):
        pg = None
        if world > 1:
            import torch.distributed as dist
            pg = dist.group.WORLD
        self.tr = Trainer(model, dims, None, batch=B, device=device, precision=precision, process_group=pg, world=world, seed=0,
                          ksplit=ksplit, direct_exchange=direct_exchange)
        self.dtype = {"bf16": "bf16", "bf16x3": "bf16x3", "fp32": "f32"}[precision]
        self.name = (f"fused(rows+wgrad HIP kernels, Adam deferred into the next step's rows kernel, {precision} MFMA operands, fp32 accumulate/master)"
                     if self.tr._defer else f"fused(rows+wgrad+apply HIP kernels, {precision} MFMA operands, fp32 accumulate/master)")
        self.model, self.dims, self.B, self.precision = model, dims, B, precision

    def step(self, x, y, e):
        return self.tr.step(x, y, e)

    def finish(self):
        """The last step's optimizer update (deferred into the next step's first kernel when there is one): applied now.  bench.py calls
        this INSIDE its timed region, so K timed steps contain K updates."""
        self.tr.flush()

    def kernel_profile(self, batches, steps):
        """Per-kernel device time over `steps` steps -> roofline dict for the dominant kernel."""
        tr = self.tr
        torch.cuda.synchronize()
        tr.profile(True)
        for i in range(steps):
            tr.step(*batches[i % len(batches)])
        prof = tr.profile_read()
        ar_us = tr.allreduce_us()
        tr.profile(False)
        # us per launch from an event pair around each launch: dispatch latency + kernel + completion signal.  rocprofv3's kernel durations
        # of the same configuration (profiles/r03_*_kernel_stats.csv) are 1.5 - 2.6 us shorter per kernel; an EMPTY event pair on this stream
        # reads 5 - 7 us, so it cannot serve as the correction -- the figures are left gross (roofline fractions understated accordingly)
        avg = {k: (ms / c * 1e3 if c else 0.0) for k, (ms, c) in prof.items()}
        dom = max(avg, key=lambda k: avg[k])
        plan = tr.plan
        B = plan.B
        y = plan.y_dim
        ye = y if self.model == "M2" else 0
        mac = 128 * (513 + ye) + 128 * 128 + 2 * 16 * 128 + 128 * (16 + y) + 128 * 128 + 513 * 128
        dxm = 128 * 128 + 2 * 16 * 128 + 16 * 128 + 128 * 128 + 513 * 128
        flops = {"rows": 2.0 * (mac + dxm) * B, "wgrad": 2.0 * mac * B, "apply": 0.0, "reduce": 0.0}
        # algorithmic bytes: what an ideally fused step must move: x, y, eps once for the
        # rows kernel; the wgrad operands (activations + their gradients) for the split design
        esz = {"bf16": 2, "bf16x3": 4, "fp32": 4}[self.precision]
        stash_rows = 513 + y + 6 * 128 + 32 + 16 + 128 * 2 + 513
        byts = {"rows": 4.0 * (513 + y + 16) * B, "wgrad": float(esz * stash_rows * B), "apply": 16.0 * plan.n_params, "reduce": 0.0}
        dur = avg[dom] * 1e-6
        peak_f = 2500.0 if self.precision in ("bf16", "bf16x3") else 157.3
        # which roofline binds: the time the ISSUED matrix work needs (bf16x3: three MFMAs per product -- the rows kernel's 6.65 GFLOP are
        # 8.0 us of matrix pipe, more than the 4.3 us its algorithmic bytes need at 8 TB/s) against the HBM time.  `achieved` / `frac` stay
        # ALGORITHMIC flops (or bytes) over the dense peak, as the bench contract asks; `mfma_frac_issued` is the share of executed MFMA work
        n_mfma = 3.0 if self.precision == "bf16x3" else 1.0
        t_m = n_mfma * flops[dom] / (peak_f * 1e12)
        t_h = byts[dom] / 8.0e12
        if t_m >= t_h:
            bound, ach, peak, unit = "mfma", flops[dom] / dur / 1e12, peak_f, "TFLOP/s"
        else:
            bound, ach, peak, unit = "hbm", byts[dom] / dur / 1e9, 8000.0, "GB/s"
        # HBM bytes per launch of the dominant kernel: NOT measured in this run -- an offline PMC figure (separate rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the gfx950 note; tools/pmc_traffic.py) committed under profiles/, used only
        # when it was taken on this exact configuration
        rows_name = "vae_rows2_kernel" if tr.plan.rows_kernel == 2 else "vae_rows_kernel"
        knames = {"rows": rows_name, "wgrad": "wgrad_kernel", "apply": "apply_kernel", "reduce": "slab_reduce_kernel"}
        traffic, traffic_source = None, None
        try:
            import json, os
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            rel = next(r for r in (os.path.join("profiles", f"traffic_r0{n}.json") for n in (5, 4, 3)) if os.path.exists(os.path.join(root, r)))
            tj = json.load(open(os.path.join(root, rel)))
            cfg = tj.get("config", {})
            if (cfg.get("model"), cfg.get("y_dim"), cfg.get("batch"), cfg.get("precision")) == (self.model, y, B, self.precision) and knames[dom] in tj["kernels"]:
                traffic = tj["kernels"][knames[dom]]["hbm_bytes_per_launch"]
                traffic_source = rel + " (offline rocprofv3 PMC passes, not this run)"
        except Exception:
            traffic, traffic_source = None, None
        return {"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": knames[dom], "allreduce_us": ar_us,
                "avg_us": avg,
                "algorithmic_flops_per_launch": flops[dom], "algorithmic_bytes_per_launch": byts[dom],
                "mfma_frac": flops[dom] / dur / 1e12 / peak_f, "hbm_frac": byts[dom] / dur / 8.0e12,
                # bf16x3 issues three MFMAs per product (two where an operand has no lo plane): against the rate of ISSUED matrix work the
                # ceiling is 2500 / 3 TFLOP/s of algorithmic work
                "mfma_frac_issued": flops[dom] / dur / 1e12 / (peak_f / 3.0) if self.precision == "bf16x3" else flops[dom] / dur / 1e12 / peak_f,
                "timing": "hipEventElapsedTime around each launch on the launch stream (includes dispatch latency and the completion signal: "
                          "1.5 - 2.6 us more per kernel than rocprofv3's kernel durations of the same run, profiles/), mean over %d steps" % steps}
