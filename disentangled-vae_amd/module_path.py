"""Whole-model autograd path of the drop-in modules.

`packages.models.models.VariationalAutoencoder.forward` / `DeepGenerativeModel.forward` on CUDA tensors at the reference
geometry (x 513, h [128, 128], z 16, y 0 / 1 / 513) run as ONE autograd Function instead of one Function per nn.Linear, and so
does the VAE body of M2_info -- `DeepGenerativeModel_v3.forward` / `DeepGenerativeModel_v5.forward` (encoder on x alone, decoder on
[z | y], y 1; "M2_DEC"): the script's `model(x, y)` call (scripts/training_M2_info_vad.py:161) with z an output that the
auxiliary classifier's loss differentiates through.  Its classifier / auxiliary MLP calls stay per-layer Functions:

    forward : dvae_module_forward  = weight-copy refresh + the 8-wave rows kernel in forward mode   (2 launches)
    backward: dvae_module_backward = rows kernel in backward mode (forward recomputed on chip, then the backward from the
              upstream gradients of r, z, mu, log_var) + weight-gradient kernel + slab sum             (3 launches)

The training scripts keep `loss.backward()` and their own stock `torch.optim.Adam` (scripts/training_M2.py:122, 142-147).
To make that cheap the engine keeps the module's 14 parameters as views of ONE flat fp32 buffer in the fused kernels'
plan layout (they stay ordinary nn.Parameters: state_dict, load_state_dict, optimizers see nothing new).  The Function's
backward RETURNS the 14 parameter gradients -- views of one flat buffer allocated per backward pass -- so everything
autograd does with a gradient stays autograd's: AccumulateGrad takes the views as `.grad` without a copy when there is
none yet (the scripts' zero_grad() sets them to None every step) and adds otherwise, tensor hooks and post-accumulate
hooks run, `torch.autograd.grad(loss, params)` works.  The forward is recomputed on chip in the backward from the CURRENT
parameters, so a parameter changed in place between forward and backward raises, like autograd's version check does.

Only grad-mode forwards take this path: inference (no_grad, or every parameter frozen -- scripts/reconstruct_M2.py:111-113)
runs the per-layer Functions of ops.py on the exact fp32 matrix cores, like direct `model.encoder(...)` /
`model.decoder(...)` calls (packages/models/mcem.py) do, so the two agree to fp32 rounding and no per-batch-size
workspace is ever built for a forward-only call.  DVAE_MODULE_PATH=layers takes the per-layer path everywhere.

MFMA operand policy: DVAE_MODULE_PRECISION (default bf16x3, the parity-grade split-bf16 policy; bf16 = fast and loose).
"""
import ctypes
import os

import torch

from . import native as N
from .trainer import MODEL_CODE, PREC_CODE, TENSOR_NAMES, TrainPlan


def enabled():
    return os.environ.get("DVAE_MODULE_PATH", "fused") != "layers"


def _precision():
    p = os.environ.get("DVAE_MODULE_PRECISION", "bf16x3")
    if p not in ("bf16x3", "bf16"):
        raise ValueError("DVAE_MODULE_PRECISION must be bf16x3 or bf16 (the fused module path has no fp32-MFMA kernel; use DVAE_MODULE_PATH=layers)")
    return p


def vae_parameters(module, model):
    """The 14 (name, parameter) pairs of the VAE in plan order.  M2_DEC (a _v3 module): encoder + decoder, not the classifier."""
    if model == "M2_DEC":
        return list(module.encoder.named_parameters(prefix="encoder")) + list(module.decoder.named_parameters(prefix="decoder"))
    return list(module.named_parameters())


class ModuleEngine:
    """Fused kernels bound to one module instance (M1, M2, or the encoder + decoder of a _v3: M2_DEC)."""

    def __init__(self, module, model, y_dim):
        self.lib = N.load()
        self.model, self.y_dim = model, int(y_dim)
        self.precision = _precision()
        named = vae_parameters(module, model)
        sd_names = [n for n, _ in named]
        if sd_names != TENSOR_NAMES:
            raise RuntimeError(f"unexpected parameter set for the fused module path: {sd_names}")
        self.params = [p for _, p in named]
        # where the module keeps each of them (submodule._parameters[name]): `current()` checks the 14 slots instead of walking
        # named_parameters() on every forward (35 us of a 400 us step at the scripts' batch size)
        mods = dict(module.named_modules())
        self.slots = []
        for n, _ in named:
            owner, _, leaf = n.rpartition(".")
            self.slots.append((mods[owner]._parameters, leaf))
        self.device = self.params[0].device
        self.plans = {}
        p0 = self._plan(128)[0]
        self.n_params = int(p0.n_params)
        self.spans = [(int(p0.tensor_offset[i]), int(p0.tensor_rows[i]) * int(p0.tensor_cols[i])) for i in range(len(self.params))]
        with torch.cuda.device(self.device):
            self.flat = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.pptrs = [self.flat.data_ptr() + 4 * o for o, _ in self.spans]
        self._alias()

    def __deepcopy__(self, memo):          # a copied module rebuilds its own engine on first use
        return None

    # ---- plans (one per batch size; each holds a training workspace: stash, gradient slabs, weight copies).  At most
    # MAX_PLANS are kept, least recently used first out (a training loop sees its batch size and the last, shorter batch)
    MAX_PLANS = 4

    def _plan(self, B):
        got = self.plans.pop(B, None)
        if got is not None:
            self.plans[B] = got                    # most recently used last
        if got is None:
            while len(self.plans) >= self.MAX_PLANS:
                self.plans.pop(next(iter(self.plans)))
            plan = TrainPlan()
            N.check(self.lib.dvae_train_plan(MODEL_CODE[self.model], self.y_dim, PREC_CODE[self.precision], int(B), 0, ctypes.byref(plan)),
                    "dvae_train_plan")
            if plan.rows_kernel != 2:
                raise RuntimeError("fused module path needs the 8-wave rows kernel (unset DVAE_ROWS)")
            ws = None
            got = [plan, ws, False]
            self.plans[B] = got
        return got

    def _ready(self, B):
        ent = self._plan(B)
        if not ent[2]:
            with torch.cuda.device(self.device):
                ent[1] = torch.empty(ent[0].workspace_bytes, dtype=torch.uint8, device=self.device)
                N.check(self.lib.dvae_train_init(ctypes.byref(ent[0]), N.ptr(self.flat), N.ptr(ent[1]), N.stream()), "dvae_train_init")
            ent[2] = True
        return ent[0], ent[1]

    # ---- parameters as views of the flat buffer ----
    def _alias(self):
        for p, (o, n), want in zip(self.params, self.spans, self.pptrs):
            if p.data_ptr() != want:
                view = self.flat[o:o + n].view(p.shape)
                with torch.no_grad():
                    view.copy_(p.detach().to(device=self.device, dtype=torch.float32))
                p.data = view

    def usable(self, x):
        return all(p.is_cuda and p.device == x.device and p.dtype == torch.float32 for p in self.params)

    def current(self):
        """The module still holds the parameter objects this engine was built on (a replaced nn.Parameter rebuilds the engine)."""
        for (d, leaf), p in zip(self.slots, self.params):
            if d.get(leaf) is not p:
                return False
        return True

    # ---- launches ----
    def forward(self, x, y, eps):
        B = x.shape[0]
        if self.params[0].device != self.device:
            raise RuntimeError("module moved to another device after the fused engine was built")
        self._alias()
        plan, ws = self._ready(B)
        with torch.cuda.device(self.device):
            r = torch.empty((B, 513), dtype=torch.float32, device=self.device)
            mlz = torch.empty((3, B, 16), dtype=torch.float32, device=self.device)
            yp, ldy = (N.ptr(y), N.ld(y)) if self.y_dim else (None, 0)
            m0 = mlz.data_ptr()                      # mu | log_var | z, [B, 16] fp32 each
            N.check(self.lib.dvae_module_forward(ctypes.byref(plan), N.ptr(self.flat), N.ptr(ws), N.ptr(x), N.ld(x), yp, ldy, N.ptr(eps),
                                                 N.ptr(r), 513, m0, m0 + 64 * B, m0 + 128 * B, 1, N.stream()), "dvae_module_forward")
        mu, lv, z = mlz.unbind(0)
        return r, z, mu, lv

    def backward(self, x, y, eps, gr, gz, gmu, glv, needs):
        """-> list of 14 parameter gradients (None where `needs` is False): views of one flat buffer of this call."""
        B = x.shape[0]
        plan, ws = self._ready(B)
        f32c = lambda t: None if t is None else _rows_ok(t.to(torch.float32))
        gr_, gz, gmu, glv = f32c(gr), f32c(gz), f32c(gmu), f32c(glv)
        if gz is not None: gz = gz.contiguous()
        if gmu is not None: gmu = gmu.contiguous()
        if glv is not None: glv = glv.contiguous()
        with torch.cuda.device(self.device):
            dst = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
            yp, ldy = (N.ptr(y), N.ld(y)) if self.y_dim else (None, 0)
            N.check(self.lib.dvae_module_backward(ctypes.byref(plan), N.ptr(self.flat), N.ptr(ws), N.ptr(x), N.ld(x), yp, ldy, N.ptr(eps),
                                                  N.ptr(gr_), 0 if gr_ is None else N.ld(gr_), N.ptr(gmu), N.ptr(glv), N.ptr(gz),
                                                  N.ptr(dst), 0, N.stream()), "dvae_module_backward")
        # one view op per gradient (as_strided on the flat buffer: contiguous [out, in] / [out] at the tensor's offset)
        return [dst.as_strided(shp, std, o) if need else None for (shp, std, o), need in zip(self._grad_views(), needs)]

    def _grad_views(self):
        gv = self.__dict__.get("_gv")
        if gv is None:
            gv = self._gv = [(tuple(p.shape), (p.shape[1], 1) if p.dim() == 2 else (1,), o) for p, (o, n) in zip(self.params, self.spans)]
        return gv


def _rows_ok(t):
    """A [B, C] operand the kernels can address: unit column stride and a row stride that covers the row (an expanded /
    broadcast tensor -- the gradient of r.sum(0), an expanded input row -- has stride 0 and is materialised)."""
    if t.dim() != 2 or (t.stride(1) == 1 and t.stride(0) >= t.shape[1]):
        return t
    return t.contiguous()


class VaeFunction(torch.autograd.Function):
    """(r, z, mu, log_var) = model(x, y) with reparametrisation noise eps; the parameters are inputs of the Function and
    backward returns their gradients (module docstring)."""

    @staticmethod
    def forward(ctx, engine, x, y, eps, *params):
        r, z, mu, lv = engine.forward(x, y, eps)
        ctx.engine = engine
        ctx.has_y = y is not None
        ctx.versions = [p._version for p in params]
        ctx.save_for_backward(x, y if y is not None else x.new_empty(0), eps)
        ctx.set_materialize_grads(False)
        return r, z, mu, lv

    @staticmethod
    def backward(ctx, gr, gz, gmu, glv):
        x, y, eps = ctx.saved_tensors
        eng = ctx.engine
        # the backward kernel recomputes the forward from the flat parameter buffer as it is NOW
        for p, v0, name in zip(eng.params, ctx.versions, TENSOR_NAMES):
            if p._version != v0:
                raise RuntimeError(f"one of the variables needed for gradient computation has been modified by an inplace operation: parameter "
                                   f"'{name}' is at version {p._version}, the forward saw version {v0} (an optimizer step or load_state_dict "
                                   f"between forward and backward); run the forward again")
        if any(p.data_ptr() != want for p, want in zip(eng.params, eng.pptrs)):
            raise RuntimeError("a parameter's storage was replaced between forward and backward (module.to() / .data assignment): run the forward again")
        grads = eng.backward(x, y if ctx.has_y else None, eps, gr, gz, gmu, glv, ctx.needs_input_grad[4:])
        return (None, None, None, None, *grads)


def engine_for(module, model, x, y):
    """The module's engine when this call can take the fused path, else None.  model: "M1" | "M2" | "M2_DEC" (module: a _v3)."""
    if not (enabled() and x.is_cuda and x.dim() == 2 and x.shape[1] == 513 and x.dtype == torch.float32 and x.shape[0] > 0):
        return None
    if x.requires_grad or (y is not None and y.requires_grad):
        return None                                  # gradients with respect to the data are a layer-path feature
    if not torch.is_grad_enabled():
        return None                                  # inference: the exact-fp32 per-layer kernels (module docstring)
    eng = module.__dict__.get("_dvae_engine")
    if eng is not None and not eng.current():
        eng = None                                   # a parameter object was replaced: a new engine on the new set
        module.__dict__.pop("_dvae_engine", None)
    if not any(p.requires_grad for p in (eng.params if eng is not None else [p for _, p in vae_parameters(module, model)])):
        return None                                  # every parameter frozen: inference as well
    if eng is None:
        if module.__dict__.get("_dvae_engine_off"):
            return None
        ok = (module.z_dim == 16 and module.flow is None
              and [tuple(l.weight.shape) for l in module.encoder.hidden] == [(128, 513 + (module.y_dim if model == "M2" else 0)), (128, 128)]
              and [tuple(l.weight.shape) for l in module.decoder.hidden] == [(128, 16 + (module.y_dim if model != "M1" else 0)), (128, 128)]
              and tuple(module.decoder.reconstruction.weight.shape) == (513, 128)
              and (model == "M1" or module.y_dim in ((1, 513) if model == "M2" else (1,)))
              and all(p.is_cuda and p.dtype == torch.float32 for _, p in vae_parameters(module, model))
              and [n for n, _ in vae_parameters(module, model)] == TENSOR_NAMES)
        if not ok:
            object.__setattr__(module, "_dvae_engine_off", True)
            return None
        eng = ModuleEngine(module, model, module.y_dim if model != "M1" else 0)
        object.__setattr__(module, "_dvae_engine", eng)
    if not eng.usable(x):
        return None
    if model != "M1" and (y is None or y.dim() != 2 or y.shape != (x.shape[0], eng.y_dim) or y.dtype != torch.float32 or y.device != x.device):
        return None
    return eng


def run(eng, x, y, eps):
    """-> (r, z, mu, log_var).  eps: [B, 16] float32 on x's device."""
    x = _rows_ok(x)
    if y is not None:
        y = _rows_ok(y)
    eps = eps.to(device=x.device, dtype=torch.float32).contiguous()
    if eps.shape != (x.shape[0], 16):
        raise RuntimeError(f"reparametrisation noise must be [{x.shape[0]}, 16], got {tuple(eps.shape)}")
    return VaeFunction.apply(eng, x, y, eps, *eng.params)
