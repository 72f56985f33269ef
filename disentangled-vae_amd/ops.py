"""torch.autograd.Function shells over the layer-level HIP kernels.

These are what packages/models/models.py and packages/models/utils.py call
for CUDA tensors, so the reference's scripts keep using `loss.backward()` and
stock `torch.optim.Adam` (SURVEY.md 8b "nn.Module protocol").  Every Function
fails loudly when the library is missing; there is no eager fallback here.
"""
import torch

from . import native as N


def _new(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def linear_act_fwd(x0, x1, W, b, act):
    lib = N.load()
    x0_ = N.as_f32_2d(x0, "linear_act: input")
    x1_ = None if x1 is None else N.as_f32_2d(x1, "linear_act: second input")
    W_ = N.as_f32_2d(W, "linear_act: weight")
    B, k0 = x0_.shape
    k1 = 0 if x1_ is None else x1_.shape[1]
    if x1_ is not None and x1_.shape[0] != B:
        raise RuntimeError("linear_act: the two inputs differ in row count")
    Nout = W_.shape[0]
    if W_.shape[1] != k0 + k1:
        raise RuntimeError(f"linear_act: weight fan-in {W_.shape[1]} != {k0}+{k1}")
    if b is not None:
        if b.dtype != torch.float32 or not b.is_cuda:
            raise TypeError("linear_act: bias must be a float32 CUDA tensor")
        b = b.contiguous()
    out = _new((B, Nout), x0_)
    N.check(lib.dvae_linear_act_fwd(N.ptr(x0_), k0, N.ld(x0_), N.ptr(x1_), k1, 0 if x1_ is None else N.ld(x1_),
                                    N.ptr(W_), N.ld(W_), N.ptr(b), N.ptr(out), Nout, B, Nout, act, N.stream()),
            "dvae_linear_act_fwd")
    return out, x0_, x1_, W_


class LinearAct(torch.autograd.Function):
    """out = act([x0 | x1] @ W.T + b)   (models.py:57-63, 102-105, 119-122, 201-202)."""

    @staticmethod
    def forward(ctx, x0, x1, W, b, act):
        out, x0_, x1_, W_ = linear_act_fwd(x0, x1, W, b, act)
        ctx.act = act
        ctx.has_bias = b is not None
        ctx.lead0 = x0.shape[:-1]
        ctx.lead1 = None if x1 is None else x1.shape[:-1]
        ctx.save_for_backward(x0_, x1_ if x1_ is not None else torch.empty(0), W_, out)
        ctx.has_x1 = x1_ is not None
        return out.reshape(*x0.shape[:-1], W_.shape[0])

    @staticmethod
    def backward(ctx, dout):
        lib = N.load()
        x0_, x1_, W_, out = ctx.saved_tensors
        if not ctx.has_x1:
            x1_ = None
        B, Nout = out.shape
        k0 = x0_.shape[1]
        k1 = 0 if x1_ is None else x1_.shape[1]
        dout_ = N.as_f32_2d(dout, "linear_act backward: grad")
        s = N.stream()
        if ctx.act == N.ACT_NONE:
            dpre = dout_
        else:
            dpre = _new((B, Nout), out)
            N.check(lib.dvae_act_bwd(N.ptr(dout_), N.ld(dout_), N.ptr(out), Nout, N.ptr(dpre), Nout, B, Nout, ctx.act, s),
                    "dvae_act_bwd")
        dx0 = dx1 = dW = db = None
        if ctx.needs_input_grad[0]:
            dx0 = _new((B, k0), out)
            N.check(lib.dvae_linear_bwd_data(N.ptr(dpre), N.ld(dpre), N.ptr(W_), N.ld(W_), 0, N.ptr(dx0), k0, B, Nout, k0, 0, s),
                    "dvae_linear_bwd_data")
            dx0 = dx0.reshape(*ctx.lead0, k0)
        if x1_ is not None and ctx.needs_input_grad[1]:
            dx1 = _new((B, k1), out)
            N.check(lib.dvae_linear_bwd_data(N.ptr(dpre), N.ld(dpre), N.ptr(W_), N.ld(W_), k0, N.ptr(dx1), k1, B, Nout, k1, 0, s),
                    "dvae_linear_bwd_data")
            dx1 = dx1.reshape(*ctx.lead1, k1)
        need_w = ctx.needs_input_grad[2]
        need_b = ctx.has_bias and ctx.needs_input_grad[3]
        if need_w or need_b:
            dW = _new((Nout, k0 + k1), out)
            db = _new((Nout,), out) if ctx.has_bias else None
            # deterministic slice combination (partials + an ordered sum) -- the reference's GPU GEMMs return the same bits for the same inputs
            nws = lib.dvae_linear_bwd_weight_workspace_bytes(B, Nout, k0 + k1, 0)
            ws = torch.empty(nws, dtype=torch.uint8, device=out.device) if nws else None
            N.check(lib.dvae_linear_bwd_weight_det(N.ptr(dpre), N.ld(dpre), N.ptr(x0_), k0, N.ld(x0_), N.ptr(x1_), k1,
                                                   0 if x1_ is None else N.ld(x1_), N.ptr(dW), k0 + k1, N.ptr(db), B, Nout, 0, N.ptr(ws), s),
                    "dvae_linear_bwd_weight_det")
            if not need_w:
                dW = None
            if not need_b:
                db = None
        return dx0, dx1, dW, db, None


def linear_act(x0, W, b, act, x1=None):
    return LinearAct.apply(x0, x1, W, b, act)


class LinearStack(torch.autograd.Function):
    """act_L(... act_1([x0 | x1] @ W1.T + b1) ...) @ ... as ONE autograd node: the kernels LinearAct launches per layer, in the same order,
    with one trip through the interpreter per direction instead of one per layer (models.py:57-63: Classifier; 102-105, 119-122: the tanh
    stacks of Encoder / Decoder).  The scripts' M2_info step runs three such stacks forward and four times backward; a Python-level
    Function costs 20 - 30 us of host time per direction, and that loop is host-bound (tools/r05/prof_modules.py)."""

    @staticmethod
    def forward(ctx, x0, x1, acts, *wb):
        saved, h, h1 = [], x0, x1
        for i, act in enumerate(acts):
            out, x0_, x1_, W_ = linear_act_fwd(h, h1, wb[2 * i], wb[2 * i + 1], act)
            saved += [x0_, W_, out]
            if i == 0:
                saved.append(x1_ if x1_ is not None else torch.empty(0))
            h, h1 = out, None
        ctx.acts = tuple(acts)
        ctx.has_x1 = x1 is not None
        ctx.has_bias = tuple(wb[2 * i + 1] is not None for i in range(len(acts)))
        ctx.lead0 = x0.shape[:-1]
        ctx.lead1 = None if x1 is None else x1.shape[:-1]
        ctx.save_for_backward(*saved)
        return h.reshape(*x0.shape[:-1], h.shape[1])

    @staticmethod
    def backward(ctx, dout):
        lib = N.load()
        sv = ctx.saved_tensors
        L = len(ctx.acts)
        # layer 0 saved (x0, W, out, x1); layer i > 0 saved (x0, W, out)
        def layer(i):
            o = 0 if i == 0 else 4 + 3 * (i - 1)
            return sv[o], sv[o + 1], sv[o + 2]
        x1_ = sv[3] if ctx.has_x1 else None
        s = N.stream()
        grads = [None] * (2 * L)
        d = N.as_f32_2d(dout, "linear_stack backward: grad")
        dx0 = dx1 = None
        for i in range(L - 1, -1, -1):
            x0_, W_, out = layer(i)
            B, Nout = out.shape
            k0 = x0_.shape[1]
            xe = x1_ if i == 0 else None
            k1 = 0 if xe is None else xe.shape[1]
            act = ctx.acts[i]
            if act == N.ACT_NONE:
                dpre = d
            else:
                dpre = _new((B, Nout), out)
                N.check(lib.dvae_act_bwd(N.ptr(d), N.ld(d), N.ptr(out), Nout, N.ptr(dpre), Nout, B, Nout, act, s), "dvae_act_bwd")
            need_w = ctx.needs_input_grad[3 + 2 * i]
            need_b = ctx.has_bias[i] and ctx.needs_input_grad[4 + 2 * i]
            if need_w or need_b:
                dW = _new((Nout, k0 + k1), out)
                db = _new((Nout,), out) if ctx.has_bias[i] else None
                nws = lib.dvae_linear_bwd_weight_workspace_bytes(B, Nout, k0 + k1, 0)
                ws = torch.empty(nws, dtype=torch.uint8, device=out.device) if nws else None
                N.check(lib.dvae_linear_bwd_weight_det(N.ptr(dpre), N.ld(dpre), N.ptr(x0_), k0, N.ld(x0_), N.ptr(xe), k1,
                                                       0 if xe is None else N.ld(xe), N.ptr(dW), k0 + k1, N.ptr(db), B, Nout, 0, N.ptr(ws), s),
                        "dvae_linear_bwd_weight_det")
                grads[2 * i] = dW if need_w else None
                grads[2 * i + 1] = db if need_b else None
            if i > 0 or ctx.needs_input_grad[0]:
                dx = _new((B, k0), out)
                N.check(lib.dvae_linear_bwd_data(N.ptr(dpre), N.ld(dpre), N.ptr(W_), N.ld(W_), 0, N.ptr(dx), k0, B, Nout, k0, 0, s),
                        "dvae_linear_bwd_data")
                if i > 0:
                    d = dx
                else:
                    dx0 = dx.reshape(*ctx.lead0, k0)
            if i == 0 and xe is not None and ctx.needs_input_grad[1]:
                dx1 = _new((B, k1), out)
                N.check(lib.dvae_linear_bwd_data(N.ptr(dpre), N.ld(dpre), N.ptr(W_), N.ld(W_), k0, N.ptr(dx1), k1, B, Nout, k1, 0, s),
                        "dvae_linear_bwd_data")
                dx1 = dx1.reshape(*ctx.lead1, k1)
        return (dx0, dx1, None, *grads)


def linear_stack(x0, layers, acts, x1=None):
    """layers: [(weight, bias), ...]; acts: activation codes, one per layer."""
    wb = []
    for W, b in layers:
        wb += [W, b]
    return LinearStack.apply(x0, x1, tuple(acts), *wb)


class Reparam(torch.autograd.Function):
    """z = mu + exp(0.5 * log_var) * epsilon   (models.py:9-22)."""

    @staticmethod
    def forward(ctx, mu, log_var, eps):
        lib = N.load()
        mu_ = N.as_f32_2d(mu, "reparam: mu").contiguous()
        lv_ = N.as_f32_2d(log_var, "reparam: log_var").contiguous()
        eps_ = N.as_f32_2d(eps, "reparam: epsilon").contiguous()
        if not (mu_.shape == lv_.shape == eps_.shape):
            raise RuntimeError("reparam: shape mismatch")
        z = torch.empty_like(mu_)
        N.check(lib.dvae_reparam_fwd(N.ptr(mu_), N.ptr(lv_), N.ptr(eps_), N.ptr(z), mu_.numel(), N.stream()), "dvae_reparam_fwd")
        ctx.save_for_backward(lv_, eps_)
        return z.reshape(mu.shape)

    @staticmethod
    def backward(ctx, dz):
        lib = N.load()
        lv_, eps_ = ctx.saved_tensors
        dz_ = N.as_f32_2d(dz, "reparam backward").contiguous()
        dmu = torch.empty_like(lv_)
        dlv = torch.empty_like(lv_)
        N.check(lib.dvae_reparam_bwd(N.ptr(dz_), N.ptr(lv_), N.ptr(eps_), N.ptr(dmu), N.ptr(dlv), lv_.numel(), N.stream()),
                "dvae_reparam_bwd")
        return dmu.reshape(dz.shape), dlv.reshape(dz.shape), None


class Elbo(torch.autograd.Function):
    """(recon + KL, recon, KL) of packages/models/utils.py:73-76 as three 0-dim tensors (views of one [3] buffer).  The three
    upstream gradients go to the backward kernel as they are (device scalars or absent): no host-side tensor arithmetic."""

    @staticmethod
    def forward(ctx, x, r, mu, logvar, eps):
        lib = N.load()
        x_ = N.as_f32_2d(x, "elbo: x")
        r_ = N.as_f32_2d(r, "elbo: r")
        mu_ = N.as_f32_2d(mu, "elbo: mu").contiguous()
        lv_ = N.as_f32_2d(logvar, "elbo: logvar").contiguous()
        B, F = x_.shape
        Z = mu_.shape[1]
        if r_.shape != x_.shape or mu_.shape[0] != B or lv_.shape != mu_.shape:
            raise RuntimeError("elbo: shape mismatch")
        out3 = _new((3,), x_)
        ws = torch.empty(lib.dvae_elbo_workspace_bytes(B), dtype=torch.uint8, device=x_.device)
        N.check(lib.dvae_elbo_fwd(N.ptr(x_), N.ld(x_), N.ptr(r_), N.ld(r_), N.ptr(mu_), N.ptr(lv_), float(eps), B, F, Z,
                                  N.ptr(out3), None, N.ptr(ws), N.stream()), "dvae_elbo_fwd")
        ctx.save_for_backward(x_, r_, mu_, lv_)
        ctx.shapes = (r.shape, mu.shape, logvar.shape)
        ctx.set_materialize_grads(False)
        return out3.unbind(0)

    @staticmethod
    def backward(ctx, g_loss, g_recon, g_kl):
        lib = N.load()
        x_, r_, mu_, lv_ = ctx.saved_tensors
        B, F = x_.shape
        Z = mu_.shape[1]
        f32 = lambda g: None if g is None else (g if g.dtype == torch.float32 else g.to(torch.float32))
        g_loss, g_recon, g_kl = f32(g_loss), f32(g_recon), f32(g_kl)
        need_r, need_mu, need_lv = ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        dr = _new((B, F), x_) if need_r else None
        dmu = torch.empty_like(mu_) if need_mu else None
        dlv = torch.empty_like(lv_) if need_lv else None
        N.check(lib.dvae_elbo_bwd3(N.ptr(x_), N.ld(x_), N.ptr(r_), N.ld(r_), N.ptr(mu_), N.ptr(lv_), N.ptr(g_loss), N.ptr(g_recon), N.ptr(g_kl),
                                   B, F, Z, N.ptr(dr), F, N.ptr(dmu), N.ptr(dlv), N.stream()), "dvae_elbo_bwd3")
        rs, ms, ls = ctx.shapes
        return (None, None if dr is None else dr.reshape(rs), None if dmu is None else dmu.reshape(ms),
                None if dlv is None else dlv.reshape(ls), None)


class Bce(torch.autograd.Function):
    """binary_cross_entropy / _v2 / _v3 of packages/models/utils.py:55-63."""

    @staticmethod
    def forward(ctx, r, t, eps, variant):
        lib = N.load()
        r_ = N.as_f32_2d(r, "bce: r").contiguous()
        t_ = None
        if variant == 0:
            t_ = N.as_f32_2d(t, "bce: target").contiguous()
            if t_.shape != r_.shape:
                raise RuntimeError("bce: shape mismatch")
        B, Y = r_.shape
        out1 = _new((1,), r_)
        ws = torch.empty(lib.dvae_elbo_workspace_bytes(B), dtype=torch.uint8, device=r_.device)
        N.check(lib.dvae_bce_fwd(N.ptr(r_), N.ptr(t_), float(eps), B, Y, variant, N.ptr(out1), N.ptr(ws), N.stream()), "dvae_bce_fwd")
        ctx.save_for_backward(r_, t_ if t_ is not None else torch.empty(0))
        ctx.variant, ctx.eps = variant, float(eps)
        ctx.rshape = r.shape
        ctx.tshape = None if t is None else t.shape
        return out1.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = N.load()
        r_, t_ = ctx.saved_tensors
        if ctx.variant != 0:
            t_ = None
        B, Y = r_.shape
        g_ = g.to(torch.float32).reshape(1).contiguous()
        dr = torch.empty_like(r_)
        need_t = ctx.variant == 0 and ctx.needs_input_grad[1]
        dt = torch.empty_like(r_) if need_t else None
        N.check(lib.dvae_bce_bwd(N.ptr(r_), N.ptr(t_), ctx.eps, N.ptr(g_), B, Y, ctx.variant, N.ptr(dr), N.ptr(dt), N.stream()),
                "dvae_bce_bwd")
        return dr.reshape(ctx.rshape), (dt.reshape(ctx.tshape) if need_t else None), None, None


class IsRows(torch.autograd.Function):
    """Per-frame (recon, KL) rows of packages/models/utils.py:68-71, 78-81 (L_loss, ikatura_saito_divergence): [B] vectors.
    mu / logvar may be None (recon rows only)."""

    @staticmethod
    def forward(ctx, x, r, mu, logvar, eps):
        lib = N.load()
        x_ = N.as_f32_2d(x, "is rows: x")
        r_ = N.as_f32_2d(r, "is rows: r")
        if r_.shape != x_.shape:
            raise RuntimeError("is rows: shape mismatch")
        B, F = x_.shape
        has_kl = mu is not None
        mu_ = N.as_f32_2d(mu, "is rows: mu").contiguous() if has_kl else None
        lv_ = N.as_f32_2d(logvar, "is rows: logvar").contiguous() if has_kl else None
        Z = mu_.shape[1] if has_kl else 0
        rec = _new((B,), x_)
        kl = _new((B,), x_) if has_kl else None
        N.check(lib.dvae_isrows_fwd(N.ptr(x_), N.ld(x_), N.ptr(r_), N.ld(r_), N.ptr(mu_), N.ptr(lv_), float(eps), B, F, Z, N.ptr(rec), N.ptr(kl),
                                    N.stream()), "dvae_isrows_fwd")
        ctx.save_for_backward(x_, r_, *( (mu_, lv_) if has_kl else ()))
        ctx.has_kl = has_kl
        ctx.lead = x.shape[:-1]
        ctx.shapes = (r.shape, None if mu is None else mu.shape, None if logvar is None else logvar.shape)
        ctx.set_materialize_grads(False)
        if has_kl:
            return rec.reshape(ctx.lead), kl.reshape(ctx.lead)
        return rec.reshape(ctx.lead)

    @staticmethod
    def backward(ctx, g_rec, g_kl=None):
        lib = N.load()
        saved = ctx.saved_tensors
        x_, r_ = saved[0], saved[1]
        mu_, lv_ = (saved[2], saved[3]) if ctx.has_kl else (None, None)
        B, F = x_.shape
        Z = mu_.shape[1] if ctx.has_kl else 0
        c = lambda g: None if g is None else g.to(torch.float32).reshape(B).contiguous()
        g_rec, g_kl = c(g_rec), c(g_kl)
        need_r = ctx.needs_input_grad[1] and g_rec is not None
        need_mu = ctx.has_kl and ctx.needs_input_grad[2] and g_kl is not None
        need_lv = ctx.has_kl and ctx.needs_input_grad[3] and g_kl is not None
        dr = _new((B, F), x_) if need_r else None
        dmu = torch.empty_like(mu_) if need_mu else None
        dlv = torch.empty_like(lv_) if need_lv else None
        if need_r or need_mu or need_lv:
            N.check(lib.dvae_isrows_bwd(N.ptr(x_), N.ld(x_), N.ptr(r_), N.ld(r_), N.ptr(mu_), N.ptr(lv_), N.ptr(g_rec), N.ptr(g_kl), B, F, Z,
                                        N.ptr(dr), F, N.ptr(dmu), N.ptr(dlv), N.stream()), "dvae_isrows_bwd")
        rs, ms, ls = ctx.shapes
        return (None, None if dr is None else dr.reshape(rs), None if dmu is None else dmu.reshape(ms),
                None if dlv is None else dlv.reshape(ls), None)


class Bce2(torch.autograd.Function):
    """binary_cross_entropy_2classes of packages/models/utils.py:65-66."""

    @staticmethod
    def forward(ctx, r1, r2, t, eps):
        lib = N.load()
        a = N.as_f32_2d(r1, "bce2: r1").contiguous()
        b = N.as_f32_2d(r2, "bce2: r2").contiguous()
        t_ = N.as_f32_2d(t, "bce2: target").contiguous()
        if a.shape != b.shape or a.shape != t_.shape:
            raise RuntimeError("bce2: shape mismatch")
        B, Y = a.shape
        out1 = _new((1,), a)
        ws = torch.empty(lib.dvae_elbo_workspace_bytes(B), dtype=torch.uint8, device=a.device)
        N.check(lib.dvae_bce2_fwd(N.ptr(a), N.ptr(b), N.ptr(t_), float(eps), B, Y, N.ptr(out1), N.ptr(ws), N.stream()), "dvae_bce2_fwd")
        ctx.save_for_backward(a, b, t_)
        ctx.eps, ctx.shapes = float(eps), (r1.shape, r2.shape, t.shape)
        return out1.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = N.load()
        a, b, t_ = ctx.saved_tensors
        B, Y = a.shape
        g_ = g.to(torch.float32).reshape(1).contiguous()
        d1 = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        d2 = torch.empty_like(a) if ctx.needs_input_grad[1] else None
        dt = torch.empty_like(a) if ctx.needs_input_grad[2] else None
        N.check(lib.dvae_bce2_bwd(N.ptr(a), N.ptr(b), N.ptr(t_), ctx.eps, N.ptr(g_), B, Y, N.ptr(d1), N.ptr(d2), N.ptr(dt), N.stream()), "dvae_bce2_bwd")
        s1, s2, st = ctx.shapes
        return (None if d1 is None else d1.reshape(s1), None if d2 is None else d2.reshape(s2), None if dt is None else dt.reshape(st), None)


class SqErr(torch.autograd.Function):
    """mean_b sum_f |d|^2 of packages/models/utils.py:107-118.  mode 0: d = (y - yhat) x; 1: d = y - yhat; 2: d = s - yhat x with complex64
    x, s (= y) and a real mask yhat (gradient with respect to the mask only: x and s are STFT data)."""

    @staticmethod
    def forward(ctx, mode, x, y, yhat):
        lib = N.load()
        if mode == 2:
            if x.dtype != torch.complex64 or y.dtype != torch.complex64:
                raise TypeError("magnitude spectrum approximation: x and s must be complex64")
            xs = torch.view_as_real(x.reshape(-1, x.shape[-1]).contiguous())
            ys = torch.view_as_real(y.reshape(-1, y.shape[-1]).contiguous())
            B, F = xs.shape[0], xs.shape[1]
        else:
            ys = N.as_f32_2d(y, "squared error: y").contiguous()
            xs = N.as_f32_2d(x, "squared error: x").contiguous() if mode == 0 else None
            B, F = ys.shape
        h = N.as_f32_2d(yhat, "squared error: yhat").contiguous()
        if h.shape != (B, F) or (xs is not None and xs.shape[:2] != (B, F)):
            raise RuntimeError("squared error: shape mismatch")
        out1 = _new((1,), h)
        ws = torch.empty(lib.dvae_elbo_workspace_bytes(B), dtype=torch.uint8, device=h.device)
        N.check(lib.dvae_sqerr_fwd(mode, N.ptr(xs), N.ptr(ys), N.ptr(h), B, F, N.ptr(out1), N.ptr(ws), N.stream()), "dvae_sqerr_fwd")
        ctx.save_for_backward(*(t for t in (xs, ys, h) if t is not None))
        ctx.mode = mode
        ctx.shapes = (None if x is None else x.shape, y.shape, yhat.shape)
        return out1.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = N.load()
        mode = ctx.mode
        saved = ctx.saved_tensors
        xs, ys, h = (saved if mode != 1 else (None, saved[0], saved[1]))
        B, F = h.shape
        g_ = g.to(torch.float32).reshape(1).contiguous()
        need_x, need_y, need_h = ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        if mode == 2 and (need_x or need_y):
            raise NotImplementedError("magnitude_spectrum_approxiamation_loss: gradients with respect to the complex spectra are not provided "
                                      "(they are STFT data in every caller); detach them")
        dh = torch.empty_like(h) if need_h else None
        dy = torch.empty_like(h) if (need_y and mode != 2) else None
        dx = torch.empty_like(h) if (need_x and mode == 0) else None
        N.check(lib.dvae_sqerr_bwd(mode, N.ptr(xs), N.ptr(ys), N.ptr(h), N.ptr(g_), B, F, N.ptr(dh), N.ptr(dy), N.ptr(dx), N.stream()), "dvae_sqerr_bwd")
        sx, sy, sh = ctx.shapes
        return (None, None if dx is None else dx.reshape(sx), None if dy is None else dy.reshape(sy), None if dh is None else dh.reshape(sh))


def adam_step_(p, g, m, v, step, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
    """In-place torch.optim.Adam update on flat fp32 CUDA buffers (scripts/training_M2.py:122)."""
    lib = N.load()
    for t in (p, g, m, v):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise TypeError("adam_step_: contiguous float32 CUDA tensors required")
    N.check(lib.dvae_adam_step(N.ptr(p), N.ptr(g), N.ptr(m), N.ptr(v), p.numel(), lr, betas[0], betas[1], eps, step,
                               grad_scale, N.stream()), "dvae_adam_step")
