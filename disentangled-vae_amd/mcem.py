"""Device side of the MCEM enhancement loop: thin Python over the C ABI of include/dvae_mcem.h.

`DecoderPack` holds the kernel-layout copy of a VAE decoder; the free functions mirror the steps of the
reference's packages/models/mcem.py (sample_posterior, compute_Vs, M_step, compute_WF) on CUDA
tensors in the reference's own shapes.  No fallback: errors from the library raise.
"""
import ctypes, os

import numpy as np
import torch

from . import native as N

F_BINS, Z_DIM, H_DIM = 513, 16, 128
PREC = {"fp32": 0, "bf16": 1, "bf16x3": 2}      # bf16x3: split-bf16 operands (hi + lo planes, three MFMAs per product): parity grade


class McemPlan(ctypes.Structure):
    _fields_ = [("y_dim", ctypes.c_int32), ("precision", ctypes.c_int32), ("x_dim", ctypes.c_int32),
                ("z_dim", ctypes.c_int32), ("h_dim", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("weights_bytes", ctypes.c_int64)]


def _f32c(t, what):
    if t is None:
        raise RuntimeError(f"{what}: required (got None)")
    if not t.is_cuda:
        raise RuntimeError(f"{what}: expected a CUDA tensor")
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: the HIP path computes in float32, got {t.dtype}")
    return t.contiguous()


def decoder_supported(decoder, y_dim):
    """The kernels cover the geometry the reference's scripts use (decoder [16+y]-128-128-513)."""
    hs = list(decoder.hidden)
    return (len(hs) == 2 and hs[0].out_features == H_DIM and hs[1].in_features == H_DIM and hs[1].out_features == H_DIM
            and hs[0].in_features == Z_DIM + y_dim and decoder.reconstruction.out_features == F_BINS
            and (y_dim == 0 or 1 <= y_dim <= 16 or y_dim == F_BINS))


class DecoderPack:
    """Kernel-layout copy of `vae.decoder` (packages/models/models.py:108-122) for the MCEM kernels."""

    def __init__(self, decoder, y_dim, precision="fp32"):
        self.lib = N.load()
        if not decoder_supported(decoder, y_dim):
            raise RuntimeError("MCEM kernels: decoder geometry not supported (need [16+y_dim]-128-128-513, y_dim 0/1..16/513)")
        self.y_dim = y_dim
        self.plan = McemPlan()
        N.check(self.lib.dvae_mcem_plan(y_dim, PREC[precision], ctypes.byref(self.plan)), "dvae_mcem_plan")
        dev = decoder.reconstruction.weight.device
        self.weights = torch.empty(self.plan.weights_bytes, dtype=torch.uint8, device=dev)
        self.repack(decoder)

    def repack(self, decoder):
        l3, l4, l5 = decoder.hidden[0], decoder.hidden[1], decoder.reconstruction
        t = [_f32c(x.detach(), "decoder parameter") for x in (l3.weight, l3.bias, l4.weight, l4.bias, l5.weight, l5.bias)]
        N.check(self.lib.dvae_mcem_pack(ctypes.byref(self.plan), N.ptr(t[0]), t[0].stride(0), N.ptr(t[1]), N.ptr(t[2]), t[2].stride(0),
                                        N.ptr(t[3]), N.ptr(t[4]), t[4].stride(0), N.ptr(t[5]), N.ptr(self.weights), N.stream()),
                "dvae_mcem_pack")
        self._keep = t

    # sample_posterior (mcem.py:207-277, 372-448) [+ compute_Vs of the kept samples when want_vs]
    def sample(self, Z, y, g, Vb, X2, noise, logu, burnin, var_rw=0.01, want_vs=True, trace=False):
        nit, L, n = noise.shape
        assert L == Z_DIM and Z.shape == (Z_DIM, n) and logu.shape == (nit, n) and Vb.shape == (F_BINS, n) and X2.shape == (F_BINS, n)
        R = nit - burnin
        Z, g, Vb, X2, noise, logu = (_f32c(a, nm) for a, nm in ((Z, "Z"), (g, "g"), (Vb, "Vb"), (X2, "X2"), (noise, "noise"), (logu, "logu")))
        y = _f32c(y, "y") if self.y_dim else None
        if y is not None:
            assert y.shape == (self.y_dim, n), (tuple(y.shape), self.y_dim, n)
        Zs = torch.empty((n, R, Z_DIM), dtype=torch.float32, device=Z.device)
        Vs = torch.empty((R, F_BINS, n), dtype=torch.float32, device=Z.device) if want_vs else None
        accp = torch.empty((nit, n), dtype=torch.float32, device=Z.device) if trace else None
        accd = torch.empty((nit, n), dtype=torch.uint8, device=Z.device) if trace else None
        N.check(self.lib.dvae_mcem_sample(ctypes.byref(self.plan), N.ptr(self.weights), N.ptr(Z), N.ptr(y), N.ptr(g), N.ptr(Vb), N.ptr(X2),
                                          N.ptr(noise), N.ptr(logu), nit, burnin, float(var_rw), n, N.ptr(Zs), N.ptr(Vs), N.ptr(accp),
                                          N.ptr(accd), N.stream()), "dvae_mcem_sample")
        return (Zs, Vs, accp, accd) if trace else (Zs, Vs)

    # compute_Vs (mcem.py:280-290): Zs (N, R, 16) -> (R, F, N)
    def decode(self, Zs, y):
        n, R, L = Zs.shape
        assert L == Z_DIM
        Zs = _f32c(Zs, "Zs")
        y = _f32c(y, "y") if self.y_dim else None
        Vs = torch.empty((R, F_BINS, n), dtype=torch.float32, device=Zs.device)
        N.check(self.lib.dvae_mcem_decode(ctypes.byref(self.plan), N.ptr(self.weights), N.ptr(Zs), N.ptr(y), R, n, N.ptr(Vs), N.stream()),
                "dvae_mcem_decode")
        return Vs


def m_step_(X2, Vs, W, H, g, Vb, want_cost=True):
    """EM.M_step (mcem.py:91-153), in place on W (F,K), H (K,N), g (N), Vb (F,N); returns the cost (1-element tensor)."""
    lib = N.load()
    R, F, n = Vs.shape
    K = W.shape[1]
    assert F == F_BINS and W.shape == (F, K) and H.shape == (K, n) and g.shape == (n,) and Vb.shape == (F, n) and X2.shape == (F, n)
    for a, nm in ((X2, "X2"), (Vs, "Vs"), (W, "W"), (H, "H"), (g, "g"), (Vb, "Vb")):
        if not (a.is_cuda and a.dtype == torch.float32 and a.is_contiguous()):
            raise RuntimeError(f"m_step_: {nm} must be a contiguous float32 CUDA tensor")
    ws = torch.empty(lib.dvae_mcem_m_step_workspace_bytes(n, K, 1), dtype=torch.uint8, device=W.device)
    cost = torch.empty(1, dtype=torch.float32, device=W.device) if want_cost else None
    N.check(lib.dvae_mcem_m_step(N.ptr(X2), N.ptr(Vs), R, n, K, N.ptr(W), N.ptr(H), N.ptr(g), N.ptr(Vb), N.ptr(cost), N.ptr(ws), N.stream()),
            "dvae_mcem_m_step")
    return cost


def wiener(Vs, g, Vb):
    """compute_WF (mcem.py:321-327) -> WFs, WFn (F, N)."""
    lib = N.load()
    R, F, n = Vs.shape
    Vs, g, Vb = _f32c(Vs, "Vs"), _f32c(g, "g"), _f32c(Vb, "Vb")
    WFs = torch.empty((F, n), dtype=torch.float32, device=Vs.device)
    WFn = torch.empty_like(WFs)
    N.check(lib.dvae_mcem_wiener(N.ptr(Vs), R, n, N.ptr(g), N.ptr(Vb), N.ptr(WFs), N.ptr(WFn), N.stream()), "dvae_mcem_wiener")
    return WFs, WFn


def m_step_batch_(X2, Vs, W, H, g, Vb, seg_start, seg_count, tile_seg):
    """M-step for U utterances laid side by side on the (padded) frame axis; in place; returns cost (U)."""
    lib = N.load()
    R, F, n = Vs.shape
    U, _, K = W.shape
    assert F == F_BINS and W.shape == (U, F, K) and H.shape == (K, n) and g.shape == (n,) and Vb.shape == (F, n) and X2.shape == (F, n)
    assert n % 32 == 0 and tile_seg.numel() == n // 32 and seg_start.numel() == U and seg_count.numel() == U
    for a, nm in ((X2, "X2"), (Vs, "Vs"), (W, "W"), (H, "H"), (g, "g"), (Vb, "Vb")):
        if not (a.is_cuda and a.dtype == torch.float32 and a.is_contiguous()):
            raise RuntimeError(f"m_step_batch_: {nm} must be a contiguous float32 CUDA tensor")
    for a in (seg_start, seg_count, tile_seg):
        if not (a.is_cuda and a.dtype == torch.int32 and a.is_contiguous()):
            raise RuntimeError("m_step_batch_: segment tables must be contiguous int32 CUDA tensors")
    ws = torch.empty(lib.dvae_mcem_m_step_workspace_bytes(n, K, U), dtype=torch.uint8, device=W.device)
    cost = torch.empty(U, dtype=torch.float32, device=W.device)
    N.check(lib.dvae_mcem_m_step_batch(N.ptr(X2), N.ptr(Vs), R, n, K, U, N.ptr(seg_start), N.ptr(seg_count), N.ptr(tile_seg),
                                       N.ptr(W), N.ptr(H), N.ptr(g), N.ptr(Vb), N.ptr(cost), N.ptr(ws), N.stream()),
            "dvae_mcem_m_step_batch")
    return cost


class McemBatch:
    """MCEM enhancement of many utterances at once (the MI355X-native form of the reference's process pool,
    scripts/evaluate_ntcd_M2.py:288-327: there, nb_devices * 2 processes each run one utterance at a time).

    The utterances are laid side by side on the frame axis, each padded to a multiple of 32 frames; the
    Metropolis-Hastings chains are independent per frame, the NMF factors per utterance.  Per EM iteration:
    one chain launch, three M-step launches, whatever the number of utterances.

    vae: a packages.models VAE (encoder / decoder / z_dim).  label_in_encoder / label_in_decoder select the
    reference variant: MCEM_M1 (False, False), MCEM_M2 (True, True), MCEM_M2v2 / M2v3 (False, True).  For the M1 variant the
    chain lengths follow what the reference actually runs (its argument-shift quirk, see __init__).
    """

    def __init__(self, vae, niter=100, nsamples_E_step=10, burnin_E_step=30, nsamples_WF=25, burnin_WF=75, var_RW=0.01,
                 nmf_rank=10, eps=2.220446049250313e-16, label_in_encoder=True, label_in_decoder=True, precision="fp32",
                 reference_m1_counts=True):
        self.vae, self.niter = vae, niter
        self.n_e, self.b_e, self.n_wf, self.b_wf = nsamples_E_step, burnin_E_step, nsamples_WF, burnin_WF
        if reference_m1_counts and not label_in_encoder and not label_in_decoder:
            # MCEM_M1 passes (Z, nsamples, burnin) positionally into sample_posterior(Z, y, nsamples=10, burnin=30)
            # (reference mcem.py:207, 297-298, 314-315): the chain it actually runs keeps `burnin` samples after the
            # default burn-in of 30.  Same chain lengths here (reference_m1_counts=False: the lengths as written).
            self.n_e, self.b_e, self.n_wf, self.b_wf = burnin_E_step, 30, burnin_WF, 30
        self.var_RW, self.K, self.eps = var_RW, nmf_rank, eps
        self.label_in_encoder, self.label_in_decoder = label_in_encoder, label_in_decoder
        self.precision = precision
        self._pack = None

    def _layout(self, counts, dev):
        starts, pos = [], 0
        for c in counts:
            starts.append(pos)
            pos += (c + 31) // 32 * 32
        tile_seg = []
        for u, c in enumerate(counts):
            tile_seg += [u] * ((c + 31) // 32)
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
        return starts, pos, i32(starts), i32(counts), i32(tile_seg)

    def init_parameters(self, X_list, y_list=None, device="cuda"):
        """X_list: complex mixture STFTs (F, N_u) (numpy); y_list: labels (y_dim, N_u) tensors/arrays or None."""
        dev = torch.device(device)
        self.X_list = X_list
        self.counts = [x.shape[1] for x in X_list]
        self.starts, self.ntot, self.seg_start, self.seg_count, self.tile_seg = self._layout(self.counts, dev)
        U, K = len(X_list), self.K
        self.X2 = torch.ones((F_BINS, self.ntot), dtype=torch.float32, device=dev)
        self.H = torch.ones((K, self.ntot), dtype=torch.float32, device=dev)
        self.Vb = torch.ones((F_BINS, self.ntot), dtype=torch.float32, device=dev)
        self.g = torch.ones(self.ntot, dtype=torch.float32, device=dev)
        self.Z = torch.zeros((Z_DIM, self.ntot), dtype=torch.float32, device=dev)
        self.W = torch.empty((U, F_BINS, K), dtype=torch.float32, device=dev)
        self.y = None
        if self.label_in_decoder:
            y_dim = y_list[0].shape[0]
            self.y = torch.zeros((y_dim, self.ntot), dtype=torch.float32, device=dev)
        for u, X in enumerate(X_list):
            s, c = self.starts[u], self.counts[u]
            self.X2[:, s:s + c] = torch.from_numpy((np.abs(X) ** 2).astype(np.float32)).to(dev)
            self.W[u] = torch.clamp_min(torch.rand(F_BINS, K, device=dev), self.eps)               # mcem.py:42
            self.H[:, s:s + c] = torch.clamp_min(torch.rand(K, c, device=dev), self.eps)           # mcem.py:43
            self.Vb[:, s:s + c] = self.W[u] @ self.H[:, s:s + c]                                    # mcem.py:52
            if self.y is not None:
                self.y[:, s:s + c] = torch.as_tensor(y_list[u], dtype=torch.float32).to(dev)
        enc_in = torch.cat([self.X2, self.y], dim=0) if self.label_in_encoder else self.X2
        with torch.no_grad():
            _, mu, _ = self.vae.encoder(torch.t(enc_in))                                            # mcem.py:200, 364
        self.Z = torch.t(mu).contiguous()
        y_dim = self.y.shape[0] if self.y is not None else 0
        self._pack = DecoderPack(self.vae.decoder, y_dim, self.precision)

    def _chain(self, nsamples, burnin, draws=None):
        nit = nsamples + burnin
        if draws is None:
            noise = torch.randn(nit, Z_DIM, self.ntot, device=self.Z.device)
            logu = torch.log(torch.rand(nit, self.ntot, device=self.Z.device))
        else:
            noise, logu = draws
        return self._pack.sample(self.Z, self.y, self.g, self.Vb, self.X2, noise, logu, burnin, var_rw=float(self.var_RW))

    def _loop_buffers(self):
        """Scratch of the EM loop, allocated once per run: the iteration's samples / variances, the M-step workspace."""
        dev = self.Z.device
        if getattr(self, "_bufs", None) is None or self._bufs[0].shape[0] != self.ntot or self._bufs[0].shape[1] != self.n_e:
            lib = self._pack.lib
            self._bufs = (torch.empty((self.ntot, self.n_e, Z_DIM), dtype=torch.float32, device=dev),
                          torch.empty((self.n_e, F_BINS, self.ntot), dtype=torch.float32, device=dev),
                          torch.empty(lib.dvae_mcem_m_step_workspace_bytes(self.ntot, self.K, len(self.counts)), dtype=torch.uint8, device=dev))
        return self._bufs

    def _iteration(self, noise_ptr, logu_ptr, cost_ptr, lazy=False):
        """One EM iteration (mcem.py:156-160) = ONE call into the library (dvae_mcem_em_iteration): chain + decoder variances of the kept
        samples, last kept sample -> Z in place, the M-step's launches, cost (U) to cost_ptr.  Draws and cost are raw device addresses: the
        loop does no tensor arithmetic between iterations (it was bound by the interpreter, not by its kernels: ~150 of 590 us per
        iteration at 25 utterances were gaps between ten launches issued from Python, profiles/r04_mcem_kernel_stats.csv)."""
        Zs, Vs, ws = self._loop_buffers()
        pk = self._pack
        if lazy:
            # two M-step launches instead of three (dvae_mcem_em_iteration_lazy): cost_ptr is where the PREVIOUS iteration's cost goes (None
            # in the first iteration); run() flushes the last one
            N.check(pk.lib.dvae_mcem_em_iteration_lazy(ctypes.byref(pk.plan), N.ptr(pk.weights), N.ptr(self.Z), N.ptr(self.y), N.ptr(self.g), N.ptr(self.Vb),
                                                       N.ptr(self.X2), noise_ptr, logu_ptr, self.n_e + self.b_e, self.b_e, float(self.var_RW), self.ntot,
                                                       self.K, len(self.counts), N.ptr(self.seg_start), N.ptr(self.seg_count), N.ptr(self.tile_seg),
                                                       N.ptr(self.W), N.ptr(self.H), N.ptr(Zs), N.ptr(Vs), cost_ptr, N.ptr(ws), N.stream()),
                    "dvae_mcem_em_iteration_lazy")
            return
        N.check(pk.lib.dvae_mcem_em_iteration(ctypes.byref(pk.plan), N.ptr(pk.weights), N.ptr(self.Z), N.ptr(self.y), N.ptr(self.g), N.ptr(self.Vb),
                                              N.ptr(self.X2), noise_ptr, logu_ptr, self.n_e + self.b_e, self.b_e, float(self.var_RW), self.ntot,
                                              self.K, len(self.counts), N.ptr(self.seg_start), N.ptr(self.seg_count), N.ptr(self.tile_seg),
                                              N.ptr(self.W), N.ptr(self.H), N.ptr(Zs), N.ptr(Vs), cost_ptr, N.ptr(ws), N.stream()),
                "dvae_mcem_em_iteration")

    def run(self, draws=None, graph=None):
        """EM.run (mcem.py:156-179) for all utterances.  draws: optional list of niter + 1 (noise, logu) pairs.
        Returns cost (niter, U); sets S_hat / N_hat (lists of complex (F, N_u) arrays).

        graph (default off; DVAE_MCEM_GRAPH=1 or graph=True turns it on): the iteration is captured ONCE into a HIP graph (after one eager
        iteration) and replayed: every buffer of the iteration is static, the generator draws inside the graph (or, with recorded draws,
        they are copied into the graph's static draw buffers before each replay: same kernels, same results).  Measured (round 4, 25
        utterances x 300 frames, 100 iterations): the loop is bound by its kernels, not by their launches -- 264 utterances / s replayed
        beside 280 eager (capture + instantiation cost more than the launches they save)."""
        dev = self.Z.device
        if graph is None:
            graph = os.environ.get("DVAE_MCEM_GRAPH", "0") == "1" and self.niter >= 8
        U = len(self.counts)
        cost = torch.empty((self.niter, U), dtype=torch.float32, device=dev)
        nit = self.n_e + self.b_e
        for a, nm in ((self.Z, "Z"), (self.g, "g"), (self.Vb, "Vb"), (self.X2, "X2"), (self.W, "W"), (self.H, "H")):
            if not (a.is_cuda and a.dtype == torch.float32 and a.is_contiguous()):
                raise RuntimeError(f"McemBatch.run: {nm} must be a contiguous float32 CUDA tensor")
        cptr = cost.data_ptr()
        # eager loops: two M-step launches per iteration where the library offers it (at most 10 kept samples, rank 10); the cost of
        # iteration i is written by iteration i + 1, the last one by a flush -- the same bits as the three-launch iteration
        lazy = (not graph and self.n_e <= 10 and self.K == 10 and self.n_e * F_BINS * self.ntot * 4 < 2 ** 31 - 1
                and os.environ.get("DVAE_MSTEP") != "3pass" and os.environ.get("DVAE_MCEM_LAZY", "1") != "0")
        cost_at = (lambda it: (cptr + 4 * U * (it - 1)) if it else None) if lazy else (lambda it: cptr + 4 * U * it)
        if draws is not None and not graph:
            for it in range(self.niter):
                noise, logu = (_f32c(a, nm) for a, nm in zip(draws[it], ("noise", "logu")))
                assert noise.shape == (nit, Z_DIM, self.ntot) and logu.shape == (nit, self.ntot)
                self._iteration(noise.data_ptr(), logu.data_ptr(), cost_at(it), lazy)
        elif not graph or self.niter < 2:
            # the generator's draws for as many iterations at a time as fit 256 MB (the reference draws inside the loop, mcem.py:244, 257:
            # the same distributions, fewer launches); the iterations themselves are one library call each
            per = nit * (Z_DIM + 1) * self.ntot * 4
            chunk = max(1, min(self.niter, (256 << 20) // per))
            for it0 in range(0, self.niter, chunk):
                c = min(chunk, self.niter - it0)
                noise = torch.randn(c, nit, Z_DIM, self.ntot, device=dev)
                logu = torch.rand(c, nit, self.ntot, device=dev).log_()
                nptr, lptr = noise.data_ptr(), logu.data_ptr()
                for j in range(c):
                    self._iteration(nptr + 4 * nit * Z_DIM * self.ntot * j, lptr + 4 * nit * self.ntot * j, cost_at(it0 + j), lazy)
        else:
            static = (torch.empty((nit, Z_DIM, self.ntot), dtype=torch.float32, device=dev), torch.empty((nit, self.ntot), dtype=torch.float32, device=dev))
            c = torch.empty(U, dtype=torch.float32, device=dev)

            def body():
                if draws is None:
                    static[0].normal_(); static[1].uniform_().log_()
                self._iteration(static[0].data_ptr(), static[1].data_ptr(), c.data_ptr())
            if draws is not None:
                static[0].copy_(draws[0][0]); static[1].copy_(draws[0][1])
            body()                                                 # eager: first-use set-up (kernel attributes, allocator) happens outside the capture
            cost[0].copy_(c)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            for it in range(1, self.niter):
                if draws is not None:
                    static[0].copy_(draws[it][0]); static[1].copy_(draws[it][1])
                g.replay()
                cost[it].copy_(c)
            torch.cuda.synchronize(dev)
            del g
        if lazy and self.niter > 0:
            N.check(self._pack.lib.dvae_mcem_cost_flush(self.n_e, self.ntot, self.K, U, N.ptr(self.seg_start), N.ptr(self.seg_count),
                                                        cptr + 4 * U * (self.niter - 1), N.ptr(self._loop_buffers()[2]), N.stream()), "dvae_mcem_cost_flush")
        Zs, Vs = self._chain(self.n_wf, self.b_wf, None if draws is None else draws[self.niter])
        self.WFs, self.WFn = wiener(Vs, self.g, self.Vb)
        WFs, WFn = self.WFs.cpu().numpy(), self.WFn.cpu().numpy()
        self.S_hat = [WFs[:, s:s + c] * X for s, c, X in zip(self.starts, self.counts, self.X_list)]
        self.N_hat = [WFn[:, s:s + c] * X for s, c, X in zip(self.starts, self.counts, self.X_list)]
        return cost.cpu().numpy()
