"""Drop-in replacement for the reference's packages/models/mcem.py (Monte-Carlo EM speech enhancement:
VAE speech prior + NMF noise model, Leglaive et al.): same classes (EM, MCEM_M1, MCEM_M2, MCEM_M2v2,
MCEM_M2v3), constructor arguments, method names, attribute names and shapes, so
scripts/evaluate_ntcd_*.py run unchanged (`mcem.init_parameters(...)`, `mcem.run()`, `.S_hat`, `.N_hat`).

  * CUDA tensors -> the kernels of include/dvae_mcem.h: one persistent launch per Metropolis-Hastings
    chain (the whole E-step), three small launches per M-step, one for the Wiener gains.  No fallback:
    a missing library raises.  The random draws come from torch's device generator: one call per
    chain instead of two per iteration, the E-step's chains drawn up to sixteen at a time.
  * host tensors -> ATen, drawing from the global generator in the reference's order (randn(L,N),
    rand(N) per iteration), so a seeded CPU run reproduces the reference's.  Selected by the tensors'
    device only.

Kept from the reference: importing this module seeds numpy and torch with 0
(Q13); MCEM_M1.E_step / compute_WF hand (Z, nsamples, burnin) to sample_posterior(Z, y, nsamples=10,
burnin=30), so M1 chains run with nsamples = its burn-in argument and burn-in 30 (Q12); Vb is not
recomputed after W and H are normalised.
"""
my_seed = 0
import numpy as np
np.random.seed(my_seed)
import torch
torch.manual_seed(my_seed)

from packages import _native


def _latent_dim(vae):
    return vae.latent_dim if hasattr(vae, 'latent_dim') else vae.z_dim


class EM:
    """NMF noise model + generic EM loop; the E-step lives in the subclasses (reference mcem.py:8-179)."""

    # matrix-core operand policy of the device path: "fp32" (exact products; the default), "bf16x3" (split bf16: the same bounds against the
    # reference's recorded runs, one utterance in 25 instead of 46 ms) or "bf16" (loose).  A class attribute, set on the instance by callers that
    # construct the object themselves; the reference's evaluate scripts run unchanged with DVAE_MCEM_PRECISION=bf16x3 in the environment.
    precision = __import__("os").environ.get("DVAE_MCEM_PRECISION", "fp32")

    def __init__(self, niter=100):
        self.niter = niter
        self.Vs = None          # speech variance (R, F, N), one slice per posterior draw
        self._Vs_scaled = None
        self._Vx = None
        self._cost = None

    # ------------------------------------------------------------------ state
    def init_parameters(self, X, S, nmf_rank, eps, device="cpu"):
        self.device = device
        F, N = X.shape
        floor_W = eps * torch.ones(F, nmf_rank, device=self.device)
        self.W = torch.max(torch.rand(F, nmf_rank, device=self.device), floor_W)       # (F, K)
        floor_H = eps * torch.ones(nmf_rank, N, device=self.device)
        self.H = torch.max(torch.rand(nmf_rank, N, device=self.device), floor_H)       # (K, N)
        self.X = X                                                                       # complex mixture STFT (F, N)
        self.X_abs_2 = torch.tensor(np.abs(X) ** 2, device=self.device)
        self.S_abs_2 = torch.tensor(np.abs(S) ** 2, device=self.device)
        self.compute_Vb()
        self.g = torch.ones(N, device=self.device)
        self.Vs = None
        self._Vs_scaled = None
        self._Vx = None
        self._cost = None

    def _on_device(self):
        return self.W.is_cuda

    def np2tensor(self, x):
        return torch.tensor(x, device=self.device)

    def tensor2np(self, x):
        return x.numpy()

    # Vs_scaled = g Vs and Vx = Vs_scaled + Vb are (R, F, N) temporaries; the device kernels never need them
    # materialised, so they are built on first access.
    @property
    def Vs_scaled(self):
        if self._Vs_scaled is None and self.Vs is not None:
            self._Vs_scaled = self.g * self.Vs
        return self._Vs_scaled

    @Vs_scaled.setter
    def Vs_scaled(self, v):
        self._Vs_scaled = v

    @property
    def Vx(self):
        if self._Vx is None and self.Vs is not None:
            self._Vx = self.Vs_scaled + self.Vb
        return self._Vx

    @Vx.setter
    def Vx(self, v):
        self._Vx = v

    def compute_expected_neg_log_like(self):
        if self._cost is not None:
            return self._cost
        return torch.mean(torch.log(self.Vx) + self.X_abs_2 / self.Vx)

    def compute_Vs(self, Z):
        pass

    def compute_Vs_scaled(self):
        self._Vs_scaled = None if self._on_device() else self.g * self.Vs
        self._cost = None

    def compute_Vx(self):
        self._Vx = None if self._on_device() else self.Vs_scaled + self.Vb
        self._cost = None

    def compute_Vb(self):
        self.Vb = self.W @ self.H

    def E_step(self):
        pass

    # ------------------------------------------------------------------ M-step (reference mcem.py:91-153)
    def M_step(self):
        if self._on_device() and self.Vs.ndim == 3:
            for nm in ("W", "H", "g", "Vb"):
                setattr(self, nm, getattr(self, nm).contiguous())
            cost = _native.mcem_dev().m_step_(self.X_abs_2.contiguous(), self.Vs.contiguous(), self.W, self.H, self.g, self.Vb)
            self._Vs_scaled = None
            self._Vx = None
            self._cost = cost[0]
            return
        squeeze = self.Vx.ndim == 2             # PEEM-style callers hand (F, N) variances
        Vs = self.Vs[None] if squeeze else self.Vs
        X2 = self.X_abs_2

        def vx():
            return self.g * Vs + self.Vb

        def inv_sums(v):
            return torch.sum(v ** -2, axis=0), torch.sum(v ** -1, axis=0)

        s2, s1 = inv_sums(self.Vx[None] if squeeze else self.Vx)
        self.W = self.W * (((X2 * s2) @ self.H.T) / (s1 @ self.H.T)) ** .5
        self.compute_Vb()
        s2, s1 = inv_sums(vx())
        self.H = self.H * ((self.W.T @ (X2 * s2)) / (self.W.T @ s1)) ** .5
        self.compute_Vb()
        v = vx()
        col = torch.sum(torch.abs(self.W), axis=0)
        self.W = self.W / col.unsqueeze(0)
        self.H = self.H * col.unsqueeze(1)
        num = torch.sum(X2 * torch.sum(Vs * v ** -2, axis=0), axis=0)
        den = torch.sum(torch.sum(Vs * v ** -1, axis=0), axis=0)
        self.g = self.g * (num / den) ** .5
        self._Vs_scaled = self.g * Vs
        self._Vx = self._Vs_scaled + self.Vb
        if squeeze:
            self._Vs_scaled, self._Vx = self._Vs_scaled[0], self._Vx[0]
        self._cost = None

    def run(self):
        cost = np.zeros(self.niter)
        for n in np.arange(self.niter):
            self.E_step()
            self.M_step()
            cost[n] = self.compute_expected_neg_log_like()
        WFs, WFn = self.compute_WF(sample=True)
        self.S_hat = self.tensor2np(WFs.cpu()) * self.X
        self.N_hat = self.tensor2np(WFn.cpu()) * self.X
        return cost


class _MCEM(EM):
    """Metropolis-Hastings E-step shared by the four variants; they differ in what the encoder and the
    decoder are fed (reference mcem.py:182-844)."""

    _label_in_decoder = True        # decoder input is [z | y]
    _label_in_encoder = True        # encoder input is [|X|^2 | y]

    def __init__(self, niter, nsamples_E_step=10, burnin_E_step=30, nsamples_WF=25, burnin_WF=75, var_RW=0.01):
        super().__init__(niter=niter)
        self.nsamples_E_step = nsamples_E_step
        self.burnin_E_step = burnin_E_step
        self.nsamples_WF = nsamples_WF
        self.burnin_WF = burnin_WF
        self.var_RW = var_RW
        self._pack = None
        self._cached = None         # (sampled-latents tensor, its decoder variances) from the last chain

    def _init_common(self, X, S, y, vae, nmf_rank, eps, device):
        if type(vae).__name__ == 'RVAE':
            raise NameError('MCEM algorithm only valid for FFNN VAE')
        EM.init_parameters(self, X=X, S=S, nmf_rank=nmf_rank, eps=eps, device=device)
        self.vae = vae
        if self._label_in_decoder:
            self.y = y
        enc_in = (lambda P: torch.cat([P, self.y], dim=0)) if self._label_in_encoder else (lambda P: P)
        _, Z, _ = self.vae.encoder(torch.t(enc_in(self.X_abs_2)))
        _, Zclean, _ = self.vae.encoder(torch.t(enc_in(self.S_abs_2)))
        self.Z = torch.t(Z)                     # last draw of the latents (L, N)
        self.Zclean = torch.t(Zclean)
        self.X_abs_2_t = self.X_abs_2.clone()
        self._pack = None
        self._cached = None
        self._draw_buf = {}

    def _draws(self, nit, L, N, dev):
        """(noise (nit, L, N), log u (nit, N)) of one chain from the device generator.  The E-step's chain runs `niter` times on one shape:
        its draws are made for up to 16 chains per generator call (three small launches per chain were 15 of an iteration's 190 us on one
        utterance); the stepwise loop and run() take their chains from the same blocks in the same order, so equal seeds give equal
        results either way.  Any other chain (the final Wiener chain: once) draws for itself."""
        e_nit = sum(self._e_counts())
        if nit != e_nit:
            return torch.randn(nit, L, N, device=dev), torch.log(torch.rand(nit, N, device=dev))
        buf = getattr(self, "_draw_buf", None)
        if buf is None:
            buf = self._draw_buf = {}
        blk = buf.get((nit, L, N))
        if blk is None or blk[2] >= blk[0].shape[0]:
            served = blk[3] if blk is not None else 0
            per_chain = nit * (L + 1) * N * 4
            B = max(1, min(16, int(self.niter) - served, (64 << 20) // max(per_chain, 1)))      # never more chains than the loop has left
            blk = buf[(nit, L, N)] = [torch.randn(B, nit, L, N, device=dev), torch.log(torch.rand(B, nit, N, device=dev)), 0, served]
        i = blk[2]
        blk[2] = i + 1
        blk[3] += 1
        return blk[0][i], blk[1][i]

    def _decoder_pack(self):
        if self._pack is None:
            y_dim = self.y.shape[0] if self._label_in_decoder else 0
            self._pack = _native.mcem_dev().DecoderPack(self.vae.decoder, y_dim, self.precision)
        return self._pack

    def _decode_cols(self, Zc, y):
        """decoder on latent columns (L, N) -> variances (F, N)."""
        inp = torch.cat([Zc, y], dim=0) if self._label_in_decoder else Zc
        return torch.t(self.vae.decoder(torch.t(inp)))

    def _chain(self, Z, y, nsamples, burnin):
        """Random-walk Metropolis-Hastings; returns Z_sampled (N, nsamples, L)."""
        L = _latent_dim(self.vae)
        N = self.X.shape[1]
        nit = nsamples + burnin
        if Z.is_cuda:
            noise, logu = self._draws(nit, L, N, Z.device)
            Zs, Vs = self._decoder_pack().sample(Z, y if self._label_in_decoder else None, self.g, self.Vb, self.X_abs_2_t,
                                                 noise, logu, burnin, var_rw=float(self.var_RW))
            self._cached_vs = Vs
            return Zs
        step = torch.sqrt(torch.tensor(np.float32(self.var_RW), device=self.device))
        Zs = torch.zeros(N, nsamples, L, device=self.device)
        with torch.no_grad():
            Zt = Z.clone()
            g_t, Vb_t = self.g.clone(), self.Vb.clone()
            Vx = g_t * self._decode_cols(Zt, y) + Vb_t
            kept = 0
            for m in range(nit):
                Zp = Zt + step * torch.randn(L, N, device=self.device)
                Vxp = g_t * self._decode_cols(Zp, y) + Vb_t
                log_ratio = (torch.sum(torch.log(Vx) - torch.log(Vxp) + (1 / Vx - 1 / Vxp) * self.X_abs_2_t, 0)
                             + .5 * torch.sum(Zt.pow(2) - Zp.pow(2), 0))
                take = torch.log(torch.rand(N, device=self.device)) < log_ratio
                # the decoder acts frame by frame: selecting columns equals re-running it on the updated Z
                Zt = torch.where(take, Zp, Zt)
                Vx = torch.where(take, Vxp, Vx)
                if m >= burnin:
                    Zs[:, kept, :] = torch.t(Zt)
                    kept += 1
        self._cached_vs = None
        return Zs

    # chain lengths (nsamples, burnin) the E-step / the final Wiener chain actually run
    def _e_counts(self):
        return self.nsamples_E_step, self.burnin_E_step

    def _wf_counts(self):
        return self.nsamples_WF, self.burnin_WF

    def run(self):
        """EM.run (reference mcem.py:156-179).  On the device, with the decoder geometry the kernels cover, the loop body -- E_step, M_step,
        cost -- is ONE library call per iteration (dvae_mcem_em_iteration) on buffers allocated once, the generator's draws are made for
        many iterations at a time, and the cost is read back once after the loop: the reference's loop stores `cost[n]` into a numpy array,
        i.e. it synchronises with the device and crosses the interpreter a dozen times in every iteration (evaluate_ntcd_M2.py:201-205
        runs one utterance per process, so those gaps are all a process sees of the GPU).  Same kernels, same arithmetic as stepping
        E_step() / M_step() by hand; DVAE_MCEM_RUN=steps keeps the stepwise loop."""
        import os
        dev_mod = _native.mcem_dev() if self._on_device() else None
        y_dim = self.y.shape[0] if self._label_in_decoder else 0
        if (dev_mod is None or os.environ.get("DVAE_MCEM_RUN", "fused") == "steps" or self.W.dtype != torch.float32
                or not dev_mod.decoder_supported(self.vae.decoder, y_dim) or self.W.shape[1] > 16):
            return EM.run(self)
        import ctypes
        Nn = dev_mod.N
        lib = Nn.load()
        pk = self._decoder_pack()
        n_e, b_e = self._e_counts()
        nit = n_e + b_e
        F, N = self.X_abs_2_t.shape
        L = _latent_dim(self.vae)
        K = self.W.shape[1]
        dev = self.W.device
        for nm in ("W", "H", "g", "Vb", "Z", "X_abs_2_t"):
            setattr(self, nm, getattr(self, nm).detach().to(torch.float32).contiguous())
        y = self.y.detach().to(torch.float32).contiguous() if self._label_in_decoder else None
        Zs = torch.empty((N, n_e, L), dtype=torch.float32, device=dev)
        Vs = torch.empty((n_e, F, N), dtype=torch.float32, device=dev)
        ws = torch.empty(lib.dvae_mcem_m_step_workspace_bytes(N, K, 1), dtype=torch.uint8, device=dev)
        cost = torch.empty(self.niter, dtype=torch.float32, device=dev)
        args = (ctypes.byref(pk.plan), Nn.ptr(pk.weights), Nn.ptr(self.Z), Nn.ptr(y), Nn.ptr(self.g), Nn.ptr(self.Vb), Nn.ptr(self.X_abs_2_t))
        tail = (Nn.ptr(self.W), Nn.ptr(self.H), Nn.ptr(Zs), Nn.ptr(Vs))
        cptr, wptr, stream = cost.data_ptr(), Nn.ptr(ws), Nn.stream()
        # two launches per M-step instead of three where the library offers it (at most 10 kept samples, rank 10): W normalised by the frames
        # kernel, the cost of iteration i formed by the W update of iteration i + 1 and by one flush after the loop -- the same bits
        lazy = (n_e <= 10 and K == 10 and n_e * F * N * 4 < 2 ** 31 - 1 and os.environ.get("DVAE_MSTEP") != "3pass"
                and os.environ.get("DVAE_MCEM_LAZY", "1") != "0")
        for it in range(self.niter):
            # the draws of _chain, from its blocks in its order (one generator stream for this loop and the stepwise one: equal results on equal seeds)
            noise, logu = self._draws(nit, L, N, dev)
            if lazy:
                Nn.check(lib.dvae_mcem_em_iteration_lazy(*args, Nn.ptr(noise), Nn.ptr(logu), nit, b_e, float(self.var_RW), N, K, 1,
                                                         None, None, None, *tail, (cptr + 4 * (it - 1)) if it else None, wptr, stream),
                         "dvae_mcem_em_iteration_lazy")
            else:
                Nn.check(lib.dvae_mcem_em_iteration(*args, Nn.ptr(noise), Nn.ptr(logu), nit, b_e, float(self.var_RW), N, K, 1,
                                                    None, None, None, *tail, cptr + 4 * it, wptr, stream), "dvae_mcem_em_iteration")
        if lazy and self.niter > 0:
            Nn.check(lib.dvae_mcem_cost_flush(n_e, N, K, 1, None, None, cptr + 4 * (self.niter - 1), wptr, stream), "dvae_mcem_cost_flush")
        self.Vs = Vs
        self._Vs_scaled = None
        self._Vx = None
        self._cost = cost[-1]
        self._cached = None
        WFs, WFn = self.compute_WF(sample=True)
        self.S_hat = self.tensor2np(WFs.cpu()) * self.X
        self.N_hat = self.tensor2np(WFn.cpu()) * self.X
        return cost.cpu().numpy().astype(np.float64)

    def compute_Vs(self, Z):
        """Z: (N, R, L [+ y_dim]) -> self.Vs (R, F, N)."""
        if self._cached is not None and self._cached[0] is Z and self._cached[1] is not None:
            self.Vs = self._cached[1]           # decoded inside the chain's launch
            return
        with torch.no_grad():
            Vs_t = self.vae.decoder(Z)
        if Vs_t.ndim == 2:
            Vs_t = Vs_t.unsqueeze(1)
        self.Vs = Vs_t.permute(1, 2, 0)         # (N, R, F) -> (R, F, N)


class MCEM_M1(_MCEM):
    _label_in_decoder = False
    _label_in_encoder = False

    def init_parameters(self, X, S, vae, nmf_rank, eps, device):
        self._init_common(X, S, None, vae, nmf_rank, eps, device)

    def sample_posterior(self, Z, y, nsamples=10, burnin=30):
        Zs = self._chain(Z, None, nsamples, burnin)
        self._cached = (Zs, self._cached_vs)
        return Zs

    # quirk Q12 (reference mcem.py:207, 297-298, 314-315): E_step / compute_WF pass (Z, nsamples, burnin) into (Z, y, nsamples=10, burnin=30),
    # so the chains keep `burnin` samples after the default burn-in of 30
    def _e_counts(self):
        return self.burnin_E_step, 30

    def _wf_counts(self):
        return self.burnin_WF, 30

    def E_step(self):
        Z_t = self.sample_posterior(self.Z, self.nsamples_E_step, self.burnin_E_step)      # sic (Q12)
        self.Z = torch.t(torch.squeeze(Z_t[:, -1, :]))
        self.compute_Vs(Z_t)
        self.compute_Vs_scaled()
        self.compute_Vx()

    def compute_WF(self, sample=False):
        if sample:
            Z_t = self.sample_posterior(self.Z, self.nsamples_WF, self.burnin_WF)          # sic (Q12)
            self.compute_Vs(Z_t)
            self.compute_Vs_scaled()
            self.compute_Vx()
        return _wiener(self)


class MCEM_M2(_MCEM):
    def init_parameters(self, X, S, y, vae, nmf_rank, eps, device):
        self._init_common(X, S, y, vae, nmf_rank, eps, device)

    def sample_posterior(self, Z, y, nsamples=10, burnin=30):
        Zs = self._chain(Z, y, nsamples, burnin)
        N, R, _ = Zs.shape
        Zs_y = torch.cat([Zs, torch.t(y).unsqueeze(1).expand(N, R, y.shape[0])], dim=2)
        self._cached = (Zs_y, self._cached_vs)
        return Zs, Zs_y

    def E_step(self):
        Z_t, Z_y_t = self.sample_posterior(self.Z, self.y, self.nsamples_E_step, self.burnin_E_step)
        self.Z = torch.t(torch.squeeze(Z_t[:, -1, :]))
        self.compute_Vs(Z_y_t)
        self.compute_Vs_scaled()
        self.compute_Vx()

    def compute_WF(self, sample=False):
        if sample:
            Z_t, Z_y_t = self.sample_posterior(self.Z, self.y, self.nsamples_WF, self.burnin_WF)
            self.compute_Vs(Z_y_t)
            self.compute_Vs_scaled()
            self.compute_Vx()
        return _wiener(self)


class MCEM_M2v2(MCEM_M2):
    """Encoder sees the spectrogram only; the decoder is still conditioned on y."""
    _label_in_encoder = False


class MCEM_M2v3(MCEM_M2v2):
    """Same data flow as M2v2; used with DeepGenerativeModel_v5.enc_dec_clf (evaluate_ntcd_M2_info_vad.py:324)."""


def _wiener(em):
    """Posterior-mean Wiener gains (reference mcem.py:321-327): mean over draws of g Vs / Vx and Vb / Vx."""
    if em._on_device() and em.Vs.ndim == 3:
        return _native.mcem_dev().wiener(em.Vs, em.g, em.Vb)
    return torch.mean(em.Vs_scaled / em.Vx, axis=0), torch.mean(em.Vb / em.Vx, axis=0)
