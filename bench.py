#!/usr/bin/env python3
"""bench.py -- spectrogram frames/sec of one full train step (forward + ELBO + backward +
Adam; + gradient all-reduce when N > 1) of the M2 VAE, 513 bins, on N MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          # no launcher: bench.py starts the N ranks itself (launch_ranks) and relays rank 0's line;
                                          # fewer than N GPUs under RCCL, or a WORLD_SIZE that is not N, is an error, never a dp1 run

A "step" is one pass of the hot path over one batch of B synthetic frames per GPU
(weak scaling: B per GPU is fixed).  Inputs are resident in HBM before the timed region
(cycled from a >= 1 GB device pool so they are not cache resident).  Rank 0 prints ONE JSON line.

Default = the parity-grade throughput mode: bf16x3 (split-bf16 MFMA operands, three MFMAs per product, fp32 accumulate and
master weights; losses ~1e-7 and gradients ~1e-4 of their maximum against float64, tests/test_gpu_fused.py).  The same line
carries, measured in the same process (N = 1): `parity_mode` (the exact-fp32-MFMA policy and the one-bf16-per-operand fast
mode), `spread` (5 more repeats of the timed region), `b128` (the reference scripts' batch size, GPU and CPU) and
`cpu_baseline` (oracle/torch_ref.py on the host cores: all granted cores, and one thread).
"""
import argparse
import importlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

# exact train FLOPs per frame (fwd + dW + dX, no dX into data): SURVEY.md 8d / BASELINE.md 2
TRAIN_FLOPS = {("M1", 0): 890112, ("M2", 1): 891136, ("M2", 513): 1415424, ("M2_info", 1): 1475072}
# mandatory HBM bytes per frame for a fused step (x, y, eps read once, fp32)
MIN_BYTES = {("M1", 0): 2116, ("M2", 1): 2120, ("M2", 513): 4168, ("M2_info", 1): 2120}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--model", default="M2", choices=["M1", "M2", "M2_info"])
    ap.add_argument("--y-dim", type=int, default=None, help="label width (M2 default 513 = IBM labels, the script default)")
    ap.add_argument("--batch", type=int, default=8192, help="frames per step per GPU")
    ap.add_argument("--impl", default="auto", choices=["auto", "fused", "modules"])
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "bf16", "fp32"],
                    help="MFMA operand policy of the fused path: bf16x3 = split bf16 (parity grade), bf16 = one bf16 per operand (fast, loose), fp32 = exact fp32 MFMA")
    ap.add_argument("--pool-gb", type=float, default=1.0)
    ap.add_argument("--ksplit", type=int, default=0, help="frame-axis slices of the weight-gradient kernel (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip parity_mode / spread / b128 (they run outside the timed region)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--prewarm-ms", type=float, default=200.0,
                    help="untimed train steps for this long BEFORE the W warmup steps, so that the K timed steps run at the clocks a training run "
                         "sees (from idle the shader clock needs tens of ms of load to settle: 20 steps right after 5 measure 87.6 us/step, "
                         "the same 20 steps after 200 ms of steps 80); reported as `prewarm_steps`; 0 disables")
    ap.add_argument("--direct-trial", action="store_true",
                    help="N > 1, opt-in (also DVAE_BENCH_DIRECT_TRIAL=1): after the RCCL measurement, a second timed region with the library's own "
                         "gradient exchange (dvae_allreduce_flat), reported under multi_gpu.direct; the headline value is always the RCCL run.  "
                         "Off by default: the exchange is unmeasured on multi-GPU hardware and a device fault there would lose the headline line")
    return ap.parse_args()


class ModulesImpl:
    """The drop-in path: packages.models modules + autograd Functions + stock torch.optim.Adam."""
    name = ("modules(packages.models drop-in modules + torch autograd + stock torch.optim.Adam; M1 / M2 forward+backward as one Function "
            "on the fused kernels unless DVAE_MODULE_PATH=layers)")

    def __init__(self, model, dims, device, world):
        synth = importlib.import_module("disentangled-vae_amd.synth")
        from packages.models import models as M
        from packages.models.utils import elbo, binary_cross_entropy
        self.M, self.elbo, self.bce = M, elbo, binary_cross_entropy
        torch.manual_seed(0)
        self.model = model
        self.m = synth.build_model(model, dims).to(device)
        self.world = world
        if model == "M2_info":
            self.opt = torch.optim.Adam(self.m.enc_dec_clf.parameters(), lr=1e-4, betas=(0.9, 0.999))
            self.opt_aux = torch.optim.Adam(self.m.auxiliary.parameters(), lr=1e-4, betas=(0.9, 0.999))
        else:
            self.opt = torch.optim.Adam(self.m.parameters(), lr=1e-4, betas=(0.9, 0.999))
        self.dtype = "bf16x3" if os.environ.get("DVAE_MODULE_PATH", "fused") != "layers" and model != "M2_info" else "f32"
        self._plist = list(self.m.parameters()) if model != "M2_info" else None
        self._pl_edc = list(self.m.enc_dec_clf.parameters()) if model == "M2_info" else None
        self._pl_aux = list(self.m.auxiliary.parameters()) if model == "M2_info" else None

    def _allreduce(self, params):
        if self.world == 1:
            return
        import torch.distributed as dist
        gs = [p.grad for p in params if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in gs])
        dist.all_reduce(flat)
        flat.div_(self.world)
        o = 0
        for g in gs:
            g.copy_(flat[o:o + g.numel()].view_as(g)); o += g.numel()

    def step(self, x, y, e):
        M = self.M
        M.Stochastic.epsilon_fn = lambda mu: e
        m = self.m
        if self.model != "M2_info":
            r, mu, lv = m(x) if self.model == "M1" else m(x, y)
            loss, recon, kl = self.elbo(x, r, mu, lv, 1e-8)
            loss.backward()
            self._allreduce(self._plist)
            self.opt.step(); self.opt.zero_grad()
            return loss
        yc = m.classify_fromX(x)
        r, z, mu, lv = m(x, y)
        ELBO, recon, kl = self.elbo(x, r, mu, lv, 1e-8)
        enc_loss = ELBO + 0.0 * self.bce(yc, y, 1e-8) - 10.0 * self.bce(m.classify_fromZ(z), y, 1e-8)
        aux_loss = 1.0 * self.bce(m.classify_fromZ(z.detach()), y, 1e-8)
        enc_loss.backward()
        self._allreduce(self._pl_edc)
        self.opt.step(); self.opt.zero_grad()
        aux_loss.backward()
        self._allreduce(self._pl_aux)
        self.opt_aux.step(); self.opt_aux.zero_grad()
        return ELBO

    def kernel_profile(self, batches, steps):
        return None


def host_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box exposes 256 logical CPUs but grants a 16-CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_steps_per_s(model, dims, B, seconds, threads, max_steps=400):
    """The reference's CPU loop body (oracle/torch_ref.py, kind "port") for `seconds` on `threads` host threads."""
    from oracle import torch_ref as tr
    synth = importlib.import_module("disentangled-vae_amd.synth")
    torch.set_num_threads(threads)
    d = dict(x_dim=dims["x_dim"], y_dim=max(dims["y_dim"], 1) if model != "M1" else 1, z_dim=dims["z_dim"], h_dim=dims["h_dim"])
    p = tr.init_params(model, seed=0, **d)
    st = tr.Stepper(model, p)
    xs = []
    for i in range(2):
        x, y, e = synth.make_batch(dims, B, 4321 + i)
        xs.append((torch.from_numpy(x), None if y is None else torch.from_numpy(y), torch.from_numpy(e)))
    for i in range(2):
        st.step(*xs[i % 2])
    n, t0 = 0, time.perf_counter()
    while True:
        st.step(*xs[n % 2]); n += 1
        dt = time.perf_counter() - t0
        if dt > seconds or n >= max_steps:
            break
    return B * n / dt, n, dt, torch.get_num_threads()


def cpu_baseline(model, dims, B, seconds):
    ncores = host_cores()
    v, n, dt, thr = cpu_steps_per_s(model, dims, B, seconds, ncores)
    v1, n1, dt1, _ = cpu_steps_per_s(model, dims, B, max(4.0, 0.6 * seconds), 1, max_steps=60)
    torch.set_num_threads(ncores)
    what = f"({model}, fp32, torch {torch.__version__} CPU ops + torch.optim.Adam)"
    return {"value": v, "unit": "frames/s", "cores": thr, "kind": "port",
            "sample": f"{n} steps of {B} frames {what}, {dt:.1f} s",
            "single_thread": {"value": v1, "unit": "frames/s", "cores": 1, "sample": f"{n1} steps of {B} frames {what}, {dt1:.1f} s"}}


def side_kernels(device):
    """The hot path's other kernels on this GPU, so that a driver run carries them: STFT / ISTFT of ten minutes of
    float64 audio (HBM-bound: algorithmic bytes per frame = 256 new samples in + 513 complex64 out) and one batched MCEM
    run (25 utterances of 300 frames side by side, fp32 parity policy: MFMA-bound decoder chains)."""
    import numpy as np
    H = importlib.import_module("disentangled-vae_amd.stft")
    M = importlib.import_module("disentangled-vae_amd.mcem")
    synth = importlib.import_module("disentangled-vae_amd.synth")

    def t_us(fn, n=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / n
    out = {}
    n = 16000 * 600
    x = torch.randn(n, dtype=torch.float64, device=device)
    T = H.frame_count(n, 1024, 256)
    w = H.window_f64("hann", 1024, device)
    S = H.stft_device(x, w, 1024, 256, T, 0)
    Sr = H.stft_device(x, w, 1024, 256, T, 2).T                           # frame-major memory: what stft() / istft() of the drop-in use
    for name, fn, byts in (("stft_complex", lambda: H.stft_device(x, w, 1024, 256, T, 2), T * (256 * 8 + 513 * 8)),
                           ("stft_complex_bin_major", lambda: H.stft_device(x, w, 1024, 256, T, 0), T * (256 * 8 + 513 * 8)),
                           ("stft_power_frames", lambda: H.stft_device(x, w, 1024, 256, T, 1), T * (256 * 8 + 513 * 4)),
                           ("istft", lambda: H.istft_device(Sr, w, 1024, 256, T, 0, n), T * (513 * 8 + 256 * 4)),
                           ("istft_bin_major", lambda: H.istft_device(S, w, 1024, 256, T, 0, n), T * (513 * 8 + 256 * 4))):
        us = t_us(fn)
        out[name] = {"us": us, "frames": T, "Mframes_per_s": T / us, "algorithmic_GB_per_s": byts / us * 1e-3, "hbm_frac": byts / us * 1e-3 / 8000.0}
    xf = x.to(torch.float32)
    for name, fn, byts in (("stft_complex_f32arith", lambda: H.stft_device_f32(xf, 1024, 256, T, 2), T * (256 * 4 + 513 * 8)),
                           ("stft_power_frames_f32arith", lambda: H.stft_device_f32(xf, 1024, 256, T, 1), T * (256 * 4 + 513 * 4))):
        us = t_us(fn)
        out[name] = {"us": us, "frames": T, "Mframes_per_s": T / us, "algorithmic_GB_per_s": byts / us * 1e-3, "hbm_frac": byts / us * 1e-3 / 8000.0,
                     "note": "float32 signal, float32 arithmetic: the transform of stft_pytorch (torch.stft on a float32 tensor), dvae_stft_f32"}
    us = t_us(lambda: H.istft_device_f32(Sr, 1024, 256, T, 0, n))
    byts = T * (513 * 8 + 256 * 4)
    out["istft_f32arith"] = {"us": us, "frames": T, "Mframes_per_s": T / us, "algorithmic_GB_per_s": byts / us * 1e-3, "hbm_frac": byts / us * 1e-3 / 8000.0,
                             "note": "complex64 spectrogram, float32 arithmetic: the transform of istft_pytorch (torch.istft on a complex64 tensor), dvae_istft_f32"}
    out["stft_note"] = "600 s of float64 audio at 16 kHz, nfft 1024, hop 256; bytes = new samples in + spectrogram out (float32 samples out for istft); stft_complex / istft: the frame-major spectrogram memory of packages.processing.stft (a Fortran-ordered [513, T] array, as librosa's), *_bin_major: the row-contiguous [513, T] form (stft_pytorch / device tensors)"
    # MCEM: scripts/evaluate_ntcd_M2.py settings (10 + 30 samples per E-step, rank 10), 25 utterances x 300 frames, 20 EM iterations
    dims = dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128))
    torch.manual_seed(0)
    vae = synth.build_model("M2", dims).to(device).eval()
    for p in vae.parameters():
        p.requires_grad = False
    rng = np.random.default_rng(5)
    F, N, U, niter = 513, 300, 25, 20
    env = np.exp(rng.standard_normal((F, 1)) * 0.7 - 1.0) * np.exp(rng.standard_normal((1, N)) * 0.5)
    Sx = np.sqrt(env / 2) * (rng.standard_normal((F, N)) + 1j * rng.standard_normal((F, N)))
    X = (Sx + 0.3 * (rng.standard_normal((F, N)) + 1j * rng.standard_normal((F, N)))).astype(np.complex64)
    y = torch.from_numpy((rng.random((1, N)) > 0.4).astype(np.float32)).to(device)
    # decoder frame passes the kernels execute at 25 utterances (32-frame tiles, csrc/mcem_resident.hip): per chain the evaluation of the
    # initial state + one pass per step, + under bf16x3 one decoder pass per kept sample, under exact fp32 ONE (the state at the end of the
    # burn-in: kept steps store their proposal's variances, rejected slots are copied); the reference's second decoder pass per step is
    # replaced by the kept likelihood; frames padded to 32
    Np = -(-N // 32) * 32
    per_chain = {"fp32": lambda steps, kept: steps + 2, "bf16x3": lambda steps, kept: steps + 1 + kept}
    for prec, peak in (("fp32", 157.3), ("bf16x3", 2500.0 / 3.0)):      # exact fp32 products / split-bf16 operands (three MFMAs per product)
        passes = U * Np * (niter * per_chain[prec](40, 10) + per_chain[prec](100, 25))
        flops = passes * 2.0 * (17 * 128 + 128 * 128 + 128 * 513)
        mb = M.McemBatch(vae, niter=2, nsamples_E_step=10, burnin_E_step=30, nsamples_WF=25, burnin_WF=75, label_in_encoder=True,
                         label_in_decoder=True, precision=prec)
        mb.init_parameters([X] * U, [y] * U); mb.run()
        mb.niter = niter
        dts = []
        for _ in range(3):                                   # the median of three timed runs (single runs of 50 ms spread by +- 10 % on one box)
            mb.init_parameters([X] * U, [y] * U)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            mb.run()
            torch.cuda.synchronize(); dts.append(time.perf_counter() - t0)
        dt = sorted(dts)[1]
        out["mcem_batched_" + prec] = {"utterances": U, "frames_each": N, "em_iterations": niter, "seconds": dt, "seconds_runs": [round(x, 5) for x in dts],
                                        "ms_per_em_iteration": 1e3 * dt / niter,
                                        "utterances_per_s": U / dt, "decoder_TFLOP_per_s": flops / dt * 1e-12, "mfma_frac": flops / dt * 1e-12 / peak,
                                        "mfma_peak_TFLOP_per_s": peak,
                                        "note": "whole run incl. M-steps and the Wiener chain; flops = decoder MACs x 2 of the chain / decode passes the kernels execute"}
    # ONE utterance -- what scripts/evaluate_ntcd_M2.py runs per process: the chain on 4-frame tiles under exact fp32 (csrc/mcem_resident4.hip,
    # the drop-in classes' default policy), on 16-frame tiles under bf16x3 (csrc/mcem_resident16.hip)
    for prec in ("fp32", "bf16x3"):
        mb = M.McemBatch(vae, niter=2, nsamples_E_step=10, burnin_E_step=30, nsamples_WF=25, burnin_WF=75, label_in_encoder=True,
                         label_in_decoder=True, precision=prec)
        mb.init_parameters([X], [y]); mb.run()
        mb.niter = 100
        mb.init_parameters([X], [y])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mb.run()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out["mcem_single_" + prec] = {"utterances": 1, "frames_each": N, "em_iterations": 100, "ms_per_utterance": 1e3 * dt, "ms_per_em_iteration": 10.0 * dt,
                                      "note": "the reference's settings (100 EM iterations, 10 + 30 chain steps each, final Wiener chain 25 + 75)"}
    return out


def direct_trial(a, trainer_mod, dims, B, device, world, dist, batches, impl):
    """N > 1, after the RCCL measurement: the same K steps with the library's own stream-ordered exchange (dvae_allreduce_flat over
    hipIpc-mapped peer buffers, include/dvae_train.h) on a second trainer, so that one driver run carries both.  Every stage that can
    fail on one rank is agreed on by all ranks before the next (no rank is left in a collective); the exchange is first checked
    against the process group's all-reduce on a random vector.  Never the headline value."""
    dp = importlib.import_module("disentangled-vae_amd.dp")
    rec = {"exchange": "dvae_allreduce_flat (reduce-scatter by pull + all-gather by push over peer pointers, one launch per rank on the step's stream)"}
    n = int(impl.tr.plan.n_params)
    dx, why = dp.DirectExchange.try_create(n, dist.group.WORLD)
    if dx is None:
        rec["error"] = "setup: " + str(why)
        return rec

    def agree(ok):      # logical AND over the ranks
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))
    try:
        g = torch.Generator(device="cpu").manual_seed(4321 + dist.get_rank())
        v = torch.randn(n, generator=g).to(device)
        got = torch.empty_like(v)
        dx.allreduce(v, 1, n, got)
        ref = v.clone()
        dp.allreduce_flat_(ref, dist.group.WORLD)
        torch.cuda.synchronize()
        dev_ = float((got - ref).abs().max().item())
        ok = (not dx.failed()) and dev_ <= 1e-5 * float(ref.abs().max().item())
        rec["check_max_abs_dev_vs_process_group"] = dev_
    except Exception as e:                              # noqa: BLE001
        ok = False
        rec["error"] = "check: " + str(e)
    if not agree(ok):
        rec.setdefault("error", "check: the exchanged vector differs from the process group's sum, or a wait ran out (on some rank)")
        dx.close()
        return rec
    impl2 = None
    try:                                                 # a rank-local failure up to here must not leave the peers in timed_steps' barrier
        impl2 = trainer_mod.BenchImpl(a.model, dims, B, device, world, a.precision, ksplit=a.ksplit, direct_exchange=dx)
        ok = True
    except Exception as e:                              # noqa: BLE001
        ok = False
        rec["error"] = "trainer: " + str(e)
    if not agree(ok):
        rec.setdefault("error", "trainer: construction failed on some rank")
        dx.close()
        return rec
    try:
        for i in range(max(a.warmup, 20)):
            impl2.step(*batches[i % len(batches)])
        torch.cuda.synchronize()
        ok = not dx.failed()
    except Exception as e:                              # noqa: BLE001
        ok = False
        rec["error"] = "warmup: " + str(e)
    if not agree(ok):
        rec.setdefault("error", "warmup: a step failed or a bounded wait ran out on some rank")
        del impl2                                        # the trainer holds the exchange: drop it, then unmap the peers' buffers
        dx.close()
        return rec
    try:
        dt2, _ = timed_steps(impl2, batches, a.warmup, a.steps, dist, device)
        ok = not dx.failed()
        rec.update({"ms_per_step": 1e3 * dt2 / a.steps, "value": world * B * a.steps / dt2, "steps": a.steps, "wait_ran_out": not ok})
        impl2.tr.profile(True)
        for i in range(min(a.steps, 50)):
            impl2.step(*batches[i % len(batches)])
        impl2.tr.profile_read()
        rec["allreduce_us"] = impl2.tr.allreduce_us()
        impl2.tr.profile(False)
    except Exception as e:                              # noqa: BLE001
        rec["error"] = "timed region: " + str(e)
    return rec


def timed_steps(impl, batches, first, steps, dist, device):
    """K steps bracketed by barrier + synchronize on both sides; max over ranks.  Returns (seconds, last loss tensor)."""
    nb = len(batches)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        last = impl.step(*batches[(first + i) % nb])
    if hasattr(impl, "finish"):
        impl.finish()                        # the deferred optimizer update of the last step: inside the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, last


def launch_ranks(a):
    """`bench.py --gpus N` with N > 1 and no launcher (WORLD_SIZE unset): this process starts N fresh ranks -- `python -m
    torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a CHILD process -- before it has made a single GPU call
    (torch.cuda.device_count() does not initialise HIP on this image; nothing else here touches the device, and nothing is ever exec'ed
    over this process), relays rank 0's one JSON line and exits with the launcher's code.  Fewer than N devices under the RCCL backend is
    an error, never a silent one-GPU run (the reference is single-device, scripts/training_M2.py:31-33: the N > 1 line is this build's
    own claim and must be what it says it is)."""
    import socket
    import subprocess
    backend = os.environ.get("DVAE_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < a.gpus:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} over RCCL needs {a.gpus} visible GPUs, this node shows {ndev}; refusing to measure fewer "
                         f"(DVAE_DIST_BACKEND=gloo rehearses the flow with the ranks sharing the visible devices)\n")
        raise SystemExit(2)
    if ndev < 1:
        sys.stderr.write("bench.py needs an MI355X (no CPU fallback for the hot path)\n")
        raise SystemExit(2)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: no launcher in the environment, starting the ranks: " + " ".join(cmd) + "\n")
    sys.stderr.flush()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in p.stdout:
        if ln.startswith('{"metric"'):
            line = ln.strip()                                     # rank 0's line (only rank 0 prints one)
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if rc != 0:
        sys.stderr.write(f"bench.py: the launcher exited with code {rc}\n")
        raise SystemExit(rc if 0 < rc < 256 else 1)
    if line is None:
        sys.stderr.write("bench.py: the ranks printed no result line\n")
        raise SystemExit(1)
    rec = json.loads(line)
    if rec.get("n_gpus") != a.gpus or (rec.get("multi_gpu") or {}).get("nranks") != a.gpus:
        sys.stderr.write(f"bench.py: asked for {a.gpus} ranks, the line says n_gpus {rec.get('n_gpus')}\n")
        raise SystemExit(1)
    print(line, flush=True)


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        launch_ranks(a)
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: start one rank per GPU (python -m torch.distributed.run --nproc-per-node {a.gpus} "
                         f"bench.py --gpus {a.gpus} ...), or unset WORLD_SIZE and bench.py starts them itself")
    backend = os.environ.get("DVAE_DIST_BACKEND", "nccl") if world > 1 else None     # "nccl" is RCCL on ROCm; gloo only for single-GPU rehearsals
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < world:
        raise SystemExit(f"--gpus {world} over RCCL needs {world} visible GPUs, this node shows {ndev}: refusing to let ranks share a device")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    dev_index = local_rank % max(ndev, 1)          # one rank per GPU on the node; wraps only under the gloo rehearsal backend (checked above)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"process group of {dist.get_world_size()} ranks, --gpus {a.gpus}")
    synth = importlib.import_module("disentangled-vae_amd.synth")
    y_dim = a.y_dim if a.y_dim is not None else {"M1": 0, "M2": 513, "M2_info": 1}[a.model]
    dims = dict(x_dim=513, y_dim=y_dim, z_dim=16, h_dim=(128, 128))
    B = a.batch
    bytes_per_batch = B * (513 + y_dim + 16) * 4
    nb = max(2, min(64, int(a.pool_gb * 2 ** 30 / bytes_per_batch) + 1))
    batches = synth.device_batches(dims, B, nb, 1234 + rank, device)

    impl_name = a.impl
    trainer_mod = None
    if impl_name in ("auto", "fused"):
        trainer_mod = importlib.import_module("disentangled-vae_amd.trainer")
        impl_name = "fused"
        if not trainer_mod.supported(a.model, dims):
            if a.impl == "fused":
                raise SystemExit(f"fused train step does not cover {a.model} {dims}")
            impl_name = "modules"
    if impl_name == "fused":
        impl = trainer_mod.BenchImpl(a.model, dims, B, device, world, a.precision, ksplit=a.ksplit)
    else:
        impl = ModulesImpl(a.model, dims, device, world)

    prewarm_steps = 0
    if a.prewarm_ms > 0:                                                            # untimed: bring the GPU to its sustained clocks
        torch.cuda.synchronize()
        t_pw = time.perf_counter()
        while True:
            for i in range(50):
                impl.step(*batches[(prewarm_steps + i) % nb])
            prewarm_steps += 50
            torch.cuda.synchronize()
            go = torch.tensor([1 if (time.perf_counter() - t_pw) * 1e3 < a.prewarm_ms else 0], dtype=torch.int32,
                              device=device if backend in (None, "nccl") else "cpu")
            if dist is not None:
                dist.broadcast(go, src=0)                                           # every rank makes the same number of steps (they all-reduce)
            if int(go.item()) == 0:
                break
    for i in range(a.warmup):
        impl.step(*batches[i % nb])
    dt, last = timed_steps(impl, batches, a.warmup, a.steps, dist, device)          # THE timed region: exactly K steps
    final_loss = float(last.reshape(-1)[0].item()) if last is not None else float("nan")

    extras = not a.no_extras
    spread = None
    if extras:                                                                      # run-to-run spread inside the process
        reps = [1e3 * timed_steps(impl, batches, a.warmup + (r + 1) * a.steps, a.steps, dist, device)[0] / a.steps for r in range(5)]
        spread = {"repeats": 5, "steps_each": a.steps, "ms_per_step_min": min(reps), "ms_per_step_median": statistics.median(reps),
                  "ms_per_step_max": max(reps), "note": "five more repeats of the timed region, after it"}

    prof = impl.kernel_profile(batches, min(a.steps, 50))
    direct_rec = None
    if (world > 1 and impl_name == "fused" and os.environ.get("DVAE_ALLREDUCE", "rccl") == "rccl"
            and (a.direct_trial or os.environ.get("DVAE_BENCH_DIRECT_TRIAL", "0") == "1")):
        direct_rec = direct_trial(a, trainer_mod, dims, B, device, world, dist, batches, impl)
    key = (a.model, y_dim)
    out = {
        "metric": "spectrogram frames/sec (train step), M2 VAE 513-bin" if a.model == "M2" else f"spectrogram frames/sec (train step), {a.model} VAE 513-bin",
        "value": world * B * a.steps / dt,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "prewarm_steps": prewarm_steps,
        "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": impl.dtype,
        "data": "synthetic",
        "config": {"workload": f"{a.model} VAE train step (fwd+ELBO+bwd+Adam), x_dim 513, y_dim {y_dim}, z 16, h [128,128], "
                               f"{B} frames/step/GPU, 16 kHz / 1024-pt STFT power frames",
                   "frames_per_step_per_gpu": B, "global_frames_per_step": B * world, "impl": impl.name,
                   "parallelism": f"dp{world}", "final_elbo": final_loss,
                   "train_flops_per_frame": TRAIN_FLOPS.get(key), "min_hbm_bytes_per_frame": MIN_BYTES.get(key)},
        "roofline": prof,
        "spread": spread,
    }
    if world > 1:
        devs = [None] * world
        dist.all_gather_object(devs, dev_index)
        out["multi_gpu"] = {"nranks": dist.get_world_size(), "backend": dist.get_backend(), "devices": len(set(devs)), "direct": direct_rec,
                            "exchange": os.environ.get("DVAE_ALLREDUCE", "rccl") + (" (dvae_allreduce_flat: peer pointers, one launch per rank)"
                                                                                    if os.environ.get("DVAE_ALLREDUCE") == "direct" else " (torch.distributed all_reduce)"),
                            "allreduce_us": None if prof is None else prof.get("allreduce_us"),
                            "allreduce_bytes": None if impl_name != "fused" else 4 * int(impl.tr.plan.n_params),
                            "note": "allreduce_us = mean device time of the flat-gradient all-reduce per step (events on the launch stream, rank 0)"}
    if rank == 0:
        if world == 1 and extras and impl_name == "fused":
            def other(precision, Bx, steps):
                bs = synth.device_batches(dims, Bx, 4, 99, device)
                o = trainer_mod.BenchImpl(a.model, dims, Bx, device, 1, precision)
                for i in range(20):
                    o.step(*bs[i % 4])
                t, _ = timed_steps(o, bs, 0, steps, None, device)
                return 1e3 * t / steps
            pm = {}
            for prec in ("fp32", "bf16", "bf16x3"):
                if prec != a.precision:
                    pm[prec + "_ms_per_step"] = other(prec, B, min(a.steps, 100))
            pm["note"] = ("same workload, other MFMA operand policies, measured in this process: fp32 = exact fp32 MFMA (<= 1e-4 on everything); "
                          "bf16 = one bf16 per operand (gradients within ~4e-2 of their maximum); the headline policy is " + a.precision)
            out["parity_mode"] = pm
            g128 = other(a.precision, 128, 200)
            out["b128"] = {"frames_per_step": 128, "gpu_ms_per_step": g128, "gpu_frames_per_s": 128 / (g128 * 1e-3),
                           "note": "the batch size of scripts/training_M2.py:60"}
        if world == 1 and extras and impl_name == "fused" and B == 8192:
            # the roofline regime of SURVEY 8(d): many tiles per workgroup, the persistent tile loop hides the input load and the launch
            # boundaries; per-frame step time against the MFMA ceiling of issued work and the HBM ceiling of the algorithmic bytes
            try:
                Bl, stepsl = 262144, 10
                bs = synth.device_batches(dims, Bl, 2, 77, device)
                o = trainer_mod.BenchImpl(a.model, dims, Bl, device, 1, a.precision)
                for i in range(4):
                    o.step(*bs[i % 2])
                tl, _ = timed_steps(o, bs, 0, stepsl, None, device)
                usl = 1e6 * tl / stepsl
                pk = 2500.0 if a.precision in ("bf16", "bf16x3") else 157.3
                tf = TRAIN_FLOPS[key] * Bl / (usl * 1e-6) * 1e-12
                out["large_batch"] = {"frames_per_step": Bl, "us_per_step": usl, "frames_per_s": Bl / (usl * 1e-6), "ns_per_frame": 1e3 * usl / Bl,
                                      "hbm_frac": MIN_BYTES[key] * Bl / (usl * 1e-6) / 8.0e12, "mfma_frac": tf / pk,
                                      "mfma_frac_issued": tf / (pk / 3.0) if a.precision == "bf16x3" else tf / pk,
                                      "note": "same model and policy at 262 144 frames per step (fractions: whole step, algorithmic flops / bytes of "
                                              "SURVEY 8d); from 131 072 frames on the weight-gradient kernel reads x and the labels from the input matrices "
                                              "instead of a stash (DVAE_RAW_INPUTS)"}
                del o, bs
                torch.cuda.empty_cache()
            except Exception as exc:
                out["large_batch"] = {"error": repr(exc)}
        if world == 1 and extras and impl_name == "fused" and a.precision in ("bf16", "bf16x3"):
            # north_star: "MFMA utilisation on the encoder GEMM".  The rows kernel stamps its phases with the 100 MHz wall clock
            # (dvae_train_debug_stamps, tools/stamp_rows.py); phase 1 is the x block of encoder layer 1: [B, 513] x [513, 128] on the chain
            # waves' MFMAs, the weights streamed from L2.  One stamped step after the timed region (the stamps cost ~0.3 us of the kernel).
            try:
                import numpy as np
                N_ = importlib.import_module("disentangled-vae_amd.native")
                tr_ = impl.tr
                if tr_.plan.rows_kernel == 2:
                    buf = torch.zeros(tr_.plan.rows_grid * 32, dtype=torch.int64, device=device)
                    bx = batches[0]
                    for _ in range(3):
                        impl.step(*bx)
                    N_.load().dvae_train_debug_stamps(N_.ptr(buf)); impl.step(*bx); impl.finish(); torch.cuda.synchronize(); N_.load().dvae_train_debug_stamps(None)
                    st = buf.cpu().numpy().reshape(-1, 32)[:, :16].astype(np.float64) * 0.01
                    us = float(np.median(st[:, 2] - st[:, 1]))
                    fl = 2.0 * 513 * 128 * B
                    nm = 3.0 if a.precision == "bf16x3" else 1.0
                    out["encoder_gemm"] = {"phase_us": us, "algorithmic_TFLOP_per_s": fl / us * 1e-6, "issued_TFLOP_per_s": nm * fl / us * 1e-6,
                                           "mfma_frac_issued": nm * fl / us * 1e-6 / 2500.0, "mfma_frac_algorithmic": fl / us * 1e-6 / 2500.0,
                                           "mfmas_per_product": nm,
                                           "note": "x block of encoder layer 1 (513 -> 128) over the step's frames, median over workgroups of the rows kernel's "
                                                   "in-kernel phase stamps; issued = MFMA instructions actually executed (split-bf16 operands: hi*hi + lo*hi + hi*lo), "
                                                   "against the dense bf16 peak of 2500 TFLOP/s at 2.4 GHz (the kernel runs at ~2.1)"}
            except Exception as exc:
                out["encoder_gemm"] = {"error": repr(exc)}
        if world == 1 and extras and impl_name == "fused" and a.model == "M2" and B == 8192:
            try:
                out["side_kernels"] = side_kernels(device)
            except Exception as exc:                                                # never lose the headline line to a side measurement
                out["side_kernels"] = {"error": repr(exc)}
        if isinstance(out["roofline"], dict):
            # the driver's record keeps `roofline` whole and drops top-level keys it does not know: the north-star sub-records travel inside it
            for k in ("spread", "encoder_gemm", "large_batch"):
                if out.get(k) is not None:
                    out["roofline"][k] = out[k]
            sk = out.get("side_kernels")
            if isinstance(sk, dict) and "error" not in sk:
                # one figure per side kernel: microseconds per launch (STFT / ISTFT of ten minutes of audio); MCEM: milliseconds per EM iteration of 25 utterances (batched), per utterance (single)
                out["roofline"]["side_kernels"] = {k: (round(v["us"], 1) if "us" in v else round(v.get("ms_per_utterance", v.get("ms_per_em_iteration", 0.0)), 3))
                                                   for k, v in sk.items() if isinstance(v, dict)}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.model, dims, B, a.cpu_seconds)
            if "b128" in out:
                v, n, dtc, thr = cpu_steps_per_s(a.model, dims, 128, 3.0, host_cores())
                out["b128"]["cpu_frames_per_s"] = v
                out["b128"]["cpu_cores"] = thr
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
