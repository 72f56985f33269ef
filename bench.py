#!/usr/bin/env python3
"""bench.py -- spectrogram frames/sec of one full train step (forward + ELBO + backward +
Adam; + gradient all-reduce when N > 1) of the M2 VAE, 513 bins, on N MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of B synthetic frames per GPU
(weak scaling: B per GPU is fixed).  Inputs are resident in HBM before the timed region
(cycled from a >= 1 GB device pool so they are not cache resident).  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch

# exact train FLOPs per frame (fwd + dW + dX, no dX into data): SURVEY.md 8d / BASELINE.md 2
TRAIN_FLOPS = {("M1", 0): 890112, ("M2", 1): 891136, ("M2", 513): 1415424, ("M2_info", 1): 1475072}
# mandatory HBM bytes per frame for a fused step (x, y, eps read once, fp32)
MIN_BYTES = {("M1", 0): 2116, ("M2", 1): 2120, ("M2", 513): 4168, ("M2_info", 1): 2120}
PEAK = {"hbm": (8000.0, "GB/s"), "mfma_f32": (157.3, "TFLOP/s"), "mfma_bf16": (2500.0, "TFLOP/s")}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--model", default="M2", choices=["M1", "M2", "M2_info"])
    ap.add_argument("--y-dim", type=int, default=None, help="label width (M2 default 513 = IBM labels, the script default)")
    ap.add_argument("--batch", type=int, default=8192, help="frames per step per GPU")
    ap.add_argument("--impl", default="auto", choices=["auto", "fused", "modules"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3", "fp32"])
    ap.add_argument("--pool-gb", type=float, default=1.0)
    ap.add_argument("--ksplit", type=int, default=0, help="frame-axis slices of the weight-gradient kernel (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def synth_batches(dims, B, nb, seed, device):
    """SURVEY.md 8d synthetic frames generated on the device (plumbing, outside the timed region)."""
    g = torch.Generator(device=device).manual_seed(seed)
    xd, yd, zd = dims["x_dim"], dims["y_dim"], dims["z_dim"]
    out = []
    for _ in range(nb):
        n1 = torch.randn((B, xd), generator=g, device=device)
        n2 = torch.randn((B, xd), generator=g, device=device)
        n3 = torch.randn((B, xd), generator=g, device=device)
        x = (torch.exp(4 * n1 - 8) * (n2 * n2 + n3 * n3) / 2).clamp_(1e-12, 1e4)
        y = None
        if yd:
            y = (torch.rand((B, yd), generator=g, device=device) < (0.6 if yd == 1 else 0.3)).float()
        e = torch.randn((B, zd), generator=g, device=device)
        out.append((x, y, e))
    return out


class ModulesImpl:
    """The drop-in path: packages.models modules + autograd Functions + stock torch.optim.Adam."""
    name = "modules(layer-level HIP kernels + torch.optim.Adam)"

    def __init__(self, model, dims, device, world):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from impl_modules import build_model
        from packages.models import models as M
        from packages.models.utils import elbo, binary_cross_entropy
        self.M, self.elbo, self.bce = M, elbo, binary_cross_entropy
        torch.manual_seed(0)
        self.model = model
        self.m = build_model(model, dims).to(device)
        self.world = world
        if model == "M2_info":
            self.opt = torch.optim.Adam(self.m.enc_dec_clf.parameters(), lr=1e-4, betas=(0.9, 0.999))
            self.opt_aux = torch.optim.Adam(self.m.auxiliary.parameters(), lr=1e-4, betas=(0.9, 0.999))
        else:
            self.opt = torch.optim.Adam(self.m.parameters(), lr=1e-4, betas=(0.9, 0.999))
        self.dtype = "f32"

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
            self._allreduce(list(m.parameters()))
            self.opt.step(); self.opt.zero_grad()
            return loss
        yc = m.classify_fromX(x)
        r, z, mu, lv = m(x, y)
        ELBO, recon, kl = self.elbo(x, r, mu, lv, 1e-8)
        enc_loss = ELBO + 0.0 * self.bce(yc, y, 1e-8) - 10.0 * self.bce(m.classify_fromZ(z), y, 1e-8)
        aux_loss = 1.0 * self.bce(m.classify_fromZ(z.detach()), y, 1e-8)
        enc_loss.backward()
        self._allreduce(list(m.enc_dec_clf.parameters()))
        self.opt.step(); self.opt.zero_grad()
        aux_loss.backward()
        self._allreduce(list(m.auxiliary.parameters()))
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


def cpu_baseline(model, dims, B, seconds):
    """The reference's CPU loop body (oracle/torch_ref.py, kind "port") on this node's host cores."""
    from oracle import torch_ref as tr
    import golden_util as gu
    ncores = host_cores()
    torch.set_num_threads(ncores)
    d = dict(x_dim=dims["x_dim"], y_dim=max(dims["y_dim"], 1) if model != "M1" else 1, z_dim=dims["z_dim"], h_dim=dims["h_dim"])
    p = tr.init_params(model, seed=0, **d)
    st = tr.Stepper(model, p)
    xs = []
    for i in range(2):
        x, y, e = gu.make_batch(dims, B, 4321 + i)
        xs.append((torch.from_numpy(x), None if y is None else torch.from_numpy(y), torch.from_numpy(e)))
    for i in range(2):
        st.step(*xs[i % 2])
    n, t0 = 0, time.perf_counter()
    while True:
        st.step(*xs[n % 2]); n += 1
        dt = time.perf_counter() - t0
        if dt > seconds or n >= 400:
            break
    return {"value": B * n / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} steps of {B} frames ({model}, fp32, torch {torch.__version__} CPU ops + torch.optim.Adam), {dt:.1f} s"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)          # one rank per GPU on the node; wraps only on test rigs with fewer GPUs
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DVAE_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; gloo only for single-GPU rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    y_dim = a.y_dim if a.y_dim is not None else {"M1": 0, "M2": 513, "M2_info": 1}[a.model]
    dims = dict(x_dim=513, y_dim=y_dim, z_dim=16, h_dim=(128, 128))
    B = a.batch
    bytes_per_batch = B * (513 + y_dim + 16) * 4
    nb = max(2, min(64, int(a.pool_gb * 2 ** 30 / bytes_per_batch) + 1))
    batches = synth_batches(dims, B, nb, 1234 + rank, device)

    impl_name = a.impl
    trainer_mod = None
    if impl_name in ("auto", "fused"):
        try:
            trainer_mod = importlib.import_module("disentangled-vae_amd.trainer")
        except ModuleNotFoundError:
            if impl_name == "fused":
                raise
        impl_name = "fused" if trainer_mod is not None else "modules"
        if trainer_mod is not None and not trainer_mod.supported(a.model, dims):
            if a.impl == "fused":
                raise SystemExit(f"fused train step does not cover {a.model} {dims}")
            impl_name = "modules"        # e.g. M2_info: layer-level HIP kernels + autograd + torch.optim.Adam
    if impl_name == "fused":
        impl = trainer_mod.BenchImpl(a.model, dims, B, device, world, a.precision, ksplit=a.ksplit)
    else:
        impl = ModulesImpl(a.model, dims, device, world)

    for i in range(a.warmup):
        impl.step(*batches[i % nb])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        last = impl.step(*batches[(a.warmup + i) % nb])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(last.reshape(-1)[0].item()) if last is not None else float("nan")

    prof = impl.kernel_profile(batches, min(a.steps, 50))
    key = (a.model, y_dim)
    roofline = None
    if prof is not None:
        roofline = prof
    out = {
        "metric": "spectrogram frames/sec (train step), M2 VAE 513-bin" if a.model == "M2" else f"spectrogram frames/sec (train step), {a.model} VAE 513-bin",
        "value": world * B * a.steps / dt,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
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
        "roofline": roofline,
    }
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.model, dims, B, a.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
