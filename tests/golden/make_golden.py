"""Capture golden vectors by running the REFERENCE ITSELF (CPU, fp32) on
deterministic inputs.  Build-container only (imports /root/reference read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py          # -> vae_golden.npz (golden_util.CASES, B <= 32)
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py --big    # -> vae_golden_b8192.npz (golden_util.BIG_CASES: the benchmarked 8192-frame batch)

Output: tests/golden/vae_golden.npz -- data only (inputs are regenerated from
seeds by tests/golden_util.py; a checksum of them is stored).  For each case
of golden_util.CASES it runs NSTEPS iterations of the reference loop body
(scripts/training_M1.py:134-139, scripts/training_M2.py:142-147,
scripts/training_M2_info_vad.py:159-198) with stock torch.optim.Adam and
stores: step-1 outputs (r, z, mu, logvar, kl_divergence, classifier outputs),
every loss scalar of every step, step-1 gradients of every parameter, and the
parameters after step 1 and after step NSTEPS.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import numpy as np
import torch

import golden_util as gu
from packages.models.models import VariationalAutoencoder, DeepGenerativeModel, DeepGenerativeModel_v5
from packages.models.utils import elbo, binary_cross_entropy

EPS = 1e-8
LR = 1e-4
ALPHA, BETA, GAMMA = 0.0, 10.0, 1.0


def build(model, dims):
    h = list(dims["h_dim"])
    if model == "M1":
        return VariationalAutoencoder([dims["x_dim"], dims["z_dim"], h])
    if model == "M2":
        return DeepGenerativeModel([dims["x_dim"], dims["y_dim"], dims["z_dim"], h], None)
    return DeepGenerativeModel_v5([dims["x_dim"], dims["y_dim"], dims["z_dim"], h])


def put(fix, key, arr, stride=gu.SAMPLE_STRIDE):
    for k, v in gu.summarize(arr, stride).items():
        fix[f"{key}/{k}"] = v


def keep(fix, key, arr, big):
    """Step-1 outputs: whole for the small cases; strided sample + moments at 8192 frames."""
    arr = arr.detach().numpy()
    if big:
        put(fix, key, arr, gu.OUT_STRIDE if arr.size > 2 ** 20 else gu.SAMPLE_STRIDE)
    else:
        fix[key] = arr


def model_call(m, model, x, y, eps_noise):
    """Inject eps_noise by making the reference's own torch.randn(mu.size()) draw it:
    Stochastic.reparametrize (packages/models/models.py:10) is the only RNG consumer."""
    real_randn = torch.randn
    calls = []

    def fake_randn(*a, **k):
        calls.append(a)
        return eps_noise.clone()
    torch.randn = fake_randn
    try:
        out = m(x) if model == "M1" else m(x, y)
    finally:
        torch.randn = real_randn
    assert len(calls) == 1
    return out


def run_case(fix, name, model, dims, B, wscale, seed, big=False):
    torch.manual_seed(0)
    m = build(model, dims)
    params = gu.make_params(model, dims, seed, wscale)
    sd_shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert sd_shapes == gu.layer_dims(model, **dims), (sd_shapes[:3], gu.layer_dims(model, **dims)[:3])
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    named = dict(m.named_parameters())
    if model == "M2_info":
        opt = torch.optim.Adam(m.enc_dec_clf.parameters(), lr=LR, betas=(0.9, 0.999))
        opt_aux = torch.optim.Adam(m.auxiliary.parameters(), lr=LR, betas=(0.9, 0.999))
    else:
        opt = torch.optim.Adam(m.parameters(), lr=LR, betas=(0.9, 0.999))
    chk = gu.checksum(params.values())
    for step in range(1, gu.NSTEPS + 1):
        xn, yn, en = gu.make_batch(dims, B, seed * 1000 + step)
        chk += gu.checksum([xn, yn, en])
        x, e = torch.from_numpy(xn), torch.from_numpy(en)
        y = None if yn is None else torch.from_numpy(yn)
        pre = f"{name}/step{step}"
        if model != "M2_info":
            r, mu, lv = model_call(m, model, x, y, e)
            loss, recon, kl = elbo(x, r, mu, lv, EPS)
            loss.backward()
            fix[pre + "/losses"] = np.array([loss.item(), recon.item(), kl.item()], dtype=np.float64)
            if step == 1:
                keep(fix, pre + "/r", r, big)
                keep(fix, pre + "/mu", mu, big)
                keep(fix, pre + "/logvar", lv, big)
                if model == "M1":
                    keep(fix, pre + "/kl_divergence", m.kl_divergence, big)
                for k, p in named.items():
                    put(fix, f"{pre}/grad/{k}", p.grad.numpy())
            opt.step(); opt.zero_grad()
        else:
            y_hat_class_soft = m.classify_fromX(x)
            r, z, mu, lv = model_call(m, model, x, y, e)
            ELBO, recon, kl = elbo(x, r, mu, lv, EPS)
            classif_loss = ALPHA * binary_cross_entropy(y_hat_class_soft, y, EPS)
            y_hat_aux_soft = m.classify_fromZ(z)
            aux_enc_loss = BETA * binary_cross_entropy(y_hat_aux_soft, y, EPS)
            enc_loss = ELBO + classif_loss - aux_enc_loss
            y_hat_aux_soft2 = m.classify_fromZ(z.detach())
            aux_loss = GAMMA * binary_cross_entropy(y_hat_aux_soft2, y, EPS)
            enc_loss.backward()
            fix[pre + "/losses"] = np.array([ELBO.item(), recon.item(), kl.item(), enc_loss.item(),
                                             classif_loss.item(), aux_loss.item(), aux_enc_loss.item()],
                                            dtype=np.float64)
            if step == 1:
                keep(fix, pre + "/r", r, big)
                keep(fix, pre + "/z", z, big)
                keep(fix, pre + "/mu", mu, big)
                keep(fix, pre + "/logvar", lv, big)
                keep(fix, pre + "/y_hat_class_soft", y_hat_class_soft, big)
                keep(fix, pre + "/y_hat_aux_soft", y_hat_aux_soft, big)
                for k, p in named.items():      # everything enc_loss.backward() deposited (quirk Q4)
                    put(fix, f"{pre}/grad_enc/{k}", p.grad.numpy())
            opt.step(); opt.zero_grad()
            aux_loss.backward()
            if step == 1:
                for k, p in named.items():
                    if k.startswith("auxiliary."):   # (gamma - beta) * dBCE
                        put(fix, f"{pre}/grad_aux_total/{k}", p.grad.numpy())
            opt_aux.step(); opt_aux.zero_grad()
        if step in (1, gu.NSTEPS):
            for k, p in named.items():
                put(fix, f"{pre}/param/{k}", p.detach().numpy())
    fix[name + "/input_checksum"] = np.float64(chk)


def main():
    big = "--big" in sys.argv
    fix = {}
    if big:
        torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
        for name, model, dims, B, wscale in gu.BIG_CASES:
            run_case(fix, name, model, dims, B, wscale, seed=gu.case_seed(name), big=True)
            print("captured", name)
        fix["torch_threads"] = np.int64(torch.get_num_threads())
        out = os.path.join(HERE, "vae_golden_b8192.npz")
    else:
        for i, (name, model, dims, B, wscale) in enumerate(gu.CASES):
            run_case(fix, name, model, dims, B, wscale, seed=100 + i)
            print("captured", name)
        # known-answer from SURVEY.md 8c (reference seeded init, torch RNG): stored for the record
        torch.manual_seed(0)
        m = VariationalAutoencoder([513, 16, [128, 128]])
        x = torch.rand(32, 513) ** 2
        r, mu, lv = m(x)
        fix["kat_m1_seed0_elbo"] = np.array([t.item() for t in elbo(x, r, mu, lv, 1e-8)])
        out = os.path.join(HERE, "vae_golden.npz")
    np.savez_compressed(out, **fix)
    print("wrote", out, os.path.getsize(out), "bytes", len(fix), "arrays")
    print("torch", torch.__version__, "numpy", np.__version__)


if __name__ == "__main__":
    main()
