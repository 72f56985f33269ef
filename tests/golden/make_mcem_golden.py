"""Capture golden vectors for the MCEM enhancement loop by running the REFERENCE's own
packages/models/mcem.py classes (CPU, fp32).  Build-container only (imports /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_mcem_golden.py

Every torch.rand / torch.randn the reference draws is recorded (in call order) and stored, so the
oracle and the HIP kernels can be run on exactly the same noise.  Output:
tests/golden/mcem_golden.npz (data only).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import numpy as np
import torch

from packages.models import mcem as ref_mcem        # seeds the global RNGs on import (mcem.py:1-5)
from packages.models.models import VariationalAutoencoder, DeepGenerativeModel, DeepGenerativeModel_v5
import golden_util as gu
import mcem_cases as mc


def build_vae(case):
    dims = mc.DIMS[case["model"]]
    h = list(dims["h_dim"])
    params = gu.make_params(case["model"], dims, case["seed"], case["wscale"])
    sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    if case["model"] == "M1":
        m = VariationalAutoencoder([dims["x_dim"], dims["z_dim"], h])
    elif case["model"] == "M2":
        m = DeepGenerativeModel([dims["x_dim"], dims["y_dim"], dims["z_dim"], h], None)
    else:
        m = DeepGenerativeModel_v5([dims["x_dim"], dims["y_dim"], dims["z_dim"], h])
    m.load_state_dict(sd)
    m.eval()
    for p in m.parameters():
        p.requires_grad = False
    return m.enc_dec_clf if case["model"] == "M2_info" else m


class Recorder:
    """Wraps torch.rand / torch.randn: passes through to the real generator and keeps every draw."""

    def __init__(self):
        self.draws = []
        self._rand, self._randn = torch.rand, torch.randn

    def __enter__(self):
        def rand(*a, **k):
            t = self._rand(*a, **k); self.draws.append(("rand", t.clone())); return t

        def randn(*a, **k):
            t = self._randn(*a, **k); self.draws.append(("randn", t.clone())); return t
        torch.rand, torch.randn = rand, randn
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randn = self._rand, self._randn


def run_case(case):
    vae = build_vae(case)
    X, S, y = mc.make_utterance(case)
    cls = {"M1": ref_mcem.MCEM_M1, "M2": ref_mcem.MCEM_M2, "M2_info": ref_mcem.MCEM_M2v3}[case["model"]]
    em = cls(niter=case["niter"], nsamples_E_step=case["n_e"], burnin_E_step=case["b_e"],
             nsamples_WF=case["n_wf"], burnin_WF=case["b_wf"], var_RW=0.01)
    torch.manual_seed(case["seed"] + 1000)
    fix = {}
    hist = dict(Z=[], W=[], H=[], g=[], Vb=[])
    with Recorder() as rec:
        if case["model"] == "M1":
            em.init_parameters(X=X, S=S, vae=vae, nmf_rank=case["K"], eps=mc.EPS, device="cpu")
        else:
            em.init_parameters(X=X, S=S, y=torch.from_numpy(y), vae=vae, nmf_rank=case["K"], eps=mc.EPS, device="cpu")
        fix["Z0"] = em.Z.numpy().copy()
        fix["Zclean"] = em.Zclean.numpy().copy()
        n_init = len(rec.draws)
        # EM.run (mcem.py:156-179), unrolled here only to snapshot the state after each iteration
        cost = np.zeros(em.niter)
        for n in range(em.niter):
            em.E_step()
            em.M_step()
            cost[n] = em.compute_expected_neg_log_like()
            hist["Z"].append(em.Z.numpy().copy()); hist["W"].append(em.W.numpy().copy())
            hist["H"].append(em.H.numpy().copy()); hist["g"].append(em.g.numpy().copy()); hist["Vb"].append(em.Vb.numpy().copy())
        WFs, WFn = em.compute_WF(sample=True)
    fix["cost"] = cost
    for k, v in hist.items():
        fix[k] = np.stack(v) if k != "Vb" else v[-1]
    fix["WFs"] = WFs.numpy(); fix["WFn"] = WFn.numpy()
    fix["S_hat"] = WFs.numpy() * X
    fix["Vs_last_sample"] = em.Vs.numpy()[:, ::9, :].copy()
    # draws: rand(F,K), rand(K,N), 2 x randn(N,L) from the encoder's reparametrisation, then per MH iteration
    # randn(L,N), rand(N)
    d = rec.draws
    assert [k for k, _ in d[:4]] == ["rand", "rand", "randn", "randn"] and n_init == 4, [k for k, _ in d[:6]]
    fix["rand_W"] = d[0][1].numpy(); fix["rand_H"] = d[1][1].numpy()
    fix["eps_X"] = d[2][1].numpy(); fix["eps_S"] = d[3][1].numpy()
    rest = d[4:]
    n_e, b_e, n_wf, b_wf = mc.effective_counts(case)
    sizes = [n_e + b_e] * case["niter"] + [n_wf + b_wf]
    assert len(rest) == 2 * sum(sizes), (len(rest), sizes)
    pos = 0
    for i, nit in enumerate(sizes):
        noise = np.stack([rest[pos + 2 * m][1].numpy() for m in range(nit)])
        u = torch.stack([rest[pos + 2 * m + 1][1] for m in range(nit)])
        assert all(rest[pos + 2 * m][0] == "randn" and rest[pos + 2 * m + 1][0] == "rand" for m in range(nit))
        fix[f"noise{i}"] = noise
        fix[f"logu{i}"] = torch.log(u).numpy()
        pos += 2 * nit
    fix["input_checksum"] = np.array(mc.checksum(X, S, y))
    return fix


def main():
    out = {}
    for case in mc.CASES:
        fix = run_case(case)
        for k, v in fix.items():
            out[f"{case['name']}/{k}"] = v
        print(case["name"], "cost", fix["cost"], "acc Z moved", float(np.abs(fix["Z"][0] - fix["Z0"]).mean()), flush=True)
    path = os.path.join(HERE, "mcem_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
