"""Soak: N train steps twice from the same state on the same batches -> bitwise identical parameters (race detector for the
hand-synchronised kernels), for every fused model variant; the same for an MCEM run on fixed draws."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch
import golden_util as gu
from impl_modules import build_model
trainer = importlib.import_module("disentangled-vae_amd.trainer"); mcem_dev = importlib.import_module("disentangled-vae_amd.mcem")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ok = True
for model, y_dim, B, prec in [("M2", 513, 8192, "bf16"), ("M2", 513, 8192, "fp32"), ("M1", 0, 8192, "bf16"), ("M2", 1, 5000, "bf16"),
                              ("M2_info", 1, 8192, "bf16"), ("M2", 513, 20000, "bf16")]:
    dims = dict(x_dim=513, y_dim=y_dim, z_dim=16, h_dim=(128, 128))
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    pool = [(torch.rand(B, 513, device="cuda", generator=g) ** 4 * 50 + 1e-3, (torch.rand(B, max(y_dim, 1), device="cuda", generator=g) > 0.5).float())
            for _ in range(4)]
    res = []
    t0 = time.perf_counter()
    for rep in range(2):
        tr = trainer.Trainer(model, dims, batch=B, precision=prec, seed=5, lr=1e-3)
        n = steps if prec == "bf16" else steps // 4
        for s in range(n):
            x, y = pool[s % 4]
            tr.step(x, y if y_dim else None)
        res.append((tr.params.clone(), tr.m.clone(), tr.v.clone(), tr.losses.clone()))
    same = all(torch.equal(a, b) for a, b in zip(*res))
    finite = bool(torch.isfinite(res[0][0]).all())
    ok &= same and finite
    print(f"{model} y={y_dim} B={B} {prec}: {n} steps x2, bitwise identical: {same}, finite: {finite}, loss {res[0][3][:3].tolist()} ({time.perf_counter() - t0:.1f} s)", flush=True)
# MCEM on fixed draws
dims = dict(x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128))
m = build_model("M2", dims); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in gu.make_params("M2", dims, 3).items()}); m.cuda().eval()
import mcem_cases as mc
utts = [mc.make_utterance(dict(seed=60 + i, N=n, model="M2")) for i, n in enumerate([300, 257, 64, 311] * 6)]
all_utts = utts
# 24 utterances: the chain on 32-frame tiles; 4 utterances: on 16-frame tiles (csrc/mcem_resident16.hip)
# (fp32 on 1 and 2 utterances: the 4-frame chain, csrc/mcem_resident4.hip; on 4 utterances -- 932 frames -- as well: 233 tiles)
for prec, utts in (("fp32", all_utts), ("bf16", all_utts), ("fp32", all_utts[:4]), ("bf16x3", all_utts[:4]), ("bf16", all_utts[:4]), ("fp32", all_utts[:1]), ("fp32", all_utts[:8])):
    outs = []
    for rep in range(2):
        mb = mcem_dev.McemBatch(m, niter=30, precision=prec)
        torch.manual_seed(7)
        mb.init_parameters([u[0] for u in utts], [u[2] for u in utts])
        gen = torch.Generator(device="cuda"); gen.manual_seed(9)
        draws = [(torch.randn(40, 16, mb.ntot, device="cuda", generator=gen), torch.log(torch.rand(40, mb.ntot, device="cuda", generator=gen))) for _ in range(30)]
        draws.append((torch.randn(100, 16, mb.ntot, device="cuda", generator=gen), torch.log(torch.rand(100, mb.ntot, device="cuda", generator=gen))))
        cost = mb.run(draws)
        outs.append((mb.W.clone(), mb.H.clone(), mb.g.clone(), mb.WFs.clone(), torch.from_numpy(cost)))
    same = all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(*outs))
    ok &= same
    print(f"MCEM batch of {len(utts)} ({prec}): 30 EM iterations x2, bitwise identical: {same}, cost {float(outs[0][4][-1].mean()):.4f}", flush=True)
print("SOAK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
