"""MCEM enhancement throughput on the MI355X at the reference's settings (scripts/evaluate_ntcd_M2.py:92-99:
niter 100, E-step 10 + 30 burn-in, Wiener 25 + 75 burn-in, NMF rank 10) next to the host path (ATen, the
reference's CPU mode) timed on a bounded number of EM iterations.

Units: one "decoder frame pass" = the decoder [16+y]-128-128-513 on one frame.  Per utterance of N frames the
reference runs N * (niter * 2 * (n_e + b_e) + niter * n_e + 2 * (n_wf + b_wf) + n_wf) of them."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import golden_util as gu, mcem_cases as mc
from impl_modules import build_model
from packages.models import mcem


def make(model, y_dim, N, device, precision):
    dims = dict(x_dim=513, y_dim=y_dim, z_dim=16, h_dim=(128, 128))
    params = gu.make_params(model, dims, 3)
    m = build_model(model, dims)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    m.eval().to(device)
    for p in m.parameters():
        p.requires_grad = False
    case = dict(seed=5, N=N, model=model)
    mc.DIMS["bench"] = dims
    X, S, y = mc.make_utterance(dict(case, model="bench"))
    return m, X, S, (torch.from_numpy(y).to(device) if y_dim else None)


def run(em, vae, X, S, y, device, niter):
    em.niter = niter
    kw = dict(X=X, S=S, vae=vae, nmf_rank=10, eps=mc.EPS, device=device)
    if y is not None:
        kw["y"] = y
    em.init_parameters(**kw)
    if device != "cpu":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    cost = em.run()
    if device != "cpu":
        torch.cuda.synchronize()
    return time.perf_counter() - t0, cost


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--y-dim", type=int, default=1)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--batch", type=int, nargs="*", default=[8, 25, 51, 102])
    a = ap.parse_args()
    model = "M2" if a.y_dim else "M1"
    cls = mcem.MCEM_M2 if a.y_dim else mcem.MCEM_M1
    n_e, b_e, n_wf, b_wf, niter = 10, 30, 25, 75, 100
    res = dict(frames=a.frames, y_dim=a.y_dim, niter=niter)
    passes_ref = a.frames * (niter * 2 * (n_e + b_e) + niter * n_e + 2 * (n_wf + b_wf) + n_wf)
    for prec in ("fp32", "bf16x3", "bf16"):
        m, X, S, y = make(model, a.y_dim, a.frames, "cuda", prec)
        em = cls(niter=niter, nsamples_E_step=n_e, burnin_E_step=b_e, nsamples_WF=n_wf, burnin_WF=b_wf)
        em.precision = prec
        run(em, m, X, S, y, "cuda", 3)                      # warm-up
        t, cost = run(em, m, X, S, y, "cuda", niter)
        res[prec] = dict(seconds_per_utterance=t, utterances_per_s=1 / t, ms_per_em_iteration=1e3 * t / niter,
                         ref_decoder_frame_passes_per_s=passes_ref / t, cost_first=float(cost[0]), cost_last=float(cost[-1]))
        # stage split
        pack = em._decoder_pack()
        noise = torch.randn(n_e + b_e, 16, a.frames, device="cuda"); logu = torch.log(torch.rand(n_e + b_e, a.frames, device="cuda"))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        reps = 20
        ev[0].record()
        for _ in range(reps):
            Zs, Vs = pack.sample(em.Z, y, em.g, em.Vb, em.X_abs_2_t, noise, logu, b_e)
        ev[1].record()
        dev = __import__("importlib").import_module("disentangled-vae_amd.mcem")
        W, H, g, Vb = em.W.clone(), em.H.clone(), em.g.clone(), em.Vb.clone()
        for _ in range(reps):
            dev.m_step_(em.X_abs_2, Vs, W, H, g, Vb)
        ev[2].record()
        torch.cuda.synchronize()
        res[prec]["e_step_us"] = ev[0].elapsed_time(ev[1]) * 1e3 / reps
        res[prec]["mh_iteration_us"] = res[prec]["e_step_us"] / (n_e + b_e + n_e + 1)     # per pass of the round-4 count (41 chain passes + 10 decode passes), kept as the divisor for comparability
        res[prec]["m_step_us"] = ev[1].elapsed_time(ev[2]) * 1e3 / reps
    # many utterances side by side (McemBatch): throughput in utterances / s
    dev = __import__("importlib").import_module("disentangled-vae_amd.mcem")
    res["batched"] = {}
    for prec in ("fp32", "bf16x3", "bf16"):
        for U in a.batch:
            m, X, S, y = make(model, a.y_dim, a.frames, "cuda", prec)
            mb = dev.McemBatch(m, niter=niter, nsamples_E_step=n_e, burnin_E_step=b_e, nsamples_WF=n_wf, burnin_WF=b_wf,
                               label_in_encoder=bool(a.y_dim), label_in_decoder=bool(a.y_dim), precision=prec)
            ys = [y] * U if y is not None else None
            mb.niter = 2
            mb.init_parameters([X] * U, ys); mb.run()
            mb.niter = niter
            ts = []
            for _ in range(3):                               # three timed runs, the median reported: single runs of 50-100 ms spread by +- 10 % on one box
                mb.init_parameters([X] * U, ys)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                cost = mb.run()
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            t = sorted(ts)[1]
            res["batched"][f"{prec}_U{U}"] = dict(seconds=t, seconds_runs=[round(x, 5) for x in ts], utterances_per_s=U / t, ms_per_em_iteration=1e3 * t / niter,
                                                  ref_decoder_frame_passes_per_s=U * passes_ref / t, cost_last=float(cost[-1].mean()))
    if not a.no_cpu:
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        m, X, S, y = make(model, a.y_dim, a.frames, "cpu", "fp32")
        em = cls(niter=a.cpu_iters, nsamples_E_step=n_e, burnin_E_step=b_e, nsamples_WF=n_wf, burnin_WF=b_wf)
        t, _ = run(em, m, X, S, y, "cpu", a.cpu_iters)
        # a.cpu_iters EM iterations + the final Wiener chain (100 MH iterations ~ 2.5 E-steps)
        per_iter = t / (a.cpu_iters + (n_wf + b_wf) / (n_e + b_e))
        res["cpu_host_path"] = dict(threads=torch.get_num_threads(), em_iterations_timed=a.cpu_iters, seconds=t,
                                    seconds_per_utterance_extrapolated=per_iter * (niter + (n_wf + b_wf) / (n_e + b_e)))
        res["speedup_fp32"] = res["cpu_host_path"]["seconds_per_utterance_extrapolated"] / res["fp32"]["seconds_per_utterance"]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
