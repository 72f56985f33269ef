"""One-off sweep: the default weight-gradient kernel against the 2 x 2 register-ring kernel over odd batch sizes and all models / policies
(same stash, only the order of the frame sums differs: fp32 rounding).  Prints the worst deviation relative to each tensor's maximum."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch
import golden_util as gu
T = importlib.import_module("disentangled-vae_amd.trainer")
worst = 0.0
for model, y_dim in (("M1", 0), ("M2", 1), ("M2", 513), ("M2_info", 1)):
    dims = dict(x_dim=513, y_dim=y_dim, z_dim=16, h_dim=(128, 128))
    params = gu.make_params(model, dims, 5)
    for B in (129, 257, 640, 1111, 4097, 12345, 40000):
        for prec in ("bf16x3", "fp32"):
            x, y, e = gu.make_batch(dims, B, B)
            t = lambda a: None if a is None else torch.from_numpy(a).cuda()
            g = {}
            for form in ("wg4", "ring"):
                os.environ["DVAE_WGRAD"] = form
                tr = T.Trainer(model, dims, params, batch=B, precision=prec)
                tr.step(t(x), t(y), t(e))
                g[form] = tr.grads_numpy()
            os.environ.pop("DVAE_WGRAD")
            dev = max(float(np.max(np.abs(g["wg4"][k] - g["ring"][k])) / (np.max(np.abs(g["ring"][k])) + 1e-30)) for k in g["ring"])
            worst = max(worst, dev)
            assert dev < 3e-5, (model, y_dim, B, prec, dev)
    print(model, y_dim, "ok", flush=True)
print("worst relative deviation", worst)
