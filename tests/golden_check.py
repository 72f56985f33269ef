"""Shared driver: replay one golden case through a step implementation and
compare every stored quantity.  Used with the numpy oracle (CPU tests) and with
the HIP path (gpu tests) so both are held to the same reference vectors."""
import numpy as np

import golden_util as gu


class StepImpl:
    """Interface a backend implements to be checked against the golden vectors.

    load(model, dims, params)            -> None (params: dict name -> np.float32 array)
    step(x, y, eps_noise)                -> dict with outputs/losses/grads (see below)
    params()                             -> dict name -> np array (current parameters)

    step() result keys.  M1/M2: r, mu, logvar, losses=(loss, recon, kl), grads{name},
    optional kl_divergence.  M2_info: r, z, mu, logvar, y_hat_class_soft, y_hat_aux_soft,
    losses=(ELBO, recon, kl, enc_loss, classif_loss, aux_loss, aux_enc_loss),
    grads_enc{name}, grads_aux_total{name}.  grads may be None for steps > 1.
    """


def check_case(impl, fix, case, rtol_out=2e-5, rtol_loss=2e-5, rtol_grad=1e-4, atol_grad=1e-7, atol_rel_grad=1e-5,
               rtol_param=1e-6, atol_param=2e-7, atol_out=1e-6, bad_frac=None):
    name, model, dims, B, wscale = case
    # see golden_util.compare_params: saturated / tiny-gradient elements are Adam-ill-conditioned
    max_bad_frac = 0.12 if (wscale > 1.0 or B < 4) else 0.01
    if bad_frac is not None:
        max_bad_frac = max(max_bad_frac, bad_frac)
    seed = gu.case_seed(name)
    params = gu.make_params(model, dims, seed, wscale)
    chk = gu.checksum(params.values())
    impl.load(model, dims, {k: v.copy() for k, v in params.items()})
    for step in range(1, gu.NSTEPS + 1):
        x, y, e = gu.make_batch(dims, B, seed * 1000 + step)
        chk += gu.checksum([x, y, e])
        out = impl.step(x, y, e)
        pre = f"{name}/step{step}"
        np.testing.assert_allclose(np.array(out["losses"], dtype=np.float64), fix[pre + "/losses"],
                                   rtol=(rtol_loss if step == 1 or max_bad_frac < 0.1 else 5e-3), atol=1e-6,
                                   err_msg=pre + "/losses")
        if step == 1:
            keys = ["r", "mu", "logvar"] + (["z", "y_hat_class_soft", "y_hat_aux_soft"] if model == "M2_info" else [])
            for k in keys:
                if out.get(k) is None:      # the fused trainer keeps r / mu / logvar on chip
                    continue
                if f"{pre}/{k}" in fix:
                    np.testing.assert_allclose(out[k], fix[f"{pre}/{k}"], rtol=rtol_out, atol=atol_out, err_msg=f"{pre}/{k}")
                else:                       # 8192-frame cases: strided sample + moments
                    gu.compare_summary(f"{pre}/{k}", out[k], fix, f"{pre}/{k}", rtol_out, atol_out,
                                       stride=gu.OUT_STRIDE if np.asarray(out[k]).size > 2 ** 20 else gu.SAMPLE_STRIDE)
            if model == "M1" and out.get("kl_divergence") is not None:
                if pre + "/kl_divergence" in fix:
                    np.testing.assert_allclose(out["kl_divergence"], fix[pre + "/kl_divergence"], rtol=rtol_out, atol=1e-5)
                else:
                    gu.compare_summary(pre + "/kl_divergence", out["kl_divergence"], fix, pre + "/kl_divergence", rtol_out, 1e-5)
            if model != "M2_info":
                for k in params:
                    gu.compare_summary(f"{pre}/grad/{k}", out["grads"][k], fix, f"{pre}/grad/{k}", rtol_grad, atol_grad, atol_rel_grad)
            else:
                for k in params:
                    if out.get("_fused_info") and k.startswith("auxiliary."):
                        continue        # the fused step only materialises the accumulated (gamma - beta) gradient
                    gu.compare_summary(f"{pre}/grad_enc/{k}", out["grads_enc"][k], fix, f"{pre}/grad_enc/{k}",
                                       rtol_grad, atol_grad, atol_rel_grad)
                    if k.startswith("auxiliary."):
                        gu.compare_summary(f"{pre}/grad_aux_total/{k}", out["grads_aux_total"][k], fix,
                                           f"{pre}/grad_aux_total/{k}", rtol_grad, atol_grad, atol_rel_grad)
        if step in (1, gu.NSTEPS):
            cur = impl.params()
            for k in params:
                gu.compare_params(f"{pre}/param/{k}", cur[k], fix, f"{pre}/param/{k}", rtol_param, atol_param,
                                  max_bad_frac, 1.05e-4 * step)
    assert abs(chk - float(fix[name + "/input_checksum"])) <= 1e-9 * abs(chk), "regenerated inputs differ from the captured ones"
