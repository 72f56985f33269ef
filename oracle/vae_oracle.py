"""numpy restatement of the reference VAE train step (TEST ORACLE, not product code).

Every function cites the reference lines it follows (paths relative to the
reference repo root).  Forward follows the reference op by op; backward is the
analytic gradient of exactly that forward (the reference uses autograd), so the
oracle is independent of torch autograd and pins the hand-written HIP backward.

Parameters are dicts keyed by the reference's ``state_dict`` names
(``encoder.hidden.0.weight`` ...; weights are nn.Linear layout ``[out, in]``).
All arithmetic runs in the dtype of the inputs (float32 mirrors the reference,
float64 gives a tighter truth for tolerance budgeting).
"""
import numpy as np

# ----------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------

def linear(x, W, b):
    """torch.nn.Linear: y = x @ W.T + b."""
    return x @ W.T + b


def _hidden_names(params, prefix):
    i = 0
    names = []
    while f"{prefix}hidden.{i}.weight" in params:
        names.append(f"{prefix}hidden.{i}")
        i += 1
    return names


def mlp_hidden_fwd(params, prefix, x, act):
    """`for layer in self.hidden: x = act(layer(x))`
    (packages/models/models.py:102-104 tanh encoder, :119-121 tanh decoder,
    :59-60 relu classifier).  Returns the list of layer outputs."""
    outs = []
    for name in _hidden_names(params, prefix):
        pre = linear(x, params[name + ".weight"], params[name + ".bias"])
        x = np.tanh(pre) if act == "tanh" else np.maximum(pre, 0)
        outs.append(x)
    return outs


def encoder_fwd(params, prefix, inp, eps_noise):
    """Encoder.forward + GaussianSample.forward + Stochastic.reparametrize
    (packages/models/models.py:102-105, 33-38, 9-22):
    z = mu + exp(0.5*log_var) * epsilon."""
    hs = mlp_hidden_fwd(params, prefix, inp, "tanh")
    h = hs[-1]
    mu = linear(h, params[prefix + "sample.mu.weight"], params[prefix + "sample.mu.bias"])
    lv = linear(h, params[prefix + "sample.log_var.weight"], params[prefix + "sample.log_var.bias"])
    std = np.exp(lv * inp.dtype.type(0.5))
    z = mu + std * eps_noise
    return dict(inp=inp, hs=hs, mu=mu, lv=lv, std=std, z=z, eps_noise=eps_noise)


def decoder_fwd(params, prefix, inp):
    """Decoder.forward (packages/models/models.py:119-122): exp(reconstruction(tanh stack))."""
    ds = mlp_hidden_fwd(params, prefix, inp, "tanh")
    a = linear(ds[-1], params[prefix + "reconstruction.weight"], params[prefix + "reconstruction.bias"])
    r = np.exp(a)
    return dict(inp=inp, ds=ds, a=a, r=r)


def classifier_fwd(params, prefix, inp):
    """Classifier.forward (packages/models/models.py:57-63): relu stack, sigmoid output."""
    hs = mlp_hidden_fwd(params, prefix, inp, "relu")
    pre = linear(hs[-1], params[prefix + "output_layer.weight"], params[prefix + "output_layer.bias"])
    p = 1.0 / (1.0 + np.exp(-pre))
    return dict(inp=inp, hs=hs, p=p.astype(inp.dtype))


# ----------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------

def elbo(x, r, mu, logvar, eps):
    """packages/models/utils.py:73-76 (Itakura-Saito recon + KL without the +1)."""
    recon = np.mean(np.sum(x / r - np.log(x + x.dtype.type(eps)) + np.log(r) - 1, axis=-1))
    kl = -0.5 * np.mean(np.sum(logvar - mu ** 2 - np.exp(logvar), axis=-1))
    return recon + kl, recon, kl


def kld_v2(mu, logvar):
    """VariationalAutoencoder._kld_v2 (packages/models/models.py:165-167), per frame."""
    return -0.5 * np.sum(logvar - mu ** 2 - np.exp(logvar), axis=-1)


def binary_cross_entropy(r, x, eps):
    """packages/models/utils.py:55-56 (eps inside the logs)."""
    e = r.dtype.type(eps)
    return -np.mean(np.sum(x * np.log(r + e) + (1 - x) * np.log(1 - r + e), axis=-1))


def binary_cross_entropy_v2(r, eps):
    """packages/models/utils.py:59-60 (targets 0.5)."""
    e = r.dtype.type(eps)
    return -np.mean(np.sum(0.5 * np.log(r + e) + 0.5 * np.log(1 - r + e), axis=-1))


def binary_cross_entropy_v3(r, eps):
    """packages/models/utils.py:62-63 (targets r: entropy)."""
    e = r.dtype.type(eps)
    return -np.mean(np.sum(r * np.log(r + e) + (1 - r) * np.log(1 - r + e), axis=-1))


# ----------------------------------------------------------------------------
# analytic backward (autograd of the forward above)
# ----------------------------------------------------------------------------

def _acc(grads, key, val):
    grads[key] = grads.get(key, 0) + val


def linear_bwd(params, grads, name, inp, dpre, need_dx=True):
    _acc(grads, name + ".weight", dpre.T @ inp)
    _acc(grads, name + ".bias", dpre.sum(axis=0))
    return dpre @ params[name + ".weight"] if need_dx else None


def mlp_hidden_bwd(params, grads, prefix, inp, outs, dout, act, need_dx):
    names = _hidden_names(params, prefix)
    for i in reversed(range(len(names))):
        o = outs[i]
        dpre = dout * (1 - o * o) if act == "tanh" else dout * (o > 0)
        layer_in = inp if i == 0 else outs[i - 1]
        dout = linear_bwd(params, grads, names[i], layer_in, dpre, need_dx or i > 0)
    return dout


def decoder_bwd(params, grads, prefix, cache, da, need_dx=True):
    dd = linear_bwd(params, grads, prefix + "reconstruction", cache["ds"][-1], da)
    return mlp_hidden_bwd(params, grads, prefix, cache["inp"], cache["ds"], dd, "tanh", need_dx)


def encoder_bwd(params, grads, prefix, cache, dz, dmu_direct, dlv_direct, need_dx=False):
    """dz flows through z = mu + std*eps; dmu_direct/dlv_direct are the KL terms."""
    dmu = dz + dmu_direct
    dlv = dz * cache["eps_noise"] * cache["std"] * 0.5 + dlv_direct
    h = cache["hs"][-1]
    dh = linear_bwd(params, grads, prefix + "sample.mu", h, dmu)
    dh = dh + linear_bwd(params, grads, prefix + "sample.log_var", h, dlv)
    return mlp_hidden_bwd(params, grads, prefix, cache["inp"], cache["hs"], dh, "tanh", need_dx)


def classifier_bwd(params, grads, prefix, cache, dp, need_dx):
    p = cache["p"]
    dpre = dp * p * (1 - p)
    dh = linear_bwd(params, grads, prefix + "output_layer", cache["hs"][-1], dpre)
    return mlp_hidden_bwd(params, grads, prefix, cache["inp"], cache["hs"], dh, "relu", need_dx)


def elbo_bwd(x, a, mu, lv, scale=1.0):
    """Gradients of elbo() wrt a = log r, mu, logvar (SURVEY 8a analytic backward)."""
    B = x.shape[0]
    da = (1 - x * np.exp(-a)) * (scale / B)
    dmu = mu * (scale / B)
    dlv = -0.5 * (1 - np.exp(lv)) * (scale / B)
    return da, dmu, dlv


def bce_bwd(p, y, eps, scale):
    """d/dp of scale * binary_cross_entropy(p, y, eps)."""
    B = p.shape[0]
    e = p.dtype.type(eps)
    return -(scale / B) * (y / (p + e) - (1 - y) / (1 - p + e))


# ----------------------------------------------------------------------------
# whole-model forward / backward
# ----------------------------------------------------------------------------

def m1_forward(params, x, eps_noise):
    """VariationalAutoencoder.forward (packages/models/models.py:172-179)."""
    enc = encoder_fwd(params, "encoder.", x, eps_noise)
    dec = decoder_fwd(params, "decoder.", enc["z"])
    return enc, dec


def m2_forward(params, x, y, eps_noise):
    """DeepGenerativeModel.forward (packages/models/models.py:200-203)."""
    enc = encoder_fwd(params, "encoder.", np.concatenate([x, y], axis=1), eps_noise)
    dec = decoder_fwd(params, "decoder.", np.concatenate([enc["z"], y], axis=1))
    return enc, dec


def m2v3_forward(params, x, y, eps_noise, prefix="enc_dec_clf."):
    """DeepGenerativeModel_v5.forward / _v3.forward (packages/models/models.py:426-433, 276-283)."""
    enc = encoder_fwd(params, prefix + "encoder.", x, eps_noise)
    dec = decoder_fwd(params, prefix + "decoder.", np.concatenate([enc["z"], y], axis=1))
    return enc, dec


def vae_loss_and_grads(model, params, x, y, eps_noise, eps=1e-8):
    """Loss + all parameter grads of one scripts/training_M1.py:134-137 /
    scripts/training_M2.py:142-145 step (forward, elbo, backward)."""
    if model == "M1":
        enc, dec = m1_forward(params, x, eps_noise)
    elif model == "M2":
        enc, dec = m2_forward(params, x, y, eps_noise)
    else:
        raise ValueError(model)
    loss, recon, kl = elbo(x, dec["r"], enc["mu"], enc["lv"], eps)
    grads = {}
    da, dmu_kl, dlv_kl = elbo_bwd(x, dec["a"], enc["mu"], enc["lv"])
    ddec_in = decoder_bwd(params, grads, "decoder.", dec, da)
    zdim = enc["z"].shape[1]
    dz = ddec_in[:, :zdim]
    encoder_bwd(params, grads, "encoder.", enc, dz, dmu_kl, dlv_kl, need_dx=False)
    out = dict(r=dec["r"], mu=enc["mu"], logvar=enc["lv"], z=enc["z"],
               loss=loss, recon=recon, kl=kl, kl_divergence=kld_v2(enc["mu"], enc["lv"]))
    return out, grads


def m2info_losses_and_grads(params, x, y, eps_noise, alpha, beta, gamma, eps=1e-8,
                            aux_enc_variant="bce", aux_weight=None, dec_label="y"):
    """One scripts/training_M2_info_vad.py:159-198 step up to the two backward
    passes.  Returns (out, grads_after_enc_backward, aux_grads_second_backward).

    `grads_after_enc_backward` holds what `enc_loss.backward()` deposits in
    EVERY parameter's .grad (including -beta*dBCE in the auxiliary net, quirk
    Q4); `aux_grads_second_backward` is what `aux_loss.backward()` then ADDS
    to the auxiliary net's .grad (the enc_dec_clf optimizer's zero_grad() does
    not touch them).
    """
    pre = "enc_dec_clf."
    clf = classifier_fwd(params, pre + "classifier.", x)
    enc, dec = m2v3_forward(params, x, y, eps_noise, pre)
    ELBO, recon, kl = elbo(x, dec["r"], enc["mu"], enc["lv"], eps)
    classif_loss = alpha * binary_cross_entropy(clf["p"], y, eps)
    aux1 = classifier_fwd(params, "auxiliary.", enc["z"])
    aux_enc_loss = beta * binary_cross_entropy(aux1["p"], y, eps)
    enc_loss = ELBO + classif_loss - aux_enc_loss
    aux2 = classifier_fwd(params, "auxiliary.", enc["z"])  # z.detach(): same values
    aux_loss = gamma * binary_cross_entropy(aux2["p"], y, eps)

    g1 = {}
    # ELBO part
    da, dmu_kl, dlv_kl = elbo_bwd(x, dec["a"], enc["mu"], enc["lv"])
    ddec_in = decoder_bwd(params, g1, pre + "decoder.", dec, da)
    zdim = enc["z"].shape[1]
    dz = ddec_in[:, :zdim]
    # -beta * BCE(aux(z), y): flows into aux params and into z
    dp_aux = bce_bwd(aux1["p"], y, eps, -beta)
    dz_aux = classifier_bwd(params, g1, "auxiliary.", aux1, dp_aux, need_dx=True)
    dz = dz + dz_aux
    encoder_bwd(params, g1, pre + "encoder.", enc, dz, dmu_kl, dlv_kl, need_dx=False)
    # + alpha * BCE(clf(x), y)
    dp_clf = bce_bwd(clf["p"], y, eps, alpha)
    classifier_bwd(params, g1, pre + "classifier.", clf, dp_clf, need_dx=False)

    g2 = {}
    dp_aux2 = bce_bwd(aux2["p"], y, eps, gamma)
    classifier_bwd(params, g2, "auxiliary.", aux2, dp_aux2, need_dx=False)

    out = dict(r=dec["r"], z=enc["z"], mu=enc["mu"], logvar=enc["lv"],
               y_hat_class_soft=clf["p"], y_hat_aux_soft=aux1["p"],
               ELBO=ELBO, recon=recon, kl=kl, classif_loss=classif_loss,
               aux_enc_loss=aux_enc_loss, enc_loss=enc_loss, aux_loss=aux_loss)
    return out, g1, g2


# ----------------------------------------------------------------------------
# Adam (torch.optim.Adam op order; scripts/training_M2.py:122)
# ----------------------------------------------------------------------------

def adam_step(p, g, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad) single-tensor op order:
    m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ;
    denom = sqrt(v)/sqrt(1-b2^t) + eps ; p -= (lr/(1-b1^t)) * m/denom."""
    dt = p.dtype.type
    m = m + (g - m) * dt(1 - b1)          # torch: exp_avg.lerp_(grad, 1-beta1)
    v = v * dt(b2) + g * g * dt(1 - b2)
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    step_size = lr / bc1
    denom = np.sqrt(v) / dt(np.sqrt(bc2)) + dt(eps)
    p = p - dt(step_size) * (m / denom)
    return p, m, v


class AdamState:
    def __init__(self, names):
        self.names = list(names)
        self.m = {}
        self.v = {}
        self.t = 0

    def step(self, params, grads, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
        self.t += 1
        for n in self.names:
            g = grads.get(n)
            if g is None:
                continue
            if isinstance(g, (int, float)):
                g = np.zeros_like(params[n])
            g = np.asarray(g, dtype=params[n].dtype).reshape(params[n].shape)
            m = self.m.get(n, np.zeros_like(params[n]))
            v = self.v.get(n, np.zeros_like(params[n]))
            params[n], self.m[n], self.v[n] = adam_step(params[n], g, m, v, self.t, lr, b1, b2, eps)


def train_step_vae(model, params, opt, x, y, eps_noise, eps=1e-8, lr=1e-4):
    """scripts/training_M1.py:134-139 / scripts/training_M2.py:142-147."""
    out, grads = vae_loss_and_grads(model, params, x, y, eps_noise, eps)
    opt.step(params, grads, lr=lr)
    return out, grads


def train_step_m2info(params, opt_edc, opt_aux, x, y, eps_noise, alpha=0.0, beta=10.0,
                      gamma=1.0, eps=1e-8, lr=1e-4):
    """scripts/training_M2_info_vad.py:159-198 incl. quirk Q4: the auxiliary net
    is stepped with (gamma - beta) * dBCE because the grads deposited by
    enc_loss.backward() are never zeroed on it."""
    out, g1, g2 = m2info_losses_and_grads(params, x, y, eps_noise, alpha, beta, gamma, eps)
    opt_edc.step(params, g1, lr=lr)                       # enc_dec_clf params only
    aux_total = {k: g1[k] + g2[k] for k in g2}            # accumulate, never zeroed in between
    opt_aux.step(params, aux_total, lr=lr)
    return out, g1, g2, aux_total
