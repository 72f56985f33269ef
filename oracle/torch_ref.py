"""torch-CPU restatement of the reference train-step loop bodies (TEST ORACLE).

Same ATen op sequence the reference executes on CPU (F.linear, tanh, exp,
the elbo expression, torch.optim.Adam), written functionally over a dict of
leaf tensors keyed by the reference state_dict names.  Used (a) by tests to
cross-check the numpy oracle through autograd, (b) by bench.py's
``cpu_baseline`` leg (kind "port"): it is the reference's CPU path timed on
the GPU node's host cores.  Never imported by the product path.

Reference lines: packages/models/models.py:9-22,33-38,57-63,102-105,119-122,
172-179,200-203,426-433; packages/models/utils.py:55-63,73-76;
scripts/training_M1.py:125-139, scripts/training_M2.py:132-147,
scripts/training_M2_info_vad.py:153-198.
"""
import math
import torch
import torch.nn.functional as F


def layer_dims(model, x_dim=513, y_dim=1, z_dim=16, h_dim=(128, 128)):
    """state_dict name -> shape, in the reference's registration order."""
    h = list(h_dim)
    rh = list(reversed(h))
    out = []

    def stack(prefix, inp, hs):
        d = inp
        for i, n in enumerate(hs):
            out.append((f"{prefix}hidden.{i}.weight", (n, d)))
            out.append((f"{prefix}hidden.{i}.bias", (n,)))
            d = n
        return d

    def enc(prefix, inp):
        d = stack(prefix, inp, h)
        for nm in ("mu", "log_var"):
            out.append((f"{prefix}sample.{nm}.weight", (z_dim, d)))
            out.append((f"{prefix}sample.{nm}.bias", (z_dim,)))

    def dec(prefix, inp):
        d = stack(prefix, inp, rh)
        out.append((f"{prefix}reconstruction.weight", (x_dim, d)))
        out.append((f"{prefix}reconstruction.bias", (x_dim,)))

    def clf(prefix, inp):
        d = stack(prefix, inp, h)
        out.append((f"{prefix}output_layer.weight", (y_dim, d)))
        out.append((f"{prefix}output_layer.bias", (y_dim,)))

    if model == "M1":
        enc("encoder.", x_dim); dec("decoder.", z_dim)
    elif model == "M2":
        enc("encoder.", x_dim + y_dim); dec("decoder.", z_dim + y_dim)
    elif model == "M2_info":
        enc("enc_dec_clf.encoder.", x_dim); dec("enc_dec_clf.decoder.", z_dim + y_dim)
        clf("enc_dec_clf.classifier.", x_dim); clf("auxiliary.", z_dim)
    else:
        raise ValueError(model)
    return out


def init_params(model, seed=0, dtype=torch.float32, **dims):
    """xavier_normal_ weights, zero bias (packages/models/models.py:137-141) from a
    private generator (NOT bit-identical to the reference's construction-order
    draw; parity tests load explicit state instead)."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, shape in layer_dims(model, **dims):
        if name.endswith("weight"):
            std = math.sqrt(2.0 / (shape[0] + shape[1]))
            p[name] = (torch.randn(shape, generator=g, dtype=dtype) * std).requires_grad_()
        else:
            p[name] = torch.zeros(shape, dtype=dtype, requires_grad=True)
    return p


def _hidden(p, prefix, x, act):
    i = 0
    while f"{prefix}hidden.{i}.weight" in p:
        x = act(F.linear(x, p[f"{prefix}hidden.{i}.weight"], p[f"{prefix}hidden.{i}.bias"]))
        i += 1
    return x


def encoder(p, prefix, x, eps_noise=None):
    h = _hidden(p, prefix, x, torch.tanh)
    mu = F.linear(h, p[prefix + "sample.mu.weight"], p[prefix + "sample.mu.bias"])
    lv = F.linear(h, p[prefix + "sample.log_var.weight"], p[prefix + "sample.log_var.bias"])
    if eps_noise is None:
        eps_noise = torch.randn(mu.size())
    std = lv.mul(0.5).exp_()
    z = mu.addcmul(std, eps_noise)
    return z, mu, lv


def decoder(p, prefix, x):
    h = _hidden(p, prefix, x, torch.tanh)
    return torch.exp(F.linear(h, p[prefix + "reconstruction.weight"], p[prefix + "reconstruction.bias"]))


def classifier(p, prefix, x):
    h = _hidden(p, prefix, x, torch.relu)
    return torch.sigmoid(F.linear(h, p[prefix + "output_layer.weight"], p[prefix + "output_layer.bias"]))


def elbo(x, r, mu, logvar, eps):
    recon = torch.mean(torch.sum(x / r - torch.log(x + eps) + torch.log(r) - 1, dim=-1))
    KL = -0.5 * torch.mean(torch.sum(logvar - mu.pow(2) - logvar.exp(), dim=-1))
    return recon + KL, recon, KL


def bce(r, x, eps):
    return -torch.mean(torch.sum(x * torch.log(r + eps) + (1 - x) * torch.log(1 - r + eps), dim=-1))


def forward(model, p, x, y=None, eps_noise=None):
    if model == "M1":
        z, mu, lv = encoder(p, "encoder.", x, eps_noise)
        return decoder(p, "decoder.", z), z, mu, lv
    if model == "M2":
        z, mu, lv = encoder(p, "encoder.", torch.cat([x, y], dim=1), eps_noise)
        return decoder(p, "decoder.", torch.cat([z, y], dim=1)), z, mu, lv
    if model == "M2_info":
        z, mu, lv = encoder(p, "enc_dec_clf.encoder.", x, eps_noise)
        return decoder(p, "enc_dec_clf.decoder.", torch.cat([z, y], dim=1)), z, mu, lv
    raise ValueError(model)


class Stepper:
    """The reference loop body with stock torch.optim.Adam (lr 1e-4, betas (0.9,0.999))."""

    def __init__(self, model, params, lr=1e-4, alpha=0.0, beta=10.0, gamma=1.0, eps=1e-8):
        self.model, self.p, self.eps = model, params, eps
        self.alpha, self.beta, self.gamma = alpha, beta, gamma
        if model == "M2_info":
            edc = [v for k, v in params.items() if k.startswith("enc_dec_clf.")]
            aux = [v for k, v in params.items() if k.startswith("auxiliary.")]
            self.opt = torch.optim.Adam(edc, lr=lr, betas=(0.9, 0.999))
            self.opt_aux = torch.optim.Adam(aux, lr=lr, betas=(0.9, 0.999))
        else:
            self.opt = torch.optim.Adam(list(params.values()), lr=lr, betas=(0.9, 0.999))

    def step(self, x, y=None, eps_noise=None):
        p = self.p
        if self.model != "M2_info":
            r, z, mu, lv = forward(self.model, p, x, y, eps_noise)
            loss, recon, kl = elbo(x, r, mu, lv, self.eps)
            loss.backward()
            self.opt.step()
            self.opt.zero_grad()
            return loss.item(), recon.item(), kl.item()
        y_hat_class_soft = classifier(p, "enc_dec_clf.classifier.", x)
        r, z, mu, lv = forward("M2_info", p, x, y, eps_noise)
        ELBO, recon, kl = elbo(x, r, mu, lv, self.eps)
        classif_loss = self.alpha * bce(y_hat_class_soft, y, self.eps)
        aux_enc_loss = self.beta * bce(classifier(p, "auxiliary.", z), y, self.eps)
        enc_loss = ELBO + classif_loss - aux_enc_loss
        aux_loss = self.gamma * bce(classifier(p, "auxiliary.", z.detach()), y, self.eps)
        enc_loss.backward()
        self.opt.step()
        self.opt.zero_grad()
        aux_loss.backward()
        self.opt_aux.step()
        self.opt_aux.zero_grad()
        return ELBO.item(), recon.item(), kl.item(), enc_loss.item(), aux_loss.item()
