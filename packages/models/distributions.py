"""Log-densities used by the (dead) `_kld` path and the unimportable SVI module of the
reference (packages/models/distributions.py:5-53).  Plain tensor ops: nothing here is on
the hot path; kept so `packages.models.models` imports resolve like the reference's."""
import math
import torch
import torch.nn.functional as F

_LOG_2PI = math.log(2 * math.pi)


def prior_categorical(batch_size, y_dim, device):
    """Uniform prior over y (softmax of ones)."""
    prior = F.softmax(torch.ones((batch_size, y_dim)).to(device), dim=1)
    prior.requires_grad = False
    return prior


def log_standard_gaussian(x):
    """log N(x | 0, I), summed over the last axis."""
    return torch.sum(-0.5 * _LOG_2PI - x ** 2 / 2, dim=-1)


def log_gaussian(x, mu, log_var):
    """log N(x | mu, exp(log_var)), summed over the last axis."""
    log_pdf = -0.5 * _LOG_2PI - log_var / 2 - (x - mu) ** 2 / (2 * torch.exp(log_var))
    return torch.sum(log_pdf, dim=-1)


def log_standard_categorical(p, eps):
    """Cross entropy between p and a 0.5 Bernoulli prior per label.  Like the reference
    (quirk Q15) this moves the prior with `p.get_device()`, which fails for host tensors."""
    prior = 0.5 * torch.ones_like(p).to(p.get_device())
    prior.requires_grad = False
    return -torch.sum((p * torch.log(prior + eps) + (1 - p) * torch.log(1 - prior + eps)), dim=1)
