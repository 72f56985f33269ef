"""Drop-in replacement for the reference's packages/models/utils.py (loss zoo).

Every loss -- `elbo` (reference utils.py:73-76), the `binary_cross_entropy` family (:55-66), `L_loss` / `U_loss` /
`ikatura_saito_divergence` (:68-105) and the squared-error mask / signal / magnitude-spectrum losses (:107-118) -- runs as HIP
reduction kernels with hand-written backward when its inputs are CUDA tensors (disentangled-vae_amd/ops.py: Elbo, Bce, Bce2,
IsRows, SqErr; csrc/losses.hip); host tensors take the reference's own ATen expression (its CPU mode).  The label helpers
(`enumerate_discrete`, `onehot`, `log_sum_exp`, `f1_loss`) are bookkeeping, not losses, and stay plain tensor code.
"""
import torch
from torch.autograd import Variable

from packages import _native


def enumerate_discrete(x, y_dim):
    """All one-hot labels repeated for x's batch size: [y_dim * B, y_dim] (reference :5-28)."""
    batch_size = x.size(0)
    generated = torch.eye(y_dim).repeat_interleave(batch_size, dim=0)
    if x.is_cuda:
        generated = generated.cuda()
    return Variable(generated.float())


def onehot(k):
    """-> function(label) = the length-k indicator vector of `label` (all zeros for label >= k; negative labels count from the end,
    like the index assignment of reference :30-42)."""
    slots = torch.arange(k)

    def encode(label):
        label = int(label)
        if label < -k:
            raise IndexError(f"index {label} is out of bounds for dimension 0 with size {k}")
        return (slots == (label if label >= 0 else label + k)).to(torch.get_default_dtype())
    return encode


def log_sum_exp(tensor, dim=-1, sum_op=torch.sum):
    """log(sum_op(exp(tensor)) + 1e-8 * exp(peak)) evaluated around the peak of `tensor` along `dim` (reference :44-53)."""
    peak = tensor.amax(dim=dim, keepdim=True)
    return peak + (sum_op((tensor - peak).exp(), dim=dim, keepdim=True) + 1e-8).log()


def _on_gpu(*ts):
    return any(t is not None and t.is_cuda for t in ts)


def _kernel_case(*ts, dtype=torch.float32, frozen=()):
    """The secondary losses (no caller in the shipped scripts) have HIP kernels for the case their experiments used: CUDA
    tensors of one dtype and ONE shape (no broadcasting), and -- `frozen` -- operands that take no gradient.  Anything else the reference
    expressions accept (a [B, 1] target against [B, F], float64, complex spectra that require grad) runs as that expression in ATen on
    the tensors' own device, as the reference would; elbo / binary_cross_entropy (the train step's losses) never take this exit."""
    ts = [t for t in ts if t is not None]
    if not ts or not all(t.is_cuda for t in ts):
        return False
    if any(t.dtype != dtype or t.shape != ts[0].shape or t.device != ts[0].device for t in ts):
        return False
    return not any(t is not None and t.requires_grad for t in frozen)


def binary_cross_entropy(r, x, eps):
    if _on_gpu(r, x):
        return _native.ops().Bce.apply(r, x, eps, 0)
    return -torch.mean(torch.sum(x * torch.log(r + eps) + (1 - x) * torch.log(1 - r + eps), dim=-1))


def binary_cross_entropy_v2(r, eps):
    if _on_gpu(r):
        return _native.ops().Bce.apply(r, None, eps, 1)
    return -torch.mean(torch.sum(0.5 * torch.log(r + eps) + 0.5 * torch.log(1 - r + eps), dim=-1))


def binary_cross_entropy_v3(r, eps):
    if _on_gpu(r):
        return _native.ops().Bce.apply(r, None, eps, 2)
    return -torch.mean(torch.sum(r * torch.log(r + eps) + (1 - r) * torch.log(1 - r + eps), dim=-1))


def binary_cross_entropy_2classes(r1, r2, x, eps):
    if _kernel_case(r1, r2, x):
        return _native.ops().Bce2.apply(r1, r2, x, eps)
    return -torch.mean(torch.sum(x * torch.log(r1 + eps) + (1 - x) * torch.log(r2 + eps), dim=-1))


def _is_rows(x, r, eps):
    # Itakura-Saito divergence per frame; note eps only inside log(x + eps) (reference :68-71)
    return torch.sum(x / r - torch.log(x + eps) + torch.log(r) - 1, dim=-1)


def _kl_rows(mu, logvar):
    # KL without the "+1" (quirk Q2)
    return -0.5 * torch.sum(logvar - mu.pow(2) - logvar.exp(), dim=-1)


def ikatura_saito_divergence(r, x, eps):
    if _kernel_case(r, x):
        return _native.ops().IsRows.apply(x, r, None, None, eps)
    return _is_rows(x, r, eps)


def elbo(x, r, mu, logvar, eps):
    """-> (recon + KL, recon, KL), 0-dim tensors (reference utils.py:73-76)."""
    if _on_gpu(x, r, mu, logvar):
        return _native.ops().Elbo.apply(x, r, mu, logvar, eps)
    recon = torch.mean(_is_rows(x, r, eps))
    KL = torch.mean(_kl_rows(mu, logvar))
    return recon + KL, recon, KL


def L_loss(x, r, mu, logvar, eps):
    if _kernel_case(x, r) and _kernel_case(mu, logvar):
        recon, KL = _native.ops().IsRows.apply(x, r, mu, logvar, eps)
        return recon + KL, recon, KL
    recon = _is_rows(x, r, eps)
    KL = _kl_rows(mu, logvar)
    return recon + KL, recon, KL


def U_loss(x, r, mu, logvar, y_hat_soft, eps):
    """Unlabelled objective of the M2v3/v4 experiments (reference :83-105)."""
    if _on_gpu(x, r, mu, logvar, y_hat_soft):
        # L_soft = sum_c [y L + (1 - y) L] is C * L whatever y is (its gradient with respect to y_hat_soft is zero term by term), and the
        # entropy term is binary_cross_entropy_v3: U = C * mean(L) + bce_v3(y_hat_soft) on the elbo and BCE kernels
        L, recon, KL = elbo(x, r, mu, logvar, eps)
        return y_hat_soft.shape[-1] * L + binary_cross_entropy_v3(y_hat_soft, eps), L, recon, KL
    recon = _is_rows(x, r, eps)
    KL = _kl_rows(mu, logvar)
    L = (recon + KL)[..., None]
    L_soft = torch.sum(torch.mul(y_hat_soft, L) + torch.mul(1 - y_hat_soft, L), dim=-1)
    H = -torch.sum(torch.mul(y_hat_soft, torch.log(y_hat_soft + eps))
                   + torch.mul(1 - y_hat_soft, torch.log(1 - y_hat_soft + eps)), dim=-1)
    return torch.mean(L_soft + H), torch.mean(L), torch.mean(recon), torch.mean(KL)


def mean_square_error_signal(x, y, y_hat):
    if _kernel_case(x, y, y_hat):
        return _native.ops().SqErr.apply(0, x, y, y_hat)
    return torch.mean(torch.sum(torch.square(torch.mul(y - y_hat, x)), axis=-1))


def mean_square_error_mask(y, y_hat):
    if _kernel_case(y, y_hat):
        return _native.ops().SqErr.apply(1, None, y, y_hat)
    return torch.mean(torch.sum(torch.square(y - y_hat), axis=-1))


def magnitude_spectrum_approxiamation_loss(x, s, y_hat):
    if _kernel_case(x, s, dtype=torch.complex64, frozen=(x, s)) and _kernel_case(y_hat) and y_hat.shape == x.shape:
        return _native.ops().SqErr.apply(2, x, s, y_hat)
    d = s - y_hat * x
    return torch.mean(torch.sum(torch.real(d * d.conj()), axis=-1))


def f1_loss(y_hat_hard: torch.Tensor, y: torch.Tensor, epsilon=1e-8) -> torch.Tensor:
    """-> (accuracy, precision, recall, f1) of hard 0/1 predictions against 0/1 truth (reference :120-159).  [N, C] predictions
    are reduced to their argmax class first.  The confusion counts come from three sums (hits, predicted positives, true positives)."""
    pred, truth = y_hat_hard.detach(), y.detach()
    if truth.ndim != 1 or pred.ndim not in (1, 2):
        raise AssertionError("f1_loss: y must be 1-D and y_hat_hard 1-D or 2-D")
    if pred.ndim == 2:
        pred = pred.argmax(dim=1)
    n = truth.numel()
    hits, n_pred, n_true = (truth * pred).sum(), pred.sum(), truth.sum()
    tp = hits.to(torch.float32)
    fp = (n_pred - hits).to(torch.float32)
    fn = (n_true - hits).to(torch.float32)
    tn = (n - n_pred - n_true + hits).to(torch.float32)
    accuracy = (tp + tn) / (tp + tn + fp + fn + epsilon)
    precision = tp / (tp + fp + epsilon)
    recall = tp / (tp + fn + epsilon)
    f1 = 2 * (precision * recall) / (precision + recall + epsilon)
    return accuracy, precision, recall, f1
