"""Drop-in replacement for the reference's packages/models/models.py.

Same class names, constructor arguments, attribute names, forward() return
tuples and state_dict keys, so scripts/training_*.py,
scripts/reconstruct_*.py and packages/models/mcem.py run unchanged.  Every
nn.Linear + activation pair is one call of `_dense`:

  * CUDA tensors  -> hand-written HIP kernels of libdvae_hip.so through
    torch.autograd.Function shells (disentangled-vae_amd/ops.py): fused
    Linear+tanh/relu/sigmoid/exp on the fp32 matrix cores, `torch.cat([x, y])`
    folded into the kernel as a two-block K loop.  No fallback: a missing
    library raises.
  * CUDA tensors at the reference geometry (x 513, h [128, 128], z 16, y 0 / 1 / 513) through the M1 / M2 model forward
    -> ONE autograd Function for the whole model (disentangled-vae_amd/module_path.py: 2 launches forward, 3 backward,
    parameters and their gradients as views of flat buffers), so `loss.backward()` + stock `torch.optim.Adam` are bound by
    PyTorch's own host cost, not by 64 launches per step.  DVAE_MODULE_PATH=layers keeps the per-layer Functions.
  * host tensors  -> the same ATen ops the reference runs on CPU (the
    reference's CPU mode; scripts select it when no GPU exists).  This is
    selected by the tensors' device only, never as a substitute for the HIP path.

Parameter construction order (and therefore the torch RNG stream consumed by
nn.Linear's default init followed by xavier_normal_) follows the reference, so
`torch.manual_seed(s)` yields the same initial weights as the reference does.
"""
import torch
from torch import nn
from torch.nn import init
import torch.nn.functional as F

from packages.models.distributions import log_gaussian, log_standard_gaussian
from packages import _native

_ACT_CODE = {"none": 0, "tanh": 1, "relu": 2, "sigmoid": 3, "exp": 4}
_ACT_HOST = {"none": lambda t: t, "tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid, "exp": torch.exp}


def _dense(layer, x, act, extra=None):
    """act(layer([x | extra])) -- one fused HIP kernel for CUDA tensors."""
    if x.is_cuda:
        return _native.ops().linear_act(x, layer.weight, layer.bias, _ACT_CODE[act], extra)
    if extra is not None:
        x = torch.cat([x, extra], dim=-1)
    return _ACT_HOST[act](F.linear(x, layer.weight, layer.bias))


def _linear_stack(widths):
    return [nn.Linear(widths[i - 1], widths[i]) for i in range(1, len(widths))]


def _stack_ok(layers, x):
    import os
    return (x.is_cuda and x.dtype == torch.float32 and len(layers) > 1 and all(isinstance(l, nn.Linear) for l in layers)
            and os.environ.get("DVAE_LINEAR_STACK", "1") != "0")


def _run_stack(layers, acts, x, extra=None):
    """The Linear + activation layers `layers` as ONE autograd node on the device (the same kernels as layer by layer: `ops.LinearStack`)."""
    return _native.ops().linear_stack(x, [(l.weight, l.bias) for l in layers], [_ACT_CODE[a] for a in acts], extra)


def _run_hidden(hidden, x, act, extra=None):
    """`for layer in self.hidden: x = act(layer(x))` (reference models.py:59-60, 103-104, 120-121)."""
    if _stack_ok(list(hidden), x):
        return _run_stack(list(hidden), [act] * len(hidden), x, extra)
    for layer in hidden:
        if isinstance(layer, nn.Linear):
            x = _dense(layer, x, act, extra)
        else:                                   # BatchNorm1d of Classifier(batch_norm=True): never used by the scripts
            x = _ACT_HOST[act](layer(x))
        extra = None
    return x


def _xavier_reset(module):
    """xavier_normal_ on every Linear weight, zero bias (reference models.py:137-141)."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            init.xavier_normal_(m.weight.data)
            if m.bias is not None:
                m.bias.data.zero_()


class Stochastic(nn.Module):
    """z = mu + exp(log_var / 2) * epsilon (reference models.py:8-22).

    epsilon is drawn like the reference does: torch.randn(mu.size()) on the HOST
    generator, then moved to mu's device (quirk Q1), so a shared seed reproduces
    the reference's noise.  `Stochastic.epsilon_fn` (callable mu -> epsilon) overrides
    the source, e.g. a device generator in throughput runs or injected noise in tests."""
    epsilon_fn = None

    @staticmethod
    def draw_epsilon(shape, device):
        """The same noise source for callers that have no mu yet (the whole-model fused forward): `epsilon_fn` sees a zero
        tensor of mu's shape on mu's device."""
        if Stochastic.epsilon_fn is not None:
            return Stochastic.epsilon_fn(torch.zeros(shape, device=device))
        epsilon = torch.randn(shape, requires_grad=False)
        return epsilon.to(device, non_blocking=True) if device.type == "cuda" else epsilon

    def reparametrize(self, mu, log_var):
        if Stochastic.epsilon_fn is not None:
            epsilon = Stochastic.epsilon_fn(mu)
        else:
            epsilon = torch.randn(mu.size(), requires_grad=False)
            if mu.is_cuda:
                epsilon = epsilon.to(mu.get_device(), non_blocking=True)
        if mu.is_cuda:
            return _native.ops().Reparam.apply(mu, log_var, epsilon.to(torch.float32))
        std = log_var.mul(0.5).exp_()
        return mu.addcmul(std, epsilon)


class GaussianSample(Stochastic):
    def __init__(self, in_features, out_features):
        super(GaussianSample, self).__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.mu = nn.Linear(in_features, out_features)
        self.log_var = nn.Linear(in_features, out_features)

    def forward(self, x):
        mu = _dense(self.mu, x, "none")
        log_var = _dense(self.log_var, x, "none")
        return self.reparametrize(mu, log_var), mu, log_var


class Classifier(nn.Module):
    """[x_dim, h_dim, y_dim]: relu hidden stack, sigmoid output (reference models.py:41-63)."""

    def __init__(self, dims, batch_norm=False):
        super(Classifier, self).__init__()
        [x_dim, h_dim, y_dim] = dims
        widths = [x_dim, *h_dim]
        layers = []
        for i in range(1, len(widths)):
            layers.append(nn.Linear(widths[i - 1], widths[i]))
            if batch_norm:
                layers.append(nn.BatchNorm1d(widths[i]))
        self.hidden = nn.ModuleList(layers)
        self.output_layer = nn.Linear(h_dim[-1], y_dim)

    def forward(self, x):
        layers = [*self.hidden, self.output_layer]
        if _stack_ok(layers, x):
            return _run_stack(layers, ["relu"] * len(self.hidden) + ["sigmoid"], x)
        x = _run_hidden(self.hidden, x, "relu")
        return _dense(self.output_layer, x, "sigmoid")


class Classifier2Classes(nn.Module):
    """Two-class softmax variant (reference models.py:65-89; no caller in the scripts)."""

    def __init__(self, dims, batch_norm=False):
        super(Classifier2Classes, self).__init__()
        [x_dim, h_dim, y_dim] = dims
        widths = [x_dim, *h_dim]
        layers = []
        for i in range(1, len(widths)):
            layers.append(nn.Linear(widths[i - 1], widths[i]))
            if batch_norm:
                layers.append(nn.BatchNorm1d(widths[i]))
        self.hidden = nn.ModuleList(layers)
        self.output_layer = nn.Linear(h_dim[-1], 2 * y_dim)
        self.softmax = nn.Softmax(dim=1)
        self.y_dim = y_dim

    def forward(self, x):
        x = _run_hidden(self.hidden, x, "relu")
        return self.softmax(_dense(self.output_layer, x, "none").view(-1, 2, self.y_dim))


class Encoder(nn.Module):
    """[x_dim, h_dim, z_dim]: tanh stack then the Gaussian sample layer (reference models.py:91-105).
    `extra` is the label block of M2's torch.cat([x, y], dim=1), consumed without the concat."""

    def __init__(self, dims, sample_layer=GaussianSample):
        super(Encoder, self).__init__()
        [x_dim, h_dim, z_dim] = dims
        self.hidden = nn.ModuleList(_linear_stack([x_dim, *h_dim]))
        self.sample = sample_layer(h_dim[-1], z_dim)

    def forward(self, x, extra=None):
        return self.sample(_run_hidden(self.hidden, x, "tanh", extra))


class Decoder(nn.Module):
    """[z_dim, h_dim, x_dim]: tanh stack, exp(Linear) output = variance / power estimate
    (reference models.py:108-122).  Accepts [N, L] and [N, R, L] inputs (mcem.py:283)."""

    def __init__(self, dims):
        super(Decoder, self).__init__()
        [z_dim, h_dim, x_dim] = dims
        self.hidden = nn.ModuleList(_linear_stack([z_dim, *h_dim]))
        self.reconstruction = nn.Linear(h_dim[-1], x_dim)

    def forward(self, x, extra=None):
        layers = [*self.hidden, self.reconstruction]
        if _stack_ok(layers, x):
            return _run_stack(layers, ["tanh"] * len(self.hidden) + ["exp"], x, extra)
        return _dense(self.reconstruction, _run_hidden(self.hidden, x, "tanh", extra), "exp")


class VariationalAutoencoder(nn.Module):
    """M1: dims = [x_dim, z_dim, h_dim] (reference models.py:125-182)."""

    def __init__(self, dims):
        super(VariationalAutoencoder, self).__init__()
        [x_dim, z_dim, h_dim] = dims
        self.z_dim = z_dim
        self.flow = None
        self.encoder = Encoder([x_dim, h_dim, z_dim])
        self.decoder = Decoder([z_dim, list(reversed(h_dim)), x_dim])
        self.kl_divergence = 0
        _xavier_reset(self)

    # kl_divergence ([B] tensor after forward, reference models.py:175) is a side value no script reads: the fused forward
    # leaves it pending and it is evaluated when somebody asks
    @property
    def kl_divergence(self):
        pend = self.__dict__.get("_kl_pending")
        if pend is not None:
            self.__dict__["_kl_value"] = self._kld_v2(None, pend)
            self.__dict__["_kl_pending"] = None
        return self.__dict__.get("_kl_value", 0)

    @kl_divergence.setter
    def kl_divergence(self, value):
        self.__dict__["_kl_value"] = value
        self.__dict__["_kl_pending"] = None

    def _kld(self, z, q_param, p_param=None):
        """log q(z|x) - log p(z) for one sample z (optionally through a normalising flow).  No script calls it
        (reference models.py:143-163 is equally unreachable); kept so the attribute exists."""
        log_q = log_gaussian(z, *q_param)
        if self.flow is not None:
            z, log_dets = self.flow(z)
            log_q = log_q - sum(log_dets)
        log_p = log_standard_gaussian(z) if p_param is None else log_gaussian(z, *p_param)
        return log_q - log_p

    def _kld_v2(self, z, q_param):
        # per-frame KL without the "+1" (quirk Q2); side-effect value only, plain tensor ops
        (mu, log_var) = q_param
        return -0.5 * torch.sum(log_var - mu.pow(2) - log_var.exp(), axis=-1)

    def add_flow(self, flow):
        self.flow = flow

    def forward(self, x, y=None):
        if x.is_cuda and type(self) is VariationalAutoencoder:
            eng = _native.module_path().engine_for(self, "M1", x, None)
            if eng is not None:
                r, z, z_mu, z_log_var = _native.module_path().run(eng, x, None, Stochastic.draw_epsilon((x.shape[0], self.z_dim), x.device))
                self.__dict__["_kl_pending"] = (z_mu.detach(), z_log_var.detach())
                return r, z_mu, z_log_var
        z, z_mu, z_log_var = self.encoder(x)
        self.kl_divergence = self._kld_v2(z, (z_mu, z_log_var))
        return self.decoder(z), z_mu, z_log_var

    def sample(self, z):
        return self.decoder(z)


def _encode_xy(encoder, x, y):
    return encoder(x, y) if x.is_cuda else encoder(torch.cat([x, y], dim=1))


def _decode_zy(decoder, z, y):
    return decoder(z, y) if z.is_cuda else decoder(torch.cat([z, y], dim=1))


class DeepGenerativeModel(VariationalAutoencoder):
    """M2: dims = [x_dim, y_dim, z_dim, h_dim]; encoder on [x|y], decoder on [z|y]
    (reference models.py:185-218).  Does not set kl_divergence, like the reference."""

    def __init__(self, dims, classifier):
        [x_dim, self.y_dim, z_dim, h_dim] = dims
        super(DeepGenerativeModel, self).__init__([x_dim, z_dim, h_dim])
        self.encoder = Encoder([x_dim + self.y_dim, h_dim, z_dim])
        self.decoder = Decoder([z_dim + self.y_dim, list(reversed(h_dim)), x_dim])
        self.classifier = classifier
        _xavier_reset(self)

    def forward(self, x, y):
        if x.is_cuda and type(self) is DeepGenerativeModel:
            eng = _native.module_path().engine_for(self, "M2", x, y)
            if eng is not None:
                r, z, z_mu, z_log_var = _native.module_path().run(eng, x, y, Stochastic.draw_epsilon((x.shape[0], self.z_dim), x.device))
                return r, z_mu, z_log_var
        z, z_mu, z_log_var = _encode_xy(self.encoder, x, y)
        return _decode_zy(self.decoder, z, y), z_mu, z_log_var

    def test(self, x):
        # the reference calls an undefined global `classify` here (quirk Q11); routed to the method
        y = self.classify(x)
        z, z_mu, z_log_var = _encode_xy(self.encoder, x, y)
        return _decode_zy(self.decoder, z, y), z_mu, z_log_var

    def classify(self, x):
        return self.classifier(x)

    def sample(self, z, y):
        return _decode_zy(self.decoder, z, y.float())


class DeepGenerativeModel_v2(VariationalAutoencoder):
    """Encoder on x only, decoder on [z|y] (reference models.py:220-242; no caller)."""

    def __init__(self, dims, classifier):
        [x_dim, self.y_dim, z_dim, h_dim] = dims
        super(DeepGenerativeModel_v2, self).__init__([x_dim, z_dim, h_dim])
        self.encoder = Encoder([x_dim, h_dim, z_dim])
        self.decoder = Decoder([z_dim + self.y_dim, list(reversed(h_dim)), x_dim])
        _xavier_reset(self)

    def forward(self, x, y):
        z, z_mu, z_log_var = self.encoder(x)
        return _decode_zy(self.decoder, z, y), z_mu, z_log_var

    def sample(self, z, y):
        return _decode_zy(self.decoder, z, y.float())


class DeepGenerativeModel_v3(nn.Module):
    """enc(x) / dec([z|y]) / classifier(x): the body of M2_info (reference models.py:245-297)."""

    def __init__(self, dims):
        [x_dim, self.y_dim, z_dim, h_dim] = dims
        self.z_dim = z_dim
        self.flow = None
        super(DeepGenerativeModel_v3, self).__init__()
        self.encoder = Encoder([x_dim, h_dim, z_dim])
        self.decoder = Decoder([z_dim + self.y_dim, list(reversed(h_dim)), x_dim])
        self.classifier = Classifier([x_dim, h_dim, self.y_dim])
        _xavier_reset(self)

    def classify(self, x):
        return self.classifier(x)

    def _forward_z(self, x, y):
        """-> (x_mu, z, z_mu, z_log_var): encoder + decoder as one autograd Function on CUDA at the reference geometry (the fused
        module path, model "M2_DEC": z is an output, so a loss on z -- the auxiliary classifier of _v5 -- differentiates through)."""
        if x.is_cuda and type(self) is DeepGenerativeModel_v3:
            eng = _native.module_path().engine_for(self, "M2_DEC", x, y)
            if eng is not None:
                return _native.module_path().run(eng, x, y, Stochastic.draw_epsilon((x.shape[0], self.z_dim), x.device))
        z, z_mu, z_log_var = self.encoder(x)
        return _decode_zy(self.decoder, z, y), z, z_mu, z_log_var

    def forward(self, x, y):
        r, _, z_mu, z_log_var = self._forward_z(x, y)
        return r, z_mu, z_log_var

    def sample(self, z, y):
        return _decode_zy(self.decoder, z, y.float())


class DeepGenerativeModel_v4(VariationalAutoencoder):
    """v3 plus an auxiliary classifier on z, 4-tuple forward (reference models.py:299-353; no caller)."""

    def __init__(self, dims):
        [x_dim, self.y_dim, z_dim, h_dim] = dims
        super(DeepGenerativeModel_v4, self).__init__([x_dim, z_dim, h_dim])
        self.encoder = Encoder([x_dim, h_dim, z_dim])
        self.decoder = Decoder([z_dim + self.y_dim, list(reversed(h_dim)), x_dim])
        self.classifier = Classifier([x_dim, h_dim, self.y_dim])
        self.auxiliary = Classifier([z_dim, h_dim, self.y_dim])
        _xavier_reset(self)

    def classify_fromX(self, x):
        return self.classifier(x)

    def classify_fromZ(self, z):
        return self.auxiliary(z)

    def forward(self, x, y):
        z, z_mu, z_log_var = self.encoder(x)
        return _decode_zy(self.decoder, z, y), z, z_mu, z_log_var

    def sample(self, z, y):
        return _decode_zy(self.decoder, z, y.float())


class Encoder_Classifier(nn.Module):
    """Encoder + classifier pair (reference models.py:355-388; no caller)."""

    def __init__(self, dims):
        [x_dim, self.y_dim, z_dim, h_dim] = dims
        super(Encoder_Classifier, self).__init__()
        self.encoder = Encoder([x_dim, h_dim, z_dim])
        self.classifier = Classifier([x_dim, h_dim, self.y_dim])
        _xavier_reset(self)

    def classify(self, x):
        return self.classifier(x)

    def forward(self, x):
        return self.encoder(x)


class DeepGenerativeModel_v5(nn.Module):
    """M2_info: enc_dec_clf (a _v3) + adversarial auxiliary classifier on z; forward returns
    (x_mu, z, z_mu, z_log_var) (reference models.py:390-444)."""

    def __init__(self, dims):
        [x_dim, self.y_dim, z_dim, h_dim] = dims
        super(DeepGenerativeModel_v5, self).__init__()
        self.enc_dec_clf = DeepGenerativeModel_v3([x_dim, self.y_dim, z_dim, h_dim])
        self.auxiliary = Classifier([z_dim, h_dim, self.y_dim])
        _xavier_reset(self)

    def classify_fromX(self, x):
        return self.enc_dec_clf.classifier(x)

    def classify_fromZ(self, z):
        return self.auxiliary(z)

    def forward(self, x, y):
        return self.enc_dec_clf._forward_z(x, y)

    def sample(self, z, y):
        return _decode_zy(self.enc_dec_clf.decoder, z, y.float())
