"""GammaVAE (models/gamma_vae.py:10-247, configs/gammavae.yaml) on the HIP path -- SURVEY.md §8f rank 4.

VanillaVAE's conv stacks with Gamma-distributed latents: the heads are Linear + Softmax (``fc_mu.0`` -> shape alpha,
``fc_var.0`` -> rate beta), the sample comes from the shape-augmentation reparameterisation of a Gamma(alpha + B, 1) draw, the
decoder ends in a Sigmoid instead of a Tanh, and the objective is mean_b(mse_b + kld_b) with the Gamma-to-Gamma KL the
reference writes out with lgamma / digamma (no KL weight).  Reparameterisation, KL and Sigmoid are kernels of csrc/gamma.hip;
the two softmaxes over the latent dimension and the Gamma draw itself (``torch.distributions.Gamma``) stay torch device ops.
``forward`` takes an optional ``zhat`` (the injected Gamma(alpha + B, 1) draw).
"""
import torch
from torch import nn

from .. import kernels as K
from .base import BaseVAE
from .blocks import ConvBNLeaky, conv_bn_leaky
from .packing import PackedBN, PackedConv, PackedLinear
from .types_ import List, Tensor


class _Holder(nn.Sequential):
    """nn.Sequential(Linear, ...) of the reference: the Linear's parameters under child "0"."""

    def __init__(self, lin):
        super().__init__()
        self.add_module("0", lin)


class _GammaFinal(nn.Module):
    """nn.Sequential(ConvTranspose2d, BatchNorm2d, LeakyReLU, Conv2d(->3), Sigmoid) (gamma_vae.py:66-78)."""

    def __init__(self, c):
        super().__init__()
        self.add_module("0", PackedConv(c, c, 3, transposed=True, bias=True))
        self.add_module("1", PackedBN(c))
        self.add_module("3", PackedConv(c, 3, 3, bias=True))
        self.spec_up = K.ConvSpec(K.CONVT, c, c, 3, 2, 1, 1, K.ACT_NONE)
        self.spec_out = K.ConvSpec(K.CONV, c, 3, 3, 1, 1, 0, K.ACT_NONE)

    def forward(self, x):
        up, bn, conv = self._modules["0"], self._modules["1"], self._modules["3"]
        h = conv_bn_leaky(x, up, bn, self.spec_up, self.training)
        return K.Sigmoid.apply(K.ConvAct.apply(h, conv.weight, conv.bias, None, self.spec_out))


class GammaVAE(BaseVAE):

    def __init__(self, in_channels: int, latent_dim: int, hidden_dims: List = None, gamma_shape: float = 8., prior_shape: float = 2.0,
                 prior_rate: float = 1., **kwargs) -> None:
        super().__init__()
        self.latent_dim, self.B = latent_dim, gamma_shape
        self.prior_alpha, self.prior_beta = torch.tensor([prior_shape]), torch.tensor([prior_rate])
        if hidden_dims is None:
            hidden_dims = [32, 64, 128, 256, 512]
        if hidden_dims[-1] != 512:
            raise ValueError("GammaVAE.decode assumes hidden_dims[-1] == 512 (gamma_vae.py:104)")
        enc, c = [], in_channels
        for h in hidden_dims:
            enc.append(ConvBNLeaky(c, h, 3, 2, 1))
            c = h
        self.encoder = nn.Sequential(*enc)
        feat = hidden_dims[-1] * 4
        self.fc_mu = _Holder(PackedLinear(feat, latent_dim))
        self.fc_var = _Holder(PackedLinear(feat, latent_dim))
        self._head_spec = K.ConvSpec(K.CONV, feat, latent_dim, 1)
        self.decoder_input = _Holder(PackedLinear(latent_dim, feat))
        self._dec_in_spec = K.ConvSpec(K.CONV, latent_dim, feat, 1)
        hidden_dims.reverse()
        dec = []
        for i in range(len(hidden_dims) - 1):
            dec.append(ConvBNLeaky(hidden_dims[i], hidden_dims[i + 1], 3, 2, 1, out_pad=1, transposed=True))
        self.decoder = nn.Sequential(*dec)
        self.final_layer = _GammaFinal(hidden_dims[-1])
        self.flatten_parameters()

    def encode(self, input: Tensor) -> List[Tensor]:
        """-> [alpha, beta]: softmax over the latent dimension of the two heads (gamma_vae.py:93-106)."""
        self.attach_grads()
        h = self.encoder(K.to_nhwc(input))
        B = h.shape[0]
        lm, lv = self.fc_mu._modules["0"], self.fc_var._modules["0"]
        a = K.flatten_linear(h, lm.weight, lm.bias, self._head_spec.co)
        b = K.flatten_linear(h, lv.weight, lv.bias, self._head_spec.co)
        return [torch.softmax(a, dim=1), torch.softmax(b, dim=1)]

    def decode(self, z: Tensor) -> Tensor:
        self.attach_grads()
        B = z.shape[0]
        lin = self.decoder_input._modules["0"]
        h = K.ConvAct.apply(z.reshape(B, 1, 1, -1), lin.weight, lin.bias, None, self._dec_in_spec)
        h = K._ToNHWC.apply(h.view(B, 512, 2, 2))
        return K.to_nchw_view(self.final_layer(self.decoder(h)))

    def reparameterize(self, alpha: Tensor, beta: Tensor, zhat: Tensor = None) -> Tensor:
        if zhat is None:
            # Gamma(alpha + B, 1).sample() without the distribution object: its argument validation reads the device (a sync that
            # a hipGraph capture of the step cannot contain); _standard_gamma is the sampler it calls
            zhat = torch._standard_gamma(alpha.detach() + self.B)
        return K.GammaReparam.apply(alpha, beta, zhat.to(alpha.device), self.B)

    def forward(self, input: Tensor, zhat: Tensor = None, **kwargs) -> List[Tensor]:
        alpha, beta = self.encode(input)
        z = self.reparameterize(alpha, beta, zhat)
        return [self.decode(z), input, alpha, beta]

    def loss_function(self, *args, **kwargs) -> dict:
        """mean_b(mse_b + kld_b) (gamma_vae.py:173-199): per-sample means of equal-sized pictures average to the plain MSE."""
        recons, input, alpha, beta = args[0], args[1], args[2], args[3]
        kld = K.GammaKL.apply(alpha, beta, float(self.prior_alpha), float(self.prior_beta))
        out = K.VAELoss.apply(K.to_nhwc(recons), K.to_nhwc(input), None, None, kld.reshape(1), 0.0)
        return {'loss': out[0]}

    def sample(self, num_samples: int, current_device: int, **kwargs) -> Tensor:
        z = torch.distributions.Gamma(self.prior_alpha, self.prior_beta).sample((num_samples, self.latent_dim))
        return self.decode(z.squeeze().to(current_device))

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        return self.forward(x)[0]
