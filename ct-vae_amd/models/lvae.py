"""LVAE -- ladder VAE (models/lvae.py:8-270, configs/lvae.yaml) on the HIP path -- SURVEY.md §8f rank 4.

Bottom-up: five Conv-BatchNorm-LeakyReLU blocks (VanillaVAE's encoder layers), each with its OWN pair of Linear heads on its
flattened feature map (latent_dims [4, 8, 16, 32, 128]).  Top-down: the top latent is sampled, then every rung predicts the
next Gaussian (``LadderBlock``: Linear + BatchNorm1d, two heads), merges it with the bottom-up posterior by precision, samples,
and adds the KL between the merged and the bottom-up Gaussian -- ``kernels.LadderMerge`` (csrc/ladder.hip), one launch per rung
each way.  Decoder and final block are VanillaVAE's; loss = mse + M_N * mean_b kl.  ``forward`` takes an optional ``eps`` list
(one [B, L] draw for the top latent, then one per rung, top-down -- injected noise).
"""
from math import floor

import torch
from torch import nn

from .. import kernels as K
from .base import BaseVAE
from .blocks import ConvBNLeaky
from .packing import PackedBN, PackedLinear, PackedLinearGroup
from .types_ import List, Tensor
from .vanilla_vae import _FinalLayer


def conv_out_shape(img_size):
    return floor((img_size + 2 - 3) / 2.) + 1


def _pad_cols(t, width):
    pad = width - t.shape[1]
    return t if pad == 0 else torch.cat([t, torch.zeros(t.shape[0], pad, dtype=t.dtype, device=t.device)], dim=1)


def _linear(x, lin):
    """nn.Linear on [B, in] through the conv kernels (zero rows up to the next multiple of 32 inputs)."""
    B = x.shape[0]
    spec = K.ConvSpec(K.CONV, lin.in_padded, lin.out_features, 1)
    return K.ConvAct.apply(_pad_cols(x, lin.in_padded).reshape(B, 1, 1, -1), lin.weight, lin.bias, None, spec).view(B, -1)


class EncoderBlock(nn.Module):
    """lvae.py:14-45: ``encoder`` = Conv k3 s2 + BatchNorm2d + LeakyReLU, ``encoder_mu`` / ``encoder_var`` on the NCHW-flattened map."""

    def __init__(self, in_channels, out_channels, latent_dim, img_size):
        super().__init__()
        self.encoder = ConvBNLeaky(in_channels, out_channels, 3, 2, 1)
        out_size = conv_out_shape(img_size)
        feat = out_channels * out_size ** 2
        self.encoder_mu = PackedLinear(feat, latent_dim)
        self.encoder_var = PackedLinear(feat, latent_dim)
        grp = PackedLinearGroup([self.encoder_mu, self.encoder_var])
        self.encoder_mu._linear_group = grp
        self.encoder_var._linear_group = grp
        self.latent_dim = latent_dim
        self._spec = K.ConvSpec(K.CONV, feat, 2 * latent_dim, 1)

    def forward(self, x):
        h = self.encoder(x)                                            # NHWC
        heads = K.flatten_linear(h, self.encoder_mu.weight, self.encoder_mu.bias, self._spec.co)
        mu, log_var = K.SplitHeads.apply(heads, self.latent_dim)
        return [h, mu, log_var]


class _LinearBN(nn.Module):
    """nn.Sequential(Linear, BatchNorm1d) (lvae.py:56-57): parameters under "0" and "1"; one conv + train-mode BatchNorm call."""

    def __init__(self, fin, fout):
        super().__init__()
        self.add_module("0", PackedLinear(fin, fout, pad_in_to=32))
        self.add_module("1", PackedBN(fout))
        self.spec = K.ConvSpec(K.CONV, self._modules["0"].in_padded, fout, 1)

    def forward(self, z):
        lin, bn = self._modules["0"], self._modules["1"]
        B = z.shape[0]
        x = _pad_cols(z, lin.in_padded).reshape(B, 1, 1, -1)
        return K.ConvBNAct.apply(x, lin.weight, lin.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training,
                                 self.spec, K.ACT_NONE, bn.num_batches_tracked).view(B, -1)


class LadderBlock(nn.Module):
    """lvae.py:47-68."""

    def __init__(self, in_channels, latent_dim):
        super().__init__()
        self.decode = _LinearBN(in_channels, latent_dim)
        self.fc_mu = PackedLinear(latent_dim, latent_dim, pad_in_to=32)
        self.fc_var = PackedLinear(latent_dim, latent_dim, pad_in_to=32)

    def forward(self, z):
        h = self.decode(z)
        return [_linear(h, self.fc_mu), _linear(h, self.fc_var)]


class LVAE(BaseVAE):

    def __init__(self, in_channels: int, latent_dims: List, hidden_dims: List, **kwargs) -> None:
        super().__init__()
        self.latent_dims, self.hidden_dims = latent_dims, hidden_dims
        self.num_rungs = len(latent_dims)
        assert len(latent_dims) == len(hidden_dims), "Length of the latent and hidden dims must be the same"
        if hidden_dims[-1] != 512 or any(l % 4 for l in latent_dims):
            raise ValueError("LVAE on the HIP path: hidden_dims[-1] == 512 (the [512,2,2] decoder seed), latent dims multiples of 4")
        mods, img_size, c = [], 64, in_channels
        for i, h in enumerate(hidden_dims):
            mods.append(EncoderBlock(c, h, latent_dims[i], img_size))
            img_size = conv_out_shape(img_size)
            c = h
        self.encoders = nn.Sequential(*mods)
        self.ladders = nn.Sequential(*[LadderBlock(latent_dims[i], latent_dims[i - 1]) for i in range(self.num_rungs - 1, 0, -1)])
        self.decoder_input = PackedLinear(latent_dims[0], hidden_dims[-1] * 4, pad_in_to=32)
        hidden_dims.reverse()
        dec = []
        for i in range(len(hidden_dims) - 1):
            dec.append(ConvBNLeaky(hidden_dims[i], hidden_dims[i + 1], 3, 2, 1, out_pad=1, transposed=True))
        self.decoder = nn.Sequential(*dec)
        self.final_layer = _FinalLayer(hidden_dims[-1], 3)
        hidden_dims.reverse()                      # the reference restores the caller's list (lvae.py:132)
        self.flatten_parameters()

    def encode(self, input: Tensor) -> List[Tensor]:
        self.attach_grads()
        h = K.to_nhwc(input)
        post_params = []
        for block in self.encoders:
            h, mu, log_var = block(h)
            post_params.append((mu, log_var))
        return post_params

    def reparameterize(self, mu: Tensor, logvar: Tensor, eps: Tensor = None) -> Tensor:
        if eps is None:
            eps = torch.randn(mu.shape, dtype=mu.dtype, device=mu.device)
        return K.Reparameterize.apply(mu, logvar, eps.to(mu.device))

    def _image(self, z):
        B = z.shape[0]
        h = _linear(z, self.decoder_input)
        h = K._ToNHWC.apply(h.view(B, self.hidden_dims[-1], 2, 2))
        return K.to_nchw_view(self.final_layer(self.decoder(h)))

    def decode(self, z: Tensor, post_params: List, eps: List = None):
        self.attach_grads()
        kl_div = 0
        post_params.reverse()
        for i, ladder_block in enumerate(self.ladders):
            mu_e, log_var_e = post_params[i]
            mu_t, log_var_t = ladder_block(z)
            e = eps[i] if eps is not None else torch.randn(mu_e.shape, dtype=mu_e.dtype, device=mu_e.device)
            z, kl = K.LadderMerge.apply(mu_e, log_var_e, mu_t, log_var_t, e.to(mu_e.device))
            kl_div = kl_div + kl
        return self._image(z), kl_div

    def forward(self, input: Tensor, eps: List = None, **kwargs) -> List[Tensor]:
        post_params = self.encode(input)
        mu, log_var = post_params.pop()
        z = self.reparameterize(mu, log_var, eps[0] if eps is not None else None)
        recons, kl_div = self.decode(z, post_params, eps[1:] if eps is not None else None)
        return [recons, input, kl_div]

    def loss_function(self, *args, **kwargs) -> dict:
        recons, input, kl_div = args[0], args[1], args[2]
        kld_loss = torch.mean(kl_div, dim=0)
        out = K.VAELoss.apply(K.to_nhwc(recons), K.to_nhwc(input), None, None, (kwargs['M_N'] * kld_loss).reshape(1), 0.0)
        return {'loss': out[0], 'Reconstruction_Loss': out[1], 'KLD': -kld_loss}

    def sample(self, num_samples: int, current_device: int, **kwargs) -> Tensor:
        z = torch.randn(num_samples, self.latent_dims[-1]).to(current_device)
        for ladder_block in self.ladders:
            mu, log_var = ladder_block(z)
            z = self.reparameterize(mu, log_var)
        return self._image(z)

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        return self.forward(x)[0]
