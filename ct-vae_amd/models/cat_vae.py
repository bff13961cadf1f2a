"""CategoricalVAE (models/cat_vae.py:9-210, configs/cat_vae.yaml) on the HIP path — SURVEY.md §8f rank 4.

VanillaVAE's five Conv-BN-LeakyReLU blocks, decoder and ``final_layer`` (same kernels, same parameter names) around a
categorical latent: one head ``fc_z`` -> logits ``[B, latent_dim, categorical_dim]``, the Gumbel-softmax
reparameterisation (``ctvae_gumbel_softmax_forward/backward``) and the KL term against the uniform categorical prior
(``ctvae_cat_kl_forward/backward``).  ``loss = alpha * mse + M_N * kld``; the temperature is annealed by
``loss_function`` exactly as the reference does (every ``anneal_interval`` batches, while training, never below the
initial temperature).  Deliberate, documented difference: ``reparameterize`` / ``forward`` take an optional ``u`` so the
uniform draws can be injected.
"""
import numpy as np
import torch
from torch import nn

from .. import kernels as K
from .base import BaseVAE
from .blocks import ConvBNLeaky
from .packing import PackedLinear
from .types_ import List, Tensor
from .vanilla_vae import _FinalLayer


class CategoricalVAE(BaseVAE):

    def __init__(self, in_channels: int, latent_dim: int, categorical_dim: int = 40, hidden_dims: List = None,
                 temperature: float = 0.5, anneal_rate: float = 3e-5, anneal_interval: int = 100, alpha: float = 30.,
                 **kwargs) -> None:
        super().__init__()
        self.latent_dim = latent_dim
        self.categorical_dim = categorical_dim
        self.temp = temperature
        self.min_temp = temperature
        self.anneal_rate = anneal_rate
        self.anneal_interval = anneal_interval
        self.alpha = alpha
        if hidden_dims is None:
            hidden_dims = [32, 64, 128, 256, 512]
        if hidden_dims[-1] != 512:
            raise ValueError("CategoricalVAE.decode assumes hidden_dims[-1] == 512 (cat_vae.py:113)")

        enc, c = [], in_channels
        for h in hidden_dims:
            enc.append(ConvBNLeaky(c, h, 3, 2, 1))
            c = h
        self.encoder = nn.Sequential(*enc)
        n_lat = latent_dim * categorical_dim
        self.fc_z = PackedLinear(hidden_dims[-1] * 4, n_lat)
        self._head_spec = K.ConvSpec(K.CONV, hidden_dims[-1] * 4, n_lat, 1)
        self.decoder_input = PackedLinear(n_lat, hidden_dims[-1] * 4)
        self._dec_in_spec = K.ConvSpec(K.CONV, n_lat, hidden_dims[-1] * 4, 1)
        hidden_dims.reverse()                  # the reference mutates the caller's list too (cat_vae.py:57)
        dec = []
        for i in range(len(hidden_dims) - 1):
            dec.append(ConvBNLeaky(hidden_dims[i], hidden_dims[i + 1], 3, 2, 1, out_pad=1, transposed=True))
        self.decoder = nn.Sequential(*dec)
        self.final_layer = _FinalLayer(hidden_dims[-1], 3)
        self.flatten_parameters()

    def encode(self, input: Tensor) -> List[Tensor]:
        """[B,C,64,64] -> [logits [B, latent_dim, categorical_dim]] (cat_vae.py:90-104)."""
        self.attach_grads()
        h = self.encoder(K.to_nhwc(input))
        z = K.flatten_linear(h, self.fc_z.weight, self.fc_z.bias, self._head_spec.co)     # torch.flatten(start_dim=1) on NCHW
        return [z.view(-1, self.latent_dim, self.categorical_dim)]

    def decode(self, z: Tensor) -> Tensor:
        """[B, latent_dim*categorical_dim] -> [B,3,64,64] (cat_vae.py:106-116)."""
        self.attach_grads()
        B = z.shape[0]
        h = K.ConvAct.apply(z.reshape(B, 1, 1, -1), self.decoder_input.weight, self.decoder_input.bias, None, self._dec_in_spec)
        h = K._ToNHWC.apply(h.view(B, 512, 2, 2))                        # .view(-1,512,2,2) is NCHW
        h = self.decoder(h)
        return K.to_nchw_view(self.final_layer(h))

    def reparameterize(self, z: Tensor, eps: float = 1e-7, u: Tensor = None) -> Tensor:
        """Gumbel-softmax sample [B, latent_dim*categorical_dim] (cat_vae.py:118-132); u defaults to fresh U[0,1) draws."""
        if u is None:
            u = torch.rand(z.shape, dtype=z.dtype, device=z.device)
        s = K.GumbelSoftmax.apply(z, u.to(z.device), self.temp, eps)
        return s.view(-1, self.latent_dim * self.categorical_dim)

    def forward(self, input: Tensor, u: Tensor = None, **kwargs) -> List[Tensor]:
        q = self.encode(input)[0]
        z = self.reparameterize(q, u=u)
        return [self.decode(z), input, q]

    def loss_function(self, *args, **kwargs) -> dict:
        """alpha * mse + M_N * KL(softmax(q) || uniform) (cat_vae.py:140-169); 'KLD' carries the reference's flipped sign."""
        recons, input, q = args[0], args[1], args[2]
        kld_weight = kwargs['M_N']
        batch_idx = kwargs['batch_idx']
        if batch_idx % self.anneal_interval == 0 and self.training:
            self.temp = np.maximum(self.temp * np.exp(- self.anneal_rate * batch_idx), self.min_temp)
        mse = K.VAELoss.apply(K.to_nhwc(recons), K.to_nhwc(input), None, None, None, 0.0)
        kld = K.CatKL.apply(q, 1e-7)
        loss = self.alpha * mse[0] + kld_weight * kld
        return {'loss': loss, 'Reconstruction_Loss': mse[0], 'KLD': -kld}

    def sample(self, num_samples: int, current_device: int, **kwargs) -> Tensor:
        """One-hot draws from the uniform categorical prior -> decoder (cat_vae.py:171-192)."""
        M = num_samples * self.latent_dim
        np_y = np.zeros((M, self.categorical_dim), dtype=np.float32)
        np_y[range(M), np.random.choice(self.categorical_dim, M)] = 1
        z = torch.from_numpy(np_y).view(num_samples, self.latent_dim * self.categorical_dim).to(current_device)
        return self.decode(z)

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        return self.forward(x)[0]
