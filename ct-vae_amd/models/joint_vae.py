"""JointVAE (models/joint_vae.py:9-277, configs/joint_vae.yaml) on the HIP path — SURVEY.md §8f rank 4.

VanillaVAE's conv stacks and kernels around a joint latent: a Gaussian part (heads ``fc_mu`` / ``fc_var``, one GEMM) and ONE
categorical variable (head ``fc_z`` -> logits ``[B, categorical_dim]``), sampled with the Gaussian and the Gumbel-softmax
reparameterisation kernels and concatenated for ``decoder_input``.  The objective is alpha*mse + M_N*(gamma_d |C_d - KL_d| +
gamma_c |C_c - KL_c|) with capacities that follow the model's call counter (``num_iter``, advanced while training): host
state per step, so the harness keeps these steps eager (``graph_safe = False``).  ``forward`` takes optional ``eps`` / ``u``
(injected noise).
"""
import numpy as np
import torch
from torch import nn

from .. import kernels as K
from .base import BaseVAE
from .blocks import ConvBNLeaky
from .packing import PackedLinear, PackedLinearGroup
from .types_ import List, Tensor
from .vanilla_vae import _FinalLayer


class JointVAE(BaseVAE):
    num_iter = 1

    def __init__(self, in_channels: int, latent_dim: int, categorical_dim: int, latent_min_capacity: float = 0.,
                 latent_max_capacity: float = 25., latent_gamma: float = 30., latent_num_iter: int = 25000,
                 categorical_min_capacity: float = 0., categorical_max_capacity: float = 25., categorical_gamma: float = 30.,
                 categorical_num_iter: int = 25000, hidden_dims: List = None, temperature: float = 0.5,
                 anneal_rate: float = 3e-5, anneal_interval: int = 100, alpha: float = 30., **kwargs) -> None:
        super().__init__()
        self.latent_dim, self.categorical_dim = latent_dim, categorical_dim
        self.temp = self.min_temp = temperature
        self.anneal_rate, self.anneal_interval, self.alpha = anneal_rate, anneal_interval, alpha
        self.cont_min, self.cont_max = latent_min_capacity, latent_max_capacity
        self.disc_min, self.disc_max = categorical_min_capacity, categorical_max_capacity
        self.cont_gamma, self.disc_gamma = latent_gamma, categorical_gamma
        self.cont_iter, self.disc_iter = latent_num_iter, categorical_num_iter
        self.graph_safe = False                # the capacities follow num_iter: a captured step would freeze them
        if hidden_dims is None:
            hidden_dims = [32, 64, 128, 256, 512]
        if hidden_dims[-1] != 512:
            raise ValueError("JointVAE.decode assumes hidden_dims[-1] == 512 (joint_vae.py:137)")
        enc, c = [], in_channels
        for h in hidden_dims:
            enc.append(ConvBNLeaky(c, h, 3, 2, 1))
            c = h
        self.encoder = nn.Sequential(*enc)
        feat = hidden_dims[-1] * 4
        self.fc_mu = PackedLinear(feat, latent_dim)
        self.fc_var = PackedLinear(feat, latent_dim)
        grp = PackedLinearGroup([self.fc_mu, self.fc_var])
        self.fc_mu._linear_group = grp
        self.fc_var._linear_group = grp
        self._head_spec = K.ConvSpec(K.CONV, feat, 2 * latent_dim, 1)
        self.fc_z = PackedLinear(feat, categorical_dim)
        self._cat_spec = K.ConvSpec(K.CONV, feat, categorical_dim, 1)
        self.decoder_input = PackedLinear(latent_dim + categorical_dim, feat, pad_in_to=32)   # 168 -> 192 rows (zeros)
        self._dec_in_spec = K.ConvSpec(K.CONV, self.decoder_input.in_padded, feat, 1)
        hidden_dims.reverse()
        dec = []
        for i in range(len(hidden_dims) - 1):
            dec.append(ConvBNLeaky(hidden_dims[i], hidden_dims[i + 1], 3, 2, 1, out_pad=1, transposed=True))
        self.decoder = nn.Sequential(*dec)
        self.final_layer = _FinalLayer(hidden_dims[-1], 3)
        self.flatten_parameters()

    def encode(self, input: Tensor) -> List[Tensor]:
        """-> [mu [B,L], log_var [B,L], logits [B,Q]] (joint_vae.py:110-127)."""
        self.attach_grads()
        h = self.encoder(K.to_nhwc(input))
        B = h.shape[0]
        heads = K.flatten_linear(h, self.fc_mu.weight, self.fc_mu.bias, self._head_spec.co)
        mu, log_var = K.SplitHeads.apply(heads, self.latent_dim)
        z = K.flatten_linear(h, self.fc_z.weight, self.fc_z.bias, self._cat_spec.co)
        return [mu, log_var, z.view(-1, self.categorical_dim)]

    def decode(self, z: Tensor) -> Tensor:
        self.attach_grads()
        B = z.shape[0]
        pad = self.decoder_input.in_padded - z.shape[1]
        if pad:
            z = torch.cat([z, torch.zeros(B, pad, dtype=z.dtype, device=z.device)], dim=1)
        h = K.ConvAct.apply(z.reshape(B, 1, 1, -1), self.decoder_input.weight, self.decoder_input.bias, None, self._dec_in_spec)
        h = K._ToNHWC.apply(h.view(B, 512, 2, 2))
        return K.to_nchw_view(self.final_layer(self.decoder(h)))

    def reparameterize(self, mu: Tensor, log_var: Tensor, q: Tensor, eps: float = 1e-7, e: Tensor = None, u: Tensor = None) -> Tensor:
        """[z ; s]: Gaussian sample and Gumbel-softmax sample side by side (joint_vae.py:141-165)."""
        if e is None:
            e = torch.randn(mu.shape, dtype=mu.dtype, device=mu.device)
        if u is None:
            u = torch.rand(q.shape, dtype=q.dtype, device=q.device)
        z = K.Reparameterize.apply(mu, log_var, e.to(mu.device))
        s = K.GumbelSoftmax.apply(q, u.to(q.device), self.temp, eps).view(-1, self.categorical_dim)
        return torch.cat([z, s], dim=1)

    def forward(self, input: Tensor, eps: Tensor = None, u: Tensor = None, **kwargs) -> List[Tensor]:
        mu, log_var, q = self.encode(input)
        z = self.reparameterize(mu, log_var, q, e=eps, u=u)
        return [self.decode(z), input, q, mu, log_var]

    def loss_function(self, *args, **kwargs) -> dict:
        recons, input, q, mu, log_var = args[0], args[1], args[2], args[3], args[4]
        kld_weight = kwargs['M_N']
        batch_idx = kwargs['batch_idx']
        if batch_idx % self.anneal_interval == 0 and self.training:
            self.temp = np.maximum(self.temp * np.exp(- self.anneal_rate * batch_idx), self.min_temp)
        disc_curr = (self.disc_max - self.disc_min) * self.num_iter / float(self.disc_iter) + self.disc_min
        disc_curr = min(disc_curr, np.log(self.categorical_dim))
        cont_curr = (self.cont_max - self.cont_min) * self.num_iter / float(self.cont_iter) + self.cont_min
        cont_curr = min(cont_curr, self.cont_max)
        r, x = K.to_nhwc(recons), K.to_nhwc(input)
        mse = K.VAELoss.apply(r, x, mu, log_var, None, 0.0)
        kl = K.VAELoss.apply(r.detach(), x, mu, log_var, None, 1.0)
        kld_cont = kl[0] - kl[1]                         # (mse + kld) - mse: gradient only through the KL term
        kld_disc = K.CatKL.apply(q.view(q.shape[0], 1, -1), 1e-7)
        capacity_loss = self.disc_gamma * torch.abs(disc_curr - kld_disc) + self.cont_gamma * torch.abs(cont_curr - kld_cont)
        loss = self.alpha * mse[0] + kld_weight * capacity_loss
        if self.training:
            self.num_iter += 1
        return {'loss': loss, 'Reconstruction_Loss': mse[0], 'Capacity_Loss': capacity_loss}

    def sample(self, num_samples: int, current_device: int, **kwargs) -> Tensor:
        """Gaussian prior draws next to one-hot draws of the categorical prior (joint_vae.py:239-268)."""
        z = torch.randn(num_samples, self.latent_dim)
        np_y = np.zeros((num_samples, self.categorical_dim), dtype=np.float32)
        np_y[range(num_samples), np.random.choice(self.categorical_dim, num_samples)] = 1
        z = torch.cat([z, torch.from_numpy(np_y)], dim=1).to(current_device)
        return self.decode(z)

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        return self.forward(x)[0]
