"""BetaVAE (models/beta_vae.py:8-175) on the HIP path: the network IS VanillaVAE's (same five Conv-BN-LeakyReLU
blocks, heads, decoder and final_layer, same parameter names), only the objective differs (SURVEY.md §8f rank 4:
"free riders on the conv kernels").

    type 'H' (Higgins et al.):  loss = mse + beta * M_N * kld
    type 'B' (Burgess et al.):  loss = mse + gamma * M_N * |kld - C|,   C = clamp(C_max / C_stop_iter * num_iter, 0, C_max)

``num_iter`` counts loss_function calls (beta_vae.py:10,135).  The returned dict carries the un-detached ``mse`` and
``kld`` values like the reference (beta_vae.py:153).
"""
import torch

from .. import kernels as K
from .types_ import List, Tensor
from .vanilla_vae import VanillaVAE


class BetaVAE(VanillaVAE):
    num_iter = 0

    def __init__(self, in_channels: int, latent_dim: int, hidden_dims: List = None, beta: int = 4, gamma: float = 1000.,
                 max_capacity: int = 25, Capacity_max_iter: int = 1e5, loss_type: str = 'B', **kwargs) -> None:
        super().__init__(in_channels, latent_dim, hidden_dims, **kwargs)
        self.beta, self.gamma, self.loss_type = beta, gamma, loss_type
        self.C_max = float(max_capacity)
        self.C_stop_iter = Capacity_max_iter
        self.graph_safe = loss_type != 'B'      # 'B': C depends on the call counter -> the harness must not capture the step

    def loss_function(self, *args, **kwargs) -> dict:
        self.num_iter += 1
        recons, input, mu, log_var = args[0], args[1], args[2], args[3]
        kld_weight = kwargs['M_N']
        r, x = K.to_nhwc(recons), self._cached_nhwc(input)
        if self.loss_type == 'H':
            out = K.VAELoss.apply(r, x, mu, log_var, None, self.beta * kld_weight)
            return {'loss': out[0], 'Reconstruction_Loss': out[1], 'KLD': out[2]}
        if self.loss_type == 'B':
            # |kld - C| needs the sign of a device value: the loss kernel runs once for the reconstruction term (weight 0
            # on the KL) and once for the KL term alone (reconstruction detached), the scalar glue stays on the device
            mse = K.VAELoss.apply(r, x, mu, log_var, None, 0.0)
            kl = K.VAELoss.apply(r.detach(), x, mu, log_var, None, 1.0)
            kld = kl[0] - kl[1]                                   # (mse + kld) - mse, gradient only through kld
            C = min(max(self.C_max / self.C_stop_iter * self.num_iter, 0.0), self.C_max)
            loss = mse[0] + self.gamma * kld_weight * (kld - C).abs()
            return {'loss': loss, 'Reconstruction_Loss': mse[1], 'KLD': kl[2]}
        raise ValueError('Undefined loss type.')
