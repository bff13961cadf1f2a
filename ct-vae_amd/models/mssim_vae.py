"""MSSIMVAE on HIP kernels -- drop-in for the reference class (models/mssim_vae.py:9-180).

VanillaVAE's network and state_dict (the reference's ``mssim_loss`` module has no parameters); the reconstruction term is the
multi-scale SSIM loss of ``MSSIM`` (mssim_vae.py:182-279), computed by ``csrc/ssim.hip`` with the reference's window (11 taps,
exp(+x^2 / 4.5): the exponent's sign is the reference's) and its product rule (the last level's ssim power multiplies each of
the four contrast powers).  ``window_size`` other than 11 and ``size_average=False`` are not served (the YAML uses neither).
"""
from typing import List

from .. import kernels as K
from .types_ import Tensor
from .vanilla_vae import VanillaVAE


class MSSIMVAE(VanillaVAE):

    def __init__(self, in_channels: int, latent_dim: int, hidden_dims: List = None, window_size: int = 11,
                 size_average: bool = True, **kwargs) -> None:
        if window_size != 11 or not size_average:
            raise ValueError("MSSIMVAE on the HIP path serves window_size = 11 with size_average (configs/mssim_vae.yaml)")
        super().__init__(in_channels, latent_dim, hidden_dims, **kwargs)

    def loss_function(self, *args, **kwargs) -> dict:
        """MS-SSIM loss + M_N * KL (mssim_vae.py:130-152); 'KLD' carries the reference's flipped sign."""
        recons, input, mu, log_var = args[0], args[1], args[2], args[3]
        kld_weight = kwargs['M_N']
        recons_loss = K.MSSIMLoss.apply(K.to_nhwc(recons), self._cached_nhwc(input))
        kld = K.GaussKL.apply(mu, log_var)
        loss = recons_loss + kld_weight * kld
        return {'loss': loss, 'Reconstruction_Loss': recons_loss.detach(), 'KLD': -kld.detach()}
