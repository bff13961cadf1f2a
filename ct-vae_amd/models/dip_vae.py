"""DIPVAE (models/dip_vae.py:8-193, configs/dip_vae.yaml) on the HIP path — SURVEY.md §8f rank 4.

VanillaVAE's network and kernels.  The objective uses SUMS (``F.mse_loss(reduction='sum')``, KL summed over the batch) plus
the DIP-II covariance regulariser on the posterior means (``ctvae_dip_forward/backward``, csrc/dip.hip — the reference's
quirks included: centring over the latent dimension, one scalar variance term).
"""
from .. import kernels as K
from .types_ import List
from .vanilla_vae import VanillaVAE


class DIPVAE(VanillaVAE):

    def __init__(self, in_channels: int, latent_dim: int, hidden_dims: List = None, lambda_diag: float = 10.,
                 lambda_offdiag: float = 5., **kwargs) -> None:
        super().__init__(in_channels, latent_dim, hidden_dims, **kwargs)
        self.lambda_diag = lambda_diag
        self.lambda_offdiag = lambda_offdiag

    def loss_function(self, *args, **kwargs) -> dict:
        recons, input, mu, log_var = args[0], args[1], args[2], args[3]
        kld_weight = kwargs['M_N']
        n, B = recons.numel(), mu.shape[0]
        # the loss kernels give means: sum-reduced mse = n * mse, batch-summed KL = B * kld  ->  n * (mse + (w B / n) kld)
        out = K.VAELoss.apply(K.to_nhwc(recons), self._cached_nhwc(input), mu, log_var, None, kld_weight * B / n)
        dip_loss = K.DIPLoss.apply(mu, log_var, self.lambda_diag, self.lambda_offdiag)
        return {'loss': n * out[0] + dip_loss, 'Reconstruction_Loss': n * out[1], 'KLD': B * out[3], 'DIP_Loss': dip_loss}
