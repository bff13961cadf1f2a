"""LogCoshVAE (models/logcosh_vae.py:8-185, configs/logcosh_vae.yaml) on the HIP path — SURVEY.md §8f rank 4.

VanillaVAE's network and kernels; the reconstruction term is 1/alpha * mean(log cosh(alpha (recons - input))), written
as the reference writes it (alpha t + log(1 + exp(-2 alpha t)) - log 2), and the KL term is weighted by beta * M_N.
One launch pair forward (``ctvae_logcosh_loss_forward``), tanh(alpha t)/n backward (``ctvae_logcosh_backward``).
"""
from .. import kernels as K
from .types_ import List
from .vanilla_vae import VanillaVAE


class LogCoshVAE(VanillaVAE):

    def __init__(self, in_channels: int, latent_dim: int, hidden_dims: List = None, alpha: float = 100., beta: float = 10.,
                 **kwargs) -> None:
        super().__init__(in_channels, latent_dim, hidden_dims, **kwargs)
        self.alpha = alpha
        self.beta = beta

    def loss_function(self, *args, **kwargs) -> dict:
        recons, input, mu, log_var = args[0], args[1], args[2], args[3]
        out = K.VAELoss.apply(K.to_nhwc(recons), self._cached_nhwc(input), mu, log_var, None, self.beta * kwargs['M_N'],
                              float(self.alpha))
        return {'loss': out[0], 'Reconstruction_Loss': out[1], 'KLD': out[3]}
