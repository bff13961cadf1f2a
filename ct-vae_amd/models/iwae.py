"""IWAE (models/iwae.py:8-190, configs/iwae.yaml) and MIWAE (models/miwae.py:8-200, configs/miwae.yaml) on the HIP path —
SURVEY.md §8f rank 4.

Both are VanillaVAE's network (same kernels, same parameter names); the decoder runs on S (IWAE) resp. M x S (MIWAE)
latent samples per image and the objective is the importance-weighted bound: per-sample reconstruction error and KL
-> log-weights -> softmax over the S samples -> mean over groups.  The per-sample reductions, the softmax and the
backward pass through weights AND log-weights are ``ctvae_iw_loss_forward/backward`` (csrc/iwloss.hip).

Deliberate, documented differences: ``forward`` takes an optional ``eps`` (injected N(0,1) noise); the
repeated ``mu`` / ``log_var`` it returns are broadcast views rather than copies (same values and shape).
"""
import torch

from .. import kernels as K
from .types_ import List, Tensor
from .vanilla_vae import VanillaVAE


class MIWAE(VanillaVAE):

    def __init__(self, in_channels: int, latent_dim: int, hidden_dims: List = None, num_samples: int = 5,
                 num_estimates: int = 5, **kwargs) -> None:
        super().__init__(in_channels, latent_dim, hidden_dims, **kwargs)
        self.num_samples = num_samples        # K
        self.num_estimates = num_estimates    # M

    def _lead(self, B):
        return (B, self.num_estimates, self.num_samples)

    def decode(self, z: Tensor) -> Tensor:
        """[B x M x S x D] -> [B x M x S x C x H x W] (miwae.py:98-112)."""
        lead = tuple(z.shape[:-1])
        r = super().decode(z.reshape(-1, self.latent_dim))
        return r.view(lead + tuple(r.shape[1:]))

    def forward(self, input: Tensor, eps: Tensor = None, **kwargs) -> List[Tensor]:
        mu, log_var = self.encode(input)
        B, L = mu.shape
        lead = self._lead(B)
        ones = (1,) * (len(lead) - 1)
        mu = mu.view((B,) + ones + (L,)).expand(lead + (L,))
        log_var = log_var.view((B,) + ones + (L,)).expand(lead + (L,))
        if eps is not None:
            eps = eps.reshape(-1, L)
        z = self.reparameterize(mu.reshape(-1, L), log_var.reshape(-1, L), eps).view(lead + (L,))
        eps_out = (z - mu) / log_var          # "prior samples" as the reference computes them (iwae.py:124)
        return [self.decode(z), input, mu, log_var, z, eps_out]

    def loss_function(self, *args, **kwargs) -> dict:
        recons, input, mu, log_var = args[0], args[1], args[2], args[3]
        L = mu.shape[-1]
        r = K.to_nhwc(recons.reshape((-1,) + tuple(recons.shape[-3:])))
        out = K.IWLoss.apply(r, self._cached_nhwc(input), mu.reshape(-1, L), log_var.reshape(-1, L), self.num_samples,
                             kwargs['M_N'])
        return {'loss': out[0], 'Reconstruction_Loss': out[1], 'KLD': out[3]}

    def sample(self, num_samples: int, current_device: int, **kwargs) -> Tensor:
        z = torch.randn((num_samples,) + (1,) * (len(self._lead(1)) - 1) + (self.latent_dim,)).to(current_device)
        return self.decode(z).squeeze()

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        """Only the first reconstructed sample (miwae.py:191-200)."""
        r = self.forward(x)[0]
        return r[(slice(None),) + (0,) * (r.dim() - 4)]


class IWAE(MIWAE):
    """One estimate: tensors are [B x S x ...] (iwae.py)."""

    def __init__(self, in_channels: int, latent_dim: int, hidden_dims: List = None, num_samples: int = 5, **kwargs) -> None:
        kwargs.pop("num_estimates", None)
        super().__init__(in_channels, latent_dim, hidden_dims, num_samples=num_samples, num_estimates=1, **kwargs)

    def _lead(self, B):
        return (B, self.num_samples)
