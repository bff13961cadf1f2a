"""SWAE (models/swae.py:8-206, configs/swae.yaml) on the HIP path -- SURVEY.md §8f rank 4.

WAE_MMD's network (VanillaVAE's conv stacks, one deterministic head ``fc_z``) with the objective
mse + l1 + reg_weight / (B (B-1)) * SWD(z, prior): the sliced Wasserstein distance along ``num_projections`` random unit
directions.  The two reconstruction terms are one pass of the loss kernels (``kernels.L2L1``), the distance -- projections,
a sort of both sample sets per direction, the rank-wise p-th power and its gradient -- one launch of ``ctvae_swd_forward``
(csrc/swd.hip).  The prior draws and the directions can be injected through ``loss_function(..., prior_z=, proj=)``
(SURVEY N1); by default they are drawn on the device (the reference draws the directions on the host and moves them).
"""
import torch

from .. import kernels as K
from .types_ import List, Tensor
from .wae_mmd import WAE_MMD


class SWAE(WAE_MMD):

    def __init__(self, in_channels: int, latent_dim: int, hidden_dims: List = None, reg_weight: int = 100,
                 wasserstein_deg: float = 2., num_projections: int = 50, projection_dist: str = 'normal', **kwargs) -> None:
        super().__init__(in_channels, latent_dim, hidden_dims, reg_weight=reg_weight)
        self.p = wasserstein_deg
        self.num_projections = num_projections
        self.proj_dist = projection_dist

    def get_random_projections(self, latent_dim: int, num_samples: int, device=None) -> Tensor:
        """[S, D] directions on the unit sphere (swae.py:128-147)."""
        if self.proj_dist == 'normal':
            r = torch.randn(num_samples, latent_dim, device=device)
        elif self.proj_dist == 'cauchy':
            r = torch.distributions.Cauchy(torch.tensor([0.0]), torch.tensor([1.0])).sample((num_samples, latent_dim)).squeeze().to(device)
        else:
            raise ValueError('Unknown projection distribution.')
        return r / r.norm(dim=1).view(-1, 1)

    def compute_swd(self, z: Tensor, p: float, reg_weight: float, prior_z: Tensor = None, proj: Tensor = None) -> Tensor:
        if prior_z is None:
            prior_z = torch.randn(z.shape, dtype=z.dtype, device=z.device)
        if proj is None:
            proj = self.get_random_projections(self.latent_dim, self.num_projections, z.device)
        return K.SWD.apply(z, prior_z.to(z.device), proj.to(z.device), p, reg_weight)

    def loss_function(self, *args, **kwargs) -> dict:
        recons, input, z = args[0], args[1], args[2]
        batch_size = input.size(0)
        reg_weight = self.reg_weight / (batch_size * (batch_size - 1))
        swd_loss = self.compute_swd(z, self.p, reg_weight, kwargs.get('prior_z'), kwargs.get('proj'))
        out = K.VAELoss.apply(K.to_nhwc(recons), K.to_nhwc(input), None, None, swd_loss.reshape(1), 0.0, K.L2L1)
        return {'loss': out[0], 'Reconstruction_Loss': out[1], 'SWD': swd_loss}
