"""TwoStageVAE (models/twostage_vae.py:8-196) on the HIP path -- SURVEY.md §8f rank 4.

In the reference the class builds a second, fully connected VAE on the latent codes (``encoder2`` / ``fc_mu2`` / ``fc_var2`` /
``decoder2``: Linear + BatchNorm1d + LeakyReLU stacks, twostage_vae.py:76-101) but its ``forward`` and ``loss_function`` use the
first stage only (:137-165): the training step IS VanillaVAE's.  The second-stage modules are kept as parameter holders under
the reference's ``state_dict`` keys (checkpoints interchange); they receive no gradient, as in the reference.
"""
from torch import nn

from .types_ import List
from .vanilla_vae import VanillaVAE


def _mlp(dims_in, dims_out):
    return nn.Sequential(*[nn.Sequential(nn.Linear(i, o), nn.BatchNorm1d(o), nn.LeakyReLU()) for i, o in zip(dims_in, dims_out)])


class TwoStageVAE(VanillaVAE):

    def __init__(self, in_channels: int, latent_dim: int, hidden_dims: List = None, hidden_dims2: List = None, **kwargs) -> None:
        super().__init__(in_channels, latent_dim, hidden_dims, **kwargs)
        if hidden_dims2 is None:
            hidden_dims2 = [1024, 1024]
        self.encoder2 = _mlp([latent_dim] + hidden_dims2[:-1], hidden_dims2)
        self.fc_mu2 = nn.Linear(hidden_dims2[-1], latent_dim)
        self.fc_var2 = nn.Linear(hidden_dims2[-1], latent_dim)
        hidden_dims2.reverse()                 # the reference mutates the caller's list too (twostage_vae.py:91)
        self.decoder2 = _mlp([latent_dim] + hidden_dims2[:-1], hidden_dims2)
        self.flatten_parameters()
