"""VampVAE (models/vampvae.py:8-194, configs/vampvae.yaml) on the HIP path -- SURVEY.md §8f rank 4.

VanillaVAE's network with the VampPrior: K learned pseudo-inputs (``embed_pseudo``: Linear(K -> 3*64*64) + Hardtanh(0, 1) on the
identity matrix) pass through the SAME encoder inside ``loss_function`` -- in training mode that is a second BatchNorm pass per
step, on the K pseudo-images' own batch statistics, exactly as in the reference (:156-160) -- and the KL term is
-(E log p_vamp(z) - E log q(z|x)) with the mixture prior over their posteriors.  The reference evaluates the mixture on a
[B, K, D] tensor; here it is ``kernels.VampKL`` (csrc/vamp.hip).  ``forward`` returns z as fifth element like the reference.
"""
import torch
from torch import nn

from .. import kernels as K
from .packing import PackedLinear
from .types_ import List, Tensor
from .vanilla_vae import VanillaVAE


class VampVAE(VanillaVAE):

    def __init__(self, in_channels: int, latent_dim: int, hidden_dims: List = None, num_components: int = 50, **kwargs) -> None:
        super().__init__(in_channels, latent_dim, hidden_dims, **kwargs)
        self.num_components = num_components
        self.embed_pseudo = nn.Sequential()
        self.embed_pseudo.add_module("0", PackedLinear(num_components, 12288, pad_in_to=32))      # "1" is the Hardtanh (no parameters)
        lin = self.embed_pseudo._modules["0"]
        self._pseudo_spec = K.ConvSpec(K.CONV, lin.in_padded, 12288, 1)
        eye = torch.zeros(num_components, lin.in_padded)
        eye[:, :num_components] = torch.eye(num_components)
        self.register_buffer("_pseudo_eye", eye, persistent=False)           # the reference keeps it as a plain attribute
        self.flatten_parameters()

    @property
    def pseudo_input(self):
        return self._pseudo_eye[:, :self.num_components]

    def forward(self, input: Tensor, eps: Tensor = None, **kwargs) -> List[Tensor]:
        mu, log_var = self.encode(input)
        z = self.reparameterize(mu, log_var, eps)
        return [self.decode(z), input, mu, log_var, z]

    def pseudo_images(self) -> Tensor:
        """[K, C, H, W]: the pseudo-inputs the prior is built from (vampvae.py:152-154)."""
        lin = self.embed_pseudo._modules["0"]
        Kc = self.num_components
        h = K.ConvAct.apply(self._pseudo_eye.reshape(Kc, 1, 1, -1), lin.weight, lin.bias, None, self._pseudo_spec).view(Kc, -1)
        return torch.clamp(h, 0.0, 1.0).view(Kc, self.in_channels, 64, 64)

    def loss_function(self, *args, **kwargs) -> dict:
        recons, input, mu, log_var, z = args[0], args[1], args[2], args[3], args[4]
        kld_weight = kwargs['M_N']
        prior_mu, prior_log_var = self.encode(self.pseudo_images())
        kld_loss = K.VampKL.apply(z, mu, log_var, prior_mu, prior_log_var)
        out = K.VAELoss.apply(K.to_nhwc(recons), K.to_nhwc(input), None, None, (kld_weight * kld_loss).reshape(1), 0.0)
        return {'loss': out[0], 'Reconstruction_Loss': out[1], 'KLD': -kld_loss}

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        return self.forward(x)[0]
