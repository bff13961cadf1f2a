"""WAE_MMD (models/wae_mmd.py:8-235, configs/wae_mmd_{imq,rbf}.yaml) and InfoVAE (models/info_vae.py:8-260,
configs/infovae.yaml) on the HIP path — SURVEY.md §8f rank 4.

Both use VanillaVAE's conv stacks and kernels.  WAE_MMD has ONE head ``fc_z`` (deterministic encoder) and the objective
mse + MMD(z, prior); InfoVAE is VanillaVAE's network with beta*mse + (1-alpha)*M_N*KL + (alpha+reg_weight-1)/(B(B-1))*MMD.
The MMD term (three pairwise kernel sums, which the reference evaluates on [N,N,D] tensors) is one pass of
``ctvae_mmd_forward`` (csrc/mmd.hip) that also leaves d mmd / d z.  ``prior`` (the N(0,1) draws of ``compute_mmd``) can be
injected through ``loss_function(..., prior_z=...)``.
"""
import torch
from torch import nn

from .. import kernels as K
from .base import BaseVAE
from .blocks import ConvBNLeaky
from .packing import PackedLinear
from .types_ import List, Tensor
from .vanilla_vae import VanillaVAE, _FinalLayer


def _mmd(z, kernel_type, z_var, w_pp, w_zz, w_pz, prior_z=None):
    if prior_z is None:
        prior_z = torch.randn(z.shape, dtype=z.dtype, device=z.device)
    return K.MMD.apply(z, prior_z.to(z.device), kernel_type, 2.0 * z.shape[1] * z_var, w_pp, w_zz, w_pz)


class WAE_MMD(BaseVAE):

    def __init__(self, in_channels: int, latent_dim: int, hidden_dims: List = None, reg_weight: int = 100,
                 kernel_type: str = 'imq', latent_var: float = 2., **kwargs) -> None:
        super().__init__()
        self.latent_dim = latent_dim
        self.reg_weight = reg_weight
        self.kernel_type = kernel_type
        self.z_var = latent_var
        if hidden_dims is None:
            hidden_dims = [32, 64, 128, 256, 512]
        if hidden_dims[-1] != 512:
            raise ValueError("WAE_MMD.decode assumes hidden_dims[-1] == 512 (wae_mmd.py:97)")
        enc, c = [], in_channels
        for h in hidden_dims:
            enc.append(ConvBNLeaky(c, h, 3, 2, 1))
            c = h
        self.encoder = nn.Sequential(*enc)
        self.fc_z = PackedLinear(hidden_dims[-1] * 4, latent_dim)
        self._head_spec = K.ConvSpec(K.CONV, hidden_dims[-1] * 4, latent_dim, 1)
        self.decoder_input = PackedLinear(latent_dim, hidden_dims[-1] * 4)
        self._dec_in_spec = K.ConvSpec(K.CONV, latent_dim, hidden_dims[-1] * 4, 1)
        hidden_dims.reverse()                  # the reference mutates the caller's list too (wae_mmd.py:49)
        dec = []
        for i in range(len(hidden_dims) - 1):
            dec.append(ConvBNLeaky(hidden_dims[i], hidden_dims[i + 1], 3, 2, 1, out_pad=1, transposed=True))
        self.decoder = nn.Sequential(*dec)
        self.final_layer = _FinalLayer(hidden_dims[-1], 3)
        self.flatten_parameters()

    def encode(self, input: Tensor) -> Tensor:
        """[B,C,64,64] -> z [B, latent_dim] (a Tensor, not a list: wae_mmd.py:81-94)."""
        self.attach_grads()
        h = self.encoder(K.to_nhwc(input))
        return K.flatten_linear(h, self.fc_z.weight, self.fc_z.bias, self._head_spec.co)

    def decode(self, z: Tensor) -> Tensor:
        self.attach_grads()
        B = z.shape[0]
        h = K.ConvAct.apply(z.reshape(B, 1, 1, -1), self.decoder_input.weight, self.decoder_input.bias, None, self._dec_in_spec)
        h = K._ToNHWC.apply(h.view(B, 512, 2, 2))
        return K.to_nchw_view(self.final_layer(self.decoder(h)))

    def forward(self, input: Tensor, **kwargs) -> List[Tensor]:
        z = self.encode(input)
        return [self.decode(z), input, z]

    def compute_mmd(self, z: Tensor, reg_weight: float, prior_z: Tensor = None) -> Tensor:
        return _mmd(z, self.kernel_type, self.z_var, reg_weight, reg_weight, reg_weight, prior_z)[0]

    def loss_function(self, *args, **kwargs) -> dict:
        recons, input, z = args[0], args[1], args[2]
        batch_size = input.size(0)
        reg_weight = self.reg_weight / (batch_size * (batch_size - 1))
        mmd_loss = self.compute_mmd(z, reg_weight, kwargs.get('prior_z'))
        out = K.VAELoss.apply(K.to_nhwc(recons), K.to_nhwc(input), None, None, mmd_loss.reshape(1), 0.0)   # mse + mmd
        return {'loss': out[0], 'Reconstruction_Loss': out[1], 'MMD': mmd_loss}

    def sample(self, num_samples: int, current_device: int, **kwargs) -> Tensor:
        return self.decode(torch.randn(num_samples, self.latent_dim).to(current_device))

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        return self.forward(x)[0]


class InfoVAE(VanillaVAE):

    def __init__(self, in_channels: int, latent_dim: int, hidden_dims: List = None, alpha: float = -0.5, beta: float = 5.0,
                 reg_weight: int = 100, kernel_type: str = 'imq', latent_var: float = 2., **kwargs) -> None:
        super().__init__(in_channels, latent_dim, hidden_dims, **kwargs)
        assert alpha <= 0, 'alpha must be negative or zero.'
        self.reg_weight, self.kernel_type, self.z_var = reg_weight, kernel_type, latent_var
        self.alpha, self.beta = alpha, beta

    def forward(self, input: Tensor, eps: Tensor = None, **kwargs) -> List[Tensor]:
        mu, log_var = self.encode(input)
        z = self.reparameterize(mu, log_var, eps)
        return [self.decode(z), input, z, mu, log_var]

    def compute_mmd(self, z: Tensor, prior_z: Tensor = None) -> Tensor:
        return _mmd(z, self.kernel_type, self.z_var, 1.0, 1.0, 1.0, prior_z)[0]

    def loss_function(self, *args, **kwargs) -> dict:
        recons, input, z, mu, log_var = args[0], args[1], args[2], args[3], args[4]
        batch_size = input.size(0)
        bias_corr = batch_size * (batch_size - 1)
        kld_weight = kwargs['M_N']
        mmd_loss = self.compute_mmd(z, kwargs.get('prior_z'))
        # the loss kernels give mse + w*kld; the weights beta / (1-alpha)*M_N and the MMD term are scalar glue on the device
        out = K.VAELoss.apply(K.to_nhwc(recons), self._cached_nhwc(input), mu, log_var, None, (1. - self.alpha) * kld_weight / self.beta)
        loss = self.beta * out[0] + (self.alpha + self.reg_weight - 1.) / bias_corr * mmd_loss
        return {'loss': loss, 'Reconstruction_Loss': out[1], 'MMD': mmd_loss, 'KLD': out[3]}
