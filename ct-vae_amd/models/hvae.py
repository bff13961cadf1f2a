"""HVAE (models/hvae.py:8-259, configs/hvae.yaml) on the HIP path -- SURVEY.md §8f rank 4.

Two latent levels: z2 ~ q(z2|x) from one VanillaVAE-style encoder, z1 ~ q(z1|x, z2) from a second one whose input carries z2 as
an extra plane (``embed_z2_code`` -> [S,S] next to the 1x1 ``embed_data``, the same construction as ConditionalVAE's label
plane), a decoder over [debed(z1) ; debed(z2)], and the objective mse + M_N * (KL(q(z1)) + KL(q(z2)) - "KL"(z1 - mu_p(z2),
logvar_p(z2))) exactly as the reference writes it (hvae.py:206-229; its unused ``z2_p_kld`` is not formed).  All three terms
have the Gaussian-KL form and run through the KL kernels (``kernels.GaussKL``); the conv stacks and the final block are
VanillaVAE's.  ``forward`` takes optional ``eps`` (z1) and ``eps2`` (z2) -- injected noise, SURVEY N1.
"""
import torch
from torch import nn

from .. import kernels as K
from .base import BaseVAE
from .blocks import ConvBNLeaky
from .packing import PackedConv, PackedLinear, PackedLinearGroup
from .types_ import List, Tensor
from .vanilla_vae import _FinalLayer


def _heads(fin, fout):
    a, b = PackedLinear(fin, fout), PackedLinear(fin, fout)
    grp = PackedLinearGroup([a, b])
    a._linear_group = grp
    b._linear_group = grp
    return a, b


class HVAE(BaseVAE):

    def __init__(self, in_channels: int, latent1_dim: int, latent2_dim: int, hidden_dims: List = None, img_size: int = 64,
                 pseudo_input_size: int = 128, **kwargs) -> None:
        super().__init__()
        self.latent1_dim, self.latent2_dim, self.img_size = latent1_dim, latent2_dim, img_size
        if hidden_dims is None:
            hidden_dims = [32, 64, 128, 256, 512]
        if hidden_dims[-1] != 512:
            raise ValueError("HVAE.forward assumes hidden_dims[-1] == 512 (hvae.py:190)")
        feat = hidden_dims[-1] * 4

        def stack(c):
            mods = []
            for h in hidden_dims:
                mods.append(ConvBNLeaky(c, h, 3, 2, 1))
                c = h
            return nn.Sequential(*mods)

        self.encoder_z2_layers = stack(in_channels)
        self.fc_z2_mu, self.fc_z2_var = _heads(feat, latent2_dim)
        self._z2_spec = K.ConvSpec(K.CONV, feat, 2 * latent2_dim, 1)
        self.embed_z2_code = PackedLinear(latent2_dim, img_size * img_size, pad_in_to=32)
        self._embed_spec = K.ConvSpec(K.CONV, self.embed_z2_code.in_padded, img_size * img_size, 1)
        self.embed_data = PackedConv(in_channels, in_channels, 1, bias=True)
        self._data_spec = K.ConvSpec(K.CONV, in_channels, in_channels, 1, 1, 0, 0, K.ACT_NONE)
        self.encoder_z1_layers = stack(in_channels + 1)
        self.fc_z1_mu, self.fc_z1_var = _heads(feat, latent1_dim)
        self._z1_spec = K.ConvSpec(K.CONV, feat, 2 * latent1_dim, 1)
        self.recons_z1_mu, self.recons_z1_log_var = _heads(latent2_dim, latent1_dim)
        self._rz1_spec = K.ConvSpec(K.CONV, latent2_dim, 2 * latent1_dim, 1)
        self.debed_z1_code = PackedLinear(latent1_dim, 1024, pad_in_to=32)
        self.debed_z2_code = PackedLinear(latent2_dim, 1024, pad_in_to=32)
        self._d1_spec = K.ConvSpec(K.CONV, self.debed_z1_code.in_padded, 1024, 1)
        self._d2_spec = K.ConvSpec(K.CONV, self.debed_z2_code.in_padded, 1024, 1)
        hidden_dims.reverse()
        dec = []
        for i in range(len(hidden_dims) - 1):
            dec.append(ConvBNLeaky(hidden_dims[i], hidden_dims[i + 1], 3, 2, 1, out_pad=1, transposed=True))
        self.decoder = nn.Sequential(*dec)
        self.final_layer = _FinalLayer(hidden_dims[-1], 3)
        self.flatten_parameters()

    # -- pieces -----------------------------------------------------------------------------------------------------
    @staticmethod
    def _pad_cols(t, width):
        pad = width - t.shape[1]
        return t if pad == 0 else torch.cat([t, torch.zeros(t.shape[0], pad, dtype=t.dtype, device=t.device)], dim=1)

    def _linear(self, x, lin, spec):
        B = x.shape[0]
        return K.ConvAct.apply(self._pad_cols(x, spec.ci).reshape(B, 1, 1, -1), lin.weight, lin.bias, None, spec).view(B, -1)

    def _stack_heads(self, layers, x_nhwc, first_head, spec, L):
        h = layers(x_nhwc)
        heads = K.flatten_linear(h, first_head.weight, first_head.bias, spec.co)
        return K.SplitHeads.apply(heads, L)

    def reparameterize(self, mu: Tensor, logvar: Tensor, eps: Tensor = None) -> Tensor:
        if eps is None:
            eps = torch.randn(mu.shape, dtype=mu.dtype, device=mu.device)
        return K.Reparameterize.apply(mu, logvar, eps.to(mu.device))

    def encode_z2(self, input: Tensor) -> List[Tensor]:
        self.attach_grads()
        return list(self._stack_heads(self.encoder_z2_layers, K.to_nhwc(input), self.fc_z2_mu, self._z2_spec, self.latent2_dim))

    def encode_z1(self, input: Tensor, z2: Tensor) -> List[Tensor]:
        self.attach_grads()
        B, S = input.shape[0], self.img_size
        plane = self._linear(z2, self.embed_z2_code, self._embed_spec).view(B, S, S, 1)
        data = K.ConvAct.apply(K.to_nhwc(input), self.embed_data.weight, self.embed_data.bias, None, self._data_spec)
        x = torch.cat([data, plane], dim=-1)
        return list(self._stack_heads(self.encoder_z1_layers, x, self.fc_z1_mu, self._z1_spec, self.latent1_dim))

    def encode(self, input: Tensor, eps2: Tensor = None) -> List[Tensor]:
        z2_mu, z2_log_var = self.encode_z2(input)
        z2 = self.reparameterize(z2_mu, z2_log_var, eps2)
        z1_mu, z1_log_var = self.encode_z1(input, z2)
        return [z1_mu, z1_log_var, z2_mu, z2_log_var, z2]

    def decode(self, input: Tensor) -> Tensor:
        """input: [B, 512, 2, 2] (logical NCHW), as in the reference (hvae.py:157-160)."""
        self.attach_grads()
        return K.to_nchw_view(self.final_layer(self.decoder(K.to_nhwc(input))))

    def _decode_codes(self, z1, z2):
        B = z1.shape[0]
        d = torch.cat([self._linear(z1, self.debed_z1_code, self._d1_spec), self._linear(z2, self.debed_z2_code, self._d2_spec)], dim=1)
        h = K._ToNHWC.apply(d.view(B, 512, 2, 2))
        return K.to_nchw_view(self.final_layer(self.decoder(h)))

    def forward(self, input: Tensor, eps: Tensor = None, eps2: Tensor = None, **kwargs) -> List[Tensor]:
        z1_mu, z1_log_var, z2_mu, z2_log_var, z2 = self.encode(input, eps2)
        z1 = self.reparameterize(z1_mu, z1_log_var, eps)
        return [self._decode_codes(z1, z2), input, z1_mu, z1_log_var, z2_mu, z2_log_var, z1, z2]

    def loss_function(self, *args, **kwargs) -> dict:
        recons, input, z1_mu, z1_log_var, z2_mu, z2_log_var, z1, z2 = args[:8]
        p_mu, p_log_var = K.SplitHeads.apply(self._linear(z2, self.recons_z1_mu, self._rz1_spec), self.latent1_dim)
        kld_weight = kwargs['M_N']
        z1_kld = K.GaussKL.apply(z1_mu, z1_log_var)
        z2_kld = K.GaussKL.apply(z2_mu, z2_log_var)
        z1_p_kld = K.GaussKL.apply(z1 - p_mu, p_log_var)
        kld_loss = -(z1_p_kld - z1_kld - z2_kld)
        out = K.VAELoss.apply(K.to_nhwc(recons), K.to_nhwc(input), None, None, (kld_weight * kld_loss).reshape(1), 0.0)
        return {'loss': out[0], 'Reconstruction Loss': out[1], 'KLD': -kld_loss}

    def sample(self, batch_size: int, current_device: int, **kwargs) -> Tensor:
        z2 = torch.randn(batch_size, self.latent2_dim).to(current_device)
        p_mu, p_log_var = K.SplitHeads.apply(self._linear(z2, self.recons_z1_mu, self._rz1_spec), self.latent1_dim)
        return self._decode_codes(self.reparameterize(p_mu, p_log_var), z2)

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        return self.forward(x)[0]
