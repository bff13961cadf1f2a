"""BetaTCVAE (models/betatc_vae.py:8-237, configs/betatc_vae.yaml) on the HIP path -- SURVEY.md §8f rank 4.

Its own small network -- four Conv2d(k4, s2) + LeakyReLU without BatchNorm, ``fc`` (512 -> 256), the two heads, a decoder of
ConvTranspose2d + LeakyReLU blocks from a [32, 4, 4] seed -- on the general conv kernels (bias + activation in the epilogue),
and the total-correlation decomposition of the KL term: mutual information, total correlation and dimension-wise KL from the
[B, B, D] matrix of log densities with minibatch-stratified importance weights.  That matrix never exists here
(``kernels.TCDecomp``, csrc/tcvae.hip).  The objective is mse_sum / B + alpha * mi + beta * tc + anneal * gamma * kld with the
reference's per-call anneal counter (host state: ``graph_safe = False``).  ``forward`` takes an optional ``eps``.
"""
import torch
from torch import nn

from .. import kernels as K
from .base import BaseVAE
from .blocks import ConvAct
from .packing import PackedConv, PackedLinear, PackedLinearGroup
from .types_ import List, Tensor


class _TCFinal(nn.Module):
    """nn.Sequential(ConvTranspose2d, LeakyReLU, Conv2d(->3), Tanh) (betatc_vae.py:73-82): parameters under "0" and "2"."""

    def __init__(self, c):
        super().__init__()
        self.add_module("0", PackedConv(c, c, 3, transposed=True, bias=True))
        self.add_module("2", PackedConv(c, 3, 3, bias=True))
        self.spec_up = K.ConvSpec(K.CONVT, c, c, 3, 2, 1, 1, K.ACT_LRELU)
        self.spec_out = K.ConvSpec(K.CONV, c, 3, 3, 1, 1, 0, K.ACT_TANH)

    def forward(self, x):
        up, conv = self._modules["0"], self._modules["2"]
        h = K.ConvAct.apply(x, up.weight, up.bias, None, self.spec_up)
        return K.ConvAct.apply(h, conv.weight, conv.bias, None, self.spec_out)


class BetaTCVAE(BaseVAE):
    num_iter = 0

    def __init__(self, in_channels: int, latent_dim: int, hidden_dims: List = None, anneal_steps: int = 200, alpha: float = 1.,
                 beta: float = 6., gamma: float = 1., **kwargs) -> None:
        super().__init__()
        self.latent_dim, self.anneal_steps = latent_dim, anneal_steps
        self.alpha, self.beta, self.gamma = alpha, beta, gamma
        self.graph_safe = False                    # the anneal rate follows the call counter
        if latent_dim > 32:
            raise ValueError("BetaTCVAE on the HIP path: latent_dim <= 32 (csrc/tcvae.hip keeps D + 1 running logsumexps in registers)")
        if hidden_dims is None:
            hidden_dims = [32, 32, 32, 32]
        if hidden_dims[-1] != 32 or len(hidden_dims) != 4:
            raise ValueError("BetaTCVAE.decode assumes a [32, 4, 4] seed (betatc_vae.py:110)")
        enc, c = [], in_channels
        for h in hidden_dims:
            enc.append(ConvAct(c, h, 4, 2, 1, K.ACT_LRELU))
            c = h
        self.encoder = nn.Sequential(*enc)
        self.fc = PackedLinear(hidden_dims[-1] * 16, 256)
        self._fc_spec = K.ConvSpec(K.CONV, hidden_dims[-1] * 16, 256, 1)
        self.fc_mu = PackedLinear(256, latent_dim)
        self.fc_var = PackedLinear(256, latent_dim)
        grp = PackedLinearGroup([self.fc_mu, self.fc_var])
        self.fc_mu._linear_group = grp
        self.fc_var._linear_group = grp
        self._head_spec = K.ConvSpec(K.CONV, 256, 2 * latent_dim, 1)
        self.decoder_input = PackedLinear(latent_dim, 256 * 2, pad_in_to=32)
        self._dec_in_spec = K.ConvSpec(K.CONV, self.decoder_input.in_padded, 512, 1)
        hidden_dims.reverse()
        dec = []
        for i in range(len(hidden_dims) - 1):
            dec.append(ConvAct(hidden_dims[i], hidden_dims[i + 1], 3, 2, 1, K.ACT_LRELU, transposed=True, out_pad=1))
        self.decoder = nn.Sequential(*dec)
        self.final_layer = _TCFinal(hidden_dims[-1])
        self._liw = {}
        self.flatten_parameters()

    def encode(self, input: Tensor) -> List[Tensor]:
        self.attach_grads()
        h = self.encoder(K.to_nhwc(input))
        B = h.shape[0]
        flat = K._ToNCHW.apply(h).view(B, 1, 1, -1)
        f = K.ConvAct.apply(flat, self.fc.weight, self.fc.bias, None, self._fc_spec)
        heads = K.ConvAct.apply(f, self.fc_mu.weight, self.fc_mu.bias, None, self._head_spec).view(B, -1)
        mu, log_var = K.SplitHeads.apply(heads, self.latent_dim)
        return [mu, log_var]

    def decode(self, z: Tensor) -> Tensor:
        self.attach_grads()
        B = z.shape[0]
        pad = self.decoder_input.in_padded - z.shape[1]
        if pad:
            z = torch.cat([z, torch.zeros(B, pad, dtype=z.dtype, device=z.device)], dim=1)
        h = K.ConvAct.apply(z.reshape(B, 1, 1, -1), self.decoder_input.weight, self.decoder_input.bias, None, self._dec_in_spec)
        h = K._ToNHWC.apply(h.view(B, 32, 4, 4))
        return K.to_nchw_view(self.final_layer(self.decoder(h)))

    def reparameterize(self, mu: Tensor, logvar: Tensor, eps: Tensor = None) -> Tensor:
        if eps is None:
            eps = torch.randn(mu.shape, dtype=mu.dtype, device=mu.device)
        return K.Reparameterize.apply(mu, logvar, eps.to(mu.device))

    def forward(self, input: Tensor, eps: Tensor = None, **kwargs) -> List[Tensor]:
        mu, log_var = self.encode(input)
        z = self.reparameterize(mu, log_var, eps)
        return [self.decode(z), input, mu, log_var, z]

    def _log_importance_weights(self, batch_size, dataset_size, device):
        """betatc_vae.py:176-183, as written there (incl. which entries the strided assignments touch)."""
        key = (batch_size, float(dataset_size), device)
        w = self._liw.get(key)
        if w is None:
            strat_weight = (dataset_size - batch_size + 1) / (dataset_size * (batch_size - 1))
            iw = torch.full((batch_size, batch_size), 1 / (batch_size - 1), dtype=torch.float32)
            iw.view(-1)[::batch_size] = 1 / dataset_size
            iw.view(-1)[1::batch_size] = strat_weight
            iw[batch_size - 2, 0] = strat_weight
            w = self._liw[key] = iw.log().to(device)
        return w

    def loss_function(self, *args, **kwargs) -> dict:
        recons, input, mu, log_var, z = args[0], args[1], args[2], args[3], args[4]
        weight = 1  # kwargs['M_N'] in the paper; the reference fixes it (betatc_vae.py:164)
        batch_size = z.shape[0]
        dataset_size = (1 / kwargs['M_N']) * batch_size
        out = K.VAELoss.apply(K.to_nhwc(recons), K.to_nhwc(input), None, None, None, 0.0)
        recons_loss = out[0] * float(recons.numel())                 # F.mse_loss(reduction='sum')
        mi_loss, tc_loss, kld_loss = K.TCDecomp.apply(z, mu, log_var, self._log_importance_weights(batch_size, dataset_size, z.device))
        if self.training:
            self.num_iter += 1
            anneal_rate = min(0 + 1 * self.num_iter / self.anneal_steps, 1)
        else:
            anneal_rate = 1.
        loss = recons_loss / batch_size + self.alpha * mi_loss + weight * (self.beta * tc_loss + anneal_rate * self.gamma * kld_loss)
        return {'loss': loss, 'Reconstruction_Loss': recons_loss, 'KLD': kld_loss, 'TC_Loss': tc_loss, 'MI_Loss': mi_loss}

    def sample(self, num_samples: int, current_device: int, **kwargs) -> Tensor:
        return self.decode(torch.randn(num_samples, self.latent_dim).to(current_device))

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        return self.forward(x)[0]
