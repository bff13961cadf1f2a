"""VanillaVAE on HIP kernels — drop-in for the reference class (models/vanilla_vae.py:8-173).

Same constructor kwargs, method signatures, return lists, loss-dict keys and ``state_dict`` keys
(``encoder.{i}.{0,1}.*``, ``fc_mu``, ``fc_var``, ``decoder_input``, ``decoder.{i}.{0,1}.*``,
``final_layer.{0,1,3}.*``).  Differences that are deliberate and documented:

* tensors returned by ``decode``/``forward`` are logical NCHW with channels_last memory;
* ``reparameterize`` takes an optional ``eps`` so that noise can be injected;
* parameters are views of one packed buffer (``packing.py``); ``fc_mu``/``fc_var`` run as one GEMM.
"""
import os
from typing import List

import torch
from torch import nn

from .. import kernels as K
from .base import BaseVAE
from .blocks import Chain, ConvBNLeaky, conv_bn_leaky
from .packing import PackedBN, PackedConv, PackedLinear, PackedLinearGroup
from .types_ import Tensor


class _FinalLayer(nn.Module):
    """nn.Sequential(ConvTranspose2d, BatchNorm2d, LeakyReLU, Conv2d(->3), Tanh) (vanilla_vae.py:64-75);
    children "0", "1", "3" carry the parameters like the reference's Sequential indices."""

    def __init__(self, c, out_channels=3):
        super().__init__()
        self.add_module("0", PackedConv(c, c, 3, transposed=True, bias=True))
        self.add_module("1", PackedBN(c))
        self.add_module("3", PackedConv(c, out_channels, 3, bias=True))
        self.spec_up = K.ConvSpec(K.CONVT, c, c, 3, 2, 1, 1, K.ACT_NONE)
        self.spec_out = K.ConvSpec(K.CONV, c, out_channels, 3, 1, 1, 0, K.ACT_TANH)

    def reads_lazy_input(self, x_shape):
        """blocks.Chain: can the final block read its input through the previous block's BatchNorm coefficients?  (Only its
        one-node form does: the transposed conv's forward / weight-gradient kernels apply them while staging their patches.)"""
        B, H, W, _ = x_shape
        ho, wo = self.spec_up.out_hw(H, W)
        if os.environ.get("CTVAE_NO_LAZY_FINAL", "0") == "1":      # diagnostic
            return False
        return K.input_transform_supported(self.spec_out, B, ho, wo) and K.lazy_bn_input_supported(self.spec_up, B, H, W)

    def forward(self, x):
        up, bn, conv = self._modules["0"], self._modules["1"], self._modules["3"]
        B, H, W, _ = x.shape
        ho, wo = self.spec_up.out_hw(H, W)
        if K.input_transform_supported(self.spec_out, B, ho, wo):
            # one node: the BatchNorm+LeakyReLU output is applied on load by the 3-channel conv, never stored
            return K.ConvBNActConvAct.apply(x, up.weight, up.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                            bn.num_batches_tracked, conv.weight, conv.bias, self.training,
                                            self.spec_up, self.spec_out, K.ACT_LRELU)
        h = conv_bn_leaky(x, up, bn, self.spec_up, self.training)
        return K.ConvAct.apply(h, conv.weight, conv.bias, None, self.spec_out)


class VanillaVAE(BaseVAE):

    def __init__(self, in_channels: int, latent_dim: int, hidden_dims: List = None, **kwargs) -> None:
        super().__init__()
        self.latent_dim = latent_dim
        if hidden_dims is None:
            hidden_dims = [32, 64, 128, 256, 512]
        if hidden_dims[-1] != 512:
            # the reference hard-codes view(-1, 512, 2, 2) in decode (vanilla_vae.py:102)
            raise ValueError("VanillaVAE.decode assumes hidden_dims[-1] == 512 (reference behaviour)")
        self.in_channels = in_channels
        self.hidden_dims_fwd = list(hidden_dims)

        enc, c = [], in_channels
        for h in hidden_dims:
            enc.append(ConvBNLeaky(c, h, 3, 2, 1))
            c = h
        self.encoder = Chain(*enc)
        self.fc_mu = PackedLinear(hidden_dims[-1] * 4, latent_dim)
        self.fc_var = PackedLinear(hidden_dims[-1] * 4, latent_dim)
        grp = PackedLinearGroup([self.fc_mu, self.fc_var])
        self.fc_mu._linear_group = grp
        self.fc_var._linear_group = grp

        self.decoder_input = PackedLinear(latent_dim, hidden_dims[-1] * 4)
        self._dec_in_spec = K.ConvSpec(K.CONV, latent_dim, hidden_dims[-1] * 4, 1)
        hidden_dims.reverse()                  # the reference mutates the caller's list too (vanilla_vae.py:45)
        dec = []
        for i in range(len(hidden_dims) - 1):
            dec.append(ConvBNLeaky(hidden_dims[i], hidden_dims[i + 1], 3, 2, 1, out_pad=1, transposed=True))
        self.decoder = Chain(*dec)
        self.final_layer = _FinalLayer(hidden_dims[-1], 3)
        self._x_cache = None
        self.flatten_parameters()

    # -- helpers ----------------------------------------------------------------------------------
    def _input_nhwc(self, input):
        x = K.to_nhwc(input)
        self._x_cache = (input, input._version, x)        # the tensor itself: an address can be reused by a later batch
        return x

    def _cached_nhwc(self, input):
        c = self._x_cache
        if c is not None and c[0] is input and c[1] == input._version:
            return c[2]
        return K.to_nhwc(input)

    # -- reference API ----------------------------------------------------------------------------
    def _encode_heads(self, input: Tensor, for_latent_node: bool = False) -> Tensor:
        """[B,C,64,64] -> [B, 2L]: fc_mu | fc_var of the flattened encoder output as one GEMM.  for_latent_node: the result goes to
        kernels.GaussianLatent and nowhere else, which may then receive the GEMM's split-K slices instead of the summed tensor."""
        self.attach_grads()
        h = K.mark_sole_consumer(self.encoder(self._input_nhwc(input)))   # [B,2,2,512] NHWC, read by the heads only
        if tuple(h.shape[1:3]) != (2, 2):
            raise RuntimeError("VanillaVAE: fc_mu / fc_var take hidden_dims[-1]*4 features, i.e. a 2x2 encoder output "
                               "(64x64 input through 5 stride-2 layers, vanilla_vae.py:36-37)")
        return K.flatten_linear(h, self.fc_mu.weight, self.fc_mu.bias, 2 * self.latent_dim, lazy_slices=for_latent_node)

    def encode(self, input: Tensor) -> List[Tensor]:
        """[B,C,64,64] -> [mu [B,L], log_var [B,L]] (vanilla_vae.py:77-92)."""
        mu, log_var = K.SplitHeads.apply(self._encode_heads(input), self.latent_dim)
        return [mu, log_var]

    def decode(self, z: Tensor) -> Tensor:
        """[B,L] -> [B,3,64,64] (vanilla_vae.py:94-105)."""
        self.attach_grads()
        B = z.shape[0]
        zr = z.reshape(B, 1, 1, -1)
        if getattr(z, "_ctvae_grad_slices_ok", False):
            K.grad_slices_ok(zr)
        if K.LinearToNHWC.supported(B, self._dec_in_spec.ci, 512, 4, z.device):
            # the Linear's own epilogue leaves .view(-1,512,2,2) as the NHWC tensor decoder.0 gathers (one launch less)
            h = K.grad_slices_ok(K.LinearToNHWC.apply(zr, self.decoder_input.weight, self.decoder_input.bias, self._dec_in_spec, 512, 2, 2))
        else:
            h = K.ConvAct.apply(zr, self.decoder_input.weight, self.decoder_input.bias, None, self._dec_in_spec)
            h = K.grad_slices_ok(K._ToNHWC.apply(h.view(B, 512, 2, 2)))      # .view(-1,512,2,2) is NCHW; read by decoder.0 only
        h = self.decoder(h, last_reader=self.final_layer)                # read by final_layer only (marked by the chain)
        return K.to_nchw_view(self.final_layer(h))

    def reparameterize(self, mu: Tensor, logvar: Tensor, eps: Tensor = None) -> Tensor:
        """eps*exp(0.5*logvar)+mu (vanilla_vae.py:107-117); eps defaults to fresh N(0,1) noise on mu's device."""
        if eps is None:
            eps = torch.randn(mu.shape, dtype=mu.dtype, device=mu.device)
        return K.Reparameterize.apply(mu, logvar, eps.to(mu.device))

    def _latent_rng(self, device):
        """Device-side Philox state (key, stream position) of the in-kernel noise: seeded once from torch's generator."""
        st = getattr(self, "_rng_state", None)
        if st is None or st.device != device:
            key = int(torch.randint(0, 2 ** 62, (1,)).item())
            st = self._rng_state = torch.tensor([key, 0], dtype=torch.int64, device=device)
        return st

    def forward(self, input: Tensor, eps: Tensor = None, **kwargs) -> List[Tensor]:
        """encode -> reparameterize -> decode (vanilla_vae.py:119-122).  The latent section runs as one node
        (kernels.GaussianLatent): with eps None and gradients on, the N(0,1) noise is drawn inside its kernel."""
        one_node = self.latent_dim % 4 == 0 and type(self).reparameterize is VanillaVAE.reparameterize
        heads = self._encode_heads(input, for_latent_node=one_node)
        if one_node:
            if eps is None and not (torch.is_grad_enabled() and heads.requires_grad):
                eps = torch.randn((heads.shape[0], self.latent_dim), dtype=heads.dtype, device=heads.device)   # no backward: no state bump
            rng = self._latent_rng(heads.device) if eps is None else None
            mu, log_var, z = K.GaussianLatent.apply(heads, eps.to(heads.device) if eps is not None else None, rng)
            K.grad_slices_ok(z)          # z's gradient comes back to that node only (decode below is its one consumer)
        else:
            mu, log_var = K.SplitHeads.apply(heads, self.latent_dim)
            z = self.reparameterize(mu, log_var, eps)
        return [self.decode(z), input, mu, log_var]

    def loss_function(self, *args, **kwargs) -> dict:
        """MSE + M_N*KL (vanilla_vae.py:124-146); 'KLD' carries the reference's flipped sign."""
        recons, input, mu, log_var = args[0], args[1], args[2], args[3]
        kld_weight = kwargs['M_N']
        out = K.VAELoss.apply(K.to_nhwc(recons), self._cached_nhwc(input), mu, log_var, None, kld_weight)
        return {'loss': out[0], 'Reconstruction_Loss': out[1].detach(), 'KLD': out[3].detach()}

    def sample(self, num_samples: int, current_device: int, **kwargs) -> Tensor:
        z = torch.randn(num_samples, self.latent_dim).to(current_device)
        return self.decode(z)

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        return self.forward(x)[0]
