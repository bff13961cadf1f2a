"""ConditionalVAE (models/cvae.py:8-176, configs/cvae.yaml) on the HIP path -- SURVEY.md §8f rank 4.

VanillaVAE's conv stacks and kernels with the label fed in twice: ``embed_class`` (Linear num_classes -> img_size^2) turns the
one-hot / attribute vector into an extra input plane next to ``embed_data`` (a 1x1 Conv2d on the image), so the first encoder
layer reads ``in_channels + 1`` channels, and the label is concatenated to z in front of ``decoder_input``.  Objective, sampling
and the rest of the network are VanillaVAE's.  ``forward`` takes ``labels`` like the reference (experiment.py:51 passes them)
and an optional ``eps`` (injected noise).

Layout notes: the label plane is the Linear's output read as a one-channel NHWC image (no copy); image and plane meet in one
``torch.cat`` on the channel axis of the NHWC tensors; ``embed_class`` / ``decoder_input`` reserve zero rows up to the next
multiple of 32 input features (the tile kernels gather whole 32-wide K chunks, ``PackedLinear.in_padded``).
"""
import torch
from torch import nn

from .. import kernels as K
from .base import BaseVAE
from .blocks import ConvBNLeaky
from .packing import PackedConv, PackedLinear, PackedLinearGroup
from .types_ import List, Tensor
from .vanilla_vae import _FinalLayer


class ConditionalVAE(BaseVAE):
    uses_labels = True          # the training harness hands the batch's labels to the captured step (experiment.py)

    def __init__(self, in_channels: int, num_classes: int, latent_dim: int, hidden_dims: List = None, img_size: int = 64,
                 **kwargs) -> None:
        super().__init__()
        self.latent_dim, self.img_size, self.num_classes, self.in_channels = latent_dim, img_size, num_classes, in_channels
        if hidden_dims is None:
            hidden_dims = [32, 64, 128, 256, 512]
        if hidden_dims[-1] != 512:
            raise ValueError("ConditionalVAE.decode assumes hidden_dims[-1] == 512 (cvae.py:102)")
        self.embed_class = PackedLinear(num_classes, img_size * img_size, pad_in_to=32)
        self._embed_spec = K.ConvSpec(K.CONV, self.embed_class.in_padded, img_size * img_size, 1)
        self.embed_data = PackedConv(in_channels, in_channels, 1, bias=True)
        self._data_spec = K.ConvSpec(K.CONV, in_channels, in_channels, 1, 1, 0, 0, K.ACT_NONE)
        enc, c = [], in_channels + 1                      # the extra label channel (cvae.py:29)
        for h in hidden_dims:
            enc.append(ConvBNLeaky(c, h, 3, 2, 1))
            c = h
        self.encoder = nn.Sequential(*enc)
        feat = hidden_dims[-1] * 4
        self.fc_mu = PackedLinear(feat, latent_dim)
        self.fc_var = PackedLinear(feat, latent_dim)
        grp = PackedLinearGroup([self.fc_mu, self.fc_var])
        self.fc_mu._linear_group = grp
        self.fc_var._linear_group = grp
        self._head_spec = K.ConvSpec(K.CONV, feat, 2 * latent_dim, 1)
        self.decoder_input = PackedLinear(latent_dim + num_classes, feat, pad_in_to=32)
        self._dec_in_spec = K.ConvSpec(K.CONV, self.decoder_input.in_padded, feat, 1)
        hidden_dims.reverse()
        dec = []
        for i in range(len(hidden_dims) - 1):
            dec.append(ConvBNLeaky(hidden_dims[i], hidden_dims[i + 1], 3, 2, 1, out_pad=1, transposed=True))
        self.decoder = nn.Sequential(*dec)
        self.final_layer = _FinalLayer(hidden_dims[-1], 3)
        self.flatten_parameters()

    @staticmethod
    def _pad_cols(t, width):
        pad = width - t.shape[1]
        return t if pad == 0 else torch.cat([t, torch.zeros(t.shape[0], pad, dtype=t.dtype, device=t.device)], dim=1)

    def _conditioned_input(self, input: Tensor, y: Tensor) -> Tensor:
        """NHWC [B, S, S, C+1]: embed_data(input) next to the label plane embed_class(y) (cvae.py:124-129)."""
        B, S = input.shape[0], self.img_size
        yp = self._pad_cols(y, self.embed_class.in_padded).reshape(B, 1, 1, -1)
        plane = K.ConvAct.apply(yp, self.embed_class.weight, self.embed_class.bias, None, self._embed_spec).view(B, S, S, 1)
        data = K.ConvAct.apply(K.to_nhwc(input), self.embed_data.weight, self.embed_data.bias, None, self._data_spec)
        return torch.cat([data, plane], dim=-1)

    def encode(self, input: Tensor) -> List[Tensor]:
        """input: the conditioned [B, C+1, S, S] tensor (logical NCHW), as in the reference (cvae.py:82-98)."""
        mu, log_var = K.SplitHeads.apply(self._encode_heads(K.to_nhwc(input)), self.latent_dim)
        return [mu, log_var]

    def _encode_heads(self, x_nhwc: Tensor) -> Tensor:
        self.attach_grads()
        h = self.encoder(x_nhwc)
        return K.flatten_linear(h, self.fc_mu.weight, self.fc_mu.bias, self._head_spec.co)

    def decode(self, z: Tensor) -> Tensor:
        """z: [B, latent_dim + num_classes] (cvae.py:100-105)."""
        self.attach_grads()
        B = z.shape[0]
        z = self._pad_cols(z, self.decoder_input.in_padded)
        h = K.ConvAct.apply(z.reshape(B, 1, 1, -1), self.decoder_input.weight, self.decoder_input.bias, None, self._dec_in_spec)
        h = K._ToNHWC.apply(h.view(B, 512, 2, 2))
        return K.to_nchw_view(self.final_layer(self.decoder(h)))

    def reparameterize(self, mu: Tensor, logvar: Tensor, eps: Tensor = None) -> Tensor:
        if eps is None:
            eps = torch.randn(mu.shape, dtype=mu.dtype, device=mu.device)
        return K.Reparameterize.apply(mu, logvar, eps.to(mu.device))

    def forward(self, input: Tensor, eps: Tensor = None, **kwargs) -> List[Tensor]:
        self.attach_grads()
        y = kwargs['labels'].to(device=input.device, dtype=torch.float32)
        heads = self._encode_heads(self._conditioned_input(input, y))
        mu, log_var = K.SplitHeads.apply(heads, self.latent_dim)
        z = self.reparameterize(mu, log_var, eps)
        return [self.decode(torch.cat([z, y], dim=1)), input, mu, log_var]

    def loss_function(self, *args, **kwargs) -> dict:
        """MSE + M_N * KL (cvae.py:132-146); 'KLD' carries the reference's flipped sign."""
        recons, input, mu, log_var = args[0], args[1], args[2], args[3]
        out = K.VAELoss.apply(K.to_nhwc(recons), K.to_nhwc(input), mu, log_var, None, kwargs['M_N'])
        return {'loss': out[0], 'Reconstruction_Loss': out[1].detach(), 'KLD': out[3].detach()}

    def sample(self, num_samples: int, current_device: int, **kwargs) -> Tensor:
        y = kwargs['labels'].float().to(current_device)
        z = torch.randn(num_samples, self.latent_dim).to(current_device)
        return self.decode(torch.cat([z, y], dim=1))

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        return self.forward(x, **kwargs)[0]
