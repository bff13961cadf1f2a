"""MCQ-VAE on HIP kernels — drop-in for models/mcq_vae.py:7-317 (VectorQuantizerMS,
MultipleCodebookVectorQuantizer, MCQVAE) with identical constructor kwargs, public methods, return
conventions and ``state_dict`` keys.  No BatchNorm; bias + LeakyReLU/ReLU/Tanh and the residual adds
are fused into the conv epilogues; the C codebooks are searched / looked up in single launches.
"""
from typing import List

import torch
from torch import nn

from .. import kernels as K
from .base import BaseVAE
from .blocks import ConvAct, LeakyReLU, ResidualLayer
from .packing import PackedEmbedding
from .types_ import Tensor


class VectorQuantizerMS(nn.Module):
    """One codebook (mcq_vae.py:7-74).  Used stand-alone it quantises all D channels of its input."""

    def __init__(self, num_embeddings: int, embedding_dim: int, beta: float = 0.25):
        super().__init__()
        self.K, self.D, self.beta = num_embeddings, embedding_dim, beta
        self.embedding = PackedEmbedding(self.K, self.D)

    def compute_inds(self, latents: Tensor) -> Tensor:
        lat = K.to_nhwc(latents)
        return K.vq_compute_inds(lat, [self.embedding.weight], self.K, 1)[:, 0]            # [B,H,W]

    def compute_latents(self, latents: Tensor, encoding_inds: Tensor):
        lat = K.to_nhwc(latents)
        B, H, W, _ = lat.shape
        q, vq_loss = K.VQLookup.apply(lat, encoding_inds.reshape(B, 1, H, W), self.beta, self.K, 1, self.embedding.weight)
        return K.to_nchw_view(q), vq_loss

    def forward(self, latents: Tensor, inds: bool = False):
        encoding_inds = self.compute_inds(latents)
        q, vq_loss = self.compute_latents(latents, encoding_inds)
        return (q, vq_loss, encoding_inds) if inds else (q, vq_loss)


class MultipleCodebookVectorQuantizer(nn.Module):
    """C codebooks sharing embedding_dim (mcq_vae.py:78-137), incl. the slice quirk latents[:, i:i+D/C]."""

    def __init__(self, num_embeddings: int, embedding_dim: int, codebooks: int, beta: float = 0.25):
        super().__init__()
        assert embedding_dim % codebooks == 0
        self.nb_codebooks = codebooks
        self.reduced_embedding_dim = embedding_dim // codebooks
        self.num_embeddings, self.beta = num_embeddings, beta
        self.quantizers = nn.ModuleList([VectorQuantizerMS(num_embeddings, self.reduced_embedding_dim, beta)
                                         for _ in range(codebooks)])

    def _codebooks(self):
        ws = [q.embedding.weight for q in self.quantizers]
        step = ws[0].numel() * 4
        for i, w in enumerate(ws):
            if w.data_ptr() != ws[0].data_ptr() + i * step or not w.is_contiguous():
                raise RuntimeError("codebooks are not stored back to back: call model.flatten_parameters()")
        return ws

    def compute_inds(self, latents: Tensor) -> Tensor:
        """[B,D,H,W] -> int64 [B,C,H,W]"""
        return K.vq_compute_inds(K.to_nhwc(latents), self._codebooks(), self.num_embeddings, self.nb_codebooks)

    def compute_latents(self, latents: Tensor, encoding_inds: Tensor):
        q, vq_loss = K.VQLookup.apply(K.to_nhwc(latents), encoding_inds, self.beta, self.num_embeddings, self.nb_codebooks,
                                      *self._codebooks())
        return K.to_nchw_view(q), vq_loss

    def forward(self, latents: Tensor, inds: bool = False):
        encoding_inds = self.compute_inds(latents)
        q, vq_loss = self.compute_latents(latents, encoding_inds)
        return (q, vq_loss, encoding_inds) if inds else (q, vq_loss)


def build_mcq_encoder(in_channels, hidden_dims, embedding_dim):
    """mcq_vae.py:166-193: [Conv k4s2 + LReLU]*n, Conv3x3 + LReLU, 6 ResidualLayers, LReLU, Conv1x1 + LReLU."""
    mods, c = [], in_channels
    for h in hidden_dims:
        mods.append(ConvAct(c, h, 4, 2, 1, K.ACT_LRELU))
        c = h
    mods.append(ConvAct(c, c, 3, 1, 1, K.ACT_LRELU))
    for j in range(6):
        mods.append(ResidualLayer(c, c, post_act=K.ACT_LRELU if j == 5 else K.ACT_NONE))
    mods.append(LeakyReLU(fused=True))          # applied in the last residual block's epilogue
    mods.append(ConvAct(c, embedding_dim, 1, 1, 0, K.ACT_LRELU))
    return nn.Sequential(*mods)


def build_mcq_decoder(out_channels, hidden_dims_fwd, embedding_dim):
    """mcq_vae.py:201-239: Conv3x3 + LReLU, 6 ResidualLayers, LReLU, [ConvT k4s2 + LReLU]*(n-1), ConvT k4s2 + Tanh."""
    top = hidden_dims_fwd[-1]
    mods = [ConvAct(embedding_dim, top, 3, 1, 1, K.ACT_LRELU)]
    for j in range(6):
        mods.append(ResidualLayer(top, top, post_act=K.ACT_LRELU if j == 5 else K.ACT_NONE))
    mods.append(LeakyReLU(fused=True))
    rev = list(reversed(hidden_dims_fwd))
    for i in range(len(rev) - 1):
        mods.append(ConvAct(rev[i], rev[i + 1], 4, 2, 1, K.ACT_LRELU, transposed=True))
    mods.append(ConvAct(rev[-1], out_channels, 4, 2, 1, K.ACT_TANH, transposed=True))
    return nn.Sequential(*mods)


class MCQVAE(BaseVAE):

    def __init__(self, in_channels: int, embedding_dim: int, num_embeddings: int, hidden_dims: List = None,
                 beta: float = 0.25, img_size: int = 64, codebooks: int = 1, **kwargs) -> None:
        super().__init__()
        self.embedding_dim, self.num_embeddings = embedding_dim, num_embeddings
        self.img_size, self.in_channels, self.beta = img_size, in_channels, beta
        if hidden_dims is None:
            hidden_dims = [128, 256]
        self.nb_conv = len(hidden_dims)
        fwd = list(hidden_dims)
        self.encoder = build_mcq_encoder(in_channels, fwd, embedding_dim)
        self.vq_layer = MultipleCodebookVectorQuantizer(num_embeddings, embedding_dim, codebooks, self.beta)
        self.decoder = build_mcq_decoder(in_channels, fwd, embedding_dim)
        hidden_dims.reverse()                   # reference side effect on the caller's list (mcq_vae.py:218)
        self._x_cache = None
        self.flatten_parameters()

    def _input_nhwc(self, input):
        x = K.to_nhwc(input)
        self._x_cache = (input, input._version, x)        # the tensor itself: an address can be reused by a later batch
        return x

    def _cached_nhwc(self, input):
        c = self._x_cache
        if c is not None and c[0] is input and c[1] == input._version:
            return c[2]
        return K.to_nhwc(input)

    def encode(self, input: Tensor) -> List[Tensor]:
        self.attach_grads()
        return [K.to_nchw_view(self.encoder(self._input_nhwc(input)))]

    def decode(self, z: Tensor) -> Tensor:
        self.attach_grads()
        return K.to_nchw_view(self.decoder(K.to_nhwc(z)))

    def forward(self, input: Tensor, **kwargs) -> List[Tensor]:
        encoding = self.encode(input)[0]
        quantized_inputs, vq_loss = self.vq_layer(encoding)
        return [self.decode(quantized_inputs), input, vq_loss]

    def loss_function(self, *args, **kwargs) -> dict:
        """mse + vq_loss (mcq_vae.py:267-284).  The reference returns Reconstruction_Loss / VQ_Loss un-detached; here
        'loss' carries the gradient (mse + vq in one kernel) and 'Reconstruction_Loss' is a reported value of that kernel."""
        recons, input, vq_loss = args[0], args[1], args[2]
        out = K.VAELoss.apply(K.to_nhwc(recons), self._cached_nhwc(input), None, None, vq_loss, 0.0)
        return {'loss': out[0], 'Reconstruction_Loss': out[1], 'VQ_Loss': vq_loss}

    def sample(self, num_samples: int, current_device: int, **kwargs) -> Tensor:
        n = self.img_size // 2 ** self.nb_conv
        z = torch.randn(num_samples, self.embedding_dim, n, n).to(current_device)
        quantized_inputs, _ = self.vq_layer(z)
        return self.decode(quantized_inputs)

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        return self.forward(x)[0]
