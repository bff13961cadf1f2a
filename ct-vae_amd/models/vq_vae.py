"""VQVAE (models/vq_vae.py:73-225) on the HIP path: MCQ-VAE's conv stacks (same builders, same parameter names) around a
single-codebook quantiser whose embedding sits directly under ``vq_layer`` (vq_vae.py:7-56) -- SURVEY.md §8f rank 4."""
import torch

from .. import kernels as K
from .base import BaseVAE
from .mcq_vae import VectorQuantizerMS, build_mcq_decoder, build_mcq_encoder
from .types_ import List, Tensor, Union


class VQVAE(BaseVAE):

    def __init__(self, in_channels: int, embedding_dim: int, num_embeddings: int, hidden_dims: List = None,
                 beta: float = 0.25, img_size: int = 64, **kwargs) -> None:
        super().__init__()
        self.embedding_dim, self.num_embeddings, self.img_size, self.beta = embedding_dim, num_embeddings, img_size, beta
        if hidden_dims is None:
            hidden_dims = [128, 256]
        fwd = list(hidden_dims)
        self.encoder = build_mcq_encoder(in_channels, fwd, embedding_dim)
        self.vq_layer = VectorQuantizerMS(num_embeddings, embedding_dim, self.beta)
        self.decoder = build_mcq_decoder(in_channels, fwd, embedding_dim)
        hidden_dims.reverse()                   # reference side effect on the caller's list (vq_vae.py:145)
        self._x_cache = None
        self.flatten_parameters()

    def _cached_nhwc(self, input):
        c = self._x_cache
        if c is not None and c[0] is input and c[1] == input._version:      # the tensor itself: addresses get reused
            return c[2]
        return K.to_nhwc(input)

    def encode(self, input: Tensor) -> List[Tensor]:
        self.attach_grads()
        x = K.to_nhwc(input)
        self._x_cache = (input, input._version, x)
        return [K.to_nchw_view(self.encoder(x))]

    def decode(self, z: Tensor) -> Tensor:
        self.attach_grads()
        return K.to_nchw_view(self.decoder(K.to_nhwc(z)))

    def forward(self, input: Tensor, **kwargs) -> List[Tensor]:
        encoding = self.encode(input)[0]
        quantized_inputs, vq_loss = self.vq_layer(encoding)
        return [self.decode(quantized_inputs), input, vq_loss]

    def loss_function(self, *args, **kwargs) -> dict:
        recons, input, vq_loss = args[0], args[1], args[2]
        out = K.VAELoss.apply(K.to_nhwc(recons), self._cached_nhwc(input), None, None, vq_loss, 0.0)
        return {'loss': out[0], 'Reconstruction_Loss': out[1], 'VQ_Loss': vq_loss}

    def sample(self, num_samples: int, current_device: Union[int, str], **kwargs) -> Tensor:
        raise Warning('VQVAE sampler is not implemented.')

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        return self.forward(x)[0]
