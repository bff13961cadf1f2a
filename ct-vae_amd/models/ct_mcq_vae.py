"""CT-MCQ-VAE on HIP kernels — drop-in for ``CTMCQVAE`` (reference: models/ct_mcq_vae.py:339-713).

The conv encoder / multi-codebook VQ / conv decoder are the MCQ-VAE HIP path (ct_mcq_vae.py:365-448 is
a textual copy of mcq_vae.py:161-239); the causal-transition layer between index search and codebook
lookup is ``causal.CausalTransition`` (torch-level + HIP Gumbel kernel; parity unpinned, see its header).
Modes, return lists, loss-dict keys and ``state_dict`` keys follow the reference.
"""
import os
from typing import List, Union

import torch
from torch.nn import functional as F

from .. import kernels as K
from .base import BaseVAE
from .blocks import run_with_companion
from .causal import CausalTransition
from .mcq_vae import MultipleCodebookVectorQuantizer, build_mcq_decoder, build_mcq_encoder
from .types_ import Tensor

_PAIR_ENCODE = os.environ.get("CTVAE_NO_PAIR_ENCODE", "0") != "1"    # diagnostic: x and y through separate encoder passes


class CTMCQVAE(BaseVAE):

    def __init__(self, in_channels: int, embedding_dim: int, action_dim: int, num_embeddings: int,
                 hidden_dims: List = None, causal_hidden_dims: List = None, beta: float = 0.25, gamma: float = 0.25,
                 img_size: int = 64, codebooks: int = 1, skip_transition=False, **kwargs) -> None:
        super().__init__()
        self.embedding_dim, self.num_embeddings = embedding_dim, num_embeddings
        self.img_size, self.in_channels = img_size, in_channels
        self.beta, self.gamma, self.codebooks, self.skip_transition = beta, gamma, codebooks, skip_transition
        if hidden_dims is None:
            hidden_dims = [128, 256]
        self.nb_latents = self.img_size // 2 ** len(hidden_dims)
        fwd = list(hidden_dims)
        self.encoder = build_mcq_encoder(in_channels, fwd, embedding_dim)
        self.vq_layer = MultipleCodebookVectorQuantizer(num_embeddings, embedding_dim, codebooks, self.beta)
        self.ct_layer = CausalTransition(num_embeddings, action_dim, causal_hidden_dims, **kwargs)
        self.decoder = build_mcq_decoder(in_channels, fwd, embedding_dim)
        hidden_dims.reverse()                   # reference side effect (ct_mcq_vae.py:427)
        self._x_cache = {}
        self.flatten_parameters()

    # -- conv path ------------------------------------------------------------------------------------
    def _nhwc(self, t):
        x = K.to_nhwc(t)
        self._x_cache[id(t)] = (t, t._version, x)     # keyed by the tensor object (held): an address can be reused by a later batch
        if len(self._x_cache) > 4:
            self._x_cache.pop(next(iter(self._x_cache)))
        return x

    def _cached_nhwc(self, t):
        c = self._x_cache.get(id(t))
        return c[2] if (c is not None and c[0] is t and c[1] == t._version) else K.to_nhwc(t)

    def encode(self, input: Tensor) -> List[Tensor]:
        self.attach_grads()
        return [K.to_nchw_view(self.encoder(self._nhwc(input)))]

    def encode_pair(self, input: Tensor, input_y: Tensor):
        """(latents of input with autograd, latents of input_y without, code indices of both or None): the two encoder passes
        of a transition sample (ct_mcq_vae.py:536-539, 565-566) through the same launches -- the encoder has no BatchNorm, so
        rows are independent and y rides along as companion rows (kernels.adjacent_rows).  Only x's pass is trained: of y the
        modes use the code indices alone."""
        self.attach_grads()
        x, y = K.to_nhwc_pair(input, input_y)
        self._x_cache[id(input)] = (input, input._version, x)
        self._x_cache[id(input_y)] = (input_y, input_y._version, y)
        while len(self._x_cache) > 4:
            self._x_cache.pop(next(iter(self._x_cache)))
        lx, ly = run_with_companion(self.encoder, x, y)
        both = K.adjacent_rows(lx, ly)
        inds = None
        if both is not None:                   # one index search over both halves
            inds = self.vq_layer.compute_inds(K.to_nchw_view(both))
        return K.to_nchw_view(lx), K.to_nchw_view(ly), inds

    def decode(self, z: Tensor) -> Tensor:
        self.attach_grads()
        return K.to_nchw_view(self.decoder(K.to_nhwc(z)))

    # -- index <-> one-hot formatting around the causal layer (ct_mcq_vae.py:472-496) -------------------
    def ct_preprocess(self, x: Tensor, latents_shape) -> Tensor:
        """[B,K,H,W] int64 -> one-hot float [B, N, K*H, W]"""
        oh = K.one_hot_f32(x, self.num_embeddings) if x.is_cuda else F.one_hot(x, num_classes=self.num_embeddings).to(dtype=torch.float32)   # [B,K,H,W,N]
        oh = oh.view((latents_shape[0], self.codebooks * latents_shape[2], latents_shape[3], self.num_embeddings))
        return oh.permute(0, 3, 1, 2)

    def ct_postprocess(self, x: Tensor, latents_shape):
        """[B, N, K*H, W] -> arg-max indices [B,K,H,W]"""
        x = x.permute(0, 2, 3, 1)
        x = x.reshape((latents_shape[0], self.codebooks, latents_shape[2], latents_shape[3], self.num_embeddings))
        return torch.argmax(x, dim=-1)

    def _const(self, v, dev):
        """0-dim constant on the device, made once (a torch.full per step and constant is a launch each)."""
        c = self.__dict__.setdefault("_consts", {})
        t = c.get((v, dev))
        if t is None:
            t = c[(v, dev)] = torch.full((), float(v), device=dev)
        return t

    # -- modes (ct_mcq_vae.py:501-591) --------------------------------------------------------------------
    def forward_base(self, input: Tensor, **kwargs) -> List[Tensor]:
        latents = self.encode(input)[0]
        encoding_inds = self.vq_layer.compute_inds(latents)
        shape = latents.shape
        one_hot = self.ct_preprocess(encoding_inds, shape)
        ct_encodings, ct_reg, *ct_metrics = self.ct_layer(one_hot)
        ct_loss = ct_reg + self.ct_layer.latent_loss(ct_encodings, one_hot, target_inds=encoding_inds)
        ct_inds = self.ct_postprocess(ct_encodings, shape)
        q, vq_loss = self.vq_layer.compute_latents(latents, encoding_inds if self.skip_transition else ct_inds)
        dev = input.device
        return [self.decode(q), input, vq_loss, ct_loss,
                {**{"causal_acc": self._const(0.0, dev), "causal_nodir_acc": self._const(0.0, dev),
                    "mode": "base", "mode_id": self._const(0.0, dev)}, **ct_metrics[0]}]

    def _encode_xy(self, input, input_y):
        """latents of x (autograd), code indices of x and of y."""
        if _PAIR_ENCODE and input_y is not None and input_y.shape == input.shape:
            latents, lat_y, inds = self.encode_pair(input, input_y)
            B = input.shape[0]
            if inds is not None:
                return latents, inds[:B], inds[B:]
            with torch.no_grad():
                return latents, self.vq_layer.compute_inds(latents), self.vq_layer.compute_inds(lat_y)
        latents = self.encode(input)[0]
        encoding_inds = self.vq_layer.compute_inds(latents)
        with torch.no_grad():                 # indices cut the graph: the encoder pass on y has no backward
            inds_y = self.vq_layer.compute_inds(self.encode(input_y)[0])
        return latents, encoding_inds, inds_y

    def forward_action(self, input: Tensor, action: Tensor, input_y: Tensor = None, **kwargs) -> List[Tensor]:
        latents, encoding_inds, inds_y = self._encode_xy(input, input_y)
        shape = latents.shape
        one_hot = self.ct_preprocess(encoding_inds, shape)
        ct_encodings, ct_reg, *ct_metrics = self.ct_layer.forward_action(one_hot, action)
        # the target is the one-hot of inds_y (ct_preprocess): its arg-max are the indices themselves
        ct_loss = ct_reg + self.ct_layer.latent_loss(ct_encodings, None, target_inds=inds_y)
        ct_inds = self.ct_postprocess(ct_encodings, shape)
        q, _ = self.vq_layer.compute_latents(latents, encoding_inds if self.skip_transition else ct_inds)
        dev = input.device
        return [self.decode(q), input_y, self._const(0.0, dev), ct_loss,
                {**{"causal_acc": self._const(0.0, dev), "causal_nodir_acc": self._const(0.0, dev),
                    "mode": "action", "mode_id": self._const(1.0, dev)}, **ct_metrics[0]}]

    def forward_causal(self, input: Tensor, input_y: Tensor, action: Tensor = None, **kwargs) -> List[Tensor]:
        lat_x, enc_x, enc_y = self._encode_xy(input, input_y)      # the indices cut the graph: no encoder backward in this mode
        shape = lat_x.shape
        recons_action, ct_reg, *ct_metrics = self.ct_layer.forward_transition(self.ct_preprocess(enc_x, shape),
                                                                               self.ct_preprocess(enc_y, shape))
        nodir = self.ct_layer.causal_undirected_accuracy(recons_action, action)
        acc = self.ct_layer.causal_accuracy(recons_action, action)
        dev = input.device
        return [recons_action, action, self._const(0.0, dev), ct_reg.to(dev),
                {**{"causal_acc": acc, "causal_nodir_acc": nodir, "mode": "causal", "mode_id": self._const(2.0, dev)},
                 **ct_metrics[0]}]

    FORWARD_MODES = {"base": forward_base, "action": forward_action, "causal": forward_causal}

    def forward(self, input: Tensor, input_y: Tensor = None, action: Tensor = None,
                mode: Union[str, List[str]] = "base", **kwargs) -> List[Tensor]:
        if type(mode) is list:                # one mode per batch (datasets/transition.py:128-190)
            mode = mode[0]
        if input_y is not None:
            input_y = input_y.to(input.device)
        if action is not None:
            action = action.to(input.device)
        return CTMCQVAE.FORWARD_MODES[mode](self, input=input, input_y=input_y, action=action)

    def loss_function(self, *args, **kwargs) -> dict:
        """recons + vq + gamma*ct (ct_mcq_vae.py:594-620); causal mode is a classification (cross-entropy)."""
        recons, input, vq_loss, ct_loss = args[0], args[1], args[2], args[3]
        metrics = {} if len(args) < 5 else args[4]
        if len(metrics) > 0 and "mode" in metrics and metrics["mode"] == "causal":
            recons_loss = F.cross_entropy(recons.clamp(min=1e-4).log(), torch.argmax(input, dim=-1))
            loss = recons_loss + vq_loss + self.gamma * ct_loss
        else:
            extra = vq_loss + self.gamma * ct_loss
            out = K.VAELoss.apply(K.to_nhwc(recons), self._cached_nhwc(input), None, None, extra, 0.0)
            loss, recons_loss = out[0], out[1]
        return {**{'loss': loss, 'Reconstruction_Loss': recons_loss, 'VQ_Loss': vq_loss, 'CT_Loss': ct_loss}, **metrics}

    def sample(self, num_samples: int, current_device: int, **kwargs) -> Tensor:
        z = torch.randn(num_samples, self.embedding_dim, self.nb_latents, self.nb_latents).to(current_device)
        quantized_inputs, _ = self.vq_layer(z)
        return self.decode(quantized_inputs)

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        if "mode" in kwargs and kwargs["mode"] == "causal":
            kwargs["mode"] = "action"
        return self.forward(x, **kwargs)[0]
