"""Deterministic parameter filler owned by this build.

Parity fixtures never rely on seed -> default-init equivalence across torch versions
(SURVEY.md §8c "Version skew"): every golden vector carries explicit weights produced here and
loaded into the reference with ``load_state_dict``.  The rules depend only on the tensor's key
and shape, so the reference modules, the CPU oracle and the HIP models all receive the same bits.
"""
import math
from collections import OrderedDict

import torch


def fill_state(specs, seed: int) -> "OrderedDict[str, torch.Tensor]":
    """specs: iterable of (key, shape, dtype) in state_dict order.  Returns CPU tensors."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    out = OrderedDict()

    def uni(shape, lo, hi):
        return torch.rand(tuple(shape), generator=g, dtype=torch.float32) * (hi - lo) + lo

    for key, shape, dtype in specs:
        shape = tuple(int(s) for s in shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t = torch.zeros(shape, dtype=torch.int64)
        elif leaf == "running_mean":
            t = uni(shape, -0.1, 0.1)
        elif leaf == "running_var":
            t = uni(shape, 0.8, 1.2)
        elif leaf == "pe":                      # PositionalEncoding buffer: keep what the module built
            continue
        elif len(shape) == 4:                   # Conv2d [Co,Ci,kh,kw] / ConvTranspose2d [Ci,Co,kh,kw]
            bound = math.sqrt(6.0 / (shape[1] * shape[2] * shape[3]))
            if ".resblock." in key:
                bound *= 0.5
            t = uni(shape, -bound, bound)
        elif len(shape) == 2 and "embedding" in key:
            t = uni(shape, -0.5, 0.5)
        elif len(shape) == 2:                   # Linear [out,in]
            bound = math.sqrt(3.0 / shape[1])
            t = uni(shape, -bound, bound)
        elif len(shape) == 1 and leaf == "weight":   # BatchNorm gamma
            t = uni(shape, 0.5, 1.5)
        elif len(shape) == 1 and leaf == "bias":
            t = uni(shape, -0.1, 0.1)
        else:
            t = uni(shape, -0.1, 0.1)
        out[key] = t.to(dtype) if dtype is not None and t.dtype != dtype and t.is_floating_point() else t
    return out


def specs_of(module) -> list:
    """(key, shape, dtype) triples of a module's state_dict, in order."""
    return [(k, tuple(v.shape), v.dtype) for k, v in module.state_dict().items()]


def synthetic_batch(seed: int, batch: int, latent_dim: int = 128, img: int = 64, channels: int = 3):
    """SURVEY.md §8d synthetic inputs: x ~ U[0,1) (ToTensor range), eps ~ N(0,1), one CPU generator."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    x = torch.rand(batch, channels, img, img, generator=g)
    eps = torch.randn(batch, latent_dim, generator=g)
    return x, eps


def synthetic_pairs(seed: int, batch: int, action_dim: int = 12, img: int = 64, channels: int = 3):
    """Transition-dataset-shaped batch (datasets/transition.py:86-106): x, y ~ U[0,1), one-hot action."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    x = torch.rand(batch, channels, img, img, generator=g)
    y = torch.rand(batch, channels, img, img, generator=g)
    a = torch.zeros(batch, action_dim)
    a[torch.arange(batch), torch.arange(batch) % action_dim] = 1.0
    return x, y, a


CT_NOISE_TAGS = ("mask_dropout", "mask_gumbel", "pos_dropout", "adj_gumbel", "kl_target", "exo_noise", "endo_noise")


def ct_noise(seed: int, tag: str, k: int, shape, p: float = 0.0):
    """The k-th injected draw of kind ``tag`` of CausalTransition (every stochastic op takes its noise from a CPU
    generator so that the reference modules, the CPU oracle and the HIP path see identical bits).

    *_dropout -> keep mask (1.0 where U >= p), *_gumbel -> standard exponential draws E (the Gumbel noise is -log E, which
    is how ``F.gumbel_softmax`` forms it), kl_target -> U[0,1) (``torch.rand`` in ct_mcq_vae.py:316), *_noise -> N(0,1)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed) * 1000003 + CT_NOISE_TAGS.index(tag) * 1009 + int(k))
    shape = tuple(int(s) for s in shape)
    if tag.endswith("_dropout"):
        return (torch.rand(shape, generator=g) >= p).to(torch.float32)
    if tag.endswith("_gumbel"):
        return torch.empty(shape, dtype=torch.float32).exponential_(generator=g)
    if tag.endswith("_noise"):
        return torch.randn(shape, generator=g)
    return torch.rand(shape, generator=g)
