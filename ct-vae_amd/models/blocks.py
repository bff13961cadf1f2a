"""HIP-backed building blocks that reproduce the reference's module tree (so ``state_dict`` keys match)
while executing fused kernels.  All blocks take and return NHWC-contiguous ``[B,H,W,C]`` tensors; the
model classes convert at the public NCHW boundary."""
import torch
from torch import nn

from .. import kernels as K
from .packing import PackedBN, PackedConv


def conv_bn_leaky(x, conv, bn, spec, training, lazy_out=False):
    """Conv (+bias) -> train/eval BatchNorm2d -> LeakyReLU on the holders `conv` / `bn`.  lazy_out: see kernels.ConvBNAct --
    only for a tensor whose one consumer is a ConvBNLeaky block that reads it through the coefficients (blocks.Chain)."""
    # num_batches_tracked is advanced by the finalize kernel (no separate launch)
    a = K.ConvBNAct.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                          training, spec, K.ACT_LRELU, bn.num_batches_tracked, lazy_out)
    link = K.pop_bn_link()
    if link is not None:
        a._ctvae_bn_link = link     # lets the consumer's dgrad emit this BatchNorm's backward sums (kernels.BNLink)
    lazy = K.pop_lazy_bn()
    if lazy is not None:
        a._ctvae_lazy_bn = lazy     # the tensor holds the raw conv output: the consumer applies BatchNorm + LeakyReLU on load
    return a


class ConvBNLeaky(nn.Module):
    """nn.Sequential(Conv2d|ConvTranspose2d, BatchNorm2d, LeakyReLU) (vanilla_vae.py:25-35,47-58) as two launches
    groups: conv (+bias) and BN statistics/apply+LeakyReLU.  Children are named "0" and "1" like the reference."""

    def __init__(self, ci, co, k, stride, pad, out_pad=0, transposed=False):
        super().__init__()
        self.add_module("0", PackedConv(ci, co, k, transposed=transposed, bias=True))
        self.add_module("1", PackedBN(co))
        self.spec = K.ConvSpec(K.CONVT if transposed else K.CONV, ci, co, k, stride, pad, out_pad, K.ACT_NONE)

    def forward(self, x, lazy_out=False):
        return conv_bn_leaky(x, self._modules["0"], self._modules["1"], self.spec, self.training, lazy_out)

    def reads_lazy_input(self, x_shape):
        """Can this block read an NHWC input of this shape through the previous block's BatchNorm coefficients?"""
        B, H, W, _ = x_shape
        return K.lazy_bn_input_supported(self.spec, B, H, W)


class Chain(nn.Sequential):
    """nn.Sequential (same children names, same state_dict) whose intermediate tensors provably have ONE consumer -- the next
    child: each is marked for kernels.mark_sole_consumer, which lets a small layer's data gradient reach the BatchNorm below it
    as split-K slices instead of a tensor.  The LAST child's output leaves the chain: its consumers are the caller's business."""

    def forward(self, x, last_reader=None):
        """last_reader: the module that is the ONE consumer of the chain's output (the caller hands it nothing else); if it has
        ``reads_lazy_input`` the last block may leave its BatchNorm + activation to it as well."""
        mods = list(self)
        last = len(mods) - 1
        for i, m in enumerate(mods):
            lazy = False
            reader = mods[i + 1] if i != last else last_reader
            if isinstance(m, ConvBNLeaky) and hasattr(reader, "reads_lazy_input") and x.dim() == 4:
                # the reader is the tensor's one consumer: if it can apply this block's BatchNorm + LeakyReLU while it loads,
                # the activated tensor and its apply launch are skipped (kernels.ConvBNAct lazy_out)
                ho, wo = m.spec.out_hw(x.shape[1], x.shape[2])
                lazy = (x.is_cuda and K.bn_apply_is_separate(m.spec, x.shape[0], x.shape[1], x.shape[2])
                        and reader.reads_lazy_input((x.shape[0], ho, wo, m.spec.co)))
            x = m(x, lazy_out=True) if lazy else m(x)
            if i != last or last_reader is not None:
                K.mark_sole_consumer(x)
        return x


class ConvAct(nn.Module):
    """nn.Sequential(Conv2d|ConvTranspose2d, LeakyReLU|Tanh) (mcq_vae.py:168-173,176-180,187-191,203-209,221-237):
    one launch, bias + activation in the epilogue.  Child "0" holds the parameters."""

    def __init__(self, ci, co, k, stride, pad, act, transposed=False, out_pad=0):
        super().__init__()
        self.add_module("0", PackedConv(ci, co, k, transposed=transposed, bias=True))
        self.spec = K.ConvSpec(K.CONVT if transposed else K.CONV, ci, co, k, stride, pad, out_pad, act)

    def forward(self, x, aux=None):
        """aux: companion rows without autograd (kernels.adjacent_rows); returns (y, y_aux) then."""
        conv = self._modules["0"]
        if aux is not None:
            return K.ConvAct.apply(x, conv.weight, conv.bias, None, self.spec, aux)
        return K.ConvAct.apply(x, conv.weight, conv.bias, None, self.spec)


class _ResBlockParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.add_module("0", PackedConv(c, c, 3, bias=False))
        self.add_module("2", PackedConv(c, c, 1, bias=False))


class ResidualLayer(nn.Module):
    """x + Conv1x1(ReLU(Conv3x3(x))), both bias-free (vq_vae.py:57-70).  ReLU is fused into the 3x3 epilogue, the skip
    add into the 1x1 epilogue; ``post_act`` additionally fuses the LeakyReLU that follows the last block
    (mcq_vae.py:185,216) when the model asks for it."""

    def __init__(self, in_channels, out_channels, post_act=K.ACT_NONE):
        super().__init__()
        assert in_channels == out_channels
        self.resblock = _ResBlockParams(in_channels)
        c = in_channels
        self.spec3 = K.ConvSpec(K.CONV, c, c, 3, 1, 1, 0, K.ACT_RELU)
        self.spec1 = K.ConvSpec(K.CONV, c, c, 1, 1, 0, 0, post_act)

    def forward(self, x, aux=None):
        w3, w1 = self.resblock._modules["0"].weight, self.resblock._modules["2"].weight
        if aux is not None:
            return K.ResBlock.apply(x, w3, w1, self.spec3, self.spec1, aux)
        return K.ResBlock.apply(x, w3, w1, self.spec3, self.spec1)


class LeakyReLU(nn.Module):
    """Parameter-free placeholder that keeps the reference's Sequential indices; ``fused=True`` means the
    producer already applied it."""

    def __init__(self, fused=False):
        super().__init__()
        self.fused = fused

    def forward(self, x, aux=None):
        if aux is not None:
            return (x, aux) if self.fused else K.ActFn.apply(x, K.ACT_LRELU, aux)
        return x if self.fused else K.ActFn.apply(x, K.ACT_LRELU)


def run_with_companion(seq, x, aux):
    """seq(x) with autograd and seq(aux) without, layer by layer through the same launches where the layer can
    (ConvAct / ResidualLayer / LeakyReLU above take ``aux``); any other module runs twice."""
    for m in seq:
        if isinstance(m, nn.Sequential):
            x, aux = run_with_companion(m, x, aux)
        elif isinstance(m, (ConvAct, ResidualLayer, LeakyReLU)):
            x, aux = m(x, aux)
        else:
            x = m(x)
            with torch.no_grad():
                aux = m(aux)
    return x, aux
