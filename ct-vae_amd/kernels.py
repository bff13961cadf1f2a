"""``torch.autograd.Function`` wrappers around the C ABI (``native.py`` / ``include/ctvae_hip.h``).

Conventions
-----------
* Activations are plain contiguous NHWC tensors ``[B,H,W,C]`` inside this module; the model layer
  (``models/``) exposes them as logical-NCHW ``permute(0,3,1,2)`` views (= torch channels_last).
* Weights are parameters whose *memory* is the packed ``[kh*kw][Ci][Co]`` layout while their logical
  shape/strides are PyTorch's (``models/packing.py``); kernels receive ``param.data_ptr()``.
* Parameter gradients are written by the wgrad kernels straight into ``param.grad`` (same packed memory
  layout, accumulate semantics like autograd); ``backward`` therefore returns ``None`` for parameters.
  DDP hooks on parameters do not fire -- use ``ctvae_amd.ddp.GradBucketAllReduce``.
* No op here has a PyTorch/CPU fallback: a missing library or a CPU tensor raises.
"""
import math
import os
import weakref
from dataclasses import dataclass

import torch
from torch.autograd import Function

from . import native

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3
CONV, CONVT = 0, 1
CONV_FLAT = CONV | 0x100   # CTVAE_W_CI_TAP: nn.Linear over torch.flatten(NCHW [B,ci,k,k]) run as a k x k conv of the NHWC tensor
BN_MOMENTUM, BN_EPS = 0.1, 1e-5      # nn.BatchNorm2d defaults (vanilla_vae.py:30)


@dataclass(frozen=True)
class ConvSpec:
    kind: int          # CONV / CONVT
    ci: int
    co: int
    k: int
    stride: int = 1
    pad: int = 0
    out_pad: int = 0
    act: int = ACT_NONE

    def out_hw(self, h, w):
        if self.kind in (CONV, CONV_FLAT):
            return (h + 2 * self.pad - self.k) // self.stride + 1, (w + 2 * self.pad - self.k) // self.stride + 1
        return ((h - 1) * self.stride - 2 * self.pad + self.k + self.out_pad,
                (w - 1) * self.stride - 2 * self.pad + self.k + self.out_pad)


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("ctvae HIP kernels need device tensors (there is no CPU fallback on the product path)")


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def grad_target(p):
    """(tensor to write the gradient into, accumulate flag).  Allocates p.grad with p's memory layout if absent."""
    if p.grad is None:
        p.grad = torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=p.device)
        return p.grad, 0
    if p.grad.stride() != p.stride():
        raise RuntimeError("parameter .grad does not share the packed layout of the parameter")
    blk = getattr(p, "_grad_block", None)
    if blk is not None and blk.fresh:      # zero_grad(lazy=True) (models/packing.py): first writer of the block overwrites
        blk.fresh = False
        return p.grad, 0
    return p.grad, 1


LAZY_ZERO_GRAD = os.environ.get("CTVAE_NO_LAZY_ZERO", "0") != "1"     # diagnostic: zero_grad(lazy=True) fills like zero_grad()


# ---------------------------------------------------------------------------------------------------
# low level launch helpers (no autograd)
# ---------------------------------------------------------------------------------------------------
_wino_floats_cache = {}


def wino_filter_floats(spec: ConvSpec, B, H, W, ws_numel) -> int:
    """Size of the Winograd filter hand-over buffer of this layer (0: its forward / data gradient are not both Winograd)."""
    key = (spec.kind, B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad, ws_numel, native.winograd_enabled())
    n = _wino_floats_cache.get(key)
    if n is None:
        n = _wino_floats_cache[key] = int(native.load().ctvae_conv_wino_filter_floats(*key[:10], ws_numel * 4))
    return n


def conv_forward_raw(x, w, b, spec: ConvSpec, add=None, act=None, in_coef=None, in_act=ACT_NONE, wino_out=None, wino_ready=None):
    """in_coef [2][Ci]: read act_in(x*scale+shift) instead of x (lazy BatchNorm apply, image-side layers only).
    wino_out: buffer of wino_filter_floats() floats that receives the data gradient's Winograd filters.
    wino_ready: the forward's transformed filters, made ahead of time for the current weights (WinoFilterCache)."""
    B, H, W, _ = x.shape
    ho, wo = spec.out_hw(H, W)
    y = torch.empty((B, ho, wo, spec.co), dtype=torch.float32, device=x.device)
    ws = native.workspace(x.device)
    sc = in_coef.data_ptr() if in_coef is not None else None
    sh = in_coef.data_ptr() + 4 * spec.ci if in_coef is not None else None
    native.call("ctvae_conv_forward", spec.kind, x.data_ptr(), w.data_ptr(), native.ptr(b), native.ptr(add), y.data_ptr(),
                B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad, spec.act if act is None else act,
                sc, sh, in_act, native.ptr(wino_out), native.ptr(wino_ready), ws.data_ptr(), ws.numel() * 4)
    return y


# ---------------------------------------------------------------------------------------------------
# Winograd filters of a whole step in one launch
# ---------------------------------------------------------------------------------------------------
_param_epoch = [0]


def bump_param_epoch():
    """Parameters may have changed by means torch does not see (the Adam kernel writes through raw pointers), or a new step
    begins (zero_grad): transformed filters made before this are not trusted any more."""
    _param_epoch[0] += 1


class _WinoFilterCache:
    """Both transformed filter sets of every 3x3 stride-1 layer that runs Winograd, refreshed by ONE launch per training step
    (ctvae_wino_filters_batch) instead of a launch per layer: the first Winograd layer of a forward pass that finds its
    entry stale (another parameter epoch -- see bump_param_epoch -- or another tensor version) refreshes EVERY registered
    entry whose weight is alive.  Used only for forward passes that will be differentiated (training); anything else keeps the
    per-layer transform.  Entries die with their weight (weak references)."""

    def __init__(self):
        self.entries = {}          # id(weight) -> dict(ref, uf, ub, epoch, version, ci, co)

    def filters(self, w, spec, n_floats):
        """(forward filters, data-gradient filters) valid for w as it is now."""
        e = self.entries.get(id(w))
        if e is None or e["ref"]() is not w or e["uf"].numel() != n_floats or e["uf"].device != w.device:
            e = self.entries[id(w)] = dict(ref=weakref.ref(w), uf=torch.empty(n_floats, dtype=torch.float32, device=w.device),
                                           ub=torch.empty(n_floats, dtype=torch.float32, device=w.device), epoch=-1, version=-1,
                                           ci=spec.ci, co=spec.co)
        if e["epoch"] != _param_epoch[0] or e["version"] != w._version:
            self.refresh(w.device)
        return e["uf"], e["ub"]

    def refresh(self, device):
        import ctypes as C
        live = []
        for k in list(self.entries):
            e = self.entries[k]
            w = e["ref"]()
            if w is None:
                del self.entries[k]
            elif w.device == device and (e["epoch"] != _param_epoch[0] or e["version"] != w._version):
                live.append((e, w))
        if not live:
            return
        n = len(live)
        native.call("ctvae_wino_filters_batch", n, (C.c_void_p * n)(*[w.data_ptr() for _, w in live]),
                    (C.c_void_p * n)(*[e["uf"].data_ptr() for e, _ in live]), (C.c_void_p * n)(*[e["ub"].data_ptr() for e, _ in live]),
                    (C.c_int * n)(*[e["ci"] for e, _ in live]), (C.c_int * n)(*[e["co"] for e, _ in live]))
        for e, w in live:
            e["epoch"], e["version"] = _param_epoch[0], w._version


wino_cache = _WinoFilterCache()
_WINO_BATCH = os.environ.get("CTVAE_NO_WINO_BATCH", "0") != "1"     # diagnostic: one filter-transform launch per layer


def _wino_filters_for(ctx_needs_grad, w, spec, B, H, W_):
    """(ready forward filters or None, buffer for / holding the data-gradient filters or None) for a Winograd layer."""
    n = wino_filter_floats(spec, B, H, W_, native.workspace(w.device).numel()) if ctx_needs_grad else 0
    if not n:
        return None, None
    if _WINO_BATCH and spec.ci % 32 == 0 and spec.co % 32 == 0:
        uf, ub = wino_cache.filters(w, spec, n)
        return uf, ub
    return None, torch.empty(n, dtype=torch.float32, device=w.device)


def conv_dgrad_raw(dy, w, spec: ConvSpec, in_hw, add=None, mask=None, mask_act=ACT_NONE, wino_filters=None):
    B = dy.shape[0]
    H, W = in_hw
    dx = torch.empty((B, H, W, spec.ci), dtype=torch.float32, device=dy.device)
    ws = native.workspace(dy.device)
    if wino_filters is not None and wino_filter_floats(spec, B, H, W, ws.numel()) != wino_filters.numel():
        wino_filters = None        # the switch was flipped between forward and backward
    native.call("ctvae_conv_dgrad", spec.kind, dy.data_ptr(), w.data_ptr(), native.ptr(add), native.ptr(mask), mask_act,
                dx.data_ptr(), B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad,
                native.ptr(wino_filters), ws.data_ptr(), ws.numel() * 4)
    return dx


class BNLink:
    """Hand-over between a train-mode ConvBNAct layer and the layer that consumes its output ``a``.

    The consumer's data-gradient kernel produces g_a; given the BatchNorm's y/mean/invstd/gamma/beta it also emits the
    per-tile backward sums (ctvae_conv_dgrad_bn), which the BatchNorm's own backward then takes instead of running a
    separate pass over (g_a, y).  The sums are only used when the gradient tensor that arrives is exactly the one the
    dgrad wrote (same storage pointer, same version counter): if autograd summed several contributions, or anything
    modified it in place, the BatchNorm falls back to its own pass."""
    __slots__ = ("y", "mean", "invstd", "gamma", "beta", "act", "part", "rows", "coef", "g_ptr", "g_ver", "g_shape", "slices", "sole", "g_hold")

    def __init__(self, y, mean, invstd, gamma, beta, act):
        self.y, self.mean, self.invstd, self.gamma, self.beta, self.act = y, mean, invstd, gamma, beta, act
        self.sole = False       # set by model code (mark_sole_consumer): the layer's output has exactly one consumer
        self.g_hold = None      # the placeholder of publish_lazy, kept alive until take_lazy (its address is not recycled meanwhile)
        self.part = None
        self.rows = 0
        self.coef = None
        self.slices = None
        self.g_ptr = self.g_ver = self.g_shape = None

    def publish_lazy(self, g, slices, n, geom):
        """Small layers (ctvae_conv_backward_lazy): the consumer's data gradient ran split-K and left its n raw slices
        (channel-major, rows in that launch's own order: geom = its (kind, B, H, W, Ci, Co, k, stride, pad, out_pad)) in ``slices``; ``g`` is a placeholder that was never written -- the BatchNorm's backward sums the slices
        itself (ctvae_bn_backward_fused).  Nothing else may consume g: take_lazy raises if another tensor arrives."""
        self.slices = (slices, n, geom)
        self.g_hold = g
        self.part, self.rows, self.coef = None, 0, None
        self.g_ptr, self.g_ver, self.g_shape = g.data_ptr(), g._version, tuple(g.shape)

    def take_lazy(self, g):
        """(slices, n, geom) when the consumer left its data gradient as raw slices, else None.  One use only."""
        sl, self.slices = self.slices, None
        self.g_hold = None
        if sl is None:
            return None
        if g.data_ptr() != self.g_ptr or g._version != self.g_ver or tuple(g.shape) != self.g_shape:
            raise RuntimeError("BNLink: the consumer left its data gradient as split-K slices for this BatchNorm's backward pass "
                               "(sole consumer of the layer's output), but a different gradient tensor arrived")
        return sl

    def publish(self, g, part, rows, coef=None):
        """coef [7][C]: the consumer's finishing launch already ran this BatchNorm's backward finalize on the sums
        (ctvae_conv_backward bn_coef_out)."""
        self.part, self.rows, self.coef = part, rows, coef
        self.g_ptr, self.g_ver, self.g_shape = g.data_ptr(), g._version, tuple(g.shape)

    def take(self, g):
        """(part, rows, coef) when ``g`` is the tensor the sums were computed for, else (None, 0, None).  One use only."""
        part, rows, coef = self.part, self.rows, self.coef
        self.part, self.rows, self.coef = None, 0, None
        if part is None or g.data_ptr() != self.g_ptr or g._version != self.g_ver or tuple(g.shape) != self.g_shape:
            return None, 0, None
        return part, rows, coef


class ActLink:
    """Hand-over between a ConvAct layer with a fused activation and the SOLE consumer of its output h (declared by the
    model code, e.g. blocks.ResidualLayer where h never leaves forward()).  The consumer's data gradient multiplies by
    act'(h) in its epilogue (the ``mask`` operand of ctvae_conv_dgrad), so the tensor that arrives at the producer's
    backward is already the gradient w.r.t. its pre-activation and the separate activation-backward pass is skipped.
    The producer checks that it received exactly that tensor; anything else cannot be repaired and raises."""
    __slots__ = ("act", "g_ptr", "g_ver", "g_shape", "done", "out_ref", "out_ver")

    def __init__(self, act):
        self.act = act
        self.done = False
        self.g_ptr = self.g_ver = self.g_shape = None
        self.out_ref = self.out_ver = None

    def publish_done(self, g):
        self.done = True
        self.g_ptr, self.g_ver, self.g_shape = g.data_ptr(), g._version, tuple(g.shape)

    def take(self, g):
        """True when ``g`` already carries the activation derivative.  One use only."""
        done, self.done = self.done, False
        if not done:
            return False
        if g.data_ptr() != self.g_ptr or g._version != self.g_ver or tuple(g.shape) != self.g_shape:
            raise RuntimeError("ActLink: the activation backward was folded into the consumer's data gradient (sole "
                               "consumer promised), but a different gradient tensor arrived")
        return True


# Output-activation links OFFERED by a producer instead of promised by model code: the producer registers the tensor it hands
# out (final_layer's Tanh output r), and the one consumer that knows how to fold act'(r) into its own backward pass -- the
# reconstruction loss, which reads r anyway -- claims the link in its forward.  If nobody claims it, or the loss is not
# asked for that gradient, the producer runs its activation backward as usual; if the loss folded it but a different
# gradient tensor arrives (r had a second consumer), ActLink.take raises.
_out_act_links = {}


def offer_out_act_link(out, act):
    link = ActLink(act)
    link.out_ref, link.out_ver = weakref.ref(out), out._version
    if len(_out_act_links) > 8:
        for k in [k for k, l in _out_act_links.items() if l.out_ref() is None]:
            del _out_act_links[k]
    _out_act_links[out.data_ptr()] = link
    return link


def claim_out_act_link(t):
    """The link offered for exactly this tensor (same storage, shape and version, producer's tensor still alive), else None."""
    link = _out_act_links.pop(t.data_ptr(), None)
    if link is None:
        return None
    o = link.out_ref()
    if o is None or o.data_ptr() != t.data_ptr() or tuple(o.shape) != tuple(t.shape) or o._version != link.out_ver:
        return None
    return link


# Gradients handed on as raw split-K slices (pixel-major) to an element-wise consumer that sums them while it reads: the producer
# (conv_backward_raw, asked by a layer whose input tensor carries ``_ctvae_grad_slices_ok``) registers the placeholder it returns,
# the consumer's backward (GaussianLatent, _ToNHWC) claims it.  The tag is a promise of model code that the tensor's gradient goes
# to that consumer and nowhere else: the placeholder is never written.
_lazy_grads = {}


def offer_lazy_grad(g, slices, n):
    if len(_lazy_grads) > 16:
        _lazy_grads.clear()
    # the entry keeps the placeholder alive: its address cannot be handed to another tensor while the entry exists, so a match
    # on (address, version, size) below IS the placeholder (or a view of it), never a newcomer at a recycled address
    _lazy_grads[g.data_ptr()] = (slices, n, g._version, g.numel(), g)


def claim_lazy_grad(g):
    """(slices, n) if ``g`` is a registered, untouched placeholder, else None."""
    e = _lazy_grads.pop(g.data_ptr(), None)
    if e is None or e[2] != g._version or e[3] != g.numel():
        return None
    return e[0], e[1]


def grad_slices_ok(t):
    """Model code: the gradient w.r.t. ``t`` is consumed by ONE backward that can sum split-K slices (see _lazy_grads)."""
    t._ctvae_grad_slices_ok = True
    return t


_last_act_link = None  # set by ConvAct.forward (fused activation), picked up by the caller of .apply right after


def pop_act_link():
    global _last_act_link
    link, _last_act_link = _last_act_link, None
    return link


_last_link = None     # set by ConvBNAct.forward, picked up by the caller of .apply (models/blocks.py) right after
_last_fwd_slices = None   # set by ConvAct.forward(lazy_slices=True): (slices, n, bias) of the placeholder it returned


def pop_fwd_slices():
    global _last_fwd_slices
    fs, _last_fwd_slices = _last_fwd_slices, None
    return fs


_last_lazy = None     # set by ConvBNAct.forward(lazy_out=True): (scale | shift [2][C], activation) of the tensor it returned


def pop_lazy_bn():
    global _last_lazy
    lz, _last_lazy = _last_lazy, None
    return lz


_apply_sep_cache = {}


def bn_apply_is_separate(spec, B, H, W) -> bool:
    """Does a training ConvBNAct of this geometry run its BatchNorm apply as a launch of its own (ctvae_conv_bn_act_apply_is_separate)?
    Only then is it worth handing the consumer the raw tensor: applying on load costs the consumer's tile kernels ~4 vector
    instructions per element next to their f32 MFMAs (+10-16 % per launch, measured)."""
    ws = native.workspace(torch.device("cuda", torch.cuda.current_device()))
    key = (spec.kind, B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad, ws.numel())
    v = _apply_sep_cache.get(key)
    if v is None:
        v = _apply_sep_cache[key] = bool(native.load().ctvae_conv_bn_act_apply_is_separate(*key[:10], ws.numel() * 4))
    return v


def lazy_bn_input_supported(spec, B, H, W) -> bool:
    """Can a train-mode ConvBNAct layer of this geometry read its input through the previous block's BatchNorm + activation
    (forward tile kernel and weight-gradient kernel both transform on load: ctvae_conv_input_transform_supported)?"""
    return _LAZY_BN and input_transform_supported(spec, B, H, W)      # (the tile kernels form max(t, slope*t): LeakyReLU / ReLU / none)


def pop_bn_link():
    global _last_link
    link, _last_link = _last_link, None
    return link


def mark_sole_consumer(x):
    """Model code promises that x -- the output of a train-mode ConvBNAct block -- is read by exactly ONE consumer (the next
    block of a chain: blocks.Chain, or the one layer the model hands it to).  That consumer may then leave its data gradient as
    split-K slices for the BatchNorm's backward launch to sum (BNLink.publish_lazy): the gradient tensor in between is never
    written, so a second consumer would add garbage -- BNLink.take_lazy raises if anything but the placeholder arrives."""
    link = getattr(x, "_ctvae_bn_link", None)
    if link is not None:
        link.sole = True
    return x


def link_of(x):
    """BNLink of a tensor that is the untouched, contiguous output of a train-mode ConvBNAct (else None)."""
    link = getattr(x, "_ctvae_bn_link", None)
    return link if (link is not None and x.is_contiguous()) else None


_bn_rows_cache = {}


def conv_dgrad_bn_raw(dy, w, spec: ConvSpec, in_hw, link: BNLink):
    """dgrad + fused BatchNorm-backward sums of the layer that produced this layer's input; None if not fusable."""
    B = dy.shape[0]
    H, W = in_hw
    ws = native.workspace(dy.device)
    key = (spec.kind, B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad, ws.numel())
    rows = _bn_rows_cache.get(key)
    if rows is None:
        rows = _bn_rows_cache[key] = native.load().ctvae_conv_dgrad_bn_rows(*key[:-1], ws.numel() * 4)
    if rows <= 0 or tuple(link.y.shape) != (B, H, W, spec.ci):
        return None
    dx = torch.empty((B, H, W, spec.ci), dtype=torch.float32, device=dy.device)
    part = torch.empty(rows * spec.ci * 2, dtype=torch.float32, device=dy.device)
    native.call("ctvae_conv_dgrad_bn", spec.kind, dy.data_ptr(), w.data_ptr(), None, None, ACT_NONE, dx.data_ptr(),
                B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad, link.y.data_ptr(),
                link.mean.data_ptr(), link.invstd.data_ptr(), link.gamma.data_ptr(), link.beta.data_ptr(), link.act,
                part.data_ptr(), rows, ws.data_ptr(), ws.numel() * 4)
    link.publish(dx, part, rows)
    return dx


def conv_wgrad_raw(x, dy, w_param, b_param, spec: ConvSpec, in_coef=None, in_act=ACT_NONE, dy_bn=None, bn_commit=None):
    """dy_bn = (y, coef[5][Co], act, gy_out): dy is g_a of the BatchNorm behind this layer; g_y is formed on load and
    written to gy_out (ctvae_conv_wgrad dy_bn_*).  gy_out None + bn_commit = (dgamma, dbeta, accumulate): the layer with no
    data gradient (encoder.0) -- g_y is never written and the kernel commits the BatchNorm's parameter gradients."""
    B, H, W, _ = x.shape
    ws = native.workspace(x.device)
    gw, acc = grad_target(w_param)
    gb = None
    if b_param is not None:
        gb, accb = grad_target(b_param)
        if accb != acc:   # keep one accumulate flag per call: materialise the fresh one as zeros
            if acc == 0:
                gw.zero_()
            else:
                gb.zero_()
            acc = 1
    sc = in_coef.data_ptr() if in_coef is not None else None
    sh = in_coef.data_ptr() + 4 * spec.ci if in_coef is not None else None
    by, bc, bact, bgy = (dy_bn[0].data_ptr(), dy_bn[1].data_ptr(), dy_bn[2], native.ptr(dy_bn[3])) if dy_bn is not None else (None, None, 0, None)
    cg, cb, cacc = (bn_commit[0].data_ptr(), bn_commit[1].data_ptr(), bn_commit[2]) if bn_commit is not None else (None, None, 0)
    native.call("ctvae_conv_wgrad", spec.kind, x.data_ptr(), dy.data_ptr(), gw.data_ptr(), native.ptr(gb),
                B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad, acc, sc, sh, in_act,
                by, bc, bact, bgy, cg, cb, cacc, ws.data_ptr(), ws.numel() * 4)


_PAIR = os.environ.get("CTVAE_NO_PAIR", "0") != "1"     # diagnostic: separate wgrad / dgrad launches
_BN_RIDER = os.environ.get("CTVAE_NO_BN_RIDER", "0") != "1"   # diagnostic: BatchNorm-backward finalize as its own launch
_OUT_ACT_LINK = os.environ.get("CTVAE_NO_OUT_ACT_LINK", "0") != "1"   # diagnostic: final Tanh backward as its own launch
# diagnostic, default off (0 = the finalize always rides in the consumer's finishing launch): with 128 the three deepest BatchNorm
# layers finalize in their own apply launch and their consumers' slab reductions leave the chain -- 1.5992 / 1.5986 ms against
# 1.5996 / 1.5988 ms (VanillaVAE bs = 256, same box): neutral, the one deferred launch grows by what the three removed ones took
_BN_BWD_MERGE_ROWS = int(os.environ.get("CTVAE_BN_BWD_MERGE_ROWS", "0"))
_GRAD_SLICES = os.environ.get("CTVAE_NO_GRAD_SLICES", "0") != "1"   # diagnostic: every split-K result is summed by its own finish launch
_LAZY_BN = os.environ.get("CTVAE_NO_LAZY_BN_APPLY", "0") != "1"   # diagnostic: every BatchNorm + activation output is materialised
_BN_LAZY = os.environ.get("CTVAE_NO_BN_LAZY", "0") != "1"     # diagnostic: small layers' data gradients summed by splitk_finish as before
_ENC_BN_ON_LOAD = os.environ.get("CTVAE_NO_ENC_BN_ON_LOAD", "0") != "1"   # diagnostic: encoder.0's BatchNorm-backward apply as its own launch


def conv_backward_raw(x, dy, w_param, b_param, spec: ConvSpec, link=None, mask=None, mask_act=ACT_NONE, wino_filters=None,
                      in_coef=None, in_act=ACT_NONE, dy_bn=None, bn_commit=None, grad_slices=False):
    """ctvae_conv_backward: weight (+bias) gradient into ``.grad`` and the data gradient of one layer in one call (the
    two GEMMs share a launch when both run their 64x64 tile kernels).  ``link``: BNLink of the layer that produced x --
    its BatchNorm-backward sums come out of the data gradient's epilogues and its finalize rides in this call's finishing
    launch.  in_coef / in_act / dy_bn: the weight gradient's options of conv_wgrad_raw.  bn_commit = (dgamma, dbeta,
    accumulate): the rider also commits that BatchNorm's parameter gradients (only where no apply launch follows)."""
    B, H, W, _ = x.shape
    ws = native.workspace(x.device)
    if wino_filters is not None and wino_filter_floats(spec, B, H, W, ws.numel() // 2) != wino_filters.numel():
        # the forward pass ran Winograd over more rows than this backward pass sees (companion rows: x and y through one launch,
        # only x differentiated), or the switch was flipped in between: the data gradient picks its own kernel and filters
        wino_filters = None
    gw, acc = grad_target(w_param)
    gb = None
    if b_param is not None:
        gb, accb = grad_target(b_param)
        if accb != acc:
            (gw if acc == 0 else gb).zero_()
            acc = 1
    dx = torch.empty((B, H, W, spec.ci), dtype=torch.float32, device=dy.device)
    part, rows = None, 0
    if (link is not None and link.sole and _BN_LAZY and tuple(link.y.shape) == (B, H, W, spec.ci) and mask is None and wino_filters is None
            and dy_bn is None and bn_commit is None):
        key = (spec.kind, B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad, ws.numel(), "lazy")
        n = _bn_rows_cache.get(key)
        if n is None:
            n = _bn_rows_cache[key] = native.load().ctvae_conv_backward_lazy_slices(*key[:-2], 1, ws.numel() * 4)
        if n > 1:
            # small layer: the data gradient stays n raw split-K slices, summed by the BatchNorm's own backward launch
            slices = torch.empty(n * B * H * W * spec.ci, dtype=torch.float32, device=dy.device)
            native.call("ctvae_conv_backward_lazy", spec.kind, x.data_ptr(), dy.data_ptr(), w_param.data_ptr(), gw.data_ptr(),
                        native.ptr(gb), slices.data_ptr(), B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad,
                        acc, in_coef.data_ptr() if in_coef is not None else None,
                        in_coef.data_ptr() + 4 * spec.ci if in_coef is not None else None, in_act, 0, ws.data_ptr(), ws.numel() * 4)
            link.publish_lazy(dx, slices, n, key[:10])
            return dx
    if (grad_slices and _GRAD_SLICES and link is None and mask is None and wino_filters is None and dy_bn is None
            and bn_commit is None):
        key = (spec.kind, B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad, ws.numel(), "slices")
        n = _bn_rows_cache.get(key)
        if n is None:
            n = _bn_rows_cache[key] = native.load().ctvae_conv_backward_lazy_slices(*key[:-2], 0, ws.numel() * 4)
        if n > 1:
            # the consumer of this gradient sums split-K slices itself (offer_lazy_grad): no finish launch, dx is a placeholder
            slices = torch.empty(n * B * H * W * spec.ci, dtype=torch.float32, device=dy.device)
            native.call("ctvae_conv_backward_lazy", spec.kind, x.data_ptr(), dy.data_ptr(), w_param.data_ptr(), gw.data_ptr(),
                        native.ptr(gb), slices.data_ptr(), B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad,
                        acc, in_coef.data_ptr() if in_coef is not None else None,
                        in_coef.data_ptr() + 4 * spec.ci if in_coef is not None else None, in_act, 1, ws.data_ptr(), ws.numel() * 4)
            offer_lazy_grad(dx, slices, n)
            return dx
    if link is not None and tuple(link.y.shape) == (B, H, W, spec.ci):
        key = (spec.kind, B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad, -ws.numel())
        rows = _bn_rows_cache.get(key)
        if rows is None:
            rows = _bn_rows_cache[key] = native.load().ctvae_conv_backward_bn_rows(*key[:-1], ws.numel() * 4)
        if rows > 0:
            part = torch.empty(rows * spec.ci * 2, dtype=torch.float32, device=dy.device)
        else:
            rows = 0
    bn = link if part is not None else None
    # few rows of sums (the deep layers): the BatchNorm's own launch finalizes AND applies (bn_bwd_finalize_apply_kernel), so this
    # call's finishing launch carries no finalize and its slab reduction can leave the chain (kernels.backward's deferral)
    rider = bn is not None and _BN_RIDER and not (0 < rows <= _BN_BWD_MERGE_ROWS and spec.ci % 32 == 0 and bn_commit is None)
    coef = torch.empty(7 * spec.ci, dtype=torch.float32, device=dy.device) if rider else None
    sc = in_coef.data_ptr() if in_coef is not None else None
    sh = in_coef.data_ptr() + 4 * spec.ci if in_coef is not None else None
    by, bc, bact, bgy = (dy_bn[0].data_ptr(), dy_bn[1].data_ptr(), dy_bn[2], dy_bn[3].data_ptr()) if dy_bn is not None else (None, None, 0, None)
    native.call("ctvae_conv_backward", spec.kind, x.data_ptr(), dy.data_ptr(), w_param.data_ptr(), gw.data_ptr(), native.ptr(gb),
                dx.data_ptr(), B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad, acc,
                native.ptr(mask), mask_act, native.ptr(wino_filters),
                native.ptr(bn.y if bn else None), native.ptr(bn.mean if bn else None), native.ptr(bn.invstd if bn else None),
                native.ptr(bn.gamma if bn else None), native.ptr(bn.beta if bn else None), bn.act if bn else 0,
                native.ptr(part), rows, native.ptr(coef),
                native.ptr(bn_commit[0]) if (bn_commit and coef is not None) else None,
                native.ptr(bn_commit[1]) if (bn_commit and coef is not None) else None, bn_commit[2] if bn_commit else 0,
                sc, sh, in_act, by, bc, bact, bgy, ws.data_ptr(), ws.numel() * 4)
    if bn is not None:
        bn.publish(dx, part, rows, coef)
    return dx


def wgrad_then_dgrad(x, g, w_param, b_param, spec, need_dgrad, link=None, wino_filters=None, in_coef=None, in_act=ACT_NONE,
                     grad_slices=False):
    """Weight gradient (accumulated straight into ``.grad``) and data gradient of one layer, on the launch stream.
    Measured on MI355X: putting the wgrad kernels on a second HIP stream (joined right after dgrad, or once at the
    end of backward) is SLOWER than back-to-back launches (2.43 vs 2.32 ms/step) -- each GEMM launch already covers
    every CU, and the fork/join edges cost more than the overlap of prologue/epilogue phases returns.  What does pay is
    ONE launch for both GEMMs (ctvae_conv_backward / conv_bwd_pair_kernel)."""
    if need_dgrad and _PAIR:
        return conv_backward_raw(x, g, w_param, b_param, spec, link=link, wino_filters=wino_filters, in_coef=in_coef, in_act=in_act,
                                 grad_slices=grad_slices)
    conv_wgrad_raw(x, g, w_param, b_param, spec, in_coef=in_coef, in_act=in_act)
    if not need_dgrad:
        return None
    if link is not None:
        dx = conv_dgrad_bn_raw(g, w_param, spec, (x.shape[1], x.shape[2]), link)
        if dx is not None:
            return dx
    return conv_dgrad_raw(g, w_param, spec, (x.shape[1], x.shape[2]), wino_filters=wino_filters)


def act_backward_raw(g_out, out, act):
    if act == ACT_NONE:
        return g_out
    g_in = torch.empty_like(out)
    native.call("ctvae_act_backward", g_out.data_ptr(), out.data_ptr(), g_in.data_ptr(), out.numel(), act)
    return g_in


def permute_raw(x, B, C, P, to_nhwc):
    out = torch.empty(x.numel(), dtype=torch.float32, device=x.device)
    native.call("ctvae_permute", x.data_ptr(), out.data_ptr(), B, C, P, 1 if to_nhwc else 0)
    return out


# ---------------------------------------------------------------------------------------------------
# layout boundary
# ---------------------------------------------------------------------------------------------------
class _ToNHWC(Function):
    """Logical NCHW contiguous -> NHWC contiguous (HIP permute)."""

    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        ctx.dims = (B, C, H, W)
        return permute_raw(_c(x), B, C, H * W, True).view(B, H, W, C)

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.dims
        lazy = claim_lazy_grad(g) if g.is_contiguous() else None
        if lazy is not None:
            # the gradient arrives as split-K slices of the consumer's data gradient: summed while the layout changes
            out = torch.empty((B, C, H, W), dtype=torch.float32, device=g.device)
            native.call("ctvae_splitk_permute", lazy[0].data_ptr(), lazy[1], out.data_ptr(), B, C, H * W)
            return out
        return permute_raw(_c(g), B, C, H * W, False).view(B, C, H, W)


class LinearToNHWC(Function):
    """``decoder_input(z).view(-1, C, h, w)`` (vanilla_vae.py:101-102) handed to the decoder as the NHWC tensor [B, h, w, C] by ONE
    launch: the Linear's GEMM stores output feature c*P + p at column p*C + c (ctvae_linear_pixmajor_forward) instead of a layout
    launch behind it.  Backward is what ConvAct + _ToNHWC did: the gradient goes back to the Linear's own feature order (summing
    the consumer's split-K slices on the way when it arrives as slices), then the Linear's paired backward launch."""

    _ok = {}
    _on = os.environ.get("CTVAE_LINEAR_NHWC", "1") != "0"     # diagnostic: 0 = the two-launch path

    @staticmethod
    def supported(B, ci, C, P, device):
        if not LinearToNHWC._on:
            return False
        key = (B, ci, C, P)
        ok = LinearToNHWC._ok.get(key)
        if ok is None:
            ws = native.workspace(device)
            ok = LinearToNHWC._ok[key] = bool(native.load().ctvae_linear_pixmajor_supported(B, ci, C, P, ws.numel() * 4))
        return ok

    @staticmethod
    def forward(ctx, x, w, b, spec, C, h, wd):
        _req_cuda(x, w)
        x = _c(x)
        B = x.shape[0]
        ctx.link_in = link_of(x)
        ctx.x_slices_ok = bool(getattr(x, "_ctvae_grad_slices_ok", False))
        ctx.spec, ctx.w, ctx.b, ctx.dims = spec, w, b, (B, C, h, wd)
        ws = native.workspace(x.device)
        y = torch.empty((B, h, wd, C), dtype=torch.float32, device=x.device)
        native.call("ctvae_linear_pixmajor_forward", x.data_ptr(), w.data_ptr(), native.ptr(b), y.data_ptr(), B, spec.ci, C, h * wd,
                    spec.act, ws.data_ptr(), ws.numel() * 4)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 7
        (x,) = ctx.saved_tensors
        B, C, h, wd = ctx.dims
        spec = ctx.spec
        if spec.act != ACT_NONE:
            raise RuntimeError("LinearToNHWC: a fused activation is not differentiated here (decoder_input has none)")
        lazy = claim_lazy_grad(g) if g.is_contiguous() else None
        if lazy is not None:       # split-K slices of the consumer's data gradient: summed while the layout changes
            g_flat = torch.empty((B, 1, 1, C * h * wd), dtype=torch.float32, device=g.device)
            native.call("ctvae_splitk_permute", lazy[0].data_ptr(), lazy[1], g_flat.data_ptr(), B, C, h * wd)
        else:
            g_flat = permute_raw(_c(g), B, C, h * wd, False).view(B, 1, 1, C * h * wd)
        g_x = wgrad_then_dgrad(x, g_flat, ctx.w, ctx.b, spec, ctx.needs_input_grad[0], ctx.link_in, None, grad_slices=ctx.x_slices_ok)
        return g_x, None, None, None, None, None, None


class _ToNCHW(Function):
    """NHWC contiguous -> NCHW contiguous."""

    @staticmethod
    def forward(ctx, x):
        B, H, W, C = x.shape
        ctx.dims = (B, C, H, W)
        return permute_raw(_c(x), B, C, H * W, False).view(B, C, H, W)

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.dims
        return permute_raw(_c(g), B, C, H * W, True).view(B, H, W, C)


def to_nhwc(x_nchw):
    """Accept a logical [B,C,H,W] tensor, return the NHWC-contiguous [B,H,W,C] tensor (zero-copy for channels_last)."""
    _req_cuda(x_nchw)
    if x_nchw.dtype != torch.float32:
        raise RuntimeError("ctvae kernels are fp32 (parity target 1e-4, SURVEY.md §7)")
    v = x_nchw.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v
    if x_nchw.is_contiguous():
        return _ToNHWC.apply(x_nchw)
    return v.contiguous()


def to_nchw_view(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2)


def staging_like(batch):
    """Static input buffer for hipGraph replay of steps on batches like ``batch``: an image batch ([B,C,H,W] fp32 on the
    device) gets channels_last memory -- the layout the encoder reads -- so that the per-step hand-over ``stage_batch`` IS the
    NCHW -> NHWC conversion instead of a copy followed by one."""
    if batch.is_cuda and batch.dim() == 4 and batch.dtype == torch.float32:
        return torch.empty_like(batch, memory_format=torch.channels_last)
    return torch.empty_like(batch)


def stage_batch(dst, src):
    """dst <- src for a buffer from ``staging_like``: one ctvae_permute launch for an NCHW-contiguous source and a
    channels_last buffer (what a DataLoader hands over / what the step reads), a plain copy otherwise."""
    if (src.is_cuda and src.dim() == 4 and src.dtype == torch.float32 and src.shape == dst.shape and src.is_contiguous()
            and dst.permute(0, 2, 3, 1).is_contiguous() and not dst.is_contiguous()):
        B, C, H, W = src.shape
        native.call("ctvae_permute", src.data_ptr(), dst.data_ptr(), B, C, H * W, 1)
        torch.autograd.graph.increment_version(dst)     # written through a raw pointer: caches keyed on (tensor, version) must see it
    else:
        dst.copy_(src, non_blocking=True)
    return dst


# ---------------------------------------------------------------------------------------------------
# Companion rows: a second batch that rides through the same launches without taking part in autograd
# ---------------------------------------------------------------------------------------------------
# CT-MCQ-VAE encodes two images per sample (ct_mcq_vae.py:536-539, 565-566): x, whose encoder pass is trained, and y, of which
# only the code indices are used (no backward).  The encoder has no BatchNorm, so its layers treat rows independently: the
# layer Functions below take y as ``aux`` and, when aux directly follows x in memory, run ONE launch over both (twice the
# rows per launch instead of twice the launches: Winograd / tile kernels at 128 images leave half the chip idle).  Only x is
# an autograd input; the aux output is marked non-differentiable and the saved tensors are the x rows.
def adjacent_rows(x, aux):
    """The [Bx+Ba, ...] tensor that x and aux form when aux starts where x ends in the same allocation, else None."""
    if (aux is None or not x.is_contiguous() or not aux.is_contiguous() or x.shape[1:] != aux.shape[1:] or x.dtype != aux.dtype
            or x.untyped_storage().data_ptr() != aux.untyped_storage().data_ptr()
            or x.data_ptr() + x.numel() * x.element_size() != aux.data_ptr()):
        return None
    return torch.as_strided(x.detach(), (x.shape[0] + aux.shape[0],) + tuple(x.shape[1:]), x.stride(), x.storage_offset())


def to_nhwc_pair(x_nchw, y_nchw):
    """NHWC tensors of two same-shaped logical [B,C,H,W] batches in ONE buffer (x rows first), one launch each."""
    _req_cuda(x_nchw, y_nchw)
    if x_nchw.shape != y_nchw.shape or x_nchw.dtype != torch.float32 or y_nchw.dtype != torch.float32 or x_nchw.requires_grad:
        return to_nhwc(x_nchw), to_nhwc(y_nchw)
    B, C, H, W = x_nchw.shape
    buf = torch.empty((2 * B, H, W, C), dtype=torch.float32, device=x_nchw.device)
    for half, src in ((buf[:B], x_nchw), (buf[B:], y_nchw)):
        if src.is_contiguous():
            native.call("ctvae_permute", src.data_ptr(), half.data_ptr(), B, C, H * W, 1)
        else:
            half.copy_(src.permute(0, 2, 3, 1))
    return buf[:B], buf[B:]


def _with_aux(x, aux, fn):
    """fn(rows) for x and aux: one call on the combined rows when they are adjacent, else one call each."""
    both = adjacent_rows(x, aux)
    if both is not None:
        out = fn(both)
        outs = out if isinstance(out, tuple) else (out,)
        B = x.shape[0]
        return tuple(o[:B] for o in outs), tuple(o[B:] for o in outs)
    ox, oa = fn(x), fn(aux)
    return (ox if isinstance(ox, tuple) else (ox,)), (oa if isinstance(oa, tuple) else (oa,))


# ---------------------------------------------------------------------------------------------------
# conv / linear (+ bias + residual + activation)
# ---------------------------------------------------------------------------------------------------
_flat_specs = {}


def flatten_linear(h, w, b, out_features, lazy_slices=False):
    """nn.Linear(C*k*k, out_features) applied to torch.flatten(h_nchw, start_dim=1), for the NHWC tensor h [B,k,k,C] -> [B, out]
    (vanilla_vae.py:87-91 and every model built on that encoder).  k = 2 with 32-aligned widths: a 2x2 stride-2 convolution
    straight on the NHWC tensor over the Linear layer's own [in][out] block (CONV_FLAT) -- no layout copies, and the data
    gradient comes back NHWC with the BatchNorm-backward sums of the layer below in its epilogue.  Anything else: the NCHW copy
    and a 1x1 GEMM."""
    B, k, k2, C = h.shape
    key = (k, k2, C, out_features)
    spec = _flat_specs.get(key)
    if spec is None:
        if k == 2 and k2 == 2 and C % 32 == 0 and out_features % 32 == 0:
            spec = ConvSpec(CONV_FLAT, C, out_features, 2, 2, 0)
        else:
            spec = ConvSpec(CONV, C * k * k2, out_features, 1)
        _flat_specs[key] = spec
    if spec.kind == CONV_FLAT:
        if lazy_slices:
            # the caller hands the result to GaussianLatent only, which sums the GEMM's split-K slices itself
            y = ConvAct.apply(h, w, b, None, spec, None, True)
            fs = pop_fwd_slices()
            out = y.view(B, -1)
            if fs is not None:
                out._ctvae_fwd_slices = fs
            return out
        return ConvAct.apply(h, w, b, None, spec).view(B, -1)
    flat = _ToNCHW.apply(h).view(B, 1, 1, -1)
    return ConvAct.apply(flat, w, b, None, spec).view(B, -1)


class ConvAct(Function):
    """y = act(conv(x, w) + b + add).  Conv2d/ConvTranspose2d/Linear forward, dgrad and wgrad on HIP."""

    @staticmethod
    def forward(ctx, x, w, b, add, spec, aux=None, lazy_slices=False):
        """aux: companion rows (see above); returns (y, y_aux) then.  lazy_slices (model code: the ONE consumer of y sums split-K
        slices itself, e.g. GaussianLatent): where the launch splits K, y is an unwritten placeholder and the raw slices (no bias)
        wait in kernels.pop_fwd_slices() for the caller to hang on it."""
        global _last_act_link, _last_fwd_slices
        _req_cuda(x, w)
        ctx.link_in = link_of(x)
        ctx.x_slices_ok = bool(getattr(x, "_ctvae_grad_slices_ok", False))
        if lazy_slices and _GRAD_SLICES and aux is None and add is None and spec.act == ACT_NONE and x.is_contiguous():
            B, H, W_, _ = x.shape
            ws = native.workspace(x.device)
            key = (spec.kind, B, H, W_, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad, ws.numel(), "fwd-slices")
            n = _bn_rows_cache.get(key)
            if n is None:
                n = _bn_rows_cache[key] = native.load().ctvae_conv_forward_lazy_slices(*key[:10], ws.numel() * 4)
            if n > 1:
                ho, wo = spec.out_hw(H, W_)
                y = torch.empty((B, ho, wo, spec.co), dtype=torch.float32, device=x.device)
                slices = torch.empty(n * y.numel(), dtype=torch.float32, device=x.device)
                native.call("ctvae_conv_forward_lazy", spec.kind, x.data_ptr(), w.data_ptr(), slices.data_ptr(), B, H, W_, spec.ci,
                            spec.co, spec.k, spec.stride, spec.pad, spec.out_pad, ws.data_ptr(), ws.numel() * 4)
                ctx.act_in = getattr(x, "_ctvae_act_link", None)
                ctx.act_out = None
                ctx.wino_u = None
                ctx.spec, ctx.w, ctx.b, ctx.has_add = spec, w, b, False
                ctx.save_for_backward(x, None)
                _last_fwd_slices = (slices, n, b)
                return y
        ctx.act_in = getattr(x, "_ctvae_act_link", None) if x.is_contiguous() else None
        ctx.act_out = _last_act_link = ActLink(spec.act) if spec.act != ACT_NONE else None
        x = _c(x)
        add_c = _c(add) if add is not None else None
        ctx.wino_u = None
        ctx.spec = spec
        ctx.w, ctx.b = w, b
        ctx.has_add = add is not None
        if aux is not None:
            if add is not None:
                raise RuntimeError("ConvAct: companion rows and a residual operand together are not supported")
            rows = adjacent_rows(x, _c(aux))
            Bt = rows.shape[0] if rows is not None else x.shape[0]
            ready, ctx.wino_u = _wino_filters_for(ctx.needs_input_grad[0], w, spec, Bt, x.shape[1], x.shape[2])
            first = [True]

            def run(t):
                wo, first[0] = (ctx.wino_u if (first[0] and ready is None) else None), False
                return conv_forward_raw(t, w, b, spec, None, wino_out=wo, wino_ready=ready)
            (y,), (y_aux,) = _with_aux(x, _c(aux), run)
            ctx.save_for_backward(x, y if spec.act != ACT_NONE else None)
            ctx.mark_non_differentiable(y_aux)
            ctx.set_materialize_grads(False)        # else backward is handed a zero-filled tensor for y_aux: a fill per layer
            return y, y_aux
        ready = None
        if add is None:
            ready, ctx.wino_u = _wino_filters_for(ctx.needs_input_grad[0], w, spec, x.shape[0], x.shape[1], x.shape[2])
        y = conv_forward_raw(x, w, b, spec, add_c, wino_out=None if ready is not None else ctx.wino_u, wino_ready=ready)
        ctx.save_for_backward(x, y if spec.act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, g_y, *_g_aux):
        if g_y is None:
            return (None,) * 7
        spec = ctx.spec
        x, y = ctx.saved_tensors
        g_y = _c(g_y)
        if spec.act == ACT_NONE or (ctx.act_out is not None and ctx.act_out.take(g_y)):
            g_pre = g_y                                   # no activation, or its derivative is already in g_y
        else:
            g_pre = act_backward_raw(g_y, y, spec.act)
        if ctx.act_in is not None and ctx.needs_input_grad[0] and ctx.link_in is None:
            # x is the activated output of the producer and this layer is its only consumer: dgrad * act'(x) in one launch
            if _PAIR:
                g_x = conv_backward_raw(x, g_pre, ctx.w, ctx.b, spec, mask=x, mask_act=ctx.act_in.act)
            else:
                conv_wgrad_raw(x, g_pre, ctx.w, ctx.b, spec)
                g_x = conv_dgrad_raw(g_pre, ctx.w, spec, (x.shape[1], x.shape[2]), mask=x, mask_act=ctx.act_in.act)
            ctx.act_in.publish_done(g_x)
        else:
            g_x = wgrad_then_dgrad(x, g_pre, ctx.w, ctx.b, spec, ctx.needs_input_grad[0], ctx.link_in, ctx.wino_u,
                                   grad_slices=ctx.x_slices_ok)
        g_add = g_pre if (ctx.has_add and ctx.needs_input_grad[3]) else None
        return g_x, None, None, g_add, None, None, None


class ResBlock(Function):
    """out = post_act(x + Conv1x1(ReLU(Conv3x3(x)))), both convs bias-free: ResidualLayer (vq_vae.py:57-70) as ONE autograd
    node.  As two ConvAct nodes the input x has two consumers (the 3x3 conv and the skip), and autograd adds their
    gradients with a launch of its own per block; here the skip gradient enters the 3x3 conv's data gradient as the
    ``add`` operand of its epilogue (direct and Winograd kernels alike), and the ReLU derivative is the ``mask`` operand of
    the 1x1 conv's data gradient."""

    @staticmethod
    def forward(ctx, x, w3, w1, spec3, spec1, aux=None):
        """aux: companion rows (see adjacent_rows); returns (out, out_aux) then."""
        _req_cuda(x, w3, w1)
        x = _c(x)
        ctx.wino_u = None
        ctx.specs = (spec3, spec1)
        ctx.w = (w3, w1)
        rows = adjacent_rows(x, _c(aux)) if aux is not None else None
        Bt = rows.shape[0] if rows is not None else x.shape[0]
        ready, ctx.wino_u = _wino_filters_for(ctx.needs_input_grad[0], w3, spec3, Bt, x.shape[1], x.shape[2])
        if aux is not None:
            first = [True]

            def run(t):
                wo, first[0] = (ctx.wino_u if (first[0] and ready is None) else None), False
                h_ = conv_forward_raw(t, w3, None, spec3, wino_out=wo, wino_ready=ready)
                return h_, conv_forward_raw(h_, w1, None, spec1, t)
            (h, out), (_h_aux, out_aux) = _with_aux(x, _c(aux), run)
            ctx.save_for_backward(x, h, out if spec1.act != ACT_NONE else None)
            ctx.mark_non_differentiable(out_aux)
            ctx.set_materialize_grads(False)
            return out, out_aux
        h = conv_forward_raw(x, w3, None, spec3, wino_out=None if ready is not None else ctx.wino_u, wino_ready=ready)
        out = conv_forward_raw(h, w1, None, spec1, x)
        ctx.save_for_backward(x, h, out if spec1.act != ACT_NONE else None)
        return out

    @staticmethod
    def backward(ctx, g_out, *_g_aux):
        if g_out is None:
            return (None,) * 6
        spec3, spec1 = ctx.specs
        w3, w1 = ctx.w
        x, h, out = ctx.saved_tensors
        g_out = _c(g_out)
        g_pre = act_backward_raw(g_out, out, spec1.act) if spec1.act != ACT_NONE else g_out
        if _PAIR:
            g_h = conv_backward_raw(h, g_pre, w1, None, spec1, mask=h, mask_act=spec3.act)
        else:
            conv_wgrad_raw(h, g_pre, w1, None, spec1)
            g_h = conv_dgrad_raw(g_pre, w1, spec1, (h.shape[1], h.shape[2]), mask=h, mask_act=spec3.act)
        conv_wgrad_raw(x, g_h, w3, None, spec3)
        g_x = None
        if ctx.needs_input_grad[0]:
            g_x = conv_dgrad_raw(g_h, w3, spec3, (x.shape[1], x.shape[2]), add=g_pre, wino_filters=ctx.wino_u)
        return g_x, None, None, None, None, None


class ConvBNAct(Function):
    """a = act(BatchNorm2d(conv(x, w) + b)) with train-mode batch statistics (vanilla_vae.py:25-35,47-75)."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, running_mean, running_var, training, spec, bn_act, num_batches_tracked=None,
                lazy_out=False):
        """lazy_out (model code, blocks.Chain: the ONE consumer of the output applies BatchNorm + activation while it loads):
        the returned tensor holds the raw conv output y and carries ``_ctvae_lazy_bn = (coef [2][C], act)``; the stand-alone
        apply launch and the activated tensor do not exist.  An input tagged that way is read through its own coefficients."""
        global _last_link
        _req_cuda(x, w, gamma)
        ctx.link_in = link_of(x)
        ctx.x_slices_ok = bool(getattr(x, "_ctvae_grad_slices_ok", False))
        lazy_in = getattr(x, "_ctvae_lazy_bn", None) if x.is_contiguous() else None
        x = _c(x)
        B, H, W, _ = x.shape
        ho, wo = spec.out_hw(H, W)
        C = spec.co
        y = torch.empty((B, ho, wo, C), dtype=torch.float32, device=x.device)
        a = None if lazy_out else torch.empty_like(y)
        coef = torch.empty(2 * C, dtype=torch.float32, device=x.device) if lazy_out else None
        ws = native.workspace(x.device)
        save_mean = torch.empty(C, dtype=torch.float32, device=x.device)
        save_invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        isc = lazy_in[0].data_ptr() if lazy_in is not None else None
        ish = lazy_in[0].data_ptr() + 4 * spec.ci if lazy_in is not None else None
        native.call("ctvae_conv_bn_act_forward", spec.kind, x.data_ptr(), w.data_ptr(), native.ptr(b), gamma.data_ptr(),
                    beta.data_ptr(), native.ptr(running_mean), native.ptr(running_var), BN_MOMENTUM, BN_EPS,
                    1 if training else 0, bn_act, y.data_ptr(), native.ptr(a), save_mean.data_ptr(), save_invstd.data_ptr(),
                    native.ptr(coef), native.ptr(num_batches_tracked) if training else None, B, H, W, spec.ci, spec.co, spec.k,
                    spec.stride, spec.pad, spec.out_pad, isc, ish, lazy_in[1] if lazy_in is not None else ACT_NONE,
                    ws.data_ptr(), ws.numel() * 4)
        ctx.spec, ctx.bn_act, ctx.training = spec, bn_act, training
        ctx.params = (w, b, gamma, beta)
        ctx.lazy_in = lazy_in
        ctx.save_for_backward(x, y, save_mean, save_invstd)
        ctx.link_out = _last_link = BNLink(y, save_mean, save_invstd, gamma, beta, bn_act) if training else None
        if lazy_out:
            global _last_lazy
            _last_lazy = (coef, bn_act)        # picked up by the caller of .apply (pop_lazy_bn), which tags the returned tensor
            return y                           # the output IS the raw conv output (also saved above for backward)
        return a

    @staticmethod
    def backward(ctx, g_a):
        if not ctx.training:
            raise RuntimeError("backward through eval-mode BatchNorm is not supported on the HIP path")
        spec = ctx.spec
        w, b, gamma, beta = ctx.params
        x, y, save_mean, save_invstd = ctx.saved_tensors
        g_a = _c(g_a)
        B, H, W, C = y.shape
        in_coef, in_act = (ctx.lazy_in[0], ctx.lazy_in[1]) if ctx.lazy_in is not None else (None, ACT_NONE)
        ws = native.workspace(x.device)
        gg, accg = grad_target(gamma)
        gbt, accb = grad_target(beta)
        if accg != accb:
            (gg if accg == 0 else gbt).zero_()
            accg = 1
        lazy = ctx.link_out.take_lazy(g_a) if ctx.link_out is not None else None
        if lazy is not None:
            # g_a was never materialised: the consumer's split-K slices are summed here, with the sums, the finalize and the apply
            g_y = torch.empty_like(y)
            native.call("ctvae_bn_backward_fused", lazy[0].data_ptr(), lazy[1], *lazy[2], y.data_ptr(), gamma.data_ptr(),
                        beta.data_ptr(), save_mean.data_ptr(), save_invstd.data_ptr(), ctx.bn_act, g_y.data_ptr(), gg.data_ptr(),
                        gbt.data_ptr(), accg)
            g_x = wgrad_then_dgrad(x, g_y, w, b, spec, ctx.needs_input_grad[0], ctx.link_in, in_coef=in_coef, in_act=in_act,
                                   grad_slices=ctx.x_slices_ok)
            return (g_x,) + (None,) * 11
        part, rows, coef = ctx.link_out.take(g_a) if ctx.link_out is not None else (None, 0, None)
        if coef is not None:
            part, rows = None, 0         # finalized by the consumer's finishing launch: apply + commit only
            if not ctx.needs_input_grad[0] and _ENC_BN_ON_LOAD and wgrad_bn_apply_mode(spec, x.shape[0], x.shape[1], x.shape[2]) == 2:
                # the first layer of the encoder: no data gradient follows, so g_y is only the weight gradient's operand -- formed
                # on load from (g_a, y, coef); the apply launch and its 33 MB output are not needed
                conv_wgrad_raw(x, g_a, w, b, spec, dy_bn=(y, coef, ctx.bn_act, None), bn_commit=(gg, gbt, accg))
                return (None,) * 12
        g_y = torch.empty_like(y)
        native.call("ctvae_bn_backward", g_a.data_ptr(), beta.data_ptr(), y.data_ptr(), B * H * W, C, gamma.data_ptr(),
                    save_mean.data_ptr(), save_invstd.data_ptr(), ctx.bn_act, g_y.data_ptr(), gg.data_ptr(), gbt.data_ptr(),
                    accg, native.ptr(part), rows, None, native.ptr(coef), ws.data_ptr(), ws.numel() * 4)
        g_x = wgrad_then_dgrad(x, g_y, w, b, spec, ctx.needs_input_grad[0], ctx.link_in, in_coef=in_coef, in_act=in_act,
                               grad_slices=ctx.x_slices_ok)
        return (g_x,) + (None,) * 11


_xform_ok_cache = {}


def input_transform_supported(spec: ConvSpec, B, H, W) -> bool:
    """Can this layer apply the previous block's BatchNorm + activation while loading (ctvae_conv_forward in_scale)?"""
    key = (spec.kind, B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad)
    ok = _xform_ok_cache.get(key)
    if ok is None:
        ok = _xform_ok_cache[key] = bool(native.load().ctvae_conv_input_transform_supported(*key))
    return ok


_wgrad_bn_ok_cache = {}


def wgrad_bn_apply_mode(spec: ConvSpec, B, H, W) -> int:
    """ctvae_conv_wgrad_bn_apply_supported: 0 no; 1 the weight-gradient kernel applies the BatchNorm backward on load and writes
    g_y for the data gradient (final_layer.0); 2 it applies it on load and nothing is written (encoder.0: no data gradient)."""
    key = (spec.kind, B, H, W, spec.ci, spec.co, spec.k, spec.stride, spec.pad, spec.out_pad)
    mode = _wgrad_bn_ok_cache.get(key)
    if mode is None:
        mode = _wgrad_bn_ok_cache[key] = int(native.load().ctvae_conv_wgrad_bn_apply_supported(*key))
    return mode


def wgrad_bn_apply_supported(spec: ConvSpec, B, H, W) -> bool:
    """Can this layer's weight-gradient kernel apply the BatchNorm backward on load and hand g_y on (ctvae_conv_wgrad dy_bn_*)?"""
    return wgrad_bn_apply_mode(spec, B, H, W) == 1


class ConvBNActConvAct(Function):
    """r = act2(conv2(act(BN(conv1(x))))) for an image-side conv2 (3 output channels): the reference's final_layer
    nn.Sequential(ConvTranspose2d, BatchNorm2d, LeakyReLU, Conv2d, Tanh) (vanilla_vae.py:64-75) as ONE autograd node.

    The BatchNorm output is never written: conv1 leaves y and the per-channel affine, conv2's forward and weight
    gradient apply affine + LeakyReLU while staging their input tiles, conv2's data gradient emits the BatchNorm's
    backward sums.  On the 64x64x32 activations of this block that removes two full passes over 134 MB per step."""

    @staticmethod
    def forward(ctx, x, w1, b1, gamma, beta, running_mean, running_var, num_batches_tracked, w2, b2, training, spec1, spec2, bn_act):
        _req_cuda(x, w1, w2, gamma)
        ctx.link_in = link_of(x)
        lazy_in = ctx.lazy_in = getattr(x, "_ctvae_lazy_bn", None) if x.is_contiguous() else None
        x = _c(x)
        B, H, W, _ = x.shape
        ho, wo = spec1.out_hw(H, W)
        C = spec1.co
        y1 = torch.empty((B, ho, wo, C), dtype=torch.float32, device=x.device)
        coef = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        save_mean = torch.empty(C, dtype=torch.float32, device=x.device)
        save_invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = native.workspace(x.device)
        native.call("ctvae_conv_bn_act_forward", spec1.kind, x.data_ptr(), w1.data_ptr(), native.ptr(b1), gamma.data_ptr(),
                    beta.data_ptr(), native.ptr(running_mean), native.ptr(running_var), BN_MOMENTUM, BN_EPS,
                    1 if training else 0, bn_act, y1.data_ptr(), None, save_mean.data_ptr(), save_invstd.data_ptr(),
                    coef.data_ptr(), native.ptr(num_batches_tracked) if training else None, B, H, W, spec1.ci, spec1.co,
                    spec1.k, spec1.stride, spec1.pad, spec1.out_pad,
                    lazy_in[0].data_ptr() if lazy_in is not None else None,
                    lazy_in[0].data_ptr() + 4 * spec1.ci if lazy_in is not None else None,
                    lazy_in[1] if lazy_in is not None else ACT_NONE, ws.data_ptr(), ws.numel() * 4)
        r = conv_forward_raw(y1, w2, b2, spec2, in_coef=coef, in_act=bn_act)
        ctx.act_out = offer_out_act_link(r, spec2.act) if (training and spec2.act != ACT_NONE and _OUT_ACT_LINK) else None
        ctx.specs, ctx.bn_act, ctx.training = (spec1, spec2), bn_act, training
        ctx.params = (w1, b1, gamma, beta, w2, b2)
        ctx.save_for_backward(x, y1, r, coef, save_mean, save_invstd)
        return r

    @staticmethod
    def backward(ctx, g_r):
        if not ctx.training:
            raise RuntimeError("backward through eval-mode BatchNorm is not supported on the HIP path")
        spec1, spec2 = ctx.specs
        w1, b1, gamma, beta, w2, b2 = ctx.params
        x, y1, r, coef, save_mean, save_invstd = ctx.saved_tensors
        g_r = _c(g_r)
        if ctx.act_out is not None and ctx.act_out.take(g_r):
            g_pre = g_r                                          # the loss's backward pass already applied act'(r)
        else:
            g_pre = act_backward_raw(g_r, r, spec2.act) if spec2.act != ACT_NONE else g_r
        B, H, W, C = y1.shape
        link = BNLink(y1, save_mean, save_invstd, gamma, beta, ctx.bn_act)
        ws = native.workspace(x.device)
        gg, accg = grad_target(gamma)
        gbt, accb = grad_target(beta)
        if accg != accb:
            (gg if accg == 0 else gbt).zero_()
            accg = 1
        lazy = wgrad_bn_apply_supported(spec1, x.shape[0], x.shape[1], x.shape[2])
        g_y = torch.empty_like(y1)
        one_call = ran = _PAIR and _BN_RIDER
        if one_call:
            # conv2's weight gradient, its data gradient (which emits the BatchNorm's backward sums) and the BatchNorm's
            # finalize in one call: the finalize rides in the slab-reduction launch (the link is local to this node, so the
            # rider commits d gamma / d beta itself when no apply launch follows)
            g_a = conv_backward_raw(y1, g_pre, w2, b2, spec2, link=link, in_coef=coef, in_act=ctx.bn_act,
                                    bn_commit=(gg, gbt, accg) if lazy else None)
            part, rows, c7 = link.take(g_a)
            one_call = c7 is not None
        if not one_call:
            if not ran:
                conv_wgrad_raw(y1, g_pre, w2, b2, spec2, in_coef=coef, in_act=ctx.bn_act)
                g_a = conv_dgrad_bn_raw(g_pre, w2, spec2, (H, W), link)
                if g_a is None:
                    g_a = conv_dgrad_raw(g_pre, w2, spec2, (H, W))
                part, rows, _ = link.take(g_a)
            bcoef = torch.empty(5 * C, dtype=torch.float32, device=x.device) if lazy else None
            native.call("ctvae_bn_backward", g_a.data_ptr(), beta.data_ptr(), y1.data_ptr(), B * H * W, C, gamma.data_ptr(),
                        save_mean.data_ptr(), save_invstd.data_ptr(), ctx.bn_act, None if lazy else g_y.data_ptr(), gg.data_ptr(),
                        gbt.data_ptr(), accg, native.ptr(part), rows, native.ptr(bcoef), None, ws.data_ptr(), ws.numel() * 4)
        elif lazy:
            bcoef = c7                                           # rows 0-4 are the coefficients the weight-gradient kernel reads
        else:
            native.call("ctvae_bn_backward", g_a.data_ptr(), beta.data_ptr(), y1.data_ptr(), B * H * W, C, gamma.data_ptr(),
                        save_mean.data_ptr(), save_invstd.data_ptr(), ctx.bn_act, g_y.data_ptr(), gg.data_ptr(), gbt.data_ptr(),
                        accg, None, 0, None, c7.data_ptr(), ws.data_ptr(), ws.numel() * 4)
        in_coef, in_act = (ctx.lazy_in[0], ctx.lazy_in[1]) if ctx.lazy_in is not None else (None, ACT_NONE)
        if lazy:
            # the weight-gradient kernel turns g_a into g_y on load and leaves g_y behind for the data gradient
            if ctx.needs_input_grad[0] and _PAIR:
                g_x = conv_backward_raw(x, g_a, w1, b1, spec1, link=ctx.link_in, dy_bn=(y1, bcoef, ctx.bn_act, g_y),
                                        in_coef=in_coef, in_act=in_act)
            else:
                conv_wgrad_raw(x, g_a, w1, b1, spec1, dy_bn=(y1, bcoef, ctx.bn_act, g_y), in_coef=in_coef, in_act=in_act)
                g_x = None
                if ctx.needs_input_grad[0]:
                    g_x = None if ctx.link_in is None else conv_dgrad_bn_raw(g_y, w1, spec1, (x.shape[1], x.shape[2]), ctx.link_in)
                    if g_x is None:
                        g_x = conv_dgrad_raw(g_y, w1, spec1, (x.shape[1], x.shape[2]))
        else:
            g_x = wgrad_then_dgrad(x, g_y, w1, b1, spec1, ctx.needs_input_grad[0], ctx.link_in, in_coef=in_coef, in_act=in_act)
        return (g_x,) + (None,) * 13


class ActFn(Function):
    """Standalone activation (nn.LeakyReLU sites mcq_vae.py:185,216)."""

    @staticmethod
    def forward(ctx, x, act, aux=None):
        _req_cuda(x)
        x = _c(x)
        ctx.act = act

        def run(t):
            o = torch.empty_like(t)
            native.call("ctvae_act_forward", t.data_ptr(), o.data_ptr(), t.numel(), act)
            return o
        if aux is not None:
            (out,), (out_aux,) = _with_aux(x, _c(aux), run)
            ctx.save_for_backward(out)
            ctx.mark_non_differentiable(out_aux)
            ctx.set_materialize_grads(False)
            return out, out_aux
        out = run(x)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g, *_g_aux):
        if g is None:
            return None, None, None
        (out,) = ctx.saved_tensors
        return act_backward_raw(_c(g), out, ctx.act), None, None


# ---------------------------------------------------------------------------------------------------
# Gaussian reparameterisation and losses
# ---------------------------------------------------------------------------------------------------
def _rows(t):
    """(tensor, row stride) for a 2-D tensor whose rows are contiguous (column slices of a wider matrix allowed)."""
    if t.dim() != 2 or t.stride(1) != 1:
        t = t.contiguous()
    return t, t.stride(0)


class SplitHeads(Function):
    """[B,2L] fused fc_mu|fc_var output -> (mu, log_var) views; backward gathers both gradients with one
    kernel instead of two zero-fills, two strided copies and an add."""

    @staticmethod
    def forward(ctx, heads, L):
        ctx.L = L
        return heads[:, :L], heads[:, L:]

    @staticmethod
    def backward(ctx, g_mu, g_lv):
        if g_mu is None:
            g_mu = torch.zeros_like(g_lv)
        if g_lv is None:
            g_lv = torch.zeros_like(g_mu)
        return torch.cat([g_mu, g_lv], dim=1), None


class Reparameterize(Function):
    """z = eps * exp(0.5*logvar) + mu (vanilla_vae.py:107-117), noise injectable."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        _req_cuda(mu, logvar, eps)
        mu_, mrs = _rows(mu)
        lv_, lrs = _rows(logvar)
        eps = _c(eps)
        B, L = mu.shape
        z = torch.empty((B, L), dtype=torch.float32, device=mu.device)
        native.call("ctvae_reparam_forward", mu_.data_ptr(), mrs, lv_.data_ptr(), lrs, eps.data_ptr(), z.data_ptr(), B, L)
        ctx.save_for_backward(lv_, eps)
        ctx.lrs = lrs
        return z

    @staticmethod
    def backward(ctx, g_z):
        lv_, eps = ctx.saved_tensors
        g_z = _c(g_z)
        B, L = g_z.shape
        g_mu = torch.empty_like(g_z)
        g_lv = torch.empty_like(g_z)
        native.call("ctvae_reparam_backward", g_z.data_ptr(), lv_.data_ptr(), ctx.lrs, eps.data_ptr(), g_mu.data_ptr(),
                    g_lv.data_ptr(), B, L)
        return g_mu, g_lv, None


class GaussianLatent(Function):
    """(mu, log_var, z) from the fused fc_mu | fc_var output heads [B,2L] as ONE autograd node (vanilla_vae.py:85-92,107-122):
    mu / log_var are views of heads, z = eps*exp(0.5*log_var) + mu.  eps None: N(0,1) drawn in the kernel from ``rng`` (a device
    uint64 pair: Philox key and stream position, advanced by the backward launch).  Backward forms g_heads in one launch;
    as separate SplitHeads / Reparameterize nodes autograd adds the KL and the reparameterisation contributions to mu and
    log_var with a launch each and concatenates with a third."""

    @staticmethod
    def forward(ctx, heads, eps, rng):
        _req_cuda(heads)
        fs = getattr(heads, "_ctvae_fwd_slices", None) if heads.is_contiguous() else None   # (slices, n, bias): heads not written yet
        heads = _c(heads)
        B, L2 = heads.shape
        L = L2 // 2
        eps_in = _c(eps) if eps is not None else None
        if eps_in is None and rng is None:
            raise RuntimeError("GaussianLatent: either eps or the rng state is needed")
        eps_out = torch.empty((B, L), dtype=torch.float32, device=heads.device)
        z = torch.empty((B, L), dtype=torch.float32, device=heads.device)
        native.call("ctvae_gauss_latent_forward", heads.data_ptr(), native.ptr(eps_in), native.ptr(rng) if eps_in is None else None,
                    eps_out.data_ptr(), z.data_ptr(), B, L, fs[0].data_ptr() if fs else None, fs[1] if fs else 0,
                    native.ptr(fs[2]) if fs else None)
        ctx.save_for_backward(heads, eps_out)
        ctx.rng = rng if eps_in is None else None
        ctx.set_materialize_grads(False)
        return heads[:, :L], heads[:, L:], z

    @staticmethod
    def backward(ctx, g_mu, g_lv, g_z):
        heads, eps = ctx.saved_tensors
        B, L2 = heads.shape
        g_mu = _c(g_mu) if g_mu is not None else None
        g_lv = _c(g_lv) if g_lv is not None else None
        lazy = claim_lazy_grad(g_z) if (g_z is not None and g_z.is_contiguous()) else None
        g_z = _c(g_z) if g_z is not None else None
        g_heads = torch.empty_like(heads)
        native.call("ctvae_gauss_latent_backward", native.ptr(g_mu), native.ptr(g_lv), lazy[0].data_ptr() if lazy else native.ptr(g_z),
                    heads.data_ptr(), eps.data_ptr(), g_heads.data_ptr(), native.ptr(ctx.rng), B, L2 // 2, lazy[1] if lazy else 0)
        return g_heads, None, None


def _scalar_outputs(ctx, out):
    """The 4-vector a loss kernel wrote, handed out as four 0-dim views.  Callers index the returned tuple, so no autograd
    select node sits between the loss and this Function: indexing a tensor output (out[0]) made ``loss.backward()`` build
    zeros[4] and copy the root gradient into it -- two launches per step for nothing.  Only the first carries gradient."""
    o = out.unbind(0)
    ctx.mark_non_differentiable(*o[1:])
    ctx.set_materialize_grads(False)
    return o


class VAELoss(Function):
    """out = [loss, mse, kld, -kld]: mse = F.mse_loss(recons, x); kld as vanilla_vae.py:143; loss = mse + M_N*kld (+ extra).
    recons/x are same-layout contiguous tensors.  Only out[0] carries gradient."""

    @staticmethod
    def forward(ctx, recons, x, mu, logvar, extra, M_N, logcosh_alpha=0.0):
        """logcosh_alpha > 0: the reconstruction term is LogCoshVAE's (logcosh_vae.py:141-150) instead of the MSE;
        logcosh_alpha == L2L1 (-1): SWAE's F.mse_loss + F.l1_loss (swae.py:121-125), no KL term."""
        _req_cuda(recons, x)
        recons, x = _c(recons), _c(x)
        if recons.shape != x.shape:
            raise RuntimeError("mse_loss: shape mismatch")
        out = torch.empty(4, dtype=torch.float32, device=recons.device)
        ws = native.workspace(recons.device)
        pre = None
        if mu is not None:
            mu_, mrs = _rows(mu)
            lv_, lrs = _rows(logvar)
            B, L = mu.shape
        else:
            mu_, lv_, mrs, lrs, B, L = None, None, 0, 0, 0, 0
        if logcosh_alpha < 0.0:
            if mu is not None:
                raise RuntimeError("l2 + l1 reconstruction term: no KL term")
            native.call("ctvae_l2l1_loss_forward", recons.data_ptr(), x.data_ptr(), recons.numel(), native.ptr(extra), out.data_ptr(),
                        ws.data_ptr(), ws.numel() * 4)
        elif logcosh_alpha > 0.0:
            if extra is not None:
                raise RuntimeError("log-cosh loss: no extra term")
            native.call("ctvae_logcosh_loss_forward", recons.data_ptr(), x.data_ptr(), recons.numel(), float(logcosh_alpha),
                        native.ptr(mu_), mrs, native.ptr(lv_), lrs, B, L, float(M_N), out.data_ptr(), ws.data_ptr(), ws.numel() * 4)
        else:
            if (_LOSS_GRAD_IN_FWD and mu is not None and ctx.needs_input_grad[0] and ctx.needs_input_grad[2] and ctx.needs_input_grad[3]):
                # the gradients for a unit upstream gradient come out of the same pass (ctvae_loss_forward_grad); backward hands
                # them over as they are when the loss is the root of the pass (kernels.backward), else it runs its own kernel
                ctx.act_link = claim_out_act_link(recons)
                ract = ctx.act_link.act if ctx.act_link is not None else ACT_NONE
                pre = (torch.empty_like(recons), torch.empty((B, L), dtype=torch.float32, device=recons.device),
                       torch.empty((B, L), dtype=torch.float32, device=recons.device))
                native.call("ctvae_loss_forward_grad", recons.data_ptr(), x.data_ptr(), recons.numel(), mu_.data_ptr(), mrs, lv_.data_ptr(),
                            lrs, B, L, float(M_N), native.ptr(extra), out.data_ptr(), pre[0].data_ptr(), pre[1].data_ptr(),
                            pre[2].data_ptr(), ract, ws.data_ptr(), ws.numel() * 4)
            else:
                native.call("ctvae_loss_forward", recons.data_ptr(), x.data_ptr(), recons.numel(), native.ptr(mu_), mrs, native.ptr(lv_),
                            lrs, B, L, float(M_N), native.ptr(extra), out.data_ptr(), ws.data_ptr(), ws.numel() * 4)
        ctx.pre = pre
        ctx.save_for_backward(recons, x, mu_, lv_)
        ctx.meta = (mrs, lrs, B, L, float(M_N), tuple(extra.shape) if extra is not None else None)
        ctx.logcosh_alpha = float(logcosh_alpha)
        if pre is None:
            ctx.act_link = claim_out_act_link(recons) if ctx.needs_input_grad[0] else None
        return _scalar_outputs(ctx, out)

    @staticmethod
    def backward(ctx, g_loss, *_unused):
        recons, x, mu_, lv_ = ctx.saved_tensors
        mrs, lrs, B, L, M_N, has_extra = ctx.meta
        if g_loss is None:
            return (None,) * 7
        if ctx.pre is not None and _is_cached_root(g_loss):
            # the loss is the root of this backward pass (kernels.backward): the forward pass has written these gradients
            g_r, g_mu, g_lv = ctx.pre
            if ctx.act_link is not None:
                ctx.act_link.publish_done(g_r)
            g_extra = g_loss.reshape(has_extra) if (has_extra is not None and ctx.needs_input_grad[4]) else None
            return g_r, None, g_mu, g_lv, g_extra, None, None
        g_loss = _c(g_loss.reshape(1))               # d/d loss; mse and kld outputs are reported detached
        want_r = ctx.needs_input_grad[0]
        want_kl = mu_ is not None and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        g_r = torch.empty_like(recons) if want_r else None
        link = ctx.act_link if want_r else None
        ract = link.act if link is not None else ACT_NONE        # recons = act(pre): hand back the gradient w.r.t. pre
        g_mu = g_lv = None
        if want_kl:
            g_mu = torch.empty((B, L), dtype=torch.float32, device=recons.device)
            g_lv = torch.empty((B, L), dtype=torch.float32, device=recons.device)
        if want_r and want_kl:      # the usual case: both gradients in one launch
            native.call("ctvae_loss_backward", recons.data_ptr(), x.data_ptr(), g_loss.data_ptr(), g_r.data_ptr(), recons.numel(),
                        ctx.logcosh_alpha, mu_.data_ptr(), mrs, lv_.data_ptr(), lrs, g_mu.data_ptr(), g_lv.data_ptr(), B, L, M_N, ract)
        else:
            if want_r:
                if ctx.logcosh_alpha < 0.0:
                    native.call("ctvae_l2l1_backward", recons.data_ptr(), x.data_ptr(), g_loss.data_ptr(), g_r.data_ptr(), recons.numel(),
                                ract)
                elif ctx.logcosh_alpha > 0.0:
                    native.call("ctvae_logcosh_backward", recons.data_ptr(), x.data_ptr(), g_loss.data_ptr(), g_r.data_ptr(),
                                recons.numel(), ctx.logcosh_alpha, ract)
                else:
                    native.call("ctvae_mse_backward", recons.data_ptr(), x.data_ptr(), g_loss.data_ptr(), g_r.data_ptr(), recons.numel(),
                                ract)
            if want_kl:
                native.call("ctvae_kl_backward", mu_.data_ptr(), mrs, lv_.data_ptr(), lrs, g_loss.data_ptr(), g_mu.data_ptr(),
                            g_lv.data_ptr(), B, L, M_N)
        if link is not None:
            link.publish_done(g_r)
        g_extra = g_loss.reshape(has_extra) if (has_extra is not None and ctx.needs_input_grad[4]) else None
        return g_r, None, g_mu, g_lv, g_extra, None, None


class GumbelSoftmax(Function):
    """s = softmax((z + g)/temp, dim=-1), g = -log(-log(u + eps) + eps): CategoricalVAE.reparameterize
    (cat_vae.py:118-132) with the uniform draws injectable.  z, u: [..., Q]."""

    @staticmethod
    def forward(ctx, z, u, temp, eps):
        _req_cuda(z, u)
        if z.shape != u.shape:
            raise RuntimeError("gumbel_softmax: logits / uniform shape mismatch")
        z, u = _c(z), _c(u)
        Q = z.shape[-1]
        s = torch.empty_like(z)
        native.call("ctvae_gumbel_softmax_forward", z.data_ptr(), u.data_ptr(), s.data_ptr(), z.numel() // Q, Q, float(temp), float(eps))
        ctx.save_for_backward(s)
        ctx.temp = float(temp)
        return s

    @staticmethod
    def backward(ctx, g_s):
        (s,) = ctx.saved_tensors
        g_s = _c(g_s)
        Q = s.shape[-1]
        g_z = torch.empty_like(s)
        native.call("ctvae_gumbel_softmax_backward", g_s.data_ptr(), s.data_ptr(), g_z.data_ptr(), s.numel() // Q, Q, ctx.temp)
        return g_z, None, None, None


class CatKL(Function):
    """kld = mean_b sum_{d,q} p (log(p + eps) - log(1/Q + eps)), p = softmax(q, -1) (cat_vae.py:147,160-167).  q: [B,D,Q]."""

    @staticmethod
    def forward(ctx, q, eps):
        _req_cuda(q)
        q = _c(q)
        B, Q = q.shape[0], q.shape[-1]
        log_prior = math.log(1.0 / Q + eps)
        out = torch.empty(1, dtype=torch.float32, device=q.device)
        ws = native.workspace(q.device)
        native.call("ctvae_cat_kl_forward", q.data_ptr(), q.numel() // Q, Q, B, float(eps), log_prior, out.data_ptr(),
                    ws.data_ptr(), ws.numel() * 4)
        ctx.save_for_backward(q)
        ctx.meta = (B, Q, float(eps), log_prior)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (q,) = ctx.saved_tensors
        B, Q, eps, log_prior = ctx.meta
        g = _c(g.reshape(1))
        g_q = torch.empty_like(q)
        native.call("ctvae_cat_kl_backward", q.data_ptr(), g.data_ptr(), g_q.data_ptr(), q.numel() // Q, Q, B, eps, log_prior)
        return g_q, None


class IWLoss(Function):
    """out = [loss, mean lp, mean kld, -mean kld] of the importance-weighted objective (iwae.py:126-155, miwae.py:130-163):
    recons [R,...] (R = B*M*S rows, same per-image layout as x [B,...]), mu / logvar [R,L], S samples per group.
    Only out[0] carries gradient (through both the softmax weights and the log-weights, as in the reference)."""

    @staticmethod
    def forward(ctx, recons, x, mu, logvar, S, M_N):
        _req_cuda(recons, x, mu, logvar)
        recons, x, mu, logvar = _c(recons), _c(x), _c(mu), _c(logvar)
        R, B = recons.shape[0], x.shape[0]
        n = recons.numel() // R
        if R % B or x.numel() != B * n or mu.shape[0] != R or mu.shape != logvar.shape:
            raise RuntimeError("iw_loss: shape mismatch")
        L = mu.numel() // R
        dev = recons.device
        rows = torch.empty(3, R, dtype=torch.float32, device=dev)      # lp, kld, coef
        out = torch.empty(4, dtype=torch.float32, device=dev)
        native.call("ctvae_iw_loss_forward", recons.data_ptr(), x.data_ptr(), n, R, R // B, mu.data_ptr(), logvar.data_ptr(), L,
                    int(S), float(M_N), rows[0].data_ptr(), rows[1].data_ptr(), rows[2].data_ptr(), out.data_ptr())
        ctx.save_for_backward(recons, x, mu, logvar, rows)
        ctx.meta = (n, R, R // B, L, float(M_N))
        return _scalar_outputs(ctx, out)

    @staticmethod
    def backward(ctx, g_loss, *_unused):
        recons, x, mu, logvar, rows = ctx.saved_tensors
        n, R, rep, L, M_N = ctx.meta
        if g_loss is None:
            return (None,) * 6
        g_loss = _c(g_loss.reshape(1))
        g_r = torch.empty_like(recons) if ctx.needs_input_grad[0] else None
        g_mu = g_lv = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            g_mu, g_lv = torch.empty_like(mu), torch.empty_like(logvar)
        native.call("ctvae_iw_loss_backward", recons.data_ptr(), x.data_ptr(), n, R, rep, mu.data_ptr(), logvar.data_ptr(), L, M_N,
                    rows[2].data_ptr(), g_loss.data_ptr(), native.ptr(g_r), native.ptr(g_mu), native.ptr(g_lv))
        return g_r, None, g_mu, g_lv, None, None


class MMD(Function):
    """out = [mmd, K(p,p), K(z,z), K(p,z)] of WAE_MMD / InfoVAE (wae_mmd.py:120-203): z, prior [N,D]; kind 'imq' | 'rbf';
    c = 2*D*latent_var.  Only out[0] carries gradient, and only towards z (the prior draws are constants)."""

    @staticmethod
    def forward(ctx, z, prior, kind, c, w_pp, w_zz, w_pz):
        _req_cuda(z, prior)
        z, prior = _c(z), _c(prior)
        if z.dim() != 2 or z.shape != prior.shape:
            raise RuntimeError("mmd: z / prior must be [N,D] of one shape")
        if kind not in ("imq", "rbf"):
            raise ValueError('Undefined kernel type.')
        N, D = z.shape
        out = torch.empty(4, dtype=torch.float32, device=z.device)
        grad = torch.empty_like(z)
        ws = native.workspace(z.device)
        native.call("ctvae_mmd_forward", z.data_ptr(), prior.data_ptr(), N, D, 0 if kind == "imq" else 1, float(c), 1e-7,
                    float(w_pp), float(w_zz), float(w_pz), out.data_ptr(), grad.data_ptr(), ws.data_ptr(), ws.numel() * 4)
        ctx.save_for_backward(grad)
        return _scalar_outputs(ctx, out)

    @staticmethod
    def backward(ctx, g_mmd, *_unused):
        (grad,) = ctx.saved_tensors
        return (grad * g_mmd if g_mmd is not None else None), None, None, None, None, None, None


L2L1 = -1.0      # VAELoss(..., logcosh_alpha=L2L1): squared + absolute error

_kl_dummy = {}


class GaussKL(Function):
    """mean_b(-0.5 * sum_d(1 + logvar - mu^2 - exp(logvar))): the Gaussian KL term on its own (hvae.py:213-222 forms three of
    them) through the loss kernels: reconstruction operands of four zeros, M_N = 1, so out = {kld, 0, kld, -kld}."""

    @staticmethod
    def forward(ctx, mu, logvar):
        _req_cuda(mu, logvar)
        mu_, mrs = _rows(mu)
        lv_, lrs = _rows(logvar)
        B, L = mu.shape
        z4 = _kl_dummy.get(mu.device)
        if z4 is None:
            z4 = _kl_dummy[mu.device] = torch.zeros(4, dtype=torch.float32, device=mu.device)
        out = torch.empty(4, dtype=torch.float32, device=mu.device)
        ws = native.workspace(mu.device)
        native.call("ctvae_loss_forward", z4.data_ptr(), z4.data_ptr(), 4, mu_.data_ptr(), mrs, lv_.data_ptr(), lrs, B, L, 1.0, None,
                    out.data_ptr(), ws.data_ptr(), ws.numel() * 4)
        ctx.save_for_backward(mu_, lv_)
        ctx.meta = (mrs, lrs, B, L)
        return out[2].clone()

    @staticmethod
    def backward(ctx, g):
        mu_, lv_ = ctx.saved_tensors
        mrs, lrs, B, L = ctx.meta
        g = _c(g.reshape(1))
        g_mu = torch.empty((B, L), dtype=torch.float32, device=mu_.device)
        g_lv = torch.empty((B, L), dtype=torch.float32, device=mu_.device)
        native.call("ctvae_kl_backward", mu_.data_ptr(), mrs, lv_.data_ptr(), lrs, g.data_ptr(), g_mu.data_ptr(), g_lv.data_ptr(), B, L, 1.0)
        return g_mu, g_lv


class LadderMerge(Function):
    """(z, kl) of one LVAE rung (lvae.py:166-204; csrc/ladder.hip): merge of the bottom-up (mu_e, lv_e) and top-down (mu_t, lv_t)
    Gaussians, reparameterised sample with eps, kl [B] between the merged and the bottom-up Gaussian."""

    @staticmethod
    def forward(ctx, mu_e, lv_e, mu_t, lv_t, eps):
        _req_cuda(mu_e, lv_e, mu_t, lv_t, eps)
        mu_e, lv_e, mu_t, lv_t, eps = (_c(t) for t in (mu_e, lv_e, mu_t, lv_t, eps))
        B, D = mu_e.shape
        z = torch.empty_like(mu_e)
        kl = torch.empty(B, dtype=torch.float32, device=mu_e.device)
        native.call("ctvae_ladder_merge_forward", mu_e.data_ptr(), lv_e.data_ptr(), mu_t.data_ptr(), lv_t.data_ptr(), eps.data_ptr(),
                    B, D, z.data_ptr(), kl.data_ptr())
        ctx.save_for_backward(mu_e, lv_e, mu_t, lv_t, eps)
        ctx.set_materialize_grads(False)
        return z, kl

    @staticmethod
    def backward(ctx, gz, gkl):
        mu_e, lv_e, mu_t, lv_t, eps = ctx.saved_tensors
        if gz is None and gkl is None:
            return (None,) * 5
        B, D = mu_e.shape
        gz = _c(gz) if gz is not None else None
        gkl = _c(gkl) if gkl is not None else None
        outs = [torch.empty_like(mu_e) for _ in range(4)]
        native.call("ctvae_ladder_merge_backward", native.ptr(gz), native.ptr(gkl), mu_e.data_ptr(), lv_e.data_ptr(), mu_t.data_ptr(),
                    lv_t.data_ptr(), eps.data_ptr(), B, D, *[o.data_ptr() for o in outs])
        return outs[0], outs[1], outs[2], outs[3], None


class GammaReparam(Function):
    """z = h(alpha + Bs, h^-1(alpha + Bs, zhat)) / beta: GammaVAE.reparameterize (gamma_vae.py:108-149) with the draw
    zhat ~ Gamma(alpha + Bs, 1) given.  csrc/gamma.hip."""

    @staticmethod
    def forward(ctx, alpha, beta, zhat, shape_b):
        _req_cuda(alpha, beta, zhat)
        alpha, beta, zhat = _c(alpha), _c(beta), _c(zhat)
        z = torch.empty_like(alpha)
        native.call("ctvae_gamma_reparam_forward", alpha.data_ptr(), beta.data_ptr(), zhat.data_ptr(), float(shape_b), z.data_ptr(),
                    alpha.numel())
        ctx.save_for_backward(alpha, beta, zhat)
        ctx.shape_b = float(shape_b)
        return z

    @staticmethod
    def backward(ctx, g):
        alpha, beta, zhat = ctx.saved_tensors
        g = _c(g)
        ga, gb = torch.empty_like(alpha), torch.empty_like(alpha)
        native.call("ctvae_gamma_reparam_backward", g.data_ptr(), alpha.data_ptr(), beta.data_ptr(), zhat.data_ptr(), ctx.shape_b,
                    ga.data_ptr(), gb.data_ptr(), alpha.numel())
        return ga, gb, None, None


class GammaKL(Function):
    """mean_b sum_d [I(c,d,c,d) - I(1/alpha, beta, c, d)], c = 1/prior_alpha, d = prior_beta (gamma_vae.py:151-171)."""

    @staticmethod
    def forward(ctx, alpha, beta, prior_alpha, prior_beta):
        _req_cuda(alpha, beta)
        alpha, beta = _c(alpha), _c(beta)
        B, D = alpha.shape
        out = torch.empty(1, dtype=torch.float32, device=alpha.device)
        ws = native.workspace(alpha.device)
        native.call("ctvae_gamma_kl_forward", alpha.data_ptr(), beta.data_ptr(), B, D, float(prior_alpha), float(prior_beta),
                    out.data_ptr(), ws.data_ptr(), ws.numel() * 4)
        ctx.save_for_backward(alpha, beta)
        ctx.prior = (float(prior_alpha), float(prior_beta))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        alpha, beta = ctx.saved_tensors
        B, D = alpha.shape
        g = _c(g.reshape(1))
        ga, gb = torch.empty_like(alpha), torch.empty_like(alpha)
        native.call("ctvae_gamma_kl_backward", g.data_ptr(), alpha.data_ptr(), beta.data_ptr(), B, D, ctx.prior[0], ctx.prior[1],
                    ga.data_ptr(), gb.data_ptr())
        return ga, gb, None, None


class Sigmoid(Function):
    """nn.Sigmoid behind GammaVAE's final conv (gamma_vae.py:78) as an elementwise pair."""

    @staticmethod
    def forward(ctx, x):
        _req_cuda(x)
        x = _c(x)
        y = torch.empty_like(x)
        native.call("ctvae_sigmoid_forward", x.data_ptr(), y.data_ptr(), x.numel())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = _c(g)
        gx = torch.empty_like(y)
        native.call("ctvae_sigmoid_backward", g.data_ptr(), y.data_ptr(), gx.data_ptr(), y.numel())
        return gx


class TCDecomp(Function):
    """(mi, tc, kld) of BetaTCVAE's KL decomposition (betatc_vae.py:128-199; csrc/tcvae.hip): z, mu, logvar [B,D] (D <= 32),
    log_iw [B,B] the log importance weights.  All three outputs carry gradient."""

    @staticmethod
    def forward(ctx, z, mu, logvar, log_iw):
        _req_cuda(z, mu, logvar, log_iw)
        z, mu, logvar, log_iw = (_c(t) for t in (z, mu, logvar, log_iw))
        B, D = z.shape
        if mu.shape != z.shape or logvar.shape != z.shape or tuple(log_iw.shape) != (B, B):
            raise RuntimeError("tc decomposition: shapes")
        out = torch.empty(3, dtype=torch.float32, device=z.device)
        lse_s = torch.empty(B, dtype=torch.float32, device=z.device)
        lse_d = torch.empty((B, D), dtype=torch.float32, device=z.device)
        ws = native.workspace(z.device)
        native.call("ctvae_tc_forward", z.data_ptr(), mu.data_ptr(), logvar.data_ptr(), log_iw.data_ptr(), B, D, out.data_ptr(),
                    lse_s.data_ptr(), lse_d.data_ptr(), ws.data_ptr(), ws.numel() * 4)
        ctx.save_for_backward(z, mu, logvar, log_iw, lse_s, lse_d)
        ctx.set_materialize_grads(False)
        return out.unbind(0)

    @staticmethod
    def backward(ctx, g_mi, g_tc, g_kld):
        z, mu, logvar, log_iw, lse_s, lse_d = ctx.saved_tensors
        if g_mi is None and g_tc is None and g_kld is None:
            return None, None, None, None
        zero = torch.zeros((), dtype=torch.float32, device=z.device)
        g3 = torch.stack([(g if g is not None else zero).reshape(()).to(torch.float32) for g in (g_mi, g_tc, g_kld)])
        B, D = z.shape
        dz, dmu, dlv = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        native.call("ctvae_tc_backward", z.data_ptr(), mu.data_ptr(), logvar.data_ptr(), log_iw.data_ptr(), lse_s.data_ptr(),
                    lse_d.data_ptr(), g3.data_ptr(), B, D, dz.data_ptr(), dmu.data_ptr(), dlv.data_ptr())
        return dz, dmu, dlv, None


class VampKL(Function):
    """VampVAE's KL term (vampvae.py:140-171): -(E_log_p - E_log_q) with the VampPrior mixture over the K pseudo-input
    posteriors.  z, mu, logvar [B,D]; prior_mu, prior_logvar [K,D].  csrc/vamp.hip, both directions."""

    @staticmethod
    def forward(ctx, z, mu, logvar, pmu, plv):
        _req_cuda(z, mu, logvar, pmu, plv)
        z, mu, logvar, pmu, plv = (_c(t) for t in (z, mu, logvar, pmu, plv))
        B, D = z.shape
        Kc = pmu.shape[0]
        if mu.shape != z.shape or logvar.shape != z.shape or pmu.shape != (Kc, D) or plv.shape != (Kc, D):
            raise RuntimeError("vamp kl: shapes")
        out = torch.empty(3, dtype=torch.float32, device=z.device)
        wgt = torch.empty((B, Kc), dtype=torch.float32, device=z.device)
        ws = native.workspace(z.device)
        native.call("ctvae_vamp_kl_forward", z.data_ptr(), mu.data_ptr(), logvar.data_ptr(), pmu.data_ptr(), plv.data_ptr(), B, D, Kc,
                    out.data_ptr(), wgt.data_ptr(), ws.data_ptr(), ws.numel() * 4)
        ctx.save_for_backward(z, mu, logvar, pmu, plv, wgt)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        z, mu, logvar, pmu, plv, wgt = ctx.saved_tensors
        B, D = z.shape
        Kc = pmu.shape[0]
        g = _c(g.reshape(1))
        dz, dmu, dlv = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        dpm, dpl = torch.empty_like(pmu), torch.empty_like(pmu)
        native.call("ctvae_vamp_kl_backward", z.data_ptr(), mu.data_ptr(), logvar.data_ptr(), pmu.data_ptr(), plv.data_ptr(),
                    wgt.data_ptr(), g.data_ptr(), B, D, Kc, dz.data_ptr(), dmu.data_ptr(), dlv.data_ptr(), dpm.data_ptr(), dpl.data_ptr())
        return dz, dmu, dlv, dpm, dpl


class SWD(Function):
    """Sliced Wasserstein distance of SWAE.compute_swd (swae.py:150-178): z, prior [N,D]; proj [S,D] unit directions; p the
    exponent; weight = reg_weight.  One launch projects, sorts both sets per direction and leaves d swd / d z (csrc/swd.hip)."""

    @staticmethod
    def forward(ctx, z, prior, proj, p, weight):
        _req_cuda(z, prior, proj)
        z, prior, proj = _c(z), _c(prior), _c(proj)
        if z.dim() != 2 or z.shape != prior.shape or proj.dim() != 2 or proj.shape[1] != z.shape[1]:
            raise RuntimeError("swd: z / prior must be [N,D] of one shape and proj [S,D]")
        N, D = z.shape
        S = proj.shape[0]
        out = torch.empty(1, dtype=torch.float32, device=z.device)
        grad = torch.empty_like(z)
        ws = native.workspace(z.device)
        native.call("ctvae_swd_forward", z.data_ptr(), prior.data_ptr(), proj.data_ptr(), N, D, S, float(p), float(weight),
                    out.data_ptr(), grad.data_ptr(), ws.data_ptr(), ws.numel() * 4)
        ctx.save_for_backward(grad)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None


class DIPLoss(Function):
    """DIP-VAE II regulariser of dip_vae.py:147-159 on mu / log_var [B,D] (csrc/dip.hip) -> 0-dim tensor."""

    @staticmethod
    def forward(ctx, mu, logvar, lambda_diag, lambda_offdiag):
        _req_cuda(mu, logvar)
        mu_, mrs = _rows(mu)
        lv_, lrs = _rows(logvar)
        B, D = mu.shape
        n = native.load().ctvae_dip_state_floats(B, D)
        state = torch.empty(n, dtype=torch.float32, device=mu.device)
        native.call("ctvae_dip_forward", mu_.data_ptr(), mrs, lv_.data_ptr(), lrs, B, D, float(lambda_diag), float(lambda_offdiag),
                    state.data_ptr())
        ctx.save_for_backward(state)
        ctx.dims = (B, D)
        return state[B * D + D * D + 3 * D].clone()

    @staticmethod
    def backward(ctx, g):
        (state,) = ctx.saved_tensors
        B, D = ctx.dims
        g = _c(g.reshape(1))
        g_mu = torch.empty((B, D), dtype=torch.float32, device=state.device)
        g_lv = torch.empty((B, D), dtype=torch.float32, device=state.device)
        native.call("ctvae_dip_backward", state.data_ptr(), g.data_ptr(), g_mu.data_ptr(), g_lv.data_ptr(), B, D)
        return g_mu, g_lv, None, None


_ones = {}
_LOSS_GRAD_IN_FWD = os.environ.get("CTVAE_NO_LOSS_GRAD_IN_FWD", "0") != "1"   # diagnostic: the loss gradient always from its own launch


def _is_cached_root(g):
    """Is g the cached ones tensor kernels.backward() hands to autograd as the root gradient (value 1 by construction)?"""
    one = _ones.get((g.device, g.dtype, tuple(g.shape)))
    return one is not None and g.data_ptr() == one.data_ptr()


class capture_graph:
    """``torch.cuda.graph(g, **kw)`` with Python's cyclic garbage collector switched off for the duration of the capture.
    torch collects once before it begins the capture; a collection that the capture's own allocations trigger in the middle
    can destroy device objects of an earlier owner (another model's graphs and pools) while the stream is capturing -- under
    the capture's global mode that is an illegal call, and it ends in ``Fatal Python error: Aborted`` from a destructor."""

    def __init__(self, graph, **kwargs):
        self._cm = torch.cuda.graph(graph, **kwargs)
        self._was = False

    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        if _DEFER_REDUCE and torch.cuda.is_available():
            _defer_arena_for(torch.device("cuda", torch.cuda.current_device()))   # never from inside the capture (its private pool)
        r = self._cm.__enter__()      # synchronizes, collects, empties the cache, begins the capture
        gc.disable()
        return r

    def __exit__(self, *exc):
        import gc
        try:
            return self._cm.__exit__(*exc)
        finally:
            if self._was:
                gc.enable()


_DEFER_REDUCE = os.environ.get("CTVAE_NO_DEFER_REDUCE", "0") != "1"   # diagnostic: every slab reduction in the backward chain
_DEFER_ARENA_BYTES = int(os.environ.get("CTVAE_DEFER_ARENA_MB", "2048")) << 20
_defer_arena = {}


def _defer_arena_for(device):
    """The slab arena of deferred weight-gradient reductions on this device: 2 GB of the 288, allocated once, outside any graph
    capture (capture_graph makes sure of that: an arena first requested inside a capture would live in that graph's pool)."""
    arena = _defer_arena.get(device)
    if arena is None:
        arena = _defer_arena[device] = torch.empty(_DEFER_ARENA_BYTES // 4, dtype=torch.float32, device=device)
    return arena


def backward(loss):
    """``loss.backward()`` with the root gradient taken from a cached ones tensor: autograd otherwise fills a fresh
    ``ones_like(loss)`` on every call (one launch per step; the harness and bench.py call this).

    The slab reductions behind the weight-gradient kernels are deferred to the end of the pass (ctvae_defer_begin / _flush:
    their slabs go into a 2 GB arena, one launch reduces all of them) -- a parameter gradient is complete when this returns."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    one = _ones.get(key)
    if one is None:
        one = _ones[key] = torch.ones_like(loss)
    if not (_DEFER_REDUCE and loss.is_cuda):
        loss.backward(gradient=one)
        return
    _lazy_grads.clear()            # placeholders of an earlier pass that nobody claimed (a pass cut short) are not valid in this one
    arena = _defer_arena_for(loss.device)
    lib = native.load()
    native.check(lib.ctvae_defer_begin(arena.data_ptr(), arena.numel() * 4), "ctvae_defer_begin")
    try:
        loss.backward(gradient=one)
    finally:
        native.check(lib.ctvae_defer_flush(native.stream_ptr()), "ctvae_defer_flush")


class PairMLP(Function):
    """out[b,i,j] = sigmoid(b2 + sum_h w2[h] * leaky_relu(u[b,i,h] + v[b,j,h])): the all-pairs tail of
    ``CausalTransition.graph_discovers[k]`` (ct_mcq_vae.py:86-95,147-151) without the [B,N,N,H] intermediates.
    w2 [H] / b2 [1]: one scorer for the batch; w2 [B,H] / b2 [B]: one scorer per sample (per-action discoverers)."""

    SLOPE = 0.01   # nn.LeakyReLU() default

    @staticmethod
    def forward(ctx, u, v, w2, b2):
        _req_cuda(u, v, w2)
        B, N, H = u.shape
        per_sample = w2.dim() == 2
        if per_sample and (tuple(w2.shape) != (B, H) or b2.numel() != B):
            raise RuntimeError("PairMLP: per-sample scorer needs w2 [B,H] and b2 [B]")
        u, v = _c(u), _c(v)
        w2 = _c(w2) if per_sample else _c(w2.reshape(-1))
        b2 = _c(b2.reshape(-1))
        out = torch.empty((B, N, N), dtype=torch.float32, device=u.device)
        native.call("ctvae_pair_mlp_forward", u.data_ptr(), v.data_ptr(), H, w2.data_ptr(), b2.data_ptr(), out.data_ptr(),
                    B, N, H, PairMLP.SLOPE, 1 if per_sample else 0, None)
        ctx.save_for_backward(u, v, w2, out)
        ctx.per_sample = per_sample
        return out

    @staticmethod
    def backward(ctx, g):
        u, v, w2, out = ctx.saved_tensors
        B, N, H = u.shape
        g = _c(g)
        du, dv = torch.empty_like(u), torch.empty_like(v)
        dw2p = torch.empty((B, H), dtype=torch.float32, device=u.device)
        db2p = torch.empty(B, dtype=torch.float32, device=u.device)
        native.call("ctvae_pair_mlp_backward", u.data_ptr(), v.data_ptr(), H, w2.data_ptr(), out.data_ptr(), g.data_ptr(),
                    du.data_ptr(), dv.data_ptr(), H, dw2p.data_ptr(), db2p.data_ptr(), B, N, H, PairMLP.SLOPE,
                    1 if ctx.per_sample else 0, None)
        if ctx.per_sample:
            return du, dv, dw2p, db2p
        return du, dv, dw2p.sum(0), db2p.sum().reshape(1)


class GATScore(Function):
    """S[b,h,r,c] = sum_k att[h,k] * leaky_relu(xl[b,r,h,k] + xr[b,c,h,k] + attr[b,r,c]*we[h,k], slope): GATv2 attention
    logits on a batch of dense graphs (ct_mcq_vae.py:103-114) without the [B,N,N,C]-per-head intermediates."""

    @staticmethod
    def forward(ctx, xl, xr, attr, we, att, slope):
        _req_cuda(xl, xr, attr, we, att)
        xl, xr, attr, we, att = _c(xl), _c(xr), _c(attr), _c(we), _c(att)
        B, N, H, C = xl.shape
        out = torch.empty((B, H, N, N), dtype=torch.float32, device=xl.device)
        native.call("ctvae_gat_score", 0, xl.data_ptr(), xr.data_ptr(), attr.data_ptr(), we.data_ptr(), att.data_ptr(),
                    out.data_ptr(), B, N, H, C, float(slope))
        ctx.save_for_backward(xl, xr, attr, we, att)
        ctx.slope = float(slope)
        return out

    @staticmethod
    def backward(ctx, g):
        xl, xr, attr, we, att = ctx.saved_tensors
        B, N, H, C = xl.shape
        g = _c(g)
        dxl, dxr = torch.empty_like(xl), torch.empty_like(xr)
        dattp = torch.empty((B, H, C), dtype=torch.float32, device=xl.device)
        dwep = torch.empty((B, H, C), dtype=torch.float32, device=xl.device)
        native.call("ctvae_gat_score_backward", xl.data_ptr(), xr.data_ptr(), attr.data_ptr(), we.data_ptr(), att.data_ptr(),
                    g.data_ptr(), dxl.data_ptr(), dxr.data_ptr(), dattp.data_ptr(), dwep.data_ptr(), B, N, H, C, ctx.slope)
        dattr = None
        if ctx.needs_input_grad[2]:
            t = torch.empty((B, H, N, N), dtype=torch.float32, device=xl.device)
            native.call("ctvae_gat_score", 1, xl.data_ptr(), xr.data_ptr(), attr.data_ptr(), we.data_ptr(), att.data_ptr(),
                        t.data_ptr(), B, N, H, C, ctx.slope)
            dattr = (g * t).sum(1)
        return dxl, dxr, dattr, dwep.sum(0), dattp.sum(0), None


def banked(params) -> bool:
    """True when the same-shaped contiguous tensors lie back to back in memory (FlatParamMixin places a module bank so)."""
    p0 = params[0]
    n = p0.numel() * p0.element_size()
    return all(p.is_contiguous() and p.shape == p0.shape and p.data_ptr() == p0.data_ptr() + k * n for k, p in enumerate(params))


_flat_buffers = []     # [(weakref(flat parameter buffer), weakref(flat gradient buffer))] of the live FlatParamMixin models
_alias_given = {}
_GRAD_ALIAS = os.environ.get("CTVAE_NO_GRAD_ALIAS", "0") != "1"   # diagnostic: bank gradients in tensors of their own, copied by gather


def register_flat_buffers(flat, gflat):
    _flat_buffers[:] = [(f, g) for f, g in _flat_buffers if f() is not None and g() is not None]
    _flat_buffers.append((weakref.ref(flat), weakref.ref(gflat)))


_alias_epoch = [-1, -1]   # parameter epoch of the last zero_grad; ... of the last zero_grad that also cleared the torch-level span


def note_zero_grad(prezeroed):
    """FlatParamMixin.zero_grad (after its epoch bump): gradient aliases may be handed out until the epoch moves again (an
    optimizer step without a zero_grad behind it leaves .grad attached, and a kernel writing into the memory autograd is about to
    add to would double the gradient); prezeroed: the whole autograd-managed span of the gradient buffer was cleared in one fill."""
    _alias_epoch[0] = _param_epoch[0]
    _alias_epoch[1] = _param_epoch[0] if prezeroed else -1


def alias_prezeroed():
    return _alias_epoch[1] == _param_epoch[0]


def flat_grad_alias(t):
    """For a contiguous tensor t that lies in a model's flat PARAMETER buffer (a bank of autograd-managed parameters): a fresh
    view of the same range of the flat GRADIENT buffer, for the gradient kernel to write into -- autograd then attaches slices
    of it as the parameters' .grad and gather_torch_grads() has nothing to copy (CT-MCQ-VAE: two multi-tensor copies of 21 and
    18 us per step).  Only in the epoch a zero_grad opened, and once per range: a second writer gets None and a tensor of its
    own, which autograd adds to the first."""
    if not _GRAD_ALIAS or t is None or not t.is_contiguous() or _alias_epoch[0] != _param_epoch[0]:
        return None
    for fr, gr in _flat_buffers:
        f, g = fr(), gr()
        if f is None or g is None or f.device != t.device:
            continue
        off = t.data_ptr() - f.data_ptr()
        if off >= 0 and off % 4 == 0 and off + 4 * t.numel() <= 4 * f.numel():
            if _alias_given.get(t.data_ptr()) == _param_epoch[0]:
                return None
            _alias_given[t.data_ptr()] = _param_epoch[0]
            return g.as_strided(tuple(t.shape), tuple(t.stride()), off // 4)
    return None


class BankView(Function):
    """The G same-shaped parameters of a module bank (back to back in memory, see ``banked``) as ONE [G, *shape] tensor
    without a copy; backward hands every parameter its slice of the bank's gradient (views, no copies either)."""

    @staticmethod
    def forward(ctx, *params):
        p0 = params[0]
        if not banked(params):
            raise RuntimeError("BankView: the parameters are not laid out as one bank")
        n = p0.numel()
        return p0.detach().as_strided((len(params),) + tuple(p0.shape), (n,) + tuple(p0.stride()))

    @staticmethod
    def backward(ctx, g):
        return tuple(g[k] for k in range(g.shape[0]))


def glinear_ok(K, N, *lds):
    return K % 4 == 0 and N % 4 == 0 and all(l % 4 == 0 for l in lds)


class GroupLinear(Function):
    """y[b, m, s*N + n] = sum_k x[b, m, k] * W_s[group_s[b]][n][koff_s + k] + bias_s[group_s[b]][n] on blocks of 64 rows per
    sample (csrc/glinear.hip).  x [B,64,K]; spec: per segment (koff, group) with group an int32 [B] tensor or None (everybody
    uses matrix 0); wb: per segment the bank W [G,N,ldw >= koff+K] and its bias [G,N] or None.  Several segments may name the
    same bank: its gradient then arrives once, at the first of them."""

    @staticmethod
    def forward(ctx, x, K, N, spec, *wb):
        import ctypes as C
        _req_cuda(x, *[t for t in wb if t is not None])
        x = _c(x)
        B, M, ldx = x.shape
        nseg = len(spec)
        if M != 64 or ldx != K or len(wb) != 2 * nseg or not 1 <= nseg <= 4:
            raise RuntimeError("GroupLinear: needs x [B,64,K] and (bank, bias) per segment, 1..4 segments")
        Ws, bs, gs = [], [], []
        for s, (koff, grp) in enumerate(spec):
            W, b = wb[2 * s], wb[2 * s + 1]
            if W.dim() != 3 or W.shape[1] != N or W.stride(2) != 1 or koff + K > W.shape[2] or (b is not None and not b.is_contiguous()):
                raise RuntimeError("GroupLinear: bank must be [G,N,ldw] with unit inner stride")
            if grp is not None and (grp.dtype != torch.int32 or grp.numel() != B or not grp.is_contiguous()):
                raise RuntimeError("GroupLinear: group ids must be a contiguous int32 [B] tensor")
            Ws.append(W)
            bs.append(b)
            gs.append(grp)
        for s in range(nseg):            # segments of one bank share one gradient tensor (backward): one bias bank per weight bank
            for t in range(s):
                if Ws[t].data_ptr() == Ws[s].data_ptr() and bs[t] is not None and bs[s] is not None and bs[t].data_ptr() != bs[s].data_ptr():
                    raise RuntimeError("GroupLinear: segments that name the same weight bank must name the same bias bank")
        y = torch.empty((B, 64, nseg * N), dtype=torch.float32, device=x.device)
        ctx.host = (
            (C.c_void_p * nseg)(*[W.data_ptr() + 4 * spec[s][0] for s, W in enumerate(Ws)]),
            (C.c_int * nseg)(*[W.stride(1) for W in Ws]),
            (C.c_int64 * nseg)(*[W.stride(0) for W in Ws]),
            (C.c_void_p * nseg)(*[native.ptr(b) for b in bs]),
            (C.c_int * nseg)(*[(b.shape[-1] if b is not None else 0) for b in bs]),
            (C.c_void_p * nseg)(*[native.ptr(g) for g in gs]),
        )
        h = ctx.host
        native.call("ctvae_glinear_forward", x.data_ptr(), ldx, K, nseg, N, h[0], h[1], h[2], h[3], h[4], h[5], y.data_ptr(),
                    nseg * N, B)
        ctx.save_for_backward(x, *[t for t in Ws], *[g for g in gs if g is not None])
        ctx.meta = (K, N, nseg, tuple(k for k, _ in spec), tuple(g is not None for g in gs), tuple(b is not None for b in bs))
        ctx.bias_banks = bs
        return y

    @staticmethod
    def backward(ctx, g):
        K, N, nseg, koffs, has_g, has_b = ctx.meta
        saved = ctx.saved_tensors
        x, Ws = saved[0], saved[1:1 + nseg]
        grp_it = iter(saved[1 + nseg:])
        gs = [next(grp_it) if hg else None for hg in has_g]
        g = _c(g)
        B = x.shape[0]
        h = ctx.host
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            native.call("ctvae_glinear_dgrad", g.data_ptr(), nseg * N, nseg, N, h[0], h[1], h[2], h[5], dx.data_ptr(), K, K, B)
        ws = native.workspace(x.device)
        # Segments that name the SAME bank (the discoverers' first layers: columns 0..D-1 / D..2D-1, matrix 0 for everybody and
        # matrix 1 + action per sample) write into ONE gradient tensor per bank, zero-filled once and accumulated into by each
        # segment's launch; the first segment's input slot carries it, the others return None.  (As separate tensors autograd
        # added them: a fill per segment and an add per extra segment over 5 MB each.)
        grads = [None] * (2 * nseg)
        shared = {}
        for s in range(nseg):
            W = Ws[s]
            G = W.shape[0]
            if not ctx.needs_input_grad[4 + 2 * s]:
                continue
            key = (W.data_ptr(), tuple(W.shape), tuple(W.stride()))
            ent = shared.get(key)
            users = [t for t in range(nseg) if Ws[t].data_ptr() == W.data_ptr() and ctx.needs_input_grad[4 + 2 * t]]
            # a segment that covers its bank completely (all columns, every group) WRITES it: when that is the bank's first
            # user -- with its bias, if the bank has one -- there is no zero fill; everybody after it accumulates
            whole = (ent is None and koffs[s] == 0 and W.shape[2] == K and (gs[s] is not None or G == 1)
                     and (has_b[s] or not any(has_b[t] for t in users)))
            if ent is None:
                dW = flat_grad_alias(W)              # the bank's own range of the flat gradient buffer, where there is one
                if dW is None:
                    dW = (torch.empty_like if whole else torch.zeros_like)(W, memory_format=torch.contiguous_format)
                elif not whole and not alias_prezeroed():
                    dW.zero_()
                db = None
                if any(has_b[t] for t in users):
                    bt = next(ctx.bias_banks[t] for t in users if has_b[t])
                    db = flat_grad_alias(bt) if tuple(bt.shape) == (G, N) else None
                    if db is None:
                        db = (torch.empty if whole else torch.zeros)((G, N), dtype=torch.float32, device=x.device)
                    elif not whole and not alias_prezeroed():
                        db.zero_()
                ent = shared[key] = (dW, db)
                grads[2 * s] = dW
                if db is not None:           # the bias gradient goes to the first segment of this bank that has a bias input
                    t = next(t for t in range(nseg) if Ws[t].data_ptr() == W.data_ptr() and has_b[t])
                    grads[2 * t + 1] = db
            dW, db = ent
            Gk = G if gs[s] is not None else 1        # no group ids: only matrix 0 is used
            native.call("ctvae_glinear_wgrad", x.data_ptr(), K, K, g.data_ptr(), nseg * N, s * N, N, native.ptr(gs[s]), Gk, B,
                        dW.data_ptr() + 4 * koffs[s], dW.stride(1), native.ptr(db) if has_b[s] else None, 0 if whole else 1,
                        ws.data_ptr(), ws.numel() * 4)
        return (dx, None, None, None) + tuple(grads)


class PairScores(Function):
    """Edge scores of every ordered node pair for the all-sample discoverer 0 and (grp given) the discoverer of each sample's
    action, from the projections uv [B,64,nd*2*H] = [u0 | v0 | u_a | v_a] (GroupLinear), reading the scorer bank w2 [G,H] / b2 [G]
    in place: out [nd,B,64,64], out[d][b,i,j] = sigmoid(b2 + sum_h w2[h] * lrelu(u[b,i,h] + v[b,j,h])) (csrc/pairmlp.hip;
    ct_mcq_vae.py:86-95,147-151)."""

    @staticmethod
    def forward(ctx, uv, w2, b2, grp, H):
        _req_cuda(uv, w2, b2)
        uv, w2, b2 = _c(uv), _c(w2), _c(b2)
        B, N, ld = uv.shape
        nd = 2 if grp is not None else 1
        if N != 64 or ld != nd * 2 * H or w2.shape[-1] != H:
            raise RuntimeError("PairScores: uv must be [B,64,nd*2*H]")
        out = torch.empty((nd, B, 64, 64), dtype=torch.float32, device=uv.device)
        p = uv.data_ptr()
        native.call("ctvae_pair_mlp_forward", p, p + 4 * H, ld, w2.data_ptr(), b2.data_ptr(), out[0].data_ptr(), B, 64, H,
                    PairMLP.SLOPE, 0, None)
        if nd == 2:
            native.call("ctvae_pair_mlp_forward", p + 8 * H, p + 12 * H, ld, w2.data_ptr(), b2.data_ptr(), out[1].data_ptr(), B, 64,
                        H, PairMLP.SLOPE, 1, grp.data_ptr())
        ctx.save_for_backward(uv, w2, b2, grp, out)
        ctx.H = H
        return out

    @staticmethod
    def backward(ctx, g):
        uv, w2, b2, grp, out = ctx.saved_tensors
        H = ctx.H
        B, _, ld = uv.shape
        nd = out.shape[0]
        G = w2.shape[0]
        g = _c(g)
        d_uv = torch.empty_like(uv)
        dw2p = torch.empty((nd, B, H), dtype=torch.float32, device=uv.device)
        db2p = torch.empty((nd, B), dtype=torch.float32, device=uv.device)
        p, d = uv.data_ptr(), d_uv.data_ptr()
        native.call("ctvae_pair_mlp_backward", p, p + 4 * H, ld, w2.data_ptr(), out[0].data_ptr(), g[0].data_ptr(), d, d + 4 * H, ld,
                    dw2p[0].data_ptr(), db2p[0].data_ptr(), B, 64, H, PairMLP.SLOPE, 0, None)
        # per-sample partials -> rows of the scorer bank: discoverer 0 takes every sample, discoverer grp[b] sample b (in order,
        # no atomics; as torch ops this was one_hot^T @ parts plus the fills / casts / cats around it)
        dw2 = flat_grad_alias(w2) if tuple(w2.shape) == (G, H) else None
        db2 = flat_grad_alias(b2) if tuple(b2.shape) == (G,) else None
        if dw2 is None:
            dw2 = torch.empty((G, H), dtype=torch.float32, device=uv.device)
        if db2 is None:
            db2 = torch.empty((G,), dtype=torch.float32, device=uv.device)
        native.call("ctvae_group_rowsum", dw2p[0].data_ptr(), 0, 1, B, H, H, None, G, dw2.data_ptr(), 0)
        native.call("ctvae_group_rowsum", db2p[0].data_ptr(), 0, 1, B, 1, 1, None, G, db2.data_ptr(), 0)
        if nd == 2:
            native.call("ctvae_pair_mlp_backward", p + 8 * H, p + 12 * H, ld, w2.data_ptr(), out[1].data_ptr(), g[1].data_ptr(),
                        d + 8 * H, d + 12 * H, ld, dw2p[1].data_ptr(), db2p[1].data_ptr(), B, 64, H, PairMLP.SLOPE, 1, grp.data_ptr())
            native.call("ctvae_group_rowsum", dw2p[1].data_ptr(), 0, 1, B, H, H, grp.data_ptr(), G, dw2.data_ptr(), 1)
            native.call("ctvae_group_rowsum", db2p[1].data_ptr(), 0, 1, B, 1, 1, grp.data_ptr(), G, db2.data_ptr(), 1)
        return d_uv, dw2, db2, None, None


class GATLayer(Function):
    """One GATv2Conv layer on B dense graphs over the 64 latent nodes (csrc/gatlayer.hip; ct_mcq_vae.py:103-114): scores,
    masked softmax over the sources and the alpha-weighted aggregation in one launch.

    xlr [B,64,2*Hs*C]: lin_l(x) of the Hs head slots, then lin_r(x) (one GEMM output); adj [B,64,64] weighted adjacency
    (adj[b,r,c] != 0: edge r -> c); we / att [H,C], bias [H*C]; head_map int32 [B,Hs] or None (slot == head).
    Returns act(out) [B,64,Hs*C]."""

    @staticmethod
    def forward(ctx, xlr, adj, we, att, bias, head_map, Hs, C, slope, act):
        _req_cuda(xlr, adj, we, att, bias)
        xlr, adj, we, att, bias = _c(xlr), _c(adj), _c(we), _c(att), _c(bias)
        B, N, ld = xlr.shape
        if N != 64 or ld != 2 * Hs * C or tuple(adj.shape) != (B, 64, 64):
            raise RuntimeError("GATLayer: needs xlr [B,64,2*Hs*C] and adj [B,64,64]")
        if head_map is not None:
            head_map = _c(head_map.to(torch.int32))
            if tuple(head_map.shape) != (B, Hs):
                raise RuntimeError("GATLayer: head_map must be [B,Hs]")
        out = torch.empty((B, 64, Hs * C), dtype=torch.float32, device=xlr.device)
        alpha = torch.empty((B, Hs, 64, 64), dtype=torch.float32, device=xlr.device)
        native.call("ctvae_gat_layer_forward", xlr.data_ptr(), xlr.data_ptr() + 4 * Hs * C, ld, adj.data_ptr(), we.data_ptr(),
                    att.data_ptr(), bias.data_ptr(), native.ptr(head_map), out.data_ptr(), Hs * C, alpha.data_ptr(), B, Hs, C,
                    float(slope), int(act))
        ctx.save_for_backward(xlr, adj, we, att, bias, head_map, out, alpha)
        ctx.meta = (Hs, C, float(slope), int(act))
        return out

    @staticmethod
    def backward(ctx, g):
        xlr, adj, we, att, bias, head_map, out, alpha = ctx.saved_tensors
        Hs, C, slope, act = ctx.meta
        B, _, ld = xlr.shape
        H = we.shape[0]
        g = _c(g)
        dev = xlr.device
        scratch = torch.empty((2, B, Hs, 64, 64), dtype=torch.float32, device=dev)
        d_xlr = torch.empty_like(xlr)
        parts = torch.empty((3, B * Hs, C), dtype=torch.float32, device=dev)
        dadj = torch.empty_like(adj) if ctx.needs_input_grad[1] else None
        native.call("ctvae_gat_layer_backward", xlr.data_ptr(), xlr.data_ptr() + 4 * Hs * C, ld, adj.data_ptr(), we.data_ptr(),
                    att.data_ptr(), bias.data_ptr(), native.ptr(head_map), out.data_ptr(), Hs * C, alpha.data_ptr(), g.data_ptr(),
                    scratch[0].data_ptr(), scratch[1].data_ptr(), d_xlr.data_ptr(), d_xlr.data_ptr() + 4 * Hs * C, ld,
                    parts[0].data_ptr(), parts[1].data_ptr(), parts[2].data_ptr(), native.ptr(dadj), 0, B, Hs, C, slope, act)
        if head_map is None:
            red = parts.view(3, B, Hs * C).sum(1).view(3, H, C)
        else:   # per-head sums of the per-(sample, slot) partials, in row order (deterministic, no atomics): one launch
            red = torch.empty((3, H, C), dtype=torch.float32, device=dev)
            native.call("ctvae_group_rowsum", parts.data_ptr(), B * Hs * C, 3, B * Hs, C, C, head_map.data_ptr(), H, red.data_ptr(), 0)
        return d_xlr, dadj, red[2], red[1], red[0].reshape(-1), None, None, None, None, None


class CTActionReg(Function):
    """beta * adjacency_KL_loss(adj) + delta * graph_size_loss(graph) + epsilon * positive_trial_loss(adj) of
    CausalTransition.forward_action (ct_mcq_vae.py:275,314-323) in one launch each way (csrc/ctmisc.hip).  adj, graph [B,64,64];
    uni [B,4096]: the uniform draws behind the KL target softmax(rand)."""

    @staticmethod
    def forward(ctx, adj, graph, uni, beta, delta, epsilon):
        _req_cuda(adj, graph, uni)
        adj, graph, uni = _c(adj), _c(graph), _c(uni)
        B, N, _ = adj.shape
        part = torch.empty((B, 4), dtype=torch.float32, device=adj.device)
        ctx.coef = (float(beta) / B, float(delta) / B, float(epsilon) / B)
        native.call("ctvae_ct_reg_forward", adj.data_ptr(), graph.data_ptr(), uni.data_ptr(), part.data_ptr(), ctx.coef[0],
                    ctx.coef[1], ctx.coef[2], B, N)
        ctx.save_for_backward(adj, graph, uni, part)
        return part[:, 3].sum()

    @staticmethod
    def backward(ctx, g):
        adj, graph, uni, part = ctx.saved_tensors
        B, N, _ = adj.shape
        g = _c(g.reshape(1))
        d_adj, d_graph = torch.empty_like(adj), torch.empty_like(graph)
        native.call("ctvae_ct_reg_backward", adj.data_ptr(), graph.data_ptr(), uni.data_ptr(), part.data_ptr(), g.data_ptr(),
                    ctx.coef[0], ctx.coef[1], ctx.coef[2], d_adj.data_ptr(), d_graph.data_ptr(), B, N)
        return d_adj, d_graph, None, None, None, None


class BlendSoftmax(Function):
    """softmax_d(y[..., 0, :] * (1 - mask) + y[..., 1, :] * mask) (two head slots) or softmax_d(y[..., 0, :]) (one): the tail of
    CausalTransition._compute_y (ct_mcq_vae.py:226-228).  y [B,S,Hs,D], mask [B,S,1] or None -> [B,S,D]."""

    @staticmethod
    def forward(ctx, y, mask):
        _req_cuda(y)
        y = _c(y)
        B, S, Hs, D = y.shape
        m = _c(mask.reshape(B * S)) if mask is not None else None
        probs = torch.empty((B, S, D), dtype=torch.float32, device=y.device)
        native.call("ctvae_ct_blend_softmax_forward", y.data_ptr(), native.ptr(m), probs.data_ptr(), B * S, Hs, D)
        ctx.save_for_backward(y, m, probs)
        ctx.mask_shape = tuple(mask.shape) if mask is not None else None
        return probs

    @staticmethod
    def backward(ctx, g):
        y, m, probs = ctx.saved_tensors
        B, S, Hs, D = y.shape
        g = _c(g)
        dy = torch.empty_like(y)
        dm = torch.empty(B * S, dtype=torch.float32, device=y.device) if (m is not None and ctx.needs_input_grad[1]) else None
        native.call("ctvae_ct_blend_softmax_backward", g.data_ptr(), probs.data_ptr(), y.data_ptr(), native.ptr(m), dy.data_ptr(),
                    native.ptr(dm), B * S, Hs, D)
        return dy, (dm.view(ctx.mask_shape) if dm is not None else None)


class LatentCE(Function):
    """F.cross_entropy(log(clamp(p, 1e-4)), target) over node rows (latent_CrossEntropy_loss, ct_mcq_vae.py:306-311).
    probs [R,D] rows contiguous, target int64 [R]."""

    @staticmethod
    def forward(ctx, probs, target):
        _req_cuda(probs, target)
        probs, target = _c(probs), _c(target)
        R, D = probs.shape
        rows = torch.empty(R, dtype=torch.float32, device=probs.device)
        native.call("ctvae_ct_latent_ce_forward", probs.data_ptr(), target.data_ptr(), rows.data_ptr(), R, D)
        ctx.save_for_backward(probs, target)
        return rows.mean()

    @staticmethod
    def backward(ctx, g):
        probs, target = ctx.saved_tensors
        R, D = probs.shape
        g = _c(g.reshape(1))
        d = torch.empty_like(probs)
        native.call("ctvae_ct_latent_ce_backward", probs.data_ptr(), target.data_ptr(), g.data_ptr(), d.data_ptr(), R, D)
        return d, None


class CTMask(Function):
    """CausalTransition._compute_mask (ct_mcq_vae.py:117-127) in one launch each way (csrc/ctmisc.hip): x [B,64,64] (one-hot
    latent, no gradient), action [B,A], pe [64,64], keep [B,64,64] or None (dropout keep mask of the positional encoding),
    W [64,A+64] / bias [64] (mask.0), expo [B,64,2] exponential draws -> straight-through sample [B,64]."""

    @staticmethod
    def forward(ctx, x, action, pe, keep, scale, W, bias, expo):
        _req_cuda(x, action, pe, W, bias, expo)
        x, action, pe, W, bias, expo = _c(x), _c(action), _c(pe), _c(W), _c(bias), _c(expo)
        keep = _c(keep) if keep is not None else None
        B, S, D = x.shape
        A = action.shape[1]
        dev = x.device
        inter = torch.empty((B, S, D), dtype=torch.float32, device=dev)
        psoft = torch.empty((3, B, S), dtype=torch.float32, device=dev)      # p, sample, soft
        native.call("ctvae_ct_mask_forward", x.data_ptr(), action.data_ptr(), pe.data_ptr(), native.ptr(keep), float(scale),
                    W.data_ptr(), bias.data_ptr(), expo.data_ptr(), B, S, D, A, inter.data_ptr(), psoft[0].data_ptr(),
                    psoft[1].data_ptr(), psoft[2].data_ptr())
        ctx.save_for_backward(x, action, pe, keep, inter, psoft)
        ctx.scale = float(scale)
        return psoft[1]

    @staticmethod
    def backward(ctx, g):
        x, action, pe, keep, inter, psoft = ctx.saved_tensors
        B, S, D = x.shape
        A = action.shape[1]
        g = _c(g)
        dWp = torch.empty((B, A + D, D), dtype=torch.float32, device=x.device)
        dbp = torch.empty((B, D), dtype=torch.float32, device=x.device)
        native.call("ctvae_ct_mask_backward", x.data_ptr(), action.data_ptr(), pe.data_ptr(), native.ptr(keep), ctx.scale,
                    inter.data_ptr(), psoft[0].data_ptr(), psoft[2].data_ptr(), g.data_ptr(), B, S, D, A, dWp.data_ptr(), dbp.data_ptr())
        return None, None, None, None, None, dWp.sum(0).t(), dbp.sum(0), None


class PosEncode(Function):
    """(x + pe) * keep * scale: PositionalEncoding.forward (ct_mcq_vae.py:33-38) with the dropout mask given; x [B,S,D],
    pe [S,D], keep [B,S,D] or None."""

    @staticmethod
    def forward(ctx, x, pe, keep, scale):
        _req_cuda(x, pe)
        x, pe = _c(x), _c(pe)
        keep = _c(keep) if keep is not None else None
        if x.numel() % pe.numel() or pe.numel() % 4 or (keep is not None and keep.shape != x.shape):
            raise RuntimeError("PosEncode: shapes")
        out = torch.empty_like(x)
        native.call("ctvae_ct_posenc_forward", x.data_ptr(), pe.data_ptr(), native.ptr(keep), float(scale), out.data_ptr(), x.numel(),
                    pe.numel())
        ctx.save_for_backward(keep)
        ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, g):
        (keep,) = ctx.saved_tensors
        if keep is None:
            return g, None, None, None
        g = _c(g)
        gx = torch.empty_like(g)
        native.call("ctvae_ct_posenc_backward", g.data_ptr(), keep.data_ptr(), ctx.scale, gx.data_ptr(), g.numel())
        return gx, None, None, None


def one_hot_f32(inds, N):
    """F.one_hot(inds, N).float() in one launch; inds int64 (any shape) -> [..., N] float32."""
    _req_cuda(inds)
    inds = _c(inds)
    if inds.dtype != torch.int64 or N % 4:
        return torch.nn.functional.one_hot(inds, N).to(torch.float32)
    out = torch.empty(tuple(inds.shape) + (N,), dtype=torch.float32, device=inds.device)
    native.call("ctvae_one_hot", inds.data_ptr(), inds.numel(), N, out.data_ptr())
    return out


class MaskBlend(Function):
    """adj = s[0] * (1 - mask) + s[1] * mask (CausalTransition._compute_adj, ct_mcq_vae.py:153): s [2,B,64,64] the scores of
    discoverer 0 and of each sample's own discoverer (PairScores), mask [B,64,1]; one launch each way (csrc/ctmisc.hip)."""

    @staticmethod
    def forward(ctx, s, mask):
        _req_cuda(s, mask)
        s, mask = _c(s), _c(mask)
        if s.dim() != 4 or s.shape[0] != 2 or s.shape[-1] != 64 or mask.numel() * 64 != s[0].numel():
            raise RuntimeError("MaskBlend: needs s [2,B,64,64] and one mask value per row of 64")
        out = torch.empty_like(s[0])
        native.call("ctvae_ct_blend_forward", s[0].data_ptr(), s[1].data_ptr(), mask.data_ptr(), out.data_ptr(), mask.numel())
        ctx.save_for_backward(s, mask)
        return out

    @staticmethod
    def backward(ctx, g):
        s, mask = ctx.saved_tensors
        g = _c(g)
        gs = torch.empty_like(s)
        gm = torch.empty_like(mask)
        native.call("ctvae_ct_blend_backward", g.data_ptr(), s[0].data_ptr(), s[1].data_ptr(), mask.data_ptr(), gs[0].data_ptr(),
                    gs[1].data_ptr(), gm.data_ptr(), mask.numel())
        return gs, gm


class CTSample(Function):
    """Straight-through Bernoulli(p) from exponential draws expo [...,2] (what F.gumbel_softmax draws: Gumbel = -log E;
    ct_mcq_vae.py:180-183) -> sample, and with ``weighted`` also p * sample (weighted_graph, :244,271) from the same launch."""

    @staticmethod
    def forward(ctx, p, expo, weighted):
        _req_cuda(p, expo)
        p, expo = _c(p), _c(expo)
        out = torch.empty_like(p)
        soft = torch.empty_like(p)
        w = torch.empty_like(p) if weighted else None
        native.call("ctvae_ct_sample_forward", p.data_ptr(), expo.data_ptr(), out.data_ptr(), soft.data_ptr(), native.ptr(w), p.numel())
        ctx.save_for_backward(p, soft, out)
        ctx.set_materialize_grads(False)
        return (out, w) if weighted else out

    @staticmethod
    def backward(ctx, g_s, g_w=None):
        p, soft, out = ctx.saved_tensors
        if g_s is None and g_w is None:
            return None, None, None
        g_s = _c(g_s) if g_s is not None else None
        g_w = _c(g_w) if g_w is not None else None
        gp = torch.empty_like(p)
        native.call("ctvae_ct_sample_backward", native.ptr(g_s), native.ptr(g_w), p.data_ptr(), soft.data_ptr(), out.data_ptr(),
                    gp.data_ptr(), p.numel())
        return gp, None, None


class GumbelBernoulliST(Function):
    """Straight-through Bernoulli(p) sample via hard 2-class Gumbel-softmax (ct_mcq_vae.py:177-183); noise [...,2]."""

    @staticmethod
    def forward(ctx, p, noise):
        _req_cuda(p, noise)
        p, noise = _c(p), _c(noise)
        out = torch.empty_like(p)
        soft = torch.empty_like(p)
        native.call("ctvae_gumbel_st_forward", p.data_ptr(), noise.data_ptr(), out.data_ptr(), soft.data_ptr(), p.numel())
        ctx.save_for_backward(p, soft)
        return out

    @staticmethod
    def backward(ctx, g):
        p, soft = ctx.saved_tensors
        g = _c(g)
        gp = torch.empty_like(p)
        native.call("ctvae_gumbel_st_backward", g.data_ptr(), p.data_ptr(), soft.data_ptr(), gp.data_ptr(), p.numel())
        return gp, None


# ---------------------------------------------------------------------------------------------------
# vector quantiser
# ---------------------------------------------------------------------------------------------------
def vq_compute_inds(latents_nhwc, codebooks, K, C):
    """latents [B,H,W,D] -> int64 [B,C,H,W] (mcq_vae.py:26-39,100-110).  No gradient (arg-min)."""
    _req_cuda(latents_nhwc)
    lat = _c(latents_nhwc.detach())
    B, H, W, D = lat.shape
    inds = torch.empty((B, C, H, W), dtype=torch.int64, device=lat.device)
    native.call("ctvae_vq_inds", lat.data_ptr(), codebooks[0].data_ptr(), inds.data_ptr(), B, H * W, D, K, C)
    return inds


class VQLookup(Function):
    """(quantized [B,H,W,D], vq_loss scalar) = compute_latents(latents, inds) (mcq_vae.py:41-64,112-127)."""

    @staticmethod
    def forward(ctx, latents, inds, beta, K, C, *codebooks):
        _req_cuda(latents, inds)
        lat = _c(latents)
        inds = _c(inds)
        B, H, W, D = lat.shape
        q = torch.empty_like(lat)
        vq_loss = torch.empty((), dtype=torch.float32, device=lat.device)
        ws = native.workspace(lat.device)
        native.call("ctvae_vq_lookup", lat.data_ptr(), codebooks[0].data_ptr(), inds.data_ptr(), q.data_ptr(),
                    vq_loss.data_ptr(), float(beta), B, H * W, D, K, C, ws.data_ptr(), ws.numel() * 4)
        ctx.meta = (float(beta), K, C)
        ctx.codebooks = codebooks
        ctx.save_for_backward(lat, inds)
        return q, vq_loss

    @staticmethod
    def backward(ctx, g_q, g_vq):
        lat, inds = ctx.saved_tensors
        beta, K, C = ctx.meta
        B, H, W, D = lat.shape
        g_q = _c(g_q) if g_q is not None else None
        g_vq = _c(g_vq) if g_vq is not None else None
        g_lat = torch.empty_like(lat) if ctx.needs_input_grad[0] else None
        cb0 = ctx.codebooks[0]
        dcb = None
        acc = 0
        if g_vq is not None and cb0.requires_grad:
            flags = [grad_target(cb) for cb in ctx.codebooks]
            if any(f[1] != flags[0][1] for f in flags):
                for (gt, a) in flags:
                    if a == 0:
                        gt.zero_()
                acc = 1
            else:
                acc = flags[0][1]
            dcb = flags[0][0]
            for i in range(1, C):
                if flags[i][0].data_ptr() != dcb.data_ptr() + i * cb0.numel() * 4:
                    raise RuntimeError("codebook gradients must be stored back to back (flat gradient buffer)")
        ws = native.workspace(lat.device)
        native.call("ctvae_vq_backward", native.ptr(g_q), native.ptr(g_vq), lat.data_ptr(), cb0.data_ptr(), inds.data_ptr(),
                    native.ptr(g_lat), native.ptr(dcb), acc, beta, B, H * W, D, K, C, ws.data_ptr(), ws.numel() * 4)
        return (g_lat, None, None, None, None) + (None,) * len(ctx.codebooks)


# ---------------------------------------------------------------------------------------------------
# flat fused Adam
# ---------------------------------------------------------------------------------------------------
_mssim_host = None


def _mssim_host_arrays():
    """(window, weights) as HOST float arrays for ctvae_mssim_*: the window exactly as the reference builds it
    (mssim_vae.py:203-206: exp(+(x - 5)^2 / (2 * 1.5^2)) in float32, normalised), the level weights of mssim_vae.py:252."""
    global _mssim_host
    if _mssim_host is None:
        import ctypes
        from math import exp
        k = torch.tensor([exp((x - 11 // 2) ** 2 / (2 * 1.5 ** 2)) for x in range(11)])
        k = (k / k.sum()).float()
        win = (ctypes.c_float * 11)(*[float(v) for v in k])
        wts = (ctypes.c_float * 5)(0.0448, 0.2856, 0.3001, 0.2363, 0.1333)
        _mssim_host = (win, wts)
    return _mssim_host


class MSSIMLoss(Function):
    """MSSIM.forward (mssim_vae.py:250-279) of NHWC pictures [B,64,64,C]: 1 - prod of the level powers, five SSIM levels with
    the reference's 11-tap window; the gradient goes to the first picture only (the second is the input image)."""

    @staticmethod
    def forward(ctx, recons, target):
        _req_cuda(recons, target)
        r, x = _c(recons), _c(target)
        B, H, W, C = r.shape
        if tuple(x.shape) != (B, H, W, C) or H != 64 or W != 64:
            raise RuntimeError("MSSIMLoss: NHWC pictures of 64 x 64 pixels (five levels down to 4 x 4, mssim_vae.py:255-266)")
        win, wts = _mssim_host_arrays()
        lib = native.load()
        part = torch.empty(int(lib.ctvae_mssim_part_floats(B, C)), dtype=torch.float32, device=r.device)
        out = torch.empty(11, dtype=torch.float32, device=r.device)          # [0] loss, [1..10] per-level coefficients
        native.call("ctvae_mssim_forward", r.data_ptr(), x.data_ptr(), win, wts, part.data_ptr(), out.data_ptr(),
                    out.data_ptr() + 4, B, C, H, W)
        ctx.save_for_backward(r, x, out)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        r, x, out = ctx.saved_tensors
        B, H, W, C = r.shape
        win, _ = _mssim_host_arrays()
        g_r = torch.empty_like(r)
        native.call("ctvae_mssim_backward", r.data_ptr(), x.data_ptr(), win, out.data_ptr() + 4, _c(g).data_ptr(), g_r.data_ptr(),
                    B, C, H, W)
        return g_r, None


def adam_state(head, device):
    """The device state of ctvae_adam_step: head = [step, lr, beta1, beta2, eps, weight_decay, beta1^step, beta2^step], followed
    by the kernel's zeroed ticket counters (ctvae_adam_state_floats() floats in all).  On the CPU (the gloo tests run a test
    double of the kernel): just the head."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return torch.tensor(head, dtype=torch.float32, device=dev)
    st = torch.zeros(int(native.load().ctvae_adam_state_floats()), dtype=torch.float32, device=dev)
    st[:8] = torch.tensor(head, dtype=torch.float32)
    return st


def adam_step(flat_params, flat_grads, exp_avg, exp_avg_sq, state, grad_scale=1.0):
    _req_cuda(flat_params, flat_grads)
    bump_param_epoch()
    native.call("ctvae_adam_step", flat_params.data_ptr(), flat_grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                state.data_ptr(), flat_params.numel(), float(grad_scale))
