"""Parameter storage for the HIP models: PyTorch-shaped views over packed, flat HBM buffers.

Every parameter keeps the reference's name and logical shape (``state_dict`` interchange with
ct-vae checkpoints, SURVEY.md §5 "Checkpoint / resume"), but its memory is what the kernels want:

* Conv2d weight ``[Co,Ci,kh,kw]``          -> memory ``[kh][kw][Ci][Co]`` (strides ``(1, Co, kw*Ci*Co, Ci*Co)``)
* ConvTranspose2d weight ``[Ci,Co,kh,kw]`` -> memory ``[kh][kw][Ci][Co]`` (strides ``(Co, 1, kw*Ci*Co, Ci*Co)``)
* Linear weight ``[out,in]``               -> memory ``[in][out]``; several Linear layers that read the same
  input (fc_mu / fc_var) share one ``[in][sum(out)]`` block so the heads run as ONE GEMM.

``FlatParamMixin.flatten_parameters`` then places all blocks back to back in ONE fp32 buffer (and the
gradients in a second one of the same layout): the DDP exchange is a single all-reduce of that buffer and
Adam is one kernel over it (288 GB HBM: no reason to scatter 50 small allocations).
"""
import math

import torch
from torch import nn


def _uniform_(t, bound):
    with torch.no_grad():
        t.uniform_(-bound, bound)


class PackedConv(nn.Module):
    """Holder for a Conv2d / ConvTranspose2d weight (+bias) in packed layout; default init = torch's
    (kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)); bias U(+-1/sqrt(fan_in)))."""

    def __init__(self, ci, co, k, transposed=False, bias=True):
        super().__init__()
        self.ci, self.co, self.k, self.transposed = ci, co, k, transposed
        store = torch.empty(k, k, ci, co)
        if transposed:
            w = store.permute(2, 3, 0, 1)      # [Ci,Co,kh,kw]
            fan_in = co * k * k                # torch computes fan_in from dim 1 of the weight tensor
        else:
            w = store.permute(3, 2, 0, 1)      # [Co,Ci,kh,kw]
            fan_in = ci * k * k
        self.weight = nn.Parameter(w)
        _uniform_(self.weight, 1.0 / math.sqrt(fan_in))
        if bias:
            self.bias = nn.Parameter(torch.empty(co))
            _uniform_(self.bias, 1.0 / math.sqrt(fan_in))
        else:
            self.register_parameter("bias", None)

    def storage_blocks(self):
        k, ci, co = self.k, self.ci, self.co
        if self.transposed:
            size, stride = (ci, co, k, k), (co, 1, k * ci * co, ci * co)
        else:
            size, stride = (co, ci, k, k), (1, co, k * ci * co, ci * co)
        blocks = [(k * k * ci * co, [(self.weight, 0, size, stride)])]
        if self.bias is not None:
            blocks.append((co, [(self.bias, 0, (co,), (1,))]))
        return blocks


class PackedLinearGroup:
    """Storage description for Linear layers sharing one input: memory [in][sum(out)] (weights), [sum(out)] (biases)."""

    def __init__(self, linears):
        self.linears = linears
        self.fin = linears[0].in_features
        self.total = sum(l.out_features for l in linears)

    def storage_blocks(self):
        wviews, bviews, off = [], [], 0
        for l in self.linears:
            wviews.append((l.weight, off, (l.out_features, self.fin), (1, self.total)))
            bviews.append((l.bias, off, (l.out_features,), (1,)))
            off += l.out_features
        return [(self.fin * self.total, wviews), (self.total, bviews)]


class PackedLinear(nn.Module):
    """``in_padded``: rows of the weight block in the flat buffer (>= in_features).  The tile kernels gather whole 32-wide
    K chunks, so a layer whose in_features is not a multiple of 32 (JointVAE.decoder_input: 128 + 40) reserves the rows up
    to the next multiple: they hold zeros, the caller pads the activation with zeros, their gradient is zero, Adam leaves
    them zero; the parameter itself keeps its logical [out, in] shape as a view of the first rows."""

    def __init__(self, fin, fout, pad_in_to=1):
        super().__init__()
        self.in_features, self.out_features = fin, fout
        self.in_padded = (fin + pad_in_to - 1) // pad_in_to * pad_in_to
        self.weight = nn.Parameter(torch.empty(fin, fout).t())      # logical [out,in], memory [in][out]
        self.bias = nn.Parameter(torch.empty(fout))
        _uniform_(self.weight, 1.0 / math.sqrt(fin))
        _uniform_(self.bias, 1.0 / math.sqrt(fin))

    def storage_blocks(self):
        w, b = PackedLinearGroup([self]).storage_blocks()
        return [(self.in_padded * self.out_features, w[1]), b]


class PackedBN(nn.Module):
    """BatchNorm2d parameters/buffers (gamma=1, beta=0, running_mean=0, running_var=1, num_batches_tracked=0)."""

    def __init__(self, c):
        super().__init__()
        self.num_features = c
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def storage_blocks(self):
        c = self.num_features
        return [(c, [(self.weight, 0, (c,), (1,))]), (c, [(self.bias, 0, (c,), (1,))])]


class PackedEmbedding(nn.Module):
    """nn.Embedding(K, D) weight, init U(-1/K, 1/K) (mcq_vae.py:22-23)."""

    def __init__(self, K, D):
        super().__init__()
        self.K, self.D = K, D
        self.weight = nn.Parameter(torch.empty(K, D))
        _uniform_(self.weight, 1.0 / K)

    def storage_blocks(self):
        return [(self.K * self.D, [(self.weight, 0, (self.K, self.D), (self.D, 1))])]


class _GradBlock:
    """One storage block of the flat gradient buffer, [lo, hi) floats.  ``fresh``: zero_grad(lazy=True) declared the block zero
    without writing it -- the first kernel that writes the block's gradient overwrites instead of accumulating
    (kernels.grad_target), and whatever is still fresh when the buffer is read gets its zeros then (settle_grads)."""
    __slots__ = ("lo", "hi", "fresh")

    def __init__(self, lo, hi):
        self.lo, self.hi, self.fresh = lo, hi, False


class FlatParamMixin:
    """Mixin for nn.Module roots: one flat parameter buffer + one flat gradient buffer."""

    def _collect_blocks(self):
        """[(numel, [(param, offset, size, stride), ...]), ...] in flat-buffer order; numel < 0 is an alignment marker (the
        next block starts at a multiple of -numel floats).  Order: the HIP-layout blocks of the conv / BN / VQ modules, then the
        banks of autograd-managed modules (``autograd_grads``: placed by the module, gradients from autograd), then every
        other parameter in registration order -- so a sub-module made of torch-level parameters and banks (ct_layer) stays
        one contiguous range (flat_range)."""
        blocks, late, seen_groups = [], [], set()
        claimed, autograd_ids = set(), set()
        for m in self.modules():
            grp = getattr(m, "_linear_group", None)
            if grp is not None:
                if id(grp) not in seen_groups:
                    seen_groups.add(id(grp))
                    blocks.extend(grp.storage_blocks())
                continue
            if hasattr(m, "storage_blocks") and m is not self:
                mb = m.storage_blocks()
                if getattr(m, "autograd_grads", False):
                    late.extend(mb)
                    autograd_ids.update(id(p) for _, views in mb for p, *_ in views)
                else:
                    blocks.extend(mb)
        blocks.append((-4, []))               # 16-byte aligned from here on: the grouped-Linear kernels read weights as float4
        blocks.extend(late)
        for _, views in blocks:
            for p, *_ in views:
                claimed.add(id(p))
        self._torch_param_ids = set(autograd_ids)
        for name, p in self.named_parameters():
            if id(p) not in claimed:   # any other parameter (e.g. torch-level sub-modules): contiguous block
                pc = p.detach().contiguous()
                blocks.append((p.numel(), [(p, 0, tuple(p.shape), tuple(pc.stride()))]))
                self._torch_param_ids.add(id(p))     # its gradient comes from autograd, not from a HIP wgrad kernel
        return blocks

    def flatten_parameters(self):
        """(Re)build the flat parameter / gradient buffers on the parameters' current device, keeping values."""
        blocks = self._collect_blocks()
        total = 0
        for n, _ in blocks:
            total = -(-total // -n) * -n if n < 0 else total + n
        dev = next(self.parameters()).device
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self._grad_views = []
        self._grad_blocks = []
        self._lazy_zero = False
        with torch.no_grad():
            for n, views in blocks:
                if n < 0:
                    off = -(-off // -n) * -n
                    continue
                if views and not any(id(p) in self._torch_param_ids for p, *_ in views):
                    # written by the HIP gradient kernels, the whole block by one call (a PackedLinearGroup's heads are ONE
                    # GEMM): its parameters share the lazy-zero flag
                    blk = _GradBlock(off, off + n)
                    self._grad_blocks.append(blk)
                    for p, *_ in views:
                        p._grad_block = blk
                for p, o, size, stride in views:
                    v = flat.as_strided(size, stride, off + o)
                    v.copy_(p.detach().to(dev))
                    old_grad = p.grad
                    p.data = v
                    g = gflat.as_strided(size, stride, off + o)
                    if old_grad is not None:
                        g.copy_(old_grad.to(dev))
                    p.grad = g
                    self._grad_views.append((p, g))
                off += n
        self._flat_params, self._flat_grads = flat, gflat
        self._torch_span_cache = False
        from .. import kernels as _K
        _K.register_flat_buffers(flat, gflat)     # kernels.flat_grad_alias: bank gradients written in place
        self._torch_grad_views = [(p, g) for p, g in self._grad_views if id(p) in self._torch_param_ids]
        return flat, gflat

    def attach_grads(self):
        """Make sure every parameter's .grad is its view of the flat gradient buffer (an optimizer's
        zero_grad(set_to_none=True) drops them; None means zero, so the view is cleared)."""
        if getattr(self, "_flat_grads", None) is None:
            self.flatten_parameters()
            return
        for p, g in self._grad_views:
            if id(p) in self._torch_param_ids:
                continue        # autograd-managed: stays detached until gather_torch_grads() (see zero_grad)
            if p.grad is not g:
                with torch.no_grad():
                    if p.grad is None:
                        g.zero_()
                    else:
                        g.copy_(p.grad)
                p.grad = g

    def flat_range(self, prefix: str) -> slice:
        """Contiguous slice of the flat buffers that holds every parameter under sub-module `prefix`
        (experiment.py:157 optimises only ``getattr(model, update_parameters).parameters()``)."""
        base = self.flat_params.data_ptr()
        spans = sorted(((p.data_ptr() - base) // 4, p.numel()) for n, p in self.named_parameters()
                       if n.startswith(prefix + "."))
        if not spans:
            raise KeyError(prefix)
        lo, hi = spans[0][0], spans[-1][0] + spans[-1][1]
        for n, p in self.named_parameters():       # (padding inside the range is fine: zero values, zero gradients)
            if not n.startswith(prefix + ".") and lo <= (p.data_ptr() - base) // 4 < hi:
                raise RuntimeError(f"parameters of '{prefix}' are not one contiguous range of the flat buffer ('{n}' lies inside)")
        return slice(lo, hi)

    @property
    def flat_params(self):
        if getattr(self, "_flat_params", None) is None:
            self.flatten_parameters()
        return self._flat_params

    @property
    def flat_grads(self):
        """The flat gradient buffer, complete: autograd-produced gradients of torch-level parameters are moved in first."""
        if getattr(self, "_flat_grads", None) is None:
            self.flatten_parameters()
        self.gather_torch_grads()
        return self._flat_grads

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._flat_params = self._flat_grads = None
        if any(True for _ in self.parameters()):
            self.flatten_parameters()
        return out

    def zero_grad(self, set_to_none: bool = False, lazy: bool = False):
        """Gradients live in the flat buffer: zero it in one memset instead of dropping the views.

        lazy: no memset.  Every block of the buffer is only DECLARED zero: the first gradient kernel that writes a block
        overwrites it instead of accumulating, and blocks nobody wrote get their zeros when the buffer is next read
        (settle_grads, called by gather_torch_grads / ``flat_grads`` / FlatAdam.step / the gradient exchange).  In a step
        where every parameter receives a gradient (VanillaVAE, MCQ-VAE) the 16 MB fill launch disappears.  ``p.grad`` of a
        parameter that got no gradient is stale until then, which is why this is the training loop's option and not the default.

        Parameters of torch-level sub-modules (CausalTransition) get their gradient from autograd: with ``.grad``
        attached, AccumulateGrad adds into it with one tiny launch per parameter (62 per CT-MCQ-VAE step); detached,
        autograd just hands the tensor over and gather_torch_grads() moves all of them into the flat buffer with one
        multi-tensor copy."""
        from .. import kernels as _K
        _K.bump_param_epoch()          # a new step: transformed Winograd filters are remade once for all layers (kernels.wino_cache)
        if getattr(self, "_flat_grads", None) is not None:
            prezeroed = False
            if lazy and _K.LAZY_ZERO_GRAD:
                for blk in self._grad_blocks:
                    blk.fresh = True
                self._lazy_zero = True
                span = self._torch_span()
                if span is not None:     # the autograd-managed tail of the buffer (CT layer): ONE fill instead of one per bank
                    self._flat_grads[span[0]:span[1]].zero_()
                    prezeroed = True
            else:
                self._flat_grads.zero_()
                for blk in self._grad_blocks:
                    blk.fresh = False
                self._lazy_zero = False
                prezeroed = True
            self._torch_prezeroed = prezeroed
            for p, _ in getattr(self, "_torch_grad_views", ()):
                p.grad = None
            _K.note_zero_grad(prezeroed)
        else:
            super().zero_grad(set_to_none=set_to_none)

    def _torch_span(self):
        """[lo, hi) floats of the flat buffers that hold exactly the autograd-managed parameters (they are laid out behind every
        kernel-managed block, _collect_blocks), or None when there are none / some kernel-managed block lies inside."""
        sp = getattr(self, "_torch_span_cache", False)
        if sp is False:
            views = [g for _, g in getattr(self, "_torch_grad_views", ()) if g.is_contiguous()]
            sp = None
            if views and len(views) == len(self._torch_grad_views):
                lo = min(g.storage_offset() for g in views)
                hi = max(g.storage_offset() + g.numel() for g in views)
                if all(b.hi <= lo or b.lo >= hi for b in self._grad_blocks):
                    sp = (lo, hi)
            self._torch_span_cache = sp
        return sp

    def settle_grads(self):
        """Write the zeros that zero_grad(lazy=True) only declared, for every block no gradient kernel has written since
        (and for torch-level parameters autograd produced no gradient for): adjacent ranges share one fill."""
        if not getattr(self, "_lazy_zero", False):
            return
        self._lazy_zero = False
        ranges = []
        for blk in self._grad_blocks:
            if blk.fresh:
                blk.fresh = False
                ranges.append((blk.lo, blk.hi))
        with torch.no_grad():
            for p, g in getattr(self, "_torch_grad_views", ()):
                if p.grad is None and not getattr(self, "_torch_prezeroed", False):
                    if g.is_contiguous():
                        ranges.append((g.storage_offset(), g.storage_offset() + g.numel()))
                    else:
                        g.zero_()
            ranges.sort()
            merged = []
            for lo, hi in ranges:
                if merged and lo <= merged[-1][1] + 4:       # alignment gaps hold zeros anyway
                    merged[-1][1] = max(merged[-1][1], hi)
                else:
                    merged.append([lo, hi])
            for lo, hi in merged:
                self._flat_grads[lo:hi].zero_()

    def gather_torch_grads(self):
        """Move autograd-produced gradients of torch-level parameters into their views of the flat gradient buffer and
        re-attach the views (call before the optimizer step / gradient exchange).  Completes a lazy zero_grad first."""
        self.settle_grads()
        # (a gradient that a kernel wrote straight into the flat buffer -- kernels.flat_grad_alias -- is already in place)
        pairs = [(p, g) for p, g in getattr(self, "_torch_grad_views", ())
                 if p.grad is not None and p.grad is not g and p.grad.data_ptr() != g.data_ptr()]
        if pairs:
            with torch.no_grad():
                torch._foreach_copy_([g for _, g in pairs], [p.grad for p, _ in pairs])
        for p, g in getattr(self, "_torch_grad_views", ()):
            p.grad = g
