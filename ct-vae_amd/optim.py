"""Flat fused Adam (+ ExponentialLR) over a model's packed parameter buffer.

Same update rule as ``torch.optim.Adam(lr, weight_decay)`` with default betas/eps, which is what the
reference's harness builds (experiment.py:158-160); the schedule is ``ExponentialLR(gamma)`` stepped once
per epoch (experiment.py:173-175).  One kernel launch per step for the whole model, hyper-parameters and the
step counter live in a small device tensor so the launch is hipGraph-capturable.

Deviations from ``torch.optim.Adam`` (documented, not pinned by a fixture):
* ONE step counter for the whole buffer, and every element is updated every step.  torch keeps a step per parameter and
  skips parameters whose ``.grad`` is None.  The two differ only for parameters that receive NO gradient in some steps --
  in this repository CT-MCQ-VAE's per-action ``graph_discovers`` whose action is absent from a batch, ``ct_layer.mask`` in
  base-mode steps, the decoder in causal-mode steps: here they see a zero gradient (first moment decays, the parameter keeps
  drifting by its momentum, bias correction follows the global count), in torch they are frozen for that step.  VanillaVAE /
  MCQ-VAE (every parameter gets a gradient every step) are unaffected: tests/test_ct_gpu.py pins their 3-step trajectory.
* beta1^t / beta2^t are accumulated in fp32 on the device (state[6..7]); torch computes them in double on the host.  After
  10^4 steps the relative difference of the bias corrections is < 1e-4 (beta2^t has decayed to 4.5e-5 by then).
"""
import torch

from . import kernels as K


class FlatAdam:
    def __init__(self, model, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, params_slice=None):
        self.model = model
        flat = model.flat_params
        self.slice = params_slice if params_slice is not None else slice(0, flat.numel())
        n = flat[self.slice].numel()
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=flat.device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=flat.device)
        self.state = K.adam_state([0.0, lr, betas[0], betas[1], eps, weight_decay, 1.0, 1.0], flat.device)
        self.lr = lr

    def set_lr(self, lr):
        self.lr = lr
        self.state[1:2].fill_(lr)

    def step(self, grad_scale=1.0):
        self.model.gather_torch_grads()
        K.adam_step(self.model.flat_params[self.slice], self.model.flat_grads[self.slice], self.exp_avg, self.exp_avg_sq,
                    self.state, grad_scale)

    def zero_grad(self, set_to_none=False):
        self.model.zero_grad()

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "state": self.state}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.state[:8].copy_(sd["state"][:8])      # (states saved before the ticket word existed have 8 entries)


class ExponentialLR:
    """lr_epoch = lr0 * gamma**epoch (torch.optim.lr_scheduler.ExponentialLR), stepped once per epoch."""

    def __init__(self, optimizer: FlatAdam, gamma: float):
        self.opt, self.gamma, self.base_lr, self.epoch = optimizer, gamma, optimizer.lr, 0

    def step(self):
        self.epoch += 1
        self.opt.set_lr(self.base_lr * self.gamma ** self.epoch)
