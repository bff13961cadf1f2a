"""Lightning-free counterpart of the reference's training harness (experiment.py:17-187).

``VAEXperiment`` keeps the reference's step semantics: batch unpacking ``(real_img, labels, *options)``,
``model(real_img, labels=..., **options)``, ``loss_function(*results, M_N=kld_weight)`` for training and
``M_N=1.0`` with a ``val_`` key prefix for validation, Adam(lr=LR, weight_decay) + ExponentialLR(gamma)
over ``model.parameters()`` or ``getattr(model, update_parameters).parameters()``.  What changes is the
machinery: one flat fused Adam launch, one bucketed RCCL all-reduce, and scalars fetched with ONE device
-> host copy every ``log_every`` steps instead of one ``.item()`` sync per key per step (experiment.py:95-96).
"""
import json
import sys
import time

import torch

from . import kernels as K
from .ddp import GradBucketAllReduce
from .optim import ExponentialLR, FlatAdam


def _graph_key(real_img, kwargs):
    """Key of a capturable step: the batch's shape plus, per option, a tensor's shape / dtype or the (one-per-batch) mode name.
    None when an option is neither (such a step stays eager)."""
    items = []
    for k in sorted(kwargs):
        v = kwargs[k]
        if torch.is_tensor(v):
            items.append((k, tuple(v.shape), v.dtype))
        elif k == "mode" and isinstance(v, (str, list)):
            items.append((k, v[0] if isinstance(v, list) else v))
        else:
            return None
    return (tuple(real_img.shape), real_img.dtype, tuple(items))


class _GraphedTrainStep:
    """zero_grad + forward + loss + backward (+ Adam when there is no gradient exchange) of one batch signature, captured
    once into a hipGraph and replayed: ~130 launches per VanillaVAE step -- ~400 per CT-MCQ-VAE step -- are host-bound when
    issued eagerly (VanillaVAE bs=256: 3.5 vs 1.8 ms; CT-MCQ-VAE at the YAML's 16 pairs per GPU: 7.5 vs 3.4 ms).
    The first WARM batches of a signature run eagerly as ordinary training steps; capture itself executes nothing, so the
    training trajectory is exactly the eager one.  A signature = input shape + the shapes of the tensor options (input_y,
    action) + the mode (datasets/transition.py hands out one mode per batch): the CT-MCQ-VAE modes each get their own graph
    (none of them reads a device value on the host any more)."""

    WARM = 3

    def __init__(self, exp, real_img, kwargs):
        self.exp = exp
        self.x = K.staging_like(real_img)            # channels_last: the hand-over below is the layout conversion too
        self.static = {k: torch.empty_like(v) for k, v in kwargs.items() if torch.is_tensor(v)}
        self.const = {k: v for k, v in kwargs.items() if not torch.is_tensor(v)}
        self.seen = 0
        self.graph = None
        self.losses = None

    def _body(self):
        exp = self.exp
        exp.model.zero_grad(lazy=True)            # no fill launch: first writers overwrite, settle_grads() fills what nobody wrote
        opts = {**self.static, **self.const}
        labels = opts.pop("labels", None)         # ConditionalVAE reads them (cvae.py:123); a static buffer like every tensor option
        results = exp.forward(self.x, labels=labels, **opts)
        losses = exp.model.loss_function(*results, M_N=exp.params['kld_weight'], optimizer_idx=0, batch_idx=0)
        K.backward(losses['loss'])
        # inside the capture: the fills of unwritten blocks AND the copies of autograd-produced gradients (CT layer: a_dense, mask,
        # positional encoding, the GATv2 vectors) into the flat buffer belong to the replayed step.  Left to ddp.all_reduce() /
        # optimizer.step() outside the graph they would run once -- after the capture step Python sees p.grad already attached to
        # its flat view and copies nothing, while every replay rewrites the graph-pool tensors autograd produced: from the second
        # replay on those parameters would be exchanged and stepped with a zero gradient
        exp.model.gather_torch_grads()
        if exp.ddp is None:
            exp.optimizer.step()
        # detached: a live loss keeps the step's autograd graph -- and with it the AccumulateGrad nodes of the parameters
        # torch accumulates itself (the CT layer's banks), bound to the stream they were made on -- alive into the next
        # signature's capture, where running them on that other stream ends the capture with a fault
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in losses.items()}

    def run(self, real_img, kwargs=None):
        exp = self.exp
        K.stage_batch(self.x, real_img)
        for k, t in self.static.items():
            t.copy_(kwargs[k], non_blocking=True)
        if self.graph is None and self.seen >= self.WARM:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with K.capture_graph(g):
                self.losses = self._body()
            self.graph = g
        if self.graph is not None:
            self.graph.replay()
            if exp.ddp is None:
                K.bump_param_epoch()       # the replayed Adam launch changed the parameters behind Python's back
            losses = self.losses
        else:
            losses = self._body()
        self.seen += 1
        if exp.ddp is not None:
            exp.ddp.all_reduce()
            exp.optimizer.step(grad_scale=exp.ddp.grad_scale)
        exp.global_step += 1
        return losses


class VAEXperiment:

    def __init__(self, vae_model, params: dict, ddp: GradBucketAllReduce = None, log_every: int = 50, log_file=None):
        self.model = vae_model
        self.params = params
        self.ddp = ddp
        self.curr_device = None
        self.log_every = log_every
        self.log_file = log_file
        self.global_step = 0
        self.optimizer, self.scheduler = self.configure_optimizers()
        if ddp is not None and "update_parameters" in params:
            ddp.restrict(self.optimizer.slice)
        self._graphed = {}          # (shape, dtype) -> _GraphedTrainStep

    def forward(self, input, **kwargs):
        return self.model(input, **kwargs)

    # -- steps (experiment.py:44-74) ---------------------------------------------------------------------
    def _unpack(self, batch):
        real_img, labels, *args = batch
        kwargs = {} if len(args) < 1 or type(args[0]) != dict else args[0]
        return real_img, labels, kwargs

    def training_step(self, batch, batch_idx, optimizer_idx=0):
        real_img, labels, kwargs = self._unpack(batch)
        self.curr_device = real_img.device
        results = self.forward(real_img, labels=labels, **kwargs)
        train_loss = self.model.loss_function(*results, M_N=self.params['kld_weight'], optimizer_idx=optimizer_idx,
                                              batch_idx=batch_idx)
        self.log_all(train_loss, batch_size=real_img.size(0), validation=False)
        return train_loss['loss']

    def validation_step(self, batch, batch_idx, optimizer_idx=0):
        real_img, labels, kwargs = self._unpack(batch)
        self.curr_device = real_img.device
        with torch.no_grad():
            results = self.forward(real_img, labels=labels, **kwargs)
            val_loss = self.model.loss_function(*results, M_N=1.0, optimizer_idx=optimizer_idx, batch_idx=batch_idx)
        return self.log_all(val_loss, batch_size=real_img.size(0), validation=True, force=True)

    def metric_func(self, x):
        x = x.to(next(self.model.parameters()).device)
        x = self.model.encode(x)[0]
        return x.reshape(x.size(0), -1)

    def log_all(self, losses: dict, batch_size, validation: bool = False, force: bool = False):
        """Scalar tensors only (strings / images are dropped like experiment.py:93-106); one fused all-reduce over
        ranks (sync_dist=True) and one D2H copy."""
        if not force and (self.global_step % self.log_every) != 0:
            return None
        prefix = "val_" if validation else ""
        scal = {prefix + k: v for k, v in losses.items()
                if isinstance(v, torch.Tensor) and (v.dim() == 0 or (v.dim() == 1 and v.size(0) == 1))}
        if self.ddp is not None:
            scal = self.ddp.reduce_scalars(scal)
        keys = sorted(scal)
        vals = torch.stack([scal[k].detach().float().reshape(()) for k in keys]).cpu().tolist() if keys else []
        rec = dict(zip(keys, vals))
        rec["step"] = self.global_step
        if self.log_file is not None:
            self.log_file.write(json.dumps(rec) + "\n")
            self.log_file.flush()
        return rec

    # -- optimisation (experiment.py:152-187) ---------------------------------------------------------------
    def configure_optimizers(self):
        sl = None
        if "update_parameters" in self.params:
            sl = self.model.flat_range(self.params["update_parameters"])
        opt = FlatAdam(self.model, lr=self.params['LR'], weight_decay=self.params.get('weight_decay', 0.0), params_slice=sl)
        sched = None
        if self.params.get('scheduler_gamma') is not None:
            sched = ExponentialLR(opt, self.params['scheduler_gamma'])
        return opt, sched

    # -- full resume (run.py:85-101: trainer_params.resume_from_checkpoint hands Lightning the optimizer, scheduler, epoch) ------
    def state_dict(self):
        """Everything beside the model's own state_dict that a run needs to continue exactly where it stopped: FlatAdam's
        moments and device state vector (step, lr, betas, eps, weight decay, beta^t), the scheduler's epoch, the global step
        and the random streams (torch CPU / device generators, the in-kernel Philox state of the latent noise).  Plain tensors
        and numbers only, so ``torch.load(..., weights_only=True)`` reads it back."""
        opt = self.optimizer.state_dict()
        sd = {"global_step": int(self.global_step),
              "optimizer": {"exp_avg": opt["exp_avg"].detach().cpu(), "exp_avg_sq": opt["exp_avg_sq"].detach().cpu(),
                            "state": opt["state"][:8].detach().cpu(), "lr": float(self.optimizer.lr),
                            "slice": [int(self.optimizer.slice.start or 0), int(self.optimizer.slice.stop)]},
              "torch_rng": torch.get_rng_state()}
        if self.scheduler is not None:
            sd["scheduler"] = {"epoch": int(self.scheduler.epoch), "base_lr": float(self.scheduler.base_lr),
                               "gamma": float(self.scheduler.gamma)}
        dev = next(self.model.parameters()).device
        if dev.type == "cuda":
            sd["device_rng"] = torch.cuda.get_rng_state(dev)
        rng = getattr(self.model, "_rng_state", None)
        if rng is not None:
            sd["model_rng"] = rng.detach().cpu()
        return sd

    def load_state_dict(self, sd):
        """Inverse of state_dict(), in place (captured steps, if any, keep pointing at the same buffers)."""
        o = sd["optimizer"]
        if [int(self.optimizer.slice.start or 0), int(self.optimizer.slice.stop)] != [int(v) for v in o["slice"]]:
            raise RuntimeError("checkpoint optimizes another parameter range than this run (update_parameters differs)")
        self.optimizer.load_state_dict(o)
        self.optimizer.lr = float(o["lr"])
        if self.scheduler is not None and "scheduler" in sd:
            self.scheduler.epoch = int(sd["scheduler"]["epoch"])
            self.scheduler.base_lr = float(sd["scheduler"]["base_lr"])
        self.global_step = int(sd["global_step"])
        torch.set_rng_state(sd["torch_rng"].cpu())
        dev = next(self.model.parameters()).device
        if "device_rng" in sd and dev.type == "cuda":
            torch.cuda.set_rng_state(sd["device_rng"].cpu(), dev)
        if "model_rng" in sd:
            self.model._rng_state = sd["model_rng"].to(dev)
        K.bump_param_epoch()

    def optimizer_step(self):
        scale = 1.0
        if self.ddp is not None:
            self.ddp.all_reduce()
            scale = self.ddp.grad_scale
        self.optimizer.step(grad_scale=scale)
        self.global_step += 1

    def fit(self, train_batches, val_batches=None, max_epochs=1, on_epoch_end=None, start_epoch=0):
        """train_batches / val_batches: callables returning an iterable of batches for one epoch.  start_epoch: first epoch
        to run (a resumed run continues at the checkpoint's epoch + 1; max_epochs counts from 0 as Lightning's does)."""
        history = []
        for epoch in range(start_epoch, max_epochs):
            self.model.train()
            t0 = time.time()
            n = 0
            for i, batch in enumerate(train_batches()):
                real_img, _labels, kwargs = self._unpack(batch)
                # graph_safe = False: the model's step depends on host state that changes per call (e.g. BetaVAE type 'B':
                # the capacity C follows the loss-call counter), so a captured step would freeze it
                if getattr(self.model, 'uses_labels', False) and torch.is_tensor(_labels):
                    kwargs = {**kwargs, "labels": _labels.to(real_img.device)}      # part of the step's signature and inputs
                key = _graph_key(real_img, kwargs) if (self.params.get('hipgraph', True) and real_img.is_cuda
                                                       and getattr(self.model, 'graph_safe', True)) else None
                if key is not None:
                    gs = self._graphed.get(key)
                    if gs is None:
                        gs = self._graphed[key] = _GraphedTrainStep(self, real_img, kwargs)
                    self.curr_device = real_img.device
                    losses = gs.run(real_img, kwargs)
                    self.global_step -= 1                  # log_all keys on the step that just ran
                    self.log_all(losses, batch_size=real_img.size(0), validation=False)
                    self.global_step += 1
                else:
                    self.model.zero_grad(lazy=True)
                    loss = self.training_step(batch, i)
                    K.backward(loss)
                    self.model.settle_grads()
                    self.optimizer_step()
                n += batch[0].size(0)
            if self.scheduler is not None:
                self.scheduler.step()                      # Lightning steps ExponentialLR once per epoch
            rec = {"epoch": epoch, "train_images": n, "epoch_seconds": time.time() - t0}
            if val_batches is not None:
                self.model.eval()
                sums, cnt = {}, 0
                for i, batch in enumerate(val_batches()):
                    r = self.validation_step(batch, i)
                    for k, v in r.items():
                        if k != "step":
                            sums[k] = sums.get(k, 0.0) + v
                    cnt += 1
                rec.update({k: v / max(cnt, 1) for k, v in sums.items()})
            history.append(rec)
            if on_epoch_end is not None:
                on_epoch_end(epoch, rec)
        return history
