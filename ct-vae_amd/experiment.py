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

from .ddp import GradBucketAllReduce
from .optim import ExponentialLR, FlatAdam


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

    def optimizer_step(self):
        scale = 1.0
        if self.ddp is not None:
            self.ddp.all_reduce()
            scale = self.ddp.grad_scale
        self.optimizer.step(grad_scale=scale)
        self.global_step += 1

    def fit(self, train_batches, val_batches=None, max_epochs=1, on_epoch_end=None):
        """train_batches / val_batches: callables returning an iterable of batches for one epoch."""
        history = []
        for epoch in range(max_epochs):
            self.model.train()
            t0 = time.time()
            n = 0
            for i, batch in enumerate(train_batches()):
                self.model.zero_grad()
                loss = self.training_step(batch, i)
                loss.backward()
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
