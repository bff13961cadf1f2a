#!/usr/bin/env python3
"""Generate tests/golden/{wae_imq,wae_rbf,infovae}_b4.npz from the REFERENCE's own ``models/wae_mmd.py`` /
``models/info_vae.py``.  TEST INFRASTRUCTURE ONLY.

Method as in gen_iw_golden.py (whose loader it uses).  ``torch.randn_like`` is patched with a queue: InfoVAE draws the
reparameterisation noise first, then the prior samples of ``compute_mmd``; WAE_MMD only the latter.  Model parameters:
configs/wae_mmd_imq.yaml, configs/wae_mmd_rbf.yaml (reg_weight 5000, kernel 'rbf'), configs/infovae.yaml.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_mmd_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_iw_golden import OUT, cks, load  # noqa: E402


def prior_draws(seed, B, L=128):
    """Injected prior samples (the rule the tests repeat: tests/helpers.py::mmd_prior)."""
    return torch.randn(B, L, generator=torch.Generator().manual_seed(seed + 4))


def main():
    from ctvae_amd import filler
    wae, info = load(["wae_mmd", "info_vae"])
    seed, B, M_N = 1265, 4, 0.00025
    cases = (("wae_imq", wae.WAE_MMD, dict(in_channels=3, latent_dim=128, reg_weight=100, kernel_type='imq'), False),
             ("wae_rbf", wae.WAE_MMD, dict(in_channels=3, latent_dim=128, reg_weight=5000, kernel_type='rbf'), False),
             ("infovae", info.InfoVAE, dict(in_channels=3, latent_dim=128, reg_weight=110, kernel_type='imq', alpha=-9.0, beta=10.5), True))
    for tag, cls, cfg, gaussian in cases:
        torch.manual_seed(0)
        model = cls(**cfg)
        model.load_state_dict(filler.fill_state(filler.specs_of(model), seed + 1))
        model.train()
        x, eps = filler.synthetic_batch(seed, B)
        queue = ([eps] if gaussian else []) + [prior_draws(seed, B)]
        orig = torch.randn_like
        torch.randn_like = lambda t, **kw: queue.pop(0).clone()
        try:
            res = model(x)
            losses = model.loss_function(*res, M_N=M_N)
        finally:
            torch.randn_like = orig
        assert not queue
        losses["loss"].backward()
        out = {"seed": np.int64(seed), "B": np.int64(B), "M_N": np.float64(M_N), "z": res[2].detach().numpy().copy(),
               "recons_cks": cks(res[0])}
        head = "fc_mu" if gaussian else "fc_z"
        out[f"grad.{head}.bias"] = getattr(model, head).bias.grad.numpy().copy()
        for k, v in losses.items():
            out["loss." + k] = np.float64(v.item())
        for k, p in model.named_parameters():
            out["gradcks." + k] = cks(p.grad)
        np.savez_compressed(os.path.join(OUT, f"{tag}_b{B}.npz"), **out)
        print(tag, {k: float(v) for k, v in losses.items()})


if __name__ == "__main__":
    main()
