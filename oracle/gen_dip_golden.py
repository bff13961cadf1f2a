#!/usr/bin/env python3
"""Generate tests/golden/dip_b4.npz from the REFERENCE's own ``models/dip_vae.py``.  TEST INFRASTRUCTURE ONLY.
Method as in gen_iw_golden.py (whose loader it uses); model parameters of configs/dip_vae.yaml.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_dip_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_iw_golden import OUT, cks, load  # noqa: E402


def main():
    from ctvae_amd import filler
    (mod,) = load(["dip_vae"])
    seed, B, M_N = 1265, 4, 0.00025
    torch.manual_seed(0)
    model = mod.DIPVAE(in_channels=3, latent_dim=128, lambda_diag=0.05, lambda_offdiag=0.1)
    model.load_state_dict(filler.fill_state(filler.specs_of(model), seed + 1))
    model.train()
    x, eps = filler.synthetic_batch(seed, B)
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: eps.clone()
    try:
        res = model(x)
    finally:
        torch.randn_like = orig
    losses = model.loss_function(*res, M_N=M_N)
    losses["loss"].backward()
    out = {"seed": np.int64(seed), "B": np.int64(B), "M_N": np.float64(M_N), "mu": res[2].detach().numpy().copy(),
           "recons_cks": cks(res[0]), "grad.fc_mu.bias": model.fc_mu.bias.grad.numpy().copy(),
           "grad.fc_var.bias": model.fc_var.bias.grad.numpy().copy()}
    for k, v in losses.items():
        out["loss." + k] = np.float64(v.item())
    for k, p in model.named_parameters():
        out["gradcks." + k] = cks(p.grad)
    np.savez_compressed(os.path.join(OUT, f"dip_b{B}.npz"), **out)
    print({k: float(v) for k, v in losses.items()})


if __name__ == "__main__":
    main()
