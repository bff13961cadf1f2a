#!/usr/bin/env python3
"""Generate tests/golden/logcosh_b2.npz from the REFERENCE's own ``models/logcosh_vae.py``.  TEST INFRASTRUCTURE ONLY.

Method as in gen_iw_golden.py (whose loader it uses).  Model parameters: configs/logcosh_vae.yaml (alpha 10, beta 1).
Records mu / log_var, the loss dict and gradient checksums of every parameter plus the full gradient of fc_mu.bias and
final_layer.3.weight.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_logcosh_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_iw_golden import OUT, cks, load  # noqa: E402


def main():
    from ctvae_amd import filler
    (mod,) = load(["logcosh_vae"])
    seed, B, M_N = 1265, 2, 0.00025
    torch.manual_seed(0)
    model = mod.LogCoshVAE(in_channels=3, latent_dim=128, alpha=10.0, beta=1.0)
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
           "log_var": res[3].detach().numpy().copy(), "recons_cks": cks(res[0]),
           "grad.fc_mu.bias": model.fc_mu.bias.grad.numpy().copy(),
           "grad.final_layer.3.weight": model.final_layer[3].weight.grad.numpy().copy()}
    for k in ("loss", "Reconstruction_Loss", "KLD"):
        out["loss." + k] = np.float64(losses[k].item())
    for k, p in model.named_parameters():
        out["gradcks." + k] = cks(p.grad)
    np.savez_compressed(os.path.join(OUT, f"logcosh_b{B}.npz"), **out)
    print({k: float(out["loss." + k]) for k in ("loss", "Reconstruction_Loss", "KLD")})


if __name__ == "__main__":
    main()
