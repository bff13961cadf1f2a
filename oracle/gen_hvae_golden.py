#!/usr/bin/env python3
"""Generate tests/golden/hvae_b4.npz from the REFERENCE's own ``models/hvae.py``.  TEST INFRASTRUCTURE ONLY.
Method as in gen_iw_golden.py (whose loader it uses); ``torch.randn_like`` hands out the injected draws in the order the
module asks for them: z2's first (hvae.py:146), then z1's (:182).  Parameters: configs/hvae.yaml.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_hvae_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_iw_golden import OUT, cks, load  # noqa: E402


def main():
    from ctvae_amd import filler
    (mod,) = load(["hvae"])
    seed, B, M_N = 1265, 4, 0.00025
    torch.manual_seed(0)
    model = mod.HVAE(in_channels=3, latent1_dim=64, latent2_dim=64, pseudo_input_size=128)
    specs = filler.specs_of(model)
    model.load_state_dict(filler.fill_state(specs, seed + 1))
    model.train()
    x, _ = filler.synthetic_batch(seed, B)
    g = torch.Generator().manual_seed(seed + 6)
    e1, e2 = torch.randn(B, 64, generator=g), torch.randn(B, 64, generator=g)
    draws = iter([e2, e1])
    o1 = torch.randn_like
    torch.randn_like = lambda t, **kw: next(draws).clone()
    try:
        res = model(x)
    finally:
        torch.randn_like = o1
    losses = model.loss_function(*res, M_N=M_N)
    losses["loss"].backward()
    out = {"seed": np.int64(seed), "B": np.int64(B), "M_N": np.float64(M_N), "keys": np.array([k for k, _, _ in specs]),
           "z1_mu": res[2].detach().numpy().copy(), "z2_mu": res[4].detach().numpy().copy(), "z1": res[6].detach().numpy().copy(),
           "recons_cks": cks(res[0]), "recons_sub": res[0].detach()[:, :, ::8, ::8].numpy().copy(),
           "grad.recons_z1_mu.bias": model.recons_z1_mu.bias.grad.numpy().copy(),
           "grad.fc_z2_var.bias": model.fc_z2_var.bias.grad.numpy().copy()}
    for k, v in losses.items():
        out["loss." + k] = np.float64(v.item())
    for k, p in model.named_parameters():
        out["gradcks." + k] = cks(p.grad)
    np.savez_compressed(os.path.join(OUT, f"hvae_b{B}.npz"), **out)
    print({k: float(v) for k, v in losses.items()})


if __name__ == "__main__":
    main()
