#!/usr/bin/env python3
"""Generate tests/golden/twostage_b2.npz from the REFERENCE's own ``models/twostage_vae.py``.  TEST INFRASTRUCTURE ONLY.
Method as in gen_iw_golden.py (whose loader it uses).  The fixture pins what the class does in a training step -- the first
stage only -- and that the second-stage parameters receive NO gradient (twostage_vae.py:137-165).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_twostage_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_iw_golden import OUT, cks, load  # noqa: E402


def main():
    from ctvae_amd import filler
    (mod,) = load(["twostage_vae"])
    seed, B, M_N = 1265, 2, 0.00025
    torch.manual_seed(0)
    model = mod.TwoStageVAE(in_channels=3, latent_dim=128)
    specs = filler.specs_of(model)
    model.load_state_dict(filler.fill_state(specs, seed + 1))
    model.train()
    x, e = filler.synthetic_batch(seed, B)
    o1 = torch.randn_like
    torch.randn_like = lambda t, **kw: e.clone()
    try:
        res = model(x)
    finally:
        torch.randn_like = o1
    losses = model.loss_function(*res, M_N=M_N)
    losses["loss"].backward()
    out = {"seed": np.int64(seed), "B": np.int64(B), "M_N": np.float64(M_N), "mu": res[2].detach().numpy().copy(),
           "recons_cks": cks(res[0]), "keys": np.array([k for k, _, _ in specs]),
           "shapes": np.array([str(tuple(sh)) for _, sh, _ in specs]),
           "no_grad": np.array([k for k, p in model.named_parameters() if p.grad is None])}
    for k, v in losses.items():
        out["loss." + k] = np.float64(v.item())
    for k, p in model.named_parameters():
        if p.grad is not None:
            out["gradcks." + k] = cks(p.grad)
    np.savez_compressed(os.path.join(OUT, f"twostage_b{B}.npz"), **out)
    print({k: float(v) for k, v in losses.items()}, len(out["no_grad"]), "parameters without gradient")


if __name__ == "__main__":
    main()
