#!/usr/bin/env python3
"""Generate tests/golden/lvae_b4.npz from the REFERENCE's own ``models/lvae.py``.  TEST INFRASTRUCTURE ONLY.
Method as in gen_iw_golden.py (whose loader it uses); ``torch.randn_like`` hands out the injected draws in the order the module
asks for them (top latent first, then one per rung, tests/helpers.py::lvae_noise).  Parameters: configs/lvae.yaml.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_lvae_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_iw_golden import OUT, cks, load  # noqa: E402


def main():
    from ctvae_amd import filler
    (mod,) = load(["lvae"])
    seed, B, M_N = 1265, 4, 0.00025
    dims = [4, 8, 16, 32, 128]
    torch.manual_seed(0)
    model = mod.LVAE(in_channels=3, latent_dims=list(dims), hidden_dims=[32, 64, 128, 256, 512])
    specs = filler.specs_of(model)
    model.load_state_dict(filler.fill_state(specs, seed + 1))
    model.train()
    x, _ = filler.synthetic_batch(seed, B)
    g = torch.Generator().manual_seed(seed + 8)
    order = [dims[-1]] + [dims[i - 1] for i in range(len(dims) - 1, 0, -1)]
    draws = iter([torch.randn(B, d, generator=g) for d in order])
    o1 = torch.randn_like
    torch.randn_like = lambda t, **kw: next(draws).clone()
    try:
        res = model(x)
    finally:
        torch.randn_like = o1
    losses = model.loss_function(*res, M_N=M_N)
    losses["loss"].backward()
    out = {"seed": np.int64(seed), "B": np.int64(B), "M_N": np.float64(M_N), "keys": np.array([k for k, _, _ in specs]),
           "kl_div": res[2].detach().numpy().copy(), "recons_cks": cks(res[0]),
           "recons_sub": res[0].detach()[:, :, ::8, ::8].numpy().copy(),
           "grad.ladders.0.fc_var.bias": model.ladders[0].fc_var.bias.grad.numpy().copy(),
           "grad.encoders.2.encoder_mu.bias": model.encoders[2].encoder_mu.bias.grad.numpy().copy()}
    for k, v in losses.items():
        out["loss." + k] = np.float64(v.item())
    for k, p in model.named_parameters():
        out["gradcks." + k] = cks(p.grad) if p.grad is not None else np.zeros(3)
    for k, b in model.named_buffers():
        if "running" in k:
            out["buf." + k] = cks(b)
    np.savez_compressed(os.path.join(OUT, f"lvae_b{B}.npz"), **out)
    print({k: float(v) for k, v in losses.items()}, [k for k, p in model.named_parameters() if p.grad is None])


if __name__ == "__main__":
    main()
