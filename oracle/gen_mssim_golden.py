#!/usr/bin/env python3
"""Generate tests/golden/mssim_b4.npz from the REFERENCE's own ``models/mssim_vae.py``.  TEST INFRASTRUCTURE ONLY.
Method as in gen_iw_golden.py (whose loader it uses): the module is exec'd where it lies, the weights come from the build's
filler (MSSIMVAE has VanillaVAE's state_dict; its MSSIM module has no parameters), the N(0,1) noise of ``reparameterize`` is
injected by patching ``torch.randn_like`` (the eps of ``filler.synthetic_batch``).  Also records the loss module on its own for a second pair of tensors (one of them
requiring a gradient), with the full gradient, so that the kernel is pinned without the network in front of it.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_mssim_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_iw_golden import OUT, cks, load  # noqa: E402


def main():
    from ctvae_amd import filler
    (mod,) = load(["mssim_vae"])
    seed, B, M_N = 1265, 4, 0.00025
    torch.manual_seed(0)
    model = mod.MSSIMVAE(in_channels=3, latent_dim=128)
    specs = filler.specs_of(model)
    model.load_state_dict(filler.fill_state(specs, seed + 1))
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
    out = {"seed": np.int64(seed), "B": np.int64(B), "M_N": np.float64(M_N), "keys": np.array([k for k, _, _ in specs]),
           "mu": res[2].detach().numpy().copy(), "recons_cks": cks(res[0]), "recons_sub": res[0].detach()[:, :, ::8, ::8].numpy().copy(),
           "grad.fc_mu.bias": model.fc_mu.bias.grad.numpy().copy(), "grad.final_layer.3.bias": model.final_layer[3].bias.grad.numpy().copy()}
    for k, v in losses.items():
        out["loss." + k] = np.float64(v.item())
    for k, p in model.named_parameters():
        out["gradcks." + k] = cks(p.grad)
    # the loss module alone: a smooth picture pair in [-1, 1] (MS-SSIM of unrelated noise is negative at the coarse levels and
    # its fractional powers are NaN in the reference too)
    g = torch.Generator().manual_seed(seed + 11)
    base = torch.tanh(torch.nn.functional.interpolate(torch.randn(3, 3, 8, 8, generator=g), size=64, mode="bilinear", align_corners=False))
    a = (base + 0.05 * torch.randn(3, 3, 64, 64, generator=g)).clamp(-1, 1).requires_grad_(True)
    b = (base + 0.05 * torch.randn(3, 3, 64, 64, generator=g)).clamp(-1, 1)
    val = model.mssim_loss(a, b)
    val.backward()
    out.update({"pair.a": a.detach().numpy().copy(), "pair.b": b.numpy().copy(), "pair.loss": np.float64(val.item()),
                "pair.grad_a": a.grad.numpy().copy()})
    np.savez_compressed(os.path.join(OUT, f"mssim_b{B}.npz"), **out)
    print({k: float(v) for k, v in losses.items()}, float(val))


if __name__ == "__main__":
    main()
