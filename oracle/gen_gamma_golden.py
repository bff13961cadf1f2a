#!/usr/bin/env python3
"""Generate tests/golden/gamma_b4.npz from the REFERENCE's own ``models/gamma_vae.py``.  TEST INFRASTRUCTURE ONLY.
Method as in gen_iw_golden.py (whose loader it uses).  The module draws zhat ~ Gamma(alpha + B, 1) with
torch.distributions.Gamma inside ``reparameterize``; the generator seeds torch's global generator right before the forward
pass, lets the reference draw, and records the draw it made (the name ``Gamma`` in the module's namespace is wrapped by a
subclass that keeps its last sample) so that the tests can inject exactly it.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_gamma_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_iw_golden import OUT, cks, load  # noqa: E402


def main():
    from ctvae_amd import filler
    (mod,) = load(["gamma_vae"])
    seed, B = 1265, 4
    drawn = []

    class Recording(mod.Gamma):
        def sample(self, *a, **k):
            s = super().sample(*a, **k)
            drawn.append(s.clone())
            return s
    mod.Gamma = Recording
    torch.manual_seed(0)
    model = mod.GammaVAE(in_channels=3, latent_dim=128, gamma_shape=8., prior_shape=2., prior_rate=1.)
    specs = filler.specs_of(model)
    model.load_state_dict(filler.fill_state(specs, seed + 1))
    model.train()
    x, _ = filler.synthetic_batch(seed, B)
    torch.manual_seed(seed + 7)
    res = model(x)
    losses = model.loss_function(*res, M_N=0.00025)
    losses["loss"].backward()
    out = {"seed": np.int64(seed), "B": np.int64(B), "keys": np.array([k for k, _, _ in specs]), "zhat": drawn[0].numpy().copy(),
           "alpha": res[2].detach().numpy().copy(), "beta": res[3].detach().numpy().copy(), "recons_cks": cks(res[0]),
           "recons_sub": res[0].detach()[:, :, ::8, ::8].numpy().copy(),
           "grad.fc_mu.0.bias": model.fc_mu[0].bias.grad.numpy().copy(), "grad.fc_var.0.bias": model.fc_var[0].bias.grad.numpy().copy()}
    for k, v in losses.items():
        out["loss." + k] = np.float64(v.item())
    for k, p in model.named_parameters():
        out["gradcks." + k] = cks(p.grad)
    np.savez_compressed(os.path.join(OUT, f"gamma_b{B}.npz"), **out)
    print({k: float(v) for k, v in losses.items()}, len(drawn))


if __name__ == "__main__":
    main()
