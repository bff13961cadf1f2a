#!/usr/bin/env python3
"""Generate tests/golden/vamp_b4.npz from the REFERENCE's own ``models/vampvae.py``.  TEST INFRASTRUCTURE ONLY.
Method as in gen_iw_golden.py (whose loader it uses); ``torch.randn_like`` patched with the injected Gaussian draws.  The
module moves its identity matrix with ``.cuda(device)`` inside ``loss_function`` (vampvae.py:150); the generator runs on the
CPU, so that one attribute is replaced by an object whose ``.cuda()`` returns the tensor unchanged -- nothing else is touched.
The embed_pseudo bias is shifted by +0.5 after the filler so that the Hardtanh(0, 1) has values on both of its flat sides and
in between (the filler's +-0.1 biases with +-0.24 weights would leave most of them clamped at 0).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_vamp_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_iw_golden import OUT, cks, load  # noqa: E402


class _StaysHere(torch.Tensor):
    def cuda(self, *a, **k):
        return self.as_subclass(torch.Tensor)


def main():
    from ctvae_amd import filler
    (mod,) = load(["vampvae"])
    seed, B, M_N = 1265, 4, 0.00025
    torch.manual_seed(0)
    model = mod.VampVAE(in_channels=3, latent_dim=128)
    specs = filler.specs_of(model)
    sd = filler.fill_state(specs, seed + 1)
    sd["embed_pseudo.0.bias"] = sd["embed_pseudo.0.bias"] + 0.5
    model.load_state_dict(sd)
    model.pseudo_input = model.pseudo_input.as_subclass(_StaysHere)
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
    out = {"seed": np.int64(seed), "B": np.int64(B), "M_N": np.float64(M_N), "keys": np.array([k for k, _, _ in specs]),
           "mu": res[2].detach().numpy().copy(), "z": res[4].detach().numpy().copy(), "recons_cks": cks(res[0]),
           "grad.embed_pseudo.0.bias_sub": model.embed_pseudo[0].bias.grad[::64].numpy().copy(),
           "grad.fc_var.bias": model.fc_var.bias.grad.numpy().copy()}
    for k, v in losses.items():
        out["loss." + k] = np.float64(v.item())
    for k, p in model.named_parameters():
        out["gradcks." + k] = cks(p.grad)
    for k, b in model.named_buffers():
        if "running" in k:
            out["buf." + k] = cks(b)
    np.savez_compressed(os.path.join(OUT, f"vamp_b{B}.npz"), **out)
    print({k: float(v) for k, v in losses.items()})


if __name__ == "__main__":
    main()
