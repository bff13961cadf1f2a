#!/usr/bin/env python3
"""Generate tests/golden/betatc_b8.npz from the REFERENCE's own ``models/betatc_vae.py``.  TEST INFRASTRUCTURE ONLY.
Method as in gen_iw_golden.py (whose loader it uses); ``torch.randn_like`` patched with the injected Gaussian draws.  Two
consecutive training-mode loss calls (the anneal rate follows the call counter).  Parameters: configs/betatc_vae.yaml.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_betatc_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_iw_golden import OUT, cks, load  # noqa: E402

CFG = dict(in_channels=3, latent_dim=10, anneal_steps=10000, alpha=1., beta=6., gamma=1.)


def main():
    from ctvae_amd import filler
    (mod,) = load(["betatc_vae"])
    seed, B, M_N = 1265, 8, 0.00025
    torch.manual_seed(0)
    model = mod.BetaTCVAE(**CFG)
    specs = filler.specs_of(model)
    model.load_state_dict(filler.fill_state(specs, seed + 1))
    model.train()
    x, e = filler.synthetic_batch(seed, B, latent_dim=10)
    o1 = torch.randn_like
    torch.randn_like = lambda t, **kw: e.clone()
    try:
        res = model(x)
    finally:
        torch.randn_like = o1
    l1 = model.loss_function(*res, M_N=M_N)
    l1["loss"].backward()
    with torch.no_grad():
        l2 = model.loss_function(*res, M_N=M_N)
    out = {"seed": np.int64(seed), "B": np.int64(B), "M_N": np.float64(M_N), "keys": np.array([k for k, _, _ in specs]),
           "mu": res[2].detach().numpy().copy(), "z": res[4].detach().numpy().copy(), "recons_cks": cks(res[0]),
           "recons_sub": res[0].detach()[:, :, ::8, ::8].numpy().copy(),
           "grad.fc_var.bias": model.fc_var.bias.grad.numpy().copy(), "grad.fc_mu.bias": model.fc_mu.bias.grad.numpy().copy()}
    for call, l in (("call1", l1), ("call2", l2)):
        for k, v in l.items():
            out[f"{call}.{k}"] = np.float64(v.item())
    for k, p in model.named_parameters():
        out["gradcks." + k] = cks(p.grad)
    np.savez_compressed(os.path.join(OUT, f"betatc_b{B}.npz"), **out)
    print({k: float(v) for k, v in l1.items()})


if __name__ == "__main__":
    main()
