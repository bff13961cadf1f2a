#!/usr/bin/env python3
"""Generate tests/golden/cvae_b4.npz from the REFERENCE's own ``models/cvae.py``.  TEST INFRASTRUCTURE ONLY.
Method as in gen_iw_golden.py (whose loader it uses): the module runs unmodified under a synthetic ``models`` package,
``torch.randn_like`` patched with the injected Gaussian draws; weights from the build's deterministic filler; labels = 0/1
attribute vectors (tests/helpers.py::cvae_labels repeats the rule).  Parameters: configs/cvae.yaml.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_cvae_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_iw_golden import OUT, cks, load  # noqa: E402

CFG = dict(in_channels=3, num_classes=40, latent_dim=128)


def main():
    from ctvae_amd import filler
    (mod,) = load(["cvae"])
    seed, B, M_N = 1265, 4, 0.00025
    torch.manual_seed(0)
    model = mod.ConditionalVAE(**CFG)
    model.load_state_dict(filler.fill_state(filler.specs_of(model), seed + 1))
    model.train()
    x, e = filler.synthetic_batch(seed, B)
    labels = (torch.rand(B, CFG["num_classes"], generator=torch.Generator().manual_seed(seed + 4)) < 0.3).float()
    o1 = torch.randn_like
    torch.randn_like = lambda t, **kw: e.clone()
    try:
        res = model(x, labels=labels)
    finally:
        torch.randn_like = o1
    losses = model.loss_function(*res, M_N=M_N)
    losses["loss"].backward()
    out = {"seed": np.int64(seed), "B": np.int64(B), "M_N": np.float64(M_N), "mu": res[2].detach().numpy().copy(),
           "log_var": res[3].detach().numpy().copy(), "recons_cks": cks(res[0]),
           "recons_sub": res[0].detach()[:, :, ::8, ::8].numpy().copy(),
           "grad.embed_data.weight": model.embed_data.weight.grad.numpy().copy(),
           "grad.embed_data.bias": model.embed_data.bias.grad.numpy().copy(),
           "grad.embed_class.bias_sub": model.embed_class.bias.grad[::16].numpy().copy(),
           "grad.decoder_input.weight_labelcols": model.decoder_input.weight.grad[::64, 128:].numpy().copy()}
    for k, v in losses.items():
        out["loss." + k] = np.float64(v.item())
    for k, p in model.named_parameters():
        out["gradcks." + k] = cks(p.grad)
    for k, b in model.named_buffers():
        if "running" in k:
            out["buf." + k] = cks(b)
    np.savez_compressed(os.path.join(OUT, f"cvae_b{B}.npz"), **out)
    print({k: float(v) for k, v in losses.items()})


if __name__ == "__main__":
    main()
