#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own modules.  TEST INFRASTRUCTURE ONLY.

Runs only in the build container (needs /root/reference).  The reference package cannot be
imported whole offline (torchvision / torch_geometric absent), so its hot-path
modules are executed unmodified under a synthetic ``models`` package, exactly in the order the
real ``models/__init__.py`` would: types_, base, vanilla_vae, vq_vae (ResidualLayer), mcq_vae.
Nothing is copied: the files are exec'd where they lie.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

The fixtures are *data*: seeds, inputs' checksums, outputs, gradients.  Weights come from
``ctvae_amd.filler`` (deterministic, owned by this build) and inputs from
``filler.synthetic_batch`` so they are regenerated, not stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("CTVAE_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def load_reference_models():
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg

    def run(name):
        spec = importlib.util.spec_from_file_location(f"models.{name}", os.path.join(REF, "models", f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"models.{name}"] = mod
        spec.loader.exec_module(mod)
        return mod

    run("types_")
    pkg.BaseVAE = run("base").BaseVAE
    vanilla = run("vanilla_vae")
    vq = run("vq_vae")
    pkg.ResidualLayer = vq.ResidualLayer
    mcq = run("mcq_vae")
    pkg.MultipleCodebookVectorQuantizer = mcq.MultipleCodebookVectorQuantizer
    return vanilla.VanillaVAE, mcq.MCQVAE


def cks(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], dtype=np.float64)


def np32(t):
    return t.detach().cpu().numpy().copy()      # copy: live buffers are updated in place later


def gen_vanilla(VanillaVAE, B, seed, kld_weight, lr):
    from ctvae_amd import filler
    torch.manual_seed(0)
    model = VanillaVAE(in_channels=3, latent_dim=128)
    sd = filler.fill_state(filler.specs_of(model), seed + 1)
    model.load_state_dict(sd)
    model.train()
    x, eps = filler.synthetic_batch(seed, B)
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: eps.clone()          # noise injection
    try:
        out = {"seed": np.int64(seed), "B": np.int64(B), "M_N": np.float64(kld_weight), "lr": np.float64(lr),
               "x_cks": cks(x), "eps_cks": cks(eps)}
        recons, inp, mu, log_var = model(x)
        losses = model.loss_function(recons, inp, mu, log_var, M_N=kld_weight)
        losses["loss"].backward()
        out["mu"] = np32(mu)
        out["log_var"] = np32(log_var)
        out["recons" if B <= 2 else "recons_strided"] = np32(recons if B <= 2 else recons[:, :, ::4, ::4])
        out["recons_cks"] = cks(recons)
        for k in ("loss", "Reconstruction_Loss", "KLD"):
            out["loss." + k] = np.float64(losses[k].item())
        names = []
        for k, p in model.named_parameters():
            names.append(k)
            out["gradcks." + k] = cks(p.grad)
        for k in ("fc_mu.bias", "encoder.0.0.weight", "final_layer.3.weight", "decoder.3.1.weight", "encoder.4.1.bias"):
            out["grad." + k] = np32(dict(model.named_parameters())[k].grad)
        for k, b in model.named_buffers():
            out["buf1." + k] = np32(b)
        # three Adam steps on the same batch (experiment.py:158-160: Adam(lr, weight_decay=0))
        model.zero_grad()
        model.load_state_dict(sd)
        opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0.0)
        step_losses = []
        for step in range(3):
            opt.zero_grad()
            r = model(x)
            l = model.loss_function(*r, M_N=kld_weight)
            l["loss"].backward()
            opt.step()
            step_losses.append(l["loss"].item())
            if step in (0, 2):
                for k, p in model.named_parameters():
                    out[f"adam{step + 1}.{k}"] = cks(p)
        out["adam_losses"] = np.array(step_losses, dtype=np.float64)
    finally:
        torch.randn_like = orig
    np.savez_compressed(os.path.join(OUT, f"vanilla_b{B}.npz"), **out)
    print("vanilla", B, {k: float(out["loss." + k]) for k in ("loss", "Reconstruction_Loss", "KLD")})


def gen_mcq(MCQVAE, tag, B, seed, cfg, lr):
    from ctvae_amd import filler
    torch.manual_seed(0)
    model = MCQVAE(**{**cfg, "hidden_dims": list(cfg["hidden_dims"])})
    sd = filler.fill_state(filler.specs_of(model), seed + 1)
    model.load_state_dict(sd)
    model.train()
    x, _ = filler.synthetic_batch(seed, B)
    out = {"seed": np.int64(seed), "B": np.int64(B), "lr": np.float64(lr), "x_cks": cks(x)}
    lat = model.encode(x)[0]
    inds = model.vq_layer.compute_inds(lat)
    q, vq_loss = model.vq_layer.compute_latents(lat, inds)
    recons = model.decode(q)
    losses = model.loss_function(recons, x, vq_loss)
    losses["loss"].backward()
    # second-best margin per row per codebook
    dc = cfg["embedding_dim"] // cfg["codebooks"]
    margins = []
    for i, qz in enumerate(model.vq_layer.quantizers):
        f = lat[:, i:i + dc].permute(0, 2, 3, 1).reshape(-1, dc).double()
        e = qz.embedding.weight.double()
        d = (f ** 2).sum(1, keepdim=True) + (e ** 2).sum(1) - 2 * f @ e.t()
        top2 = torch.topk(d, 2, dim=1, largest=False).values
        margins.append((top2[:, 1] - top2[:, 0]).view(B, 8, 8))
    out["latents" if B <= 2 else "latents_cks"] = np32(lat) if B <= 2 else cks(lat)
    out["latents_cks"] = cks(lat)
    out["inds"] = np32(inds)
    out["margin"] = np32(torch.stack(margins, 1).float())
    out["quantized_cks"] = cks(q)
    out["recons" if B <= 2 else "recons_strided"] = np32(recons if B <= 2 else recons[:, :, ::4, ::4])
    out["recons_cks"] = cks(recons)
    for k in ("loss", "Reconstruction_Loss", "VQ_Loss"):
        out["loss." + k] = np.float64(losses[k].item())
    for k, p in model.named_parameters():
        out["gradcks." + k] = cks(p.grad if p.grad is not None else torch.zeros_like(p))
    full = ["encoder.0.0.weight", "encoder.0.0.bias", "vq_layer.quantizers.0.embedding.weight",
            "decoder.10.0.weight", "encoder.11.0.bias", "decoder.3.resblock.2.weight"]
    for k in full:
        out["grad." + k] = np32(dict(model.named_parameters())[k].grad)
    model.zero_grad()
    model.load_state_dict(sd)
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0.0)
    step_losses = []
    for step in range(3):
        opt.zero_grad()
        r = model(x)
        l = model.loss_function(*r)
        l["loss"].backward()
        opt.step()
        step_losses.append(l["loss"].item())
        if step in (0, 2):
            for k, p in model.named_parameters():
                out[f"adam{step + 1}.{k}"] = cks(p)
    out["adam_losses"] = np.array(step_losses, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, f"{tag}_b{B}.npz"), **out)
    print(tag, B, {k: float(out["loss." + k]) for k in ("loss", "Reconstruction_Loss", "VQ_Loss")},
          "min margin", float(out["margin"].min()))


MCQ_CFG = dict(in_channels=3, embedding_dim=128, hidden_dims=[64, 128, 256], num_embeddings=64, img_size=64,
               codebooks=4, beta=0.25)                                   # configs/mcq_vae.yaml:1-9
CT_CONV_CFG = dict(in_channels=3, embedding_dim=128, hidden_dims=[64, 128, 256], num_embeddings=64, img_size=64,
                   codebooks=1, beta=0.1)                                # configs/ct_mcq_vae.yaml:1-12 (conv/VQ part)

if __name__ == "__main__":
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    VanillaVAE, MCQVAE = load_reference_models()
    for B in (2, 4):
        gen_vanilla(VanillaVAE, B, 1265, 0.00025, 0.005)                 # configs/vae.yaml:14-20
        gen_mcq(MCQVAE, "mcq", B, 1320, MCQ_CFG, 0.0005)                 # configs/mcq_vae.yaml:22-27
        gen_mcq(MCQVAE, "ctconv", B, 1250, CT_CONV_CFG, 0.0005)          # configs/ct_mcq_vae.yaml:29-37
