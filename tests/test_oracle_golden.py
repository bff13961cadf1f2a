"""CPU: the oracle restatement (oracle/vae_cpu.py) against golden vectors produced by the reference's
own modules (oracle/gen_golden.py).  This is what pins the oracle (tier rule 3)."""
import numpy as np
import pytest
import torch

from ctvae_amd import filler
from oracle import vae_cpu as O
from tests import helpers as H

TOL = 1e-4      # north_star: outputs match the reference CPU path within 1e-4 fp32


@pytest.mark.parametrize("B", [2, 4])
def test_vanilla_forward_loss_grads(golden, B):
    g = golden(f"vanilla_b{B}")
    sd = filler.fill_state(H.vanilla_specs(), int(g["seed"]) + 1)
    x, eps = filler.synthetic_batch(int(g["seed"]), B)
    np.testing.assert_allclose(H.cks(x), g["x_cks"], rtol=1e-12)
    np.testing.assert_allclose(H.cks(eps), g["eps_cks"], rtol=1e-12)
    losses, grads, nb, out = O.vanilla_step(sd, x, eps, float(g["M_N"]))
    np.testing.assert_allclose(out["mu"].numpy(), g["mu"], atol=TOL, rtol=0)
    np.testing.assert_allclose(out["log_var"].numpy(), g["log_var"], atol=TOL, rtol=0)
    if B <= 2:
        np.testing.assert_allclose(out["recons"].numpy(), g["recons"], atol=TOL, rtol=0)
    else:
        np.testing.assert_allclose(out["recons"][:, :, ::4, ::4].numpy(), g["recons_strided"], atol=TOL, rtol=0)
    for k in ("loss", "Reconstruction_Loss", "KLD"):
        assert abs(losses[k].item() - float(g["loss." + k])) <= TOL * max(1.0, abs(float(g["loss." + k])))
    for k, gr in grads.items():
        H.assert_cks_close(H.cks(gr), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)
    for k in ("fc_mu.bias", "encoder.0.0.weight", "final_layer.3.weight", "decoder.3.1.weight", "encoder.4.1.bias"):
        np.testing.assert_allclose(grads[k].numpy(), g["grad." + k], atol=TOL, rtol=1e-3)
    for k, v in nb.items():
        np.testing.assert_allclose(v.numpy(), g["buf1." + k], atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("tag,cfg", [("mcq", H.MCQ_CFG), ("ctconv", H.CT_CONV_CFG)])
@pytest.mark.parametrize("B", [2, 4])
def test_mcq_forward_loss_grads(golden, tag, cfg, B):
    g = golden(f"{tag}_b{B}")
    sd = filler.fill_state(H.mcq_specs(cfg), int(g["seed"]) + 1)
    x, _ = filler.synthetic_batch(int(g["seed"]), B)
    np.testing.assert_allclose(H.cks(x), g["x_cks"], rtol=1e-12)
    losses, grads, aux = O.mcq_step(sd, x, cfg["codebooks"], cfg["beta"], len(cfg["hidden_dims"]))
    inds = aux["inds"].numpy()
    bad = inds != g["inds"]
    assert not (bad & (g["margin"] > 1e-5)).any(), "index mismatch on a row whose margin is not a near-tie (SURVEY N2)"
    if B <= 2:
        np.testing.assert_allclose(aux["latents"].numpy(), g["latents"], atol=TOL, rtol=0)
        np.testing.assert_allclose(aux["recons"].numpy(), g["recons"], atol=TOL, rtol=0)
    else:
        np.testing.assert_allclose(aux["recons"][:, :, ::4, ::4].numpy(), g["recons_strided"], atol=TOL, rtol=0)
    for k in ("loss", "Reconstruction_Loss", "VQ_Loss"):
        assert abs(losses[k].item() - float(g["loss." + k])) <= TOL
    for k, gr in grads.items():
        H.assert_cks_close(H.cks(gr), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)
    for k in [k[5:] for k in g if k.startswith("grad.")]:
        np.testing.assert_allclose(grads[k].numpy(), g["grad." + k], atol=TOL, rtol=1e-3)


def test_slice_offset_quirk():
    """mcq_vae.py:104,117: codebook i reads channels [i, i+D/C) -> channels >= C-1+D/C never get gradient."""
    cfg = H.MCQ_CFG
    sd = filler.fill_state(H.mcq_specs(cfg), 7)
    lat = torch.randn(2, 128, 8, 8, generator=torch.Generator().manual_seed(3)).requires_grad_(True)
    inds = O.mcq_compute_inds(sd, lat, 4)
    q, loss = O.mcq_compute_latents(sd, lat, inds, 4, 0.25)
    (q.sum() + loss).backward()
    assert lat.grad[:, 35:].abs().max().item() == 0.0
    assert lat.grad[:, :35].abs().min().item() > 0.0


def test_vanilla_adam_steps(golden):
    """experiment.py:158-160 Adam; 3 steps on one batch: the loss trajectory is the well-conditioned observable."""
    g = golden("vanilla_b2")
    sd = filler.fill_state(H.vanilla_specs(), int(g["seed"]) + 1)
    x, eps = filler.synthetic_batch(int(g["seed"]), 2)
    cur = O.leafify(sd)
    params = [k for k, v in cur.items() if v.requires_grad]
    opt = torch.optim.Adam([cur[k] for k in params], lr=float(g["lr"]))
    got = []
    for step in range(3):
        opt.zero_grad()
        nb = {}
        r = O.vanilla_forward(cur, x, eps, True, nb)
        l = O.vanilla_loss(*r, float(g["M_N"]))
        l["loss"].backward()
        opt.step()
        for k, v in nb.items():
            cur[k] = v
        got.append(l["loss"].item())
    np.testing.assert_allclose(got, g["adam_losses"], rtol=2e-3, atol=1e-4)


BETA_CFG = {"H": dict(loss_type="H", beta=10.0), "B": dict(loss_type="B", gamma=10.0, max_capacity=25, capacity_max_iter=10000)}


@pytest.mark.parametrize("tag", ["H", "B"])
def test_beta_vae_losses(golden, tag):
    """BetaVAE = VanillaVAE's network + beta / capacity objective: oracle against the reference's beta_vae.py fixture
    (two consecutive loss calls: type 'B' depends on the call counter)."""
    g = golden(f"beta_{tag}_b2")
    sd = O.leafify(filler.fill_state(H.vanilla_specs(), int(g["seed"]) + 1))
    x, eps = filler.synthetic_batch(int(g["seed"]), 2)
    recons, inp, mu, log_var = O.vanilla_forward(sd, x, eps, True, {})
    np.testing.assert_allclose(mu.detach().numpy(), g["mu"], atol=TOL, rtol=0)
    np.testing.assert_allclose(H.cks(recons), g["recons_cks"], rtol=1e-4)
    l1 = O.beta_loss(recons, inp, mu, log_var, float(g["M_N"]), num_iter=1, **BETA_CFG[tag])
    l2 = O.beta_loss(recons, inp, mu, log_var, float(g["M_N"]), num_iter=2, **BETA_CFG[tag])
    for call, l in (("call1", l1), ("call2", l2)):
        for k in ("loss", "Reconstruction_Loss", "KLD"):
            want = float(g[f"{call}.{k}"])
            assert abs(l[k].item() - want) <= TOL * max(1.0, abs(want)), (call, k)
    l1["loss"].backward()
    for k, v in sd.items():
        if v.requires_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)


CAT_CFG = dict(in_channels=3, latent_dim=64, categorical_dim=40, temperature=0.5, anneal_rate=0.00003, anneal_interval=100, alpha=1.0)


def test_categorical_vae_forward_loss_grads(golden):
    """CategoricalVAE (Gumbel-softmax latent): oracle restatement against the reference's own cat_vae.py fixture
    (oracle/gen_cat_golden.py), uniform draws injected."""
    g = golden("cat_b2")
    seed = int(g["seed"])
    sd = O.leafify(filler.fill_state(H.cat_specs(CAT_CFG["latent_dim"], CAT_CFG["categorical_dim"]), seed + 1))
    x, _ = filler.synthetic_batch(seed, 2)
    u = H.cat_uniform(seed, 2, CAT_CFG["latent_dim"], CAT_CFG["categorical_dim"])
    recons, inp, q = O.categorical_forward(sd, x, u, CAT_CFG["latent_dim"], CAT_CFG["categorical_dim"], CAT_CFG["temperature"], True, {})
    np.testing.assert_allclose(q.detach().numpy(), g["q"], atol=TOL, rtol=0)
    np.testing.assert_allclose(recons.detach()[:, :, ::8, ::8].numpy(), g["recons_slice"], atol=TOL, rtol=0)
    l = O.categorical_loss(recons, inp, q, float(g["M_N"]), CAT_CFG["alpha"])
    for k in ("loss", "Reconstruction_Loss", "KLD"):
        want = float(g["call1." + k])
        assert abs(l[k].item() - want) <= TOL * max(1.0, abs(want)), k
        assert abs(float(g["call2." + k]) - want) <= 1e-7       # annealing never drops below the initial temperature
    assert float(g["temp_after"]) == CAT_CFG["temperature"]
    l["loss"].backward()
    np.testing.assert_allclose(sd["fc_z.bias"].grad.numpy(), g["grad.fc_z.bias"], atol=TOL, rtol=1e-3)
    for k, v in sd.items():
        if v.requires_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)


@pytest.mark.parametrize("tag", ["iwae", "miwae"])
def test_iwae_miwae_forward_loss_grads(golden, tag):
    """IWAE / MIWAE (VanillaVAE's network, importance-weighted bound): oracle restatement against the reference's own
    iwae.py / miwae.py fixtures (oracle/gen_iw_golden.py), noise injected."""
    g = golden(f"{tag}_b2")
    seed = int(g["seed"])
    sd = O.leafify(filler.fill_state(H.vanilla_specs(), seed + 1))
    x, _ = filler.synthetic_batch(seed, 2)
    eps = H.iw_noise(seed, (2,) + H.IW_CASES[tag][2])
    res = O.miwae_forward(sd, x, eps, True, {})
    assert tuple(res[0].shape) == tuple(g["recons_shape"])
    np.testing.assert_allclose(res[2].detach().numpy(), g["mu"], atol=TOL, rtol=0)
    np.testing.assert_allclose(res[0].detach()[..., ::16, ::16].numpy(), g["recons_slice"], atol=TOL, rtol=0)
    l = O.iw_loss(res[0], res[1], res[2], res[3], float(g["M_N"]))
    for k in ("loss", "Reconstruction_Loss", "KLD"):
        want = float(g["loss." + k])
        assert abs(l[k].item() - want) <= TOL * max(1.0, abs(want)), k
    l["loss"].backward()
    for k in ("fc_mu.bias", "fc_var.bias"):
        np.testing.assert_allclose(sd[k].grad.numpy(), g["grad." + k], atol=1e-6, rtol=1e-3)
    for k, v in sd.items():
        if v.requires_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)


def test_logcosh_vae_loss_grads(golden):
    """LogCoshVAE (VanillaVAE's network, log-cosh reconstruction term): oracle against the reference's logcosh_vae.py fixture."""
    g = golden("logcosh_b2")
    seed = int(g["seed"])
    sd = O.leafify(filler.fill_state(H.vanilla_specs(), seed + 1))
    x, eps = filler.synthetic_batch(seed, 2)
    recons, inp, mu, log_var = O.vanilla_forward(sd, x, eps, True, {})
    np.testing.assert_allclose(mu.detach().numpy(), g["mu"], atol=TOL, rtol=0)
    l = O.logcosh_loss(recons, inp, mu, log_var, float(g["M_N"]), 10.0, 1.0)
    for k in ("loss", "Reconstruction_Loss", "KLD"):
        want = float(g["loss." + k])
        assert abs(l[k].item() - want) <= TOL * max(1.0, abs(want)), k
    l["loss"].backward()
    for k in ("fc_mu.bias", "final_layer.3.weight"):
        np.testing.assert_allclose(sd[k].grad.numpy(), g["grad." + k], atol=1e-5, rtol=1e-3)
    for k, v in sd.items():
        if v.requires_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)


@pytest.mark.parametrize("tag", ["wae_imq", "wae_rbf", "infovae"])
def test_mmd_models_forward_loss_grads(golden, tag):
    """WAE_MMD (imq / rbf) and InfoVAE: oracle restatement against the reference's own wae_mmd.py / info_vae.py fixtures
    (oracle/gen_mmd_golden.py), prior samples injected."""
    g = golden(f"{tag}_b4")
    seed, B = int(g["seed"]), 4
    name, cfg = H.MMD_CASES[tag]
    gaussian = name == "InfoVAE"
    sd = O.leafify(filler.fill_state(H.vanilla_specs() if gaussian else H.wae_specs(), seed + 1))
    x, eps = filler.synthetic_batch(seed, B)
    prior = H.mmd_prior(seed, B)
    if gaussian:
        recons, inp, mu, log_var = O.vanilla_forward(sd, x, eps, True, {})
        z = O.vanilla_reparameterize(mu, log_var, eps)
        l = O.infovae_loss(recons, inp, z, mu, log_var, prior, float(g["M_N"]), cfg["alpha"], cfg["beta"], cfg["reg_weight"], cfg["kernel_type"])
    else:
        recons, inp, z = O.wae_forward(sd, x, True, {})
        l = O.wae_loss(recons, inp, z, prior, cfg["reg_weight"], cfg["kernel_type"])
    np.testing.assert_allclose(z.detach().numpy(), g["z"], atol=TOL, rtol=0)
    for k, v in l.items():
        want = float(g["loss." + k])
        assert abs(v.item() - want) <= TOL * max(1.0, abs(want)), k
    l["loss"].backward()
    head = "fc_mu" if gaussian else "fc_z"
    np.testing.assert_allclose(sd[head + ".bias"].grad.numpy(), g[f"grad.{head}.bias"], atol=1e-5, rtol=1e-3)
    for k, v in sd.items():
        if v.requires_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)


def test_mssim_vae_loss_grads(golden):
    """MSSIMVAE (VanillaVAE's network, multi-scale SSIM reconstruction term): oracle against the reference's mssim_vae.py fixture,
    and the loss module on its own against the recorded pair (value and full gradient)."""
    g = golden("mssim_b4")
    seed = int(g["seed"])
    sd = O.leafify(filler.fill_state(H.vanilla_specs(), seed + 1))
    x, eps = filler.synthetic_batch(seed, 4)
    recons, inp, mu, log_var = O.vanilla_forward(sd, x, eps, True, {})
    np.testing.assert_allclose(mu.detach().numpy(), g["mu"], atol=TOL, rtol=0)
    l = O.mssimvae_loss(recons, inp, mu, log_var, float(g["M_N"]))
    for k in ("loss", "Reconstruction_Loss", "KLD"):
        want = float(g["loss." + k])
        assert abs(l[k].item() - want) <= TOL * max(1.0, abs(want)), k
    l["loss"].backward()
    np.testing.assert_allclose(sd["fc_mu.bias"].grad.numpy(), g["grad.fc_mu.bias"], atol=1e-5, rtol=1e-3)
    for k, v in sd.items():
        if v.requires_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)
    a = torch.from_numpy(g["pair.a"]).requires_grad_(True)
    val = O.mssim_loss(a, torch.from_numpy(g["pair.b"]))
    assert abs(val.item() - float(g["pair.loss"])) <= 1e-6
    val.backward()
    np.testing.assert_allclose(a.grad.numpy(), g["pair.grad_a"], atol=1e-8, rtol=1e-4)


def test_dip_vae_loss_grads(golden):
    """DIPVAE (VanillaVAE's network, sum-reduced objective + DIP-II regulariser): oracle against the reference's dip_vae.py fixture."""
    g = golden("dip_b4")
    seed = int(g["seed"])
    sd = O.leafify(filler.fill_state(H.vanilla_specs(), seed + 1))
    x, eps = filler.synthetic_batch(seed, 4)
    recons, inp, mu, log_var = O.vanilla_forward(sd, x, eps, True, {})
    np.testing.assert_allclose(mu.detach().numpy(), g["mu"], atol=TOL, rtol=0)
    l = O.dip_loss(recons, inp, mu, log_var, float(g["M_N"]), 0.05, 0.1)
    for k, v in l.items():
        want = float(g["loss." + k])
        assert abs(v.item() - want) <= TOL * max(1.0, abs(want)), k
    l["loss"].backward()
    for k in ("fc_mu.bias", "fc_var.bias"):
        np.testing.assert_allclose(sd[k].grad.numpy(), g["grad." + k], atol=1e-3, rtol=1e-3)
    for k, v in sd.items():
        if v.requires_grad and not (k.endswith(".0.bias") and not k.startswith("final_layer.3")):   # see the GPU twin of this test
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-3, what=k)


def test_joint_vae_forward_loss_grads(golden):
    """JointVAE (Gaussian + one categorical latent, capacity objective): oracle against the reference's joint_vae.py fixture
    (two consecutive loss calls: the capacities follow the call counter)."""
    g = golden("joint_b4")
    seed = int(g["seed"])
    sd = O.leafify(filler.fill_state(H.joint_specs(), seed + 1))
    x, e = filler.synthetic_batch(seed, 4)
    res = O.joint_forward(sd, x, e, H.joint_uniform(seed, 4), 0.5, True, {})
    np.testing.assert_allclose(res[2].detach().numpy(), g["q"], atol=TOL, rtol=0)
    np.testing.assert_allclose(res[3].detach().numpy(), g["mu"], atol=TOL, rtol=0)
    caps = dict(cont=(0.0, 20.0, 10.0, 25000), disc=(0.0, 20.0, 10.0, 25000))
    l1 = O.joint_loss(*res, float(g["M_N"]), 1, 10.0, **caps)
    l2 = O.joint_loss(*res, float(g["M_N"]), 2, 10.0, **caps)
    for call, l in (("call1", l1), ("call2", l2)):
        for k, v in l.items():
            want = float(g[f"{call}.{k}"])
            assert abs(v.item() - want) <= TOL * max(1.0, abs(want)), (call, k)
    l1["loss"].backward()
    for k in ("fc_z.bias", "fc_var.bias"):
        np.testing.assert_allclose(sd[k].grad.numpy(), g["grad." + k], atol=1e-6, rtol=1e-3)
    for k, v in sd.items():
        if v.requires_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)


def test_vqvae_forward_loss_grads(golden):
    """VQVAE = MCQ-VAE's stacks around ONE codebook: oracle (single-codebook path of the MCQ restatement) against the
    reference's own vq_vae.py fixture."""
    g = golden("vqvae_b2")
    cfg = H.VQVAE_CFG
    sd = filler.fill_state(H.vqvae_specs(), int(g["seed"]) + 1)
    sd = type(sd)((k.replace("vq_layer.embedding", "vq_layer.quantizers.0.embedding"), v) for k, v in sd.items())
    x, _ = filler.synthetic_batch(int(g["seed"]), 2)
    losses, grads, aux = O.mcq_step(sd, x, 1, cfg["beta"], len(cfg["hidden_dims"]))
    inds = aux["inds"].numpy()[:, 0]
    assert not ((inds != g["inds"]) & (g["margin"] > 1e-5)).any()
    np.testing.assert_allclose(aux["latents"].numpy(), g["latents"], atol=TOL, rtol=0)
    np.testing.assert_allclose(H.cks(aux["recons"]), g["recons_cks"], rtol=1e-4)
    for k in ("loss", "Reconstruction_Loss", "VQ_Loss"):
        assert abs(losses[k].item() - float(g["loss." + k])) <= TOL
    for k, gr in grads.items():
        H.assert_cks_close(H.cks(gr), g["gradcks." + k.replace("vq_layer.quantizers.0.embedding", "vq_layer.embedding")],
                           rtol=1e-3, atol=1e-5, what=k)


def test_conditional_vae_forward_loss_grads(golden):
    """ConditionalVAE (label as an extra input plane and next to z): oracle against the reference's own cvae.py fixture."""
    g = golden("cvae_b4")
    seed = int(g["seed"])
    sd = O.leafify(filler.fill_state(H.cvae_specs(), seed + 1))
    x, e = filler.synthetic_batch(seed, 4)
    res = O.cvae_forward(sd, x, H.cvae_labels(seed, 4), e, True, {})
    np.testing.assert_allclose(res[2].detach().numpy(), g["mu"], atol=TOL, rtol=0)
    np.testing.assert_allclose(res[3].detach().numpy(), g["log_var"], atol=TOL, rtol=0)
    np.testing.assert_allclose(res[0].detach()[:, :, ::8, ::8].numpy(), g["recons_sub"], atol=TOL, rtol=0)
    losses = O.vanilla_loss(*res, float(g["M_N"]))
    for k, v in losses.items():
        want = float(g["loss." + k])
        assert abs(v.item() - want) <= TOL * max(1.0, abs(want)), k
    losses["loss"].backward()
    np.testing.assert_allclose(sd["embed_data.weight"].grad.numpy(), g["grad.embed_data.weight"], atol=1e-6, rtol=1e-3)
    np.testing.assert_allclose(sd["decoder_input.weight"].grad[::64, 128:].numpy(), g["grad.decoder_input.weight_labelcols"], atol=1e-6, rtol=1e-3)
    for k, v in sd.items():
        if v.requires_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)


def test_swae_forward_loss_grads(golden):
    """SWAE (WAE's network, mse + l1 + sliced Wasserstein distance): oracle against the reference's own swae.py fixture with the
    prior draws and the projection directions injected."""
    g = golden("swae_b8")
    seed, B = int(g["seed"]), int(g["B"])
    sd = O.leafify(filler.fill_state(H.wae_specs(), seed + 1))
    x, _ = filler.synthetic_batch(seed, B)
    res = O.wae_forward(sd, x, True, {})
    np.testing.assert_allclose(res[2].detach().numpy(), g["z"], atol=TOL, rtol=0)
    prior, proj = H.swae_draws(seed, B)
    losses = O.swae_loss(*res, prior, proj, 100, 2.0)
    for k, v in losses.items():
        want = float(g["loss." + k])
        assert abs(v.item() - want) <= TOL * max(1.0, abs(want)), (k, v.item(), want)
    losses["loss"].backward()
    np.testing.assert_allclose(sd["fc_z.bias"].grad.numpy(), g["grad.fc_z.bias"], atol=1e-6, rtol=1e-3)
    for k, v in sd.items():
        if v.requires_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)


def test_twostage_vae_is_the_first_stage(golden):
    """TwoStageVAE: the reference's step uses the first stage only -- the oracle's VanillaVAE functions on the first-stage keys
    reproduce the fixture of the reference's own twostage_vae.py; key order / shapes of the whole state_dict as recorded."""
    g = golden("twostage_b2")
    seed = int(g["seed"])
    specs = H.twostage_specs()
    assert [k for k, _, _ in specs] == list(g["keys"]) and [str(tuple(sh)) for _, sh, _ in specs] == list(g["shapes"])
    sd = O.leafify(filler.fill_state(specs, seed + 1))
    x, e = filler.synthetic_batch(seed, 2)
    res = O.vanilla_forward(sd, x, e, True, {})
    np.testing.assert_allclose(res[2].detach().numpy(), g["mu"], atol=TOL, rtol=0)
    losses = O.vanilla_loss(*res, float(g["M_N"]))
    for k, v in losses.items():
        assert abs(v.item() - float(g["loss." + k])) <= TOL * max(1.0, abs(float(g["loss." + k]))), k
    losses["loss"].backward()
    no_grad = set(g["no_grad"])
    assert no_grad == {k for k, v in sd.items() if v.requires_grad and k.split(".")[0] in ("encoder2", "fc_mu2", "fc_var2", "decoder2")}
    for k, v in sd.items():
        if v.requires_grad and k not in no_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)
        elif v.requires_grad:
            assert v.grad is None


def test_hvae_forward_loss_grads(golden):
    """HVAE (two latent levels, second encoder conditioned on z2, three Gaussian-KL terms): oracle against the reference's own
    hvae.py fixture with both noise draws injected."""
    g = golden("hvae_b4")
    seed, B = int(g["seed"]), int(g["B"])
    specs = H.hvae_specs()
    assert [k for k, _, _ in specs] == list(g["keys"])
    sd = O.leafify(filler.fill_state(specs, seed + 1))
    x, _ = filler.synthetic_batch(seed, B)
    e1, e2 = H.hvae_noise(seed, B)
    res = O.hvae_forward(sd, x, e1, e2, True, {})
    np.testing.assert_allclose(res[2].detach().numpy(), g["z1_mu"], atol=TOL, rtol=0)
    np.testing.assert_allclose(res[4].detach().numpy(), g["z2_mu"], atol=TOL, rtol=0)
    np.testing.assert_allclose(res[6].detach().numpy(), g["z1"], atol=TOL, rtol=0)
    losses = O.hvae_loss(sd, *res, float(g["M_N"]))
    for k, v in losses.items():
        want = float(g["loss." + k])
        assert abs(v.item() - want) <= TOL * max(1.0, abs(want)), (k, v.item(), want)
    losses["loss"].backward()
    np.testing.assert_allclose(sd["recons_z1_mu.bias"].grad.numpy(), g["grad.recons_z1_mu.bias"], atol=1e-7, rtol=1e-3)
    for k, v in sd.items():
        if v.requires_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)


def _vamp_state(seed):
    sd = filler.fill_state(H.vamp_specs(), seed + 1)
    sd["embed_pseudo.0.bias"] = sd["embed_pseudo.0.bias"] + 0.5      # as oracle/gen_vamp_golden.py: the Hardtanh gets all three regimes
    return sd


def test_vamp_vae_forward_loss_grads(golden):
    """VampVAE (VampPrior over K pseudo-inputs encoded by the same network): oracle against the reference's own vampvae.py
    fixture."""
    g = golden("vamp_b4")
    seed, B = int(g["seed"]), int(g["B"])
    assert [k for k, _, _ in H.vamp_specs()] == list(g["keys"])
    sd = O.leafify(_vamp_state(seed))
    x, e = filler.synthetic_batch(seed, B)
    res = O.vanilla_forward(sd, x, e, True, {})
    z = O.vanilla_reparameterize(res[2], res[3], e)
    np.testing.assert_allclose(res[2].detach().numpy(), g["mu"], atol=TOL, rtol=0)
    np.testing.assert_allclose(z.detach().numpy(), g["z"], atol=TOL, rtol=0)
    losses = O.vamp_loss(sd, *res, z, float(g["M_N"]), 50, True, {})
    for k, v in losses.items():
        want = float(g["loss." + k])
        assert abs(v.item() - want) <= TOL * max(1.0, abs(want)), (k, v.item(), want)
    losses["loss"].backward()
    np.testing.assert_allclose(sd["embed_pseudo.0.bias"].grad[::64].numpy(), g["grad.embed_pseudo.0.bias_sub"], atol=1e-9, rtol=2e-3)
    for k, v in sd.items():
        if v.requires_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)


def test_betatc_vae_forward_loss_grads(golden):
    """BetaTCVAE (own conv net without BatchNorm; mutual information / total correlation / dimension-wise KL from the [B,B,D]
    log-density matrix with stratified importance weights): oracle against the reference's own betatc_vae.py fixture, two
    consecutive loss calls (anneal counter)."""
    g = golden("betatc_b8")
    seed, B = int(g["seed"]), int(g["B"])
    specs = H.betatc_specs()
    assert [k for k, _, _ in specs] == list(g["keys"])
    sd = O.leafify(filler.fill_state(specs, seed + 1))
    x, e = filler.synthetic_batch(seed, B, latent_dim=10)
    res = O.betatc_forward(sd, x, e)
    np.testing.assert_allclose(res[2].detach().numpy(), g["mu"], atol=TOL, rtol=0)
    np.testing.assert_allclose(res[4].detach().numpy(), g["z"], atol=TOL, rtol=0)
    np.testing.assert_allclose(res[0].detach()[:, :, ::8, ::8].numpy(), g["recons_sub"], atol=TOL, rtol=0)
    c = H.BETATC_CFG
    for call, it in (("call1", 1), ("call2", 2)):
        l = O.betatc_loss(*res, float(g["M_N"]), it, c["anneal_steps"], c["alpha"], c["beta"], c["gamma"])
        for k, v in l.items():
            want = float(g[f"{call}.{k}"])
            assert abs(v.item() - want) <= TOL * max(1.0, abs(want)), (call, k, v.item(), want)
        if it == 1:
            l["loss"].backward()
    np.testing.assert_allclose(sd["fc_var.bias"].grad.numpy(), g["grad.fc_var.bias"], atol=1e-5, rtol=1e-3)
    for k, v in sd.items():
        if v.requires_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-4, what=k)


def test_gamma_vae_forward_loss_grads(golden):
    """GammaVAE (softmax heads, shape-augmentation reparameterisation of an injected Gamma draw, Gamma KL with lgamma / digamma,
    Sigmoid output): oracle against the reference's own gamma_vae.py fixture."""
    g = golden("gamma_b4")
    seed, B = int(g["seed"]), int(g["B"])
    specs = H.gamma_specs()
    assert [k for k, _, _ in specs] == list(g["keys"])
    sd = O.leafify(filler.fill_state(specs, seed + 1))
    x, _ = filler.synthetic_batch(seed, B)
    res = O.gamma_forward(sd, x, torch.from_numpy(g["zhat"]), 8.0, True, {})
    np.testing.assert_allclose(res[2].detach().numpy(), g["alpha"], atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(res[3].detach().numpy(), g["beta"], atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(res[0].detach()[:, :, ::8, ::8].numpy(), g["recons_sub"], atol=TOL, rtol=0)
    losses = O.gamma_loss(*res)
    want = float(g["loss.loss"])
    assert abs(losses["loss"].item() - want) <= TOL * max(1.0, abs(want)), (losses["loss"].item(), want)
    losses["loss"].backward()
    np.testing.assert_allclose(sd["fc_var.0.bias"].grad.numpy(), g["grad.fc_var.0.bias"], atol=1e-6, rtol=1e-3)
    for k, v in sd.items():
        if v.requires_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)


def test_lvae_forward_loss_grads(golden):
    """LVAE (ladder VAE: per-level heads bottom-up; precision merge, sample and KL per rung top-down): oracle against the
    reference's own lvae.py fixture with all five noise draws injected."""
    from ctvae_amd.models import vae_models
    g = golden("lvae_b4")
    seed, B = int(g["seed"]), int(g["B"])
    specs = filler.specs_of(vae_models["LVAE"](**{k: (list(v) if isinstance(v, list) else v) for k, v in H.LVAE_CFG.items()}))
    assert [k for k, _, _ in specs] == list(g["keys"])           # the product mirrors the reference's state_dict, key for key
    sd = O.leafify(filler.fill_state(specs, seed + 1))
    x, _ = filler.synthetic_batch(seed, B)
    res = O.lvae_forward(sd, x, H.lvae_noise(seed, B), training=True, new_buffers={})
    np.testing.assert_allclose(res[2].detach().numpy(), g["kl_div"], atol=1e-3, rtol=1e-4)
    np.testing.assert_allclose(res[0].detach()[:, :, ::8, ::8].numpy(), g["recons_sub"], atol=TOL, rtol=0)
    losses = O.lvae_loss(*res, float(g["M_N"]))
    for k, v in losses.items():
        want = float(g["loss." + k])
        assert abs(v.item() - want) <= TOL * max(1.0, abs(want)), (k, v.item(), want)
    losses["loss"].backward()
    np.testing.assert_allclose(sd["ladders.0.fc_var.bias"].grad.numpy(), g["grad.ladders.0.fc_var.bias"], atol=1e-7, rtol=1e-3)
    for k, v in sd.items():
        if v.requires_grad:
            H.assert_cks_close(H.cks(v.grad), g["gradcks." + k], rtol=1e-3, atol=1e-5, what=k)
