"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A functional, pure-torch fp32 restatement of the reference's VAE training-step arithmetic
(Strong-AI-Lab/ct-vae).  It exists to *check* the HIP path; nothing under ``ct-vae_amd/`` may
import it.  Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the reference's own
unmodified ``models/{types_,base,vanilla_vae,vq_vae,mcq_vae}.py`` in the build container
(``oracle/gen_golden.py``) and ``tests/test_oracle_golden.py`` checks every function below against
them.  ``CausalTransition`` / ``CTMCQVAE`` (ct_mcq_vae.py:42-620) are restated in ``oracle/causal_cpu.py`` and pinned by
``oracle/gen_ct_golden.py`` (the reference's own module with an empty stub for the absent torch_geometric 2.2.0 and a test
double for its GATv2Conv); only the arithmetic inside GATv2Conv is "parity unpinned" (see that file's header).

Everything here works on an ordered ``state_dict``-shaped mapping ``sd`` whose keys and PyTorch
layouts equal the reference's (Conv2d [Co,Ci,kh,kw], ConvTranspose2d [Ci,Co,kh,kw], Linear
[out,in]) so that the same tensors can be loaded into the reference modules.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

LEAKY = 0.01          # nn.LeakyReLU() default slope, every site (vanilla_vae.py:31, mcq_vae.py:172 ...)
BN_EPS = 1e-5         # nn.BatchNorm2d defaults (vanilla_vae.py:30)
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------------
# VanillaVAE  (models/vanilla_vae.py:8-173)
# --------------------------------------------------------------------------------------------
def _bn_lrelu(sd, pfx, y, training, new_buffers):
    """BatchNorm2d (train: biased batch var; running stats momentum .1 with unbiased var) + LeakyReLU."""
    rm = sd[pfx + ".running_mean"].detach().clone()
    rv = sd[pfx + ".running_var"].detach().clone()
    out = F.batch_norm(y, rm, rv, sd[pfx + ".weight"], sd[pfx + ".bias"], training, BN_MOMENTUM, BN_EPS)
    if training and new_buffers is not None:
        new_buffers[pfx + ".running_mean"] = rm
        new_buffers[pfx + ".running_var"] = rv
        new_buffers[pfx + ".num_batches_tracked"] = sd[pfx + ".num_batches_tracked"] + 1
    return F.leaky_relu(out, LEAKY)


def vanilla_encode(sd, x, training=True, new_buffers=None, n_layers=5):
    """vanilla_vae.py:77-92: 5 x [Conv k3 s2 p1 -> BN -> LeakyReLU], NCHW flatten, two Linear heads."""
    h = x
    for i in range(n_layers):
        h = F.conv2d(h, sd[f"encoder.{i}.0.weight"], sd[f"encoder.{i}.0.bias"], stride=2, padding=1)
        h = _bn_lrelu(sd, f"encoder.{i}.1", h, training, new_buffers)
    flat = torch.flatten(h, start_dim=1)
    mu = F.linear(flat, sd["fc_mu.weight"], sd["fc_mu.bias"])
    log_var = F.linear(flat, sd["fc_var.weight"], sd["fc_var.bias"])
    return mu, log_var


def vanilla_reparameterize(mu, log_var, eps):
    """vanilla_vae.py:107-117 with the noise injected."""
    return eps * torch.exp(0.5 * log_var) + mu


def vanilla_decode(sd, z, training=True, new_buffers=None, n_layers=4):
    """vanilla_vae.py:94-105: Linear -> view(-1,512,2,2) -> 4 x [ConvT k3 s2 p1 op1 -> BN -> LReLU] -> final_layer."""
    h = F.linear(z, sd["decoder_input.weight"], sd["decoder_input.bias"]).view(-1, 512, 2, 2)
    for i in range(n_layers):
        h = F.conv_transpose2d(h, sd[f"decoder.{i}.0.weight"], sd[f"decoder.{i}.0.bias"],
                               stride=2, padding=1, output_padding=1)
        h = _bn_lrelu(sd, f"decoder.{i}.1", h, training, new_buffers)
    h = F.conv_transpose2d(h, sd["final_layer.0.weight"], sd["final_layer.0.bias"],
                           stride=2, padding=1, output_padding=1)
    h = _bn_lrelu(sd, "final_layer.1", h, training, new_buffers)
    h = F.conv2d(h, sd["final_layer.3.weight"], sd["final_layer.3.bias"], padding=1)
    return torch.tanh(h)


def vanilla_forward(sd, x, eps, training=True, new_buffers=None):
    """vanilla_vae.py:119-122 -> [recons, input, mu, log_var]."""
    mu, log_var = vanilla_encode(sd, x, training, new_buffers)
    z = vanilla_reparameterize(mu, log_var, eps)
    return [vanilla_decode(sd, z, training, new_buffers), x, mu, log_var]


def vanilla_loss(recons, x, mu, log_var, M_N):
    """vanilla_vae.py:124-146 (note 'KLD' is returned with flipped sign)."""
    recons_loss = F.mse_loss(recons, x)
    kld = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
    return {"loss": recons_loss + M_N * kld, "Reconstruction_Loss": recons_loss.detach(), "KLD": -kld.detach()}


def mmd_terms(z, prior_z, kernel_type, z_var, eps=1e-7):
    """compute_kernel on (p,p), (z,z), (p,z) (wae_mmd.py:120-198 == info_vae.py:150-216): imq sums every off-diagonal pair
    (the cross term's diagonal is removed too), rbf is the full [N,N] matrix, reduced by .mean() in compute_mmd."""
    def kern(a, b):
        d2 = (a.unsqueeze(-2) - b.unsqueeze(-3)).pow(2)
        c = 2. * a.size(1) * z_var
        if kernel_type == 'rbf':
            return torch.exp(-(d2.mean(-1) / c)).mean()
        if kernel_type == 'imq':
            k = c / (eps + c + d2.sum(dim=-1))
            return k.sum() - k.diag().sum()
        raise ValueError('Undefined kernel type.')
    return kern(prior_z, prior_z), kern(z, z), kern(prior_z, z)


def wae_forward(sd, x, training=True, new_buffers=None):
    """WAE_MMD.forward (wae_mmd.py:81-103): deterministic encoder with one head fc_z -> [recons, input, z]."""
    h = x
    for i in range(5):
        h = F.conv2d(h, sd[f"encoder.{i}.0.weight"], sd[f"encoder.{i}.0.bias"], stride=2, padding=1)
        h = _bn_lrelu(sd, f"encoder.{i}.1", h, training, new_buffers)
    z = F.linear(torch.flatten(h, start_dim=1), sd["fc_z.weight"], sd["fc_z.bias"])
    return [vanilla_decode(sd, z, training, new_buffers), x, z]


def wae_loss(recons, x, z, prior_z, reg_weight, kernel_type, z_var=2.0):
    """wae_mmd.py:105-118,192-203."""
    B = x.size(0)
    w = reg_weight / (B * (B - 1))
    pp, zz, pz = mmd_terms(z, prior_z, kernel_type, z_var)
    mmd = w * pp + w * zz - 2 * w * pz
    rl = F.mse_loss(recons, x)
    return {"loss": rl + mmd, "Reconstruction_Loss": rl, "MMD": mmd}


def _conv_stack(sd, pfx, h, training, new_buffers, n_layers=5):
    for i in range(n_layers):
        h = F.conv2d(h, sd[f"{pfx}.{i}.0.weight"], sd[f"{pfx}.{i}.0.bias"], stride=2, padding=1)
        h = _bn_lrelu(sd, f"{pfx}.{i}.1", h, training, new_buffers)
    return torch.flatten(h, start_dim=1)


def hvae_forward(sd, x, e1, e2, training=True, new_buffers=None, img_size=64):
    """HVAE.forward (hvae.py:112-199) with both Gaussian draws injected (e2 for z2, which is drawn first; e1 for z1)
    -> [recons, input, z1_mu, z1_log_var, z2_mu, z2_log_var, z1, z2]."""
    lin = lambda name, t: F.linear(t, sd[name + ".weight"], sd[name + ".bias"])
    f2 = _conv_stack(sd, "encoder_z2_layers", x, training, new_buffers)
    z2_mu, z2_lv = lin("fc_z2_mu", f2), lin("fc_z2_var", f2)
    z2 = vanilla_reparameterize(z2_mu, z2_lv, e2)
    plane = lin("embed_z2_code", z2).view(-1, img_size, img_size).unsqueeze(1)
    data = F.conv2d(x, sd["embed_data.weight"], sd["embed_data.bias"])
    f1 = _conv_stack(sd, "encoder_z1_layers", torch.cat([data, plane], dim=1), training, new_buffers)
    z1_mu, z1_lv = lin("fc_z1_mu", f1), lin("fc_z1_var", f1)
    z1 = vanilla_reparameterize(z1_mu, z1_lv, e1)
    h = torch.cat([lin("debed_z1_code", z1), lin("debed_z2_code", z2)], dim=1).view(-1, 512, 2, 2)
    for i in range(4):
        h = F.conv_transpose2d(h, sd[f"decoder.{i}.0.weight"], sd[f"decoder.{i}.0.bias"], stride=2, padding=1, output_padding=1)
        h = _bn_lrelu(sd, f"decoder.{i}.1", h, training, new_buffers)
    h = F.conv_transpose2d(h, sd["final_layer.0.weight"], sd["final_layer.0.bias"], stride=2, padding=1, output_padding=1)
    h = _bn_lrelu(sd, "final_layer.1", h, training, new_buffers)
    recons = torch.tanh(F.conv2d(h, sd["final_layer.3.weight"], sd["final_layer.3.bias"], padding=1))
    return [recons, x, z1_mu, z1_lv, z2_mu, z2_lv, z1, z2]


def hvae_loss(sd, recons, x, z1_mu, z1_lv, z2_mu, z2_lv, z1, z2, M_N):
    """hvae.py:201-229 (the keys as the reference spells them)."""
    p_mu = F.linear(z2, sd["recons_z1_mu.weight"], sd["recons_z1_mu.bias"])
    p_lv = F.linear(z2, sd["recons_z1_log_var.weight"], sd["recons_z1_log_var.bias"])
    kl = lambda m, l: torch.mean(-0.5 * torch.sum(1 + l - m ** 2 - l.exp(), dim=1), dim=0)
    kld_loss = -(kl(z1 - p_mu, p_lv) - kl(z1_mu, z1_lv) - kl(z2_mu, z2_lv))
    rl = F.mse_loss(recons, x)
    return {"loss": rl + M_N * kld_loss, "Reconstruction Loss": rl, "KLD": -kld_loss}


def lvae_forward(sd, x, eps, latent_dims=(4, 8, 16, 32, 128), training=True, new_buffers=None):
    """LVAE.forward (lvae.py:137-222) with the Gaussian draws injected (eps[0]: top latent; eps[1:]: the rungs, top-down)
    -> [recons, input, kl_div [B]]."""
    n = len(latent_dims)
    h, post = x, []
    for i in range(n):
        h = F.conv2d(h, sd[f"encoders.{i}.encoder.0.weight"], sd[f"encoders.{i}.encoder.0.bias"], stride=2, padding=1)
        h = _bn_lrelu(sd, f"encoders.{i}.encoder.1", h, training, new_buffers)
        f = torch.flatten(h, start_dim=1)
        post.append((F.linear(f, sd[f"encoders.{i}.encoder_mu.weight"], sd[f"encoders.{i}.encoder_mu.bias"]),
                     F.linear(f, sd[f"encoders.{i}.encoder_var.weight"], sd[f"encoders.{i}.encoder_var.bias"])))
    mu, lv = post.pop()
    z = vanilla_reparameterize(mu, lv, eps[0])
    post.reverse()
    kl_div = 0
    for i in range(n - 1):
        mu_e, lv_e = post[i]
        p = f"ladders.{i}"
        d = F.linear(z, sd[p + ".decode.0.weight"], sd[p + ".decode.0.bias"])
        rm, rv = sd[p + ".decode.1.running_mean"].detach().clone(), sd[p + ".decode.1.running_var"].detach().clone()
        d = F.batch_norm(d, rm, rv, sd[p + ".decode.1.weight"], sd[p + ".decode.1.bias"], training, BN_MOMENTUM, BN_EPS)
        if training and new_buffers is not None:
            new_buffers[p + ".decode.1.running_mean"], new_buffers[p + ".decode.1.running_var"] = rm, rv
        mu_t = F.linear(d, sd[p + ".fc_mu.weight"], sd[p + ".fc_mu.bias"])
        lv_t = F.linear(d, sd[p + ".fc_var.weight"], sd[p + ".fc_var.bias"])
        p1, p2 = 1. / (lv_e.exp() + 1e-7), 1. / (lv_t.exp() + 1e-7)
        mu_m, lv_m = (mu_e * p1 + mu_t * p2) / (p1 + p2), torch.log(1. / (p1 + p2))
        z = vanilla_reparameterize(mu_m, lv_m, eps[1 + i])
        kl = (lv_e - lv_m) + (lv_m.exp() + (mu_m - mu_e) ** 2) / (2 * lv_e.exp()) - 0.5
        kl_div = kl_div + torch.sum(kl, dim=-1)
    return [vanilla_decode(sd, z, training, new_buffers), x, kl_div]


def lvae_loss(recons, x, kl_div, M_N):
    """lvae.py:224-244."""
    rl, kld = F.mse_loss(recons, x), torch.mean(kl_div, dim=0)
    return {"loss": rl + M_N * kld, "Reconstruction_Loss": rl, "KLD": -kld}


def gamma_forward(sd, x, zhat, shape_b=8.0, training=True, new_buffers=None):
    """GammaVAE.forward (gamma_vae.py:93-149) with the Gamma(alpha + B, 1) draw injected -> [recons, input, alpha, beta]."""
    h = x
    for i in range(5):
        h = F.conv2d(h, sd[f"encoder.{i}.0.weight"], sd[f"encoder.{i}.0.bias"], stride=2, padding=1)
        h = _bn_lrelu(sd, f"encoder.{i}.1", h, training, new_buffers)
    flat = torch.flatten(h, start_dim=1)
    alpha = torch.softmax(F.linear(flat, sd["fc_mu.0.weight"], sd["fc_mu.0.bias"]), dim=1)
    beta = torch.softmax(F.linear(flat, sd["fc_var.0.weight"], sd["fc_var.0.bias"]), dim=1)
    a = alpha + shape_b
    eps = torch.sqrt(9. * a - 3.) * ((zhat / (a - 1. / 3.)) ** (1. / 3.) - 1.)
    z = (a - 1. / 3.) * (1 + eps / torch.sqrt(9. * a - 3.)) ** 3 / beta
    h = F.linear(z, sd["decoder_input.0.weight"], sd["decoder_input.0.bias"]).view(-1, 512, 2, 2)
    for i in range(4):
        h = F.conv_transpose2d(h, sd[f"decoder.{i}.0.weight"], sd[f"decoder.{i}.0.bias"], stride=2, padding=1, output_padding=1)
        h = _bn_lrelu(sd, f"decoder.{i}.1", h, training, new_buffers)
    h = F.conv_transpose2d(h, sd["final_layer.0.weight"], sd["final_layer.0.bias"], stride=2, padding=1, output_padding=1)
    h = _bn_lrelu(sd, "final_layer.1", h, training, new_buffers)
    return [torch.sigmoid(F.conv2d(h, sd["final_layer.3.weight"], sd["final_layer.3.bias"], padding=1)), x, alpha, beta]


def gamma_loss(recons, x, alpha, beta, prior_alpha=2.0, prior_beta=1.0):
    """gamma_vae.py:151-199."""
    def I(a, b, c, d):
        return -c * d / a - b * torch.log(a) - torch.lgamma(b) + (b - 1) * (torch.digamma(d) + torch.log(c))
    c, d = 1 / torch.tensor([prior_alpha]), torch.tensor([prior_beta])
    kld = torch.sum(I(c, d, c, d) - I(1 / alpha, beta, c, d), dim=1)
    rl = torch.mean(F.mse_loss(recons, x, reduction='none'), dim=(1, 2, 3))
    return {"loss": torch.mean(rl + kld, dim=0)}


def betatc_forward(sd, x, e):
    """BetaTCVAE.forward (betatc_vae.py:84-126) with the Gaussian draws injected -> [recons, input, mu, log_var, z]."""
    h = x
    for i in range(4):
        h = F.leaky_relu(F.conv2d(h, sd[f"encoder.{i}.0.weight"], sd[f"encoder.{i}.0.bias"], stride=2, padding=1), LEAKY)
    f = F.linear(torch.flatten(h, start_dim=1), sd["fc.weight"], sd["fc.bias"])
    mu, lv = F.linear(f, sd["fc_mu.weight"], sd["fc_mu.bias"]), F.linear(f, sd["fc_var.weight"], sd["fc_var.bias"])
    z = vanilla_reparameterize(mu, lv, e)
    h = F.linear(z, sd["decoder_input.weight"], sd["decoder_input.bias"]).view(-1, 32, 4, 4)
    for i in range(3):
        h = F.leaky_relu(F.conv_transpose2d(h, sd[f"decoder.{i}.0.weight"], sd[f"decoder.{i}.0.bias"], stride=2, padding=1,
                                            output_padding=1), LEAKY)
    h = F.leaky_relu(F.conv_transpose2d(h, sd["final_layer.0.weight"], sd["final_layer.0.bias"], stride=2, padding=1, output_padding=1), LEAKY)
    recons = torch.tanh(F.conv2d(h, sd["final_layer.2.weight"], sd["final_layer.2.bias"], padding=1))
    return [recons, x, mu, lv, z]


def betatc_loss(recons, x, mu, log_var, z, M_N, num_iter, anneal_steps, alpha, beta, gamma):
    """betatc_vae.py:142-205; num_iter = the counter value AFTER this (training) call's increment."""
    import math

    def logn(v, m, l):
        return -0.5 * (math.log(2 * math.pi) + l) - 0.5 * ((v - m) ** 2 * torch.exp(-l))
    B, D = z.shape
    rl = F.mse_loss(recons, x, reduction='sum')
    log_q_zx = logn(z, mu, log_var).sum(dim=1)
    log_p_z = logn(z, torch.zeros_like(z), torch.zeros_like(z)).sum(dim=1)
    mat = logn(z.view(B, 1, D), mu.view(1, B, D), log_var.view(1, B, D))
    N = (1 / M_N) * B
    strat = (N - B + 1) / (N * (B - 1))
    iw = torch.full((B, B), 1 / (B - 1))
    iw.view(-1)[::B] = 1 / N
    iw.view(-1)[1::B] = strat
    iw[B - 2, 0] = strat
    mat = mat + iw.log().view(B, B, 1)
    log_q_z = torch.logsumexp(mat.sum(2), dim=1)
    log_prod = torch.logsumexp(mat, dim=1).sum(1)
    mi, tc, kld = (log_q_zx - log_q_z).mean(), (log_q_z - log_prod).mean(), (log_prod - log_p_z).mean()
    anneal = min(num_iter / anneal_steps, 1)
    return {"loss": rl / B + alpha * mi + beta * tc + anneal * gamma * kld, "Reconstruction_Loss": rl, "KLD": kld, "TC_Loss": tc,
            "MI_Loss": mi}


def vamp_loss(sd, recons, x, mu, log_var, z, M_N, K, training=True, new_buffers=None):
    """vampvae.py:126-172: the pseudo-inputs go through the same encoder (train mode: their own batch statistics)."""
    pseudo = torch.clamp(F.linear(torch.eye(K), sd["embed_pseudo.0.weight"], sd["embed_pseudo.0.bias"]), 0.0, 1.0)
    prior_mu, prior_lv = vanilla_encode(sd, pseudo.view(-1, x.size(1), x.size(2), x.size(3)), training, new_buffers)
    rl = F.mse_loss(recons, x)
    e_log_q = torch.mean(torch.sum(-0.5 * (log_var + (z - mu) ** 2) / log_var.exp(), dim=1), dim=0)
    e_log_p = torch.sum(-0.5 * (prior_lv.unsqueeze(0) + (z.unsqueeze(1) - prior_mu.unsqueeze(0)) ** 2) / prior_lv.unsqueeze(0).exp(),
                        dim=2) - torch.log(torch.tensor(K).float())
    e_log_p = torch.mean(torch.logsumexp(e_log_p, dim=1), dim=0)
    kld = -(e_log_p - e_log_q)
    return {"loss": rl + M_N * kld, "Reconstruction_Loss": rl, "KLD": -kld}


def swae_loss(recons, x, z, prior_z, proj, reg_weight, p=2.0):
    """swae.py:109-126,150-178 with the prior draws and the unit directions proj [S, D] injected."""
    B = x.size(0)
    w = reg_weight / (B * (B - 1))
    lat, pri = z.matmul(proj.t()), prior_z.matmul(proj.t())                  # [N, S]
    w_dist = torch.sort(lat.t(), dim=1)[0] - torch.sort(pri.t(), dim=1)[0]
    swd = w * w_dist.pow(p).mean()
    rl = F.mse_loss(recons, x) + F.l1_loss(recons, x)
    return {"loss": rl + swd, "Reconstruction_Loss": rl, "SWD": swd}


def infovae_loss(recons, x, z, mu, log_var, prior_z, M_N, alpha, beta, reg_weight, kernel_type, z_var=2.0):
    """info_vae.py:128-148,218-229."""
    B = x.size(0)
    pp, zz, pz = mmd_terms(z, prior_z, kernel_type, z_var)
    mmd = pp + zz - 2 * pz
    rl = F.mse_loss(recons, x)
    kld = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
    loss = beta * rl + (1. - alpha) * M_N * kld + (alpha + reg_weight - 1.) / (B * (B - 1)) * mmd
    return {"loss": loss, "Reconstruction_Loss": rl, "MMD": mmd, "KLD": -kld}


def cvae_forward(sd, x, labels, e, training=True, new_buffers=None, img_size=64):
    """ConditionalVAE.forward (cvae.py:122-130) with the Gaussian noise injected -> [recons, input, mu, log_var]; the loss is
    VanillaVAE's (cvae.py:132-146 == vanilla_loss)."""
    y = labels.float()
    plane = F.linear(y, sd["embed_class.weight"], sd["embed_class.bias"]).view(-1, img_size, img_size).unsqueeze(1)
    data = F.conv2d(x, sd["embed_data.weight"], sd["embed_data.bias"])
    mu, log_var = vanilla_encode(sd, torch.cat([data, plane], dim=1), training, new_buffers)
    z = vanilla_reparameterize(mu, log_var, e)
    return [vanilla_decode(sd, torch.cat([z, y], dim=1), training, new_buffers), x, mu, log_var]


def joint_forward(sd, x, e, u, temp, training=True, new_buffers=None, eps=1e-7):
    """JointVAE.forward (joint_vae.py:110-170) with both noises injected -> [recons, input, q, mu, log_var]."""
    h = x
    for i in range(5):
        h = F.conv2d(h, sd[f"encoder.{i}.0.weight"], sd[f"encoder.{i}.0.bias"], stride=2, padding=1)
        h = _bn_lrelu(sd, f"encoder.{i}.1", h, training, new_buffers)
    flat = torch.flatten(h, start_dim=1)
    mu = F.linear(flat, sd["fc_mu.weight"], sd["fc_mu.bias"])
    log_var = F.linear(flat, sd["fc_var.weight"], sd["fc_var.bias"])
    q = F.linear(flat, sd["fc_z.weight"], sd["fc_z.bias"])
    z = e * torch.exp(0.5 * log_var) + mu
    g = -torch.log(-torch.log(u + eps) + eps)
    s = F.softmax((q + g) / temp, dim=-1)
    return [vanilla_decode(sd, torch.cat([z, s], dim=1), training, new_buffers), x, q, mu, log_var]


def joint_loss(recons, x, q, mu, log_var, M_N, num_iter, alpha, cont=(0.0, 25.0, 30.0, 25000), disc=(0.0, 25.0, 30.0, 25000), eps=1e-7):
    """joint_vae.py:172-237; cont / disc = (min capacity, max capacity, gamma, iterations); num_iter = counter value used."""
    import math
    Q = q.shape[-1]
    q_p = F.softmax(q, dim=-1)
    rl = F.mse_loss(recons, x)
    disc_curr = min((disc[1] - disc[0]) * num_iter / float(disc[3]) + disc[0], math.log(Q))
    kld_disc = torch.mean(torch.sum(q_p * torch.log(q_p + eps) - q_p * math.log(1. / Q + eps), dim=1), dim=0)
    cont_curr = min((cont[1] - cont[0]) * num_iter / float(cont[3]) + cont[0], cont[1])
    kld_cont = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
    cap = disc[2] * torch.abs(disc_curr - kld_disc) + cont[2] * torch.abs(cont_curr - kld_cont)
    return {"loss": alpha * rl + M_N * cap, "Reconstruction_Loss": rl, "Capacity_Loss": cap}


def dip_loss(recons, x, mu, log_var, M_N, lambda_diag, lambda_offdiag):
    """DIPVAE.loss_function (dip_vae.py:136-165), quirks included (centring over dim 1, scalar variance term)."""
    recons_loss = F.mse_loss(recons, x, reduction='sum')
    kld = torch.sum(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
    centered = mu - mu.mean(dim=1, keepdim=True)
    cov_mu = centered.t().matmul(centered).squeeze()
    cov_z = cov_mu + torch.mean(torch.diagonal((2. * log_var).exp(), dim1=0), dim=0)
    cov_diag = torch.diag(cov_z)
    cov_off = cov_z - torch.diag(cov_diag)
    dip = lambda_offdiag * torch.sum(cov_off ** 2) + lambda_diag * torch.sum((cov_diag - 1) ** 2)
    return {"loss": recons_loss + M_N * kld + dip, "Reconstruction_Loss": recons_loss, "KLD": -kld, "DIP_Loss": dip}


def logcosh_loss(recons, x, mu, log_var, M_N, alpha, beta):
    """LogCoshVAE.loss_function (logcosh_vae.py:135-155), written as the reference writes it."""
    t = recons - x
    rl = alpha * t + torch.log(1. + torch.exp(-2 * alpha * t)) - torch.log(torch.tensor(2.0))
    rl = (1. / alpha) * rl.mean()
    kld = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
    return {"loss": rl + beta * M_N * kld, "Reconstruction_Loss": rl, "KLD": -kld}


def miwae_forward(sd, x, eps, training=True, new_buffers=None):
    """IWAE.forward (iwae.py:119-124) / MIWAE.forward (miwae.py:124-130) with the noise injected: eps [B,S,D] or
    [B,M,S,D] fixes the sample dimensions -> [recons [B,(M,)S,C,H,W], input, mu, log_var, z, eps_out]."""
    mu, log_var = vanilla_encode(sd, x, training, new_buffers)
    lead, L = tuple(eps.shape[:-1]), eps.shape[-1]
    shape = (lead[0],) + (1,) * (len(lead) - 1) + (L,)
    mu = mu.view(shape).expand(lead + (L,))
    log_var = log_var.view(shape).expand(lead + (L,))
    z = eps * torch.exp(0.5 * log_var) + mu
    r = vanilla_decode(sd, z.reshape(-1, L), training, new_buffers)
    return [r.view(lead + tuple(r.shape[1:])), x, mu, log_var, z, (z - mu) / log_var]


def iw_loss(recons, x, mu, log_var, M_N):
    """iwae.py:126-160 / miwae.py:130-163: softmax over the LAST sample dimension, mean over everything in front."""
    lead = tuple(recons.shape[:-3])
    xin = x.view((x.shape[0],) + (1,) * (len(lead) - 1) + tuple(x.shape[1:])).expand(lead + tuple(x.shape[1:]))
    log_p_x_z = ((recons - xin) ** 2).flatten(len(lead)).mean(-1)
    kld = -0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=-1)
    log_weight = log_p_x_z + M_N * kld
    weight = F.softmax(log_weight, dim=-1)
    loss = torch.sum(weight * log_weight, dim=-1).mean()
    return {"loss": loss, "Reconstruction_Loss": log_p_x_z.mean(), "KLD": -kld.mean()}


def categorical_forward(sd, x, u, latent_dim, categorical_dim, temp, training=True, new_buffers=None, eps=1e-7):
    """CategoricalVAE.forward (cat_vae.py:90-138) with the uniform draws injected: -> [recons, input, q]."""
    h = x
    for i in range(5):
        h = F.conv2d(h, sd[f"encoder.{i}.0.weight"], sd[f"encoder.{i}.0.bias"], stride=2, padding=1)
        h = _bn_lrelu(sd, f"encoder.{i}.1", h, training, new_buffers)
    q = F.linear(torch.flatten(h, start_dim=1), sd["fc_z.weight"], sd["fc_z.bias"]).view(-1, latent_dim, categorical_dim)
    g = -torch.log(-torch.log(u + eps) + eps)
    s = F.softmax((q + g) / temp, dim=-1).view(-1, latent_dim * categorical_dim)
    return [vanilla_decode(sd, s, training, new_buffers), x, q]


def categorical_loss(recons, x, q, M_N, alpha, eps=1e-7):
    """cat_vae.py:140-169 (temperature annealing is host state of the model, not part of the arithmetic)."""
    import math
    q_p = F.softmax(q, dim=-1)
    recons_loss = F.mse_loss(recons, x)
    h1 = q_p * torch.log(q_p + eps)
    h2 = q_p * math.log(1.0 / q.shape[-1] + eps)
    kld = torch.mean(torch.sum(h1 - h2, dim=(1, 2)), dim=0)
    return {"loss": alpha * recons_loss + M_N * kld, "Reconstruction_Loss": recons_loss, "KLD": -kld}


# --------------------------------------------------------------------------------------------
# Vector quantisers  (models/mcq_vae.py:7-137)
# --------------------------------------------------------------------------------------------
def beta_loss(recons, x, mu, log_var, M_N, loss_type, num_iter, beta=4, gamma=1000.0, max_capacity=25, capacity_max_iter=1e5):
    """BetaVAE.loss_function (beta_vae.py:132-153); num_iter = value of the call counter AFTER its increment."""
    mse = F.mse_loss(recons, x)
    kld = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
    if loss_type == 'H':
        loss = mse + beta * M_N * kld
    elif loss_type == 'B':
        c_max = torch.tensor([float(max_capacity)])
        C = torch.clamp(c_max / capacity_max_iter * num_iter, 0, c_max[0])
        loss = mse + gamma * M_N * (kld - C).abs()
    else:
        raise ValueError('Undefined loss type.')
    return {'loss': loss, 'Reconstruction_Loss': mse, 'KLD': kld}


def vq_compute_inds(codebook, latents):
    """mcq_vae.py:26-39: expanded-form distances, first-min argmin.  latents [B,Dc,H,W] -> [B,H,W] i64."""
    lat = latents.permute(0, 2, 3, 1).contiguous()
    flat = lat.view(-1, codebook.shape[1])
    dist = torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1) \
        - 2 * torch.matmul(flat, codebook.t())
    return torch.argmin(dist, dim=1).view(lat.shape[:3])


def vq_compute_latents(codebook, latents, inds, beta):
    """mcq_vae.py:41-64: row gather, beta*commitment + embedding loss, straight-through x + (q - x)."""
    lat = latents.permute(0, 2, 3, 1).contiguous()
    q = codebook[inds.reshape(-1)].view(lat.shape)
    commitment = F.mse_loss(q.detach(), lat)
    embedding = F.mse_loss(q, lat.detach())
    vq_loss = commitment * beta + embedding
    q = lat + (q - lat).detach()
    return q.permute(0, 3, 1, 2).contiguous(), vq_loss


def mcq_compute_inds(sd, latents, codebooks, pfx="vq_layer"):
    """mcq_vae.py:100-110, including the slice-offset quirk latents[:, i:i+D/C]."""
    dc = latents.shape[1] // codebooks
    out = [vq_compute_inds(sd[f"{pfx}.quantizers.{i}.embedding.weight"], latents[:, i:i + dc]) for i in range(codebooks)]
    return torch.stack(out, 1)


def mcq_compute_latents(sd, latents, inds, codebooks, beta, pfx="vq_layer"):
    """mcq_vae.py:112-127."""
    dc = latents.shape[1] // codebooks
    qs, losses = [], []
    for i in range(codebooks):
        q, l = vq_compute_latents(sd[f"{pfx}.quantizers.{i}.embedding.weight"], latents[:, i:i + dc], inds[:, i], beta)
        qs.append(q)
        losses.append(l)
    return torch.cat(qs, 1), sum(losses)


# --------------------------------------------------------------------------------------------
# MCQVAE and the conv/VQ path of CTMCQVAE  (models/mcq_vae.py:142-317, ct_mcq_vae.py:365-469)
# --------------------------------------------------------------------------------------------
def _res(sd, pfx, x):
    """vq_vae.py:57-70: x + Conv1x1(ReLU(Conv3x3(x))), both bias-free."""
    h = F.relu(F.conv2d(x, sd[pfx + ".resblock.0.weight"], None, padding=1))
    return x + F.conv2d(h, sd[pfx + ".resblock.2.weight"], None)


def mcq_encode(sd, x, n_down=3):
    """mcq_vae.py:166-193."""
    h = x
    for i in range(n_down):
        h = F.leaky_relu(F.conv2d(h, sd[f"encoder.{i}.0.weight"], sd[f"encoder.{i}.0.bias"], stride=2, padding=1), LEAKY)
    h = F.leaky_relu(F.conv2d(h, sd[f"encoder.{n_down}.0.weight"], sd[f"encoder.{n_down}.0.bias"], padding=1), LEAKY)
    for j in range(6):
        h = _res(sd, f"encoder.{n_down + 1 + j}", h)
    h = F.leaky_relu(h, LEAKY)
    k = n_down + 8
    return F.leaky_relu(F.conv2d(h, sd[f"encoder.{k}.0.weight"], sd[f"encoder.{k}.0.bias"]), LEAKY)


def mcq_decode(sd, q, n_down=3):
    """mcq_vae.py:201-239."""
    h = F.leaky_relu(F.conv2d(q, sd["decoder.0.0.weight"], sd["decoder.0.0.bias"], padding=1), LEAKY)
    for j in range(6):
        h = _res(sd, f"decoder.{1 + j}", h)
    h = F.leaky_relu(h, LEAKY)
    for i in range(n_down - 1):
        h = F.leaky_relu(F.conv_transpose2d(h, sd[f"decoder.{8 + i}.0.weight"], sd[f"decoder.{8 + i}.0.bias"],
                                            stride=2, padding=1), LEAKY)
    k = 8 + n_down - 1
    return torch.tanh(F.conv_transpose2d(h, sd[f"decoder.{k}.0.weight"], sd[f"decoder.{k}.0.bias"], stride=2, padding=1))


def mcq_forward(sd, x, codebooks, beta, n_down=3, return_aux=False):
    """mcq_vae.py:262-265 -> [recons, input, vq_loss]."""
    lat = mcq_encode(sd, x, n_down)
    inds = mcq_compute_inds(sd, lat, codebooks)
    q, vq_loss = mcq_compute_latents(sd, lat, inds, codebooks, beta)
    out = [mcq_decode(sd, q, n_down), x, vq_loss]
    return (out, {"latents": lat, "inds": inds, "quantized": q}) if return_aux else out


def mcq_loss(recons, x, vq_loss):
    """mcq_vae.py:267-284 (values are NOT detached)."""
    recons_loss = F.mse_loss(recons, x)
    return {"loss": recons_loss + vq_loss, "Reconstruction_Loss": recons_loss, "VQ_Loss": vq_loss}


def ct_forward_conv_path(sd, x, y, codebooks, beta, n_down=3):
    """Conv+VQ+decoder part of CTMCQVAE.forward_action (ct_mcq_vae.py:525-546) with the causal layer
    replaced by identity on the indices (== skip_transition=True data flow): encoder on x (with grad),
    encoder on y (indices only, no grad path), decoder once, recon compared with y, vq_loss forced 0."""
    lat = mcq_encode(sd, x, n_down)
    inds = mcq_compute_inds(sd, lat, codebooks)
    with torch.no_grad():
        inds_y = mcq_compute_inds(sd, mcq_encode(sd, y, n_down), codebooks)
    q, _ = mcq_compute_latents(sd, lat, inds, codebooks, beta)
    return [mcq_decode(sd, q, n_down), y, torch.tensor(0.0)], {"inds": inds, "inds_y": inds_y}


# --------------------------------------------------------------------------------------------
# step helpers used by tests / the CPU baseline
# --------------------------------------------------------------------------------------------
def leafify(sd):
    """Detached float leaves with requires_grad (integer buffers are passed through)."""
    out = OrderedDict()
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            out[k] = v.detach().clone().requires_grad_(True)
        else:
            out[k] = v.detach().clone()
    return out


def vanilla_step(sd, x, eps, M_N):
    """forward + loss + backward.  Returns (loss dict, grads dict, new BN buffers, outputs)."""
    sd = leafify(sd)
    nb = OrderedDict()
    recons, _, mu, log_var = vanilla_forward(sd, x, eps, True, nb)
    losses = vanilla_loss(recons, x, mu, log_var, M_N)
    losses["loss"].backward()
    grads = OrderedDict((k, v.grad) for k, v in sd.items() if v.requires_grad)
    return losses, grads, nb, {"recons": recons.detach(), "mu": mu.detach(), "log_var": log_var.detach()}


def mcq_step(sd, x, codebooks, beta, n_down=3):
    sd = leafify(sd)
    (recons, _, vq_loss), aux = mcq_forward(sd, x, codebooks, beta, n_down, return_aux=True)
    losses = mcq_loss(recons, x, vq_loss)
    losses["loss"].backward()
    grads = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v))) for k, v in sd.items() if v.requires_grad)
    aux = {k: v.detach() for k, v in aux.items()}
    aux["recons"] = recons.detach()
    return {k: v.detach() for k, v in losses.items()}, grads, aux


def adam_steps(params, grads_fn, n_steps, lr, weight_decay=0.0, gamma=None):
    """experiment.py:152-187: optim.Adam(lr, weight_decay) default betas/eps; grads_fn(params)->grads."""
    ps = [p.detach().clone().requires_grad_(True) for p in params]
    opt = torch.optim.Adam(ps, lr=lr, weight_decay=weight_decay)
    for _ in range(n_steps):
        gs = grads_fn(ps)
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        opt.step()
    return [p.detach() for p in ps]


# ---- MSSIMVAE (mssim_vae.py:182-279) ------------------------------------------------------------------------------
MSSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def mssim_window(window_size=11, sigma=1.5):
    """mssim_vae.py:203-206: NOT a Gaussian -- the exponent is +x^2 / (2 sigma^2) (no minus sign), normalised to sum 1."""
    import math
    k = torch.tensor([math.exp((x - window_size // 2) ** 2 / (2 * sigma ** 2)) for x in range(window_size)])
    return k / k.sum()


def mssim_ssim(img1, img2, window_size=11):
    """mssim_vae.py:214-248 with size_average=True -> (mean ssim map, mean contrast-sensitivity map)."""
    C = img1.shape[1]
    w1 = mssim_window(window_size).unsqueeze(1)
    window = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0).expand(C, 1, window_size, window_size).contiguous()
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=C)
    mu2 = F.conv2d(img2, window, padding=pad, groups=C)
    s11 = F.conv2d(img1 * img1, window, padding=pad, groups=C) - mu1 * mu1
    s22 = F.conv2d(img2 * img2, window, padding=pad, groups=C) - mu2 * mu2
    s12 = F.conv2d(img1 * img2, window, padding=pad, groups=C) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    v1, v2 = 2.0 * s12 + C2, s11 + s22 + C2
    cs = torch.mean(v1 / v2)
    ssim = torch.mean(((2 * mu1 * mu2 + C1) * v1) / ((mu1 * mu1 + mu2 * mu2 + C1) * v2))
    return ssim, cs


def mssim_loss(img1, img2, window_size=11):
    """MSSIM.forward (mssim_vae.py:250-279): five levels, 2x2 average pooling between them;
    output = prod(pow1[:-1] * pow2[-1]) -- the last level's ssim power multiplies EACH of the four contrast powers."""
    weights = torch.tensor(MSSIM_WEIGHTS)
    ms, mc = [], []
    for _ in range(len(MSSIM_WEIGHTS)):
        s, c = mssim_ssim(img1, img2, window_size)
        ms.append(s)
        mc.append(c)
        img1, img2 = F.avg_pool2d(img1, (2, 2)), F.avg_pool2d(img2, (2, 2))
    ms, mc = torch.stack(ms), torch.stack(mc)
    pow1, pow2 = mc ** weights, ms ** weights
    return 1 - torch.prod(pow1[:-1] * pow2[-1])


def mssimvae_loss(recons, x, mu, log_var, M_N):
    """MSSIMVAE.loss_function (mssim_vae.py:130-152)."""
    rl = mssim_loss(recons, x)
    kld = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
    return {"loss": rl + M_N * kld, "Reconstruction_Loss": rl, "KLD": -kld}
