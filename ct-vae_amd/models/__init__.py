"""Mirror of the reference's ``models`` package for the hot-path models (models/__init__.py:1-56):
``vae_models[name](**config['model_params'])`` is how run.py:52 builds a model."""
from .base import BaseVAE
from .blocks import ResidualLayer
from .mcq_vae import MCQVAE, MultipleCodebookVectorQuantizer, VectorQuantizerMS
from .vanilla_vae import VanillaVAE
from .beta_vae import BetaVAE
from .vq_vae import VQVAE
from .cat_vae import CategoricalVAE
from .iwae import IWAE, MIWAE
from .logcosh_vae import LogCoshVAE
from .wae_mmd import WAE_MMD, InfoVAE
from .dip_vae import DIPVAE
from .joint_vae import JointVAE
from .cvae import ConditionalVAE
from .swae import SWAE
from .twostage_vae import TwoStageVAE
from .hvae import HVAE
from .vampvae import VampVAE
from .betatc_vae import BetaTCVAE
from .gamma_vae import GammaVAE
from .mssim_vae import MSSIMVAE
from .lvae import LVAE

# Aliases (models/__init__.py:29-32)
VAE = VanillaVAE
GaussianVAE = VanillaVAE
GumbelVAE = CategoricalVAE

vae_models = {
    'VanillaVAE': VanillaVAE,
    'VAE': VanillaVAE,
    'GaussianVAE': VanillaVAE,
    'MCQVAE': MCQVAE,
    'BetaVAE': BetaVAE,       # same network as VanillaVAE, beta / capacity objectives (beta_vae.py)
    'VQVAE': VQVAE,           # MCQ-VAE's conv stacks around one codebook (vq_vae.py)
    'IWAE': IWAE,             # VanillaVAE's network, importance-weighted bound over S samples (iwae.py)
    'MIWAE': MIWAE,           # ... over M estimates x S samples (miwae.py)
    'LogCoshVAE': LogCoshVAE, # VanillaVAE's network, log-cosh reconstruction term (logcosh_vae.py)
    'WAE_MMD': WAE_MMD,       # VanillaVAE's stacks, one deterministic head, mse + MMD (wae_mmd.py)
    'InfoVAE': InfoVAE,       # VanillaVAE's network, beta*mse + (1-alpha)*KL + MMD (info_vae.py)
    'DIPVAE': DIPVAE,         # VanillaVAE's network, sum-reduced objective + DIP-II covariance regulariser (dip_vae.py)
    'JointVAE': JointVAE,     # VanillaVAE's stacks, Gaussian + one categorical latent, capacity objective (joint_vae.py)
    'TwoStageVAE': TwoStageVAE,   # VanillaVAE's step; the second-stage MLPs are parameter holders, as in the reference (twostage_vae.py)
    'LVAE': LVAE,             # ladder VAE: per-level heads bottom-up, merge / sample / KL per rung top-down (lvae.py)
    'MSSIMVAE': MSSIMVAE,     # VanillaVAE's network, multi-scale SSIM reconstruction loss (mssim_vae.py)
    'GammaVAE': GammaVAE,     # VanillaVAE's stacks, Gamma latents (shape-augmentation reparameterisation, Gamma KL), Sigmoid output (gamma_vae.py)
    'BetaTCVAE': BetaTCVAE,   # own small conv net (no BatchNorm), total-correlation decomposition of the KL term (betatc_vae.py)
    'VampVAE': VampVAE,       # VanillaVAE's network, VampPrior over K learned pseudo-inputs (vampvae.py)
    'HVAE': HVAE,             # two latent levels, the second encoder conditioned on z2; three Gaussian-KL terms (hvae.py)
    'SWAE': SWAE,             # WAE_MMD's network, mse + l1 + sliced Wasserstein distance (swae.py)
    'ConditionalVAE': ConditionalVAE,   # VanillaVAE's stacks, the label as an extra input plane and next to z (cvae.py)
    'CategoricalVAE': CategoricalVAE,   # VanillaVAE's stacks around a Gumbel-softmax categorical latent (cat_vae.py)
}

try:  # CTMCQVAE needs nothing beyond torch, but keep the registry usable if it is being developed
    from .ct_mcq_vae import CTMCQVAE
    vae_models['CTMCQVAE'] = CTMCQVAE
except ImportError:  # pragma: no cover
    pass
