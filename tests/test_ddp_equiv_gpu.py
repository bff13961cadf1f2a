"""GPU (one device): data-parallel equivalence of a REAL training step (SURVEY §4 iii, run.py:99 / §2.3 C1).

Two ranks with half the batch each are emulated one after the other on one GPU: each half-batch's backward writes the flat
gradient buffer, the two buffers are summed (what the RCCL SUM all-reduce of ddp.GradBucketAllReduce leaves on every rank)
and Adam runs with grad_scale = 1/2 (FlatAdam folds DDP's division by the world size into the update).  For models without
BatchNorm (MCQ-VAE, CT-MCQ-VAE's conv path) every loss term is a mean over samples, so the result must equal the
single-process step on the concatenated batch.  (VanillaVAE: BatchNorm statistics are per rank in the reference — plain DDP,
no SyncBN — so DDP(2 x B/2) != single(B) there by design; not tested.)"""
import numpy as np
import pytest
import torch

from ctvae_amd import filler
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def _build(dev, seed):
    from ctvae_amd.models import vae_models
    cfg = H.MCQ_CFG
    m = vae_models["MCQVAE"](**{**cfg, "hidden_dims": list(cfg["hidden_dims"])})
    m.load_state_dict(filler.fill_state(H.mcq_specs(cfg), seed + 1))
    return m.to(dev).train()


def _step_grads(m, x):
    from ctvae_amd import kernels as K
    m.zero_grad()
    out = m(x)
    losses = m.loss_function(*out)
    K.backward(losses["loss"])
    return m.flat_grads.clone(), float(losses["loss"])


@pytest.mark.parametrize("B", [8, 128])
def test_mcq_two_half_batches_equal_the_full_batch_step(dev, B):
    from ctvae_amd.optim import FlatAdam
    seed = 1320
    x, _ = filler.synthetic_batch(seed, B)
    xd = x.to(dev)
    full = _build(dev, seed)
    g_full, loss_full = _step_grads(full, xd)
    half = _build(dev, seed)
    g0, l0 = _step_grads(half, xd[:B // 2].contiguous())
    g1, l1 = _step_grads(half, xd[B // 2:].contiguous())
    assert abs(0.5 * (l0 + l1) - loss_full) <= 1e-5 * max(1.0, abs(loss_full))
    summed = g0 + g1                                       # the SUM all-reduce
    scale = float(g_full.abs().max())
    err = float((0.5 * summed - g_full).abs().max())
    rel = float((0.5 * summed - g_full).norm() / g_full.norm())
    # the half batches run other launches than the full batch (tile shapes, split-K, Winograd variants follow the batch), so
    # the comparison is between differently ordered fp32 sums (and Winograd against direct convolution at B = 128 / 64:
    # 4e-5 measured): relative L2 and every element within the 1e-4 parity bound
    assert rel <= 1e-4, f"mean of the half-batch gradients differs from the full-batch gradient: rel L2 {rel}"
    assert err <= 1e-4 * scale, f"max |diff| {err} (scale {scale})"
    # one Adam step: DDP path (summed gradient, grad_scale 1/2) against the single-process step
    opt_full, opt_half = FlatAdam(full, lr=5e-4), FlatAdam(half, lr=5e-4)
    full.flat_grads.copy_(g_full)
    opt_full.step()
    half.flat_grads.copy_(summed)
    opt_half.step(grad_scale=0.5)
    torch.cuda.synchronize()
    p_full, p_half = full.flat_params, half.flat_params
    # Adam normalises the gradient: where |g| is at rounding level the update direction is noise on both sides; compare
    # with an absolute bound of a fraction of one lr step on those, tightly elsewhere
    big = g_full.abs() > 1e-3 * scale
    assert float((p_full - p_half)[big].abs().max()) <= 2e-3 * 5e-4
    assert float((p_full - p_half).abs().max()) <= 2 * 5e-4
    np.testing.assert_allclose(float((p_full - p_half).abs().mean()), 0.0, atol=1e-2 * 5e-4)
