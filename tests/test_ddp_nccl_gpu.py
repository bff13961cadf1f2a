"""GPU (one device, ONE process): the N > 1 training step of the harness under RCCL.

A 1-rank ``nccl`` (= RCCL) process group, ``VAEXperiment(ddp=GradBucketAllReduce(force=True))``: the step is then exactly what
every rank of an N-GPU job runs (run.py:99 DDPStrategy, experiment.py:152-160) -- hipGraph(zero_grad + forward + loss + backward
+ settling / gathering the flat gradient buffer), the bucketed SUM all-reduce of that buffer OUTSIDE the graph on the process
group's stream, then ``FlatAdam.step(grad_scale = 1 / world)``.  With one rank the all-reduce is the identity and the scale is 1,
so the parameter trajectory must equal the ``ddp=None`` harness (Adam inside the graph) bit for bit -- for VanillaVAE, MCQVAE
and CTMCQVAE (one mode per batch, ``update_parameters: ct_layer`` as in ct_mcq_vae.yaml:37, which also restricts the exchanged
range).  This is the test that catches a gradient which only reaches the flat buffer in eager Python (the CT layer's
autograd-produced gradients): from the second replay of a captured step on it would be exchanged and stepped as zero."""
import os
import socket

import pytest
import torch

from ctvae_amd import filler
from tests import helpers as H
from tests.test_ct_gpu import _FixedNoise, build_ct

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def one_rank_rccl(dev):
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield dist
    torch.cuda.synchronize()
    dist.destroy_process_group()


def _vanilla(dev):
    from ctvae_amd.models import vae_models
    m = vae_models["VanillaVAE"](in_channels=3, latent_dim=128)
    m.load_state_dict(filler.fill_state(H.vanilla_specs(), 1266))
    return m.to(dev).train()


def _mcq(dev):
    from ctvae_amd.models import vae_models
    m = vae_models["MCQVAE"](**{**H.MCQ_CFG, "hidden_dims": list(H.MCQ_CFG["hidden_dims"])})
    m.load_state_dict(filler.fill_state(H.mcq_specs(H.MCQ_CFG), 1321))
    return m.to(dev).train()


def _image_batches(dev, n, B):
    return [(filler.synthetic_batch(300 + i, B)[0].to(dev), torch.zeros(B, device=dev)) for i in range(n)]


def _run(build, batches, params, use_ddp):
    from ctvae_amd.ddp import GradBucketAllReduce
    from ctvae_amd.experiment import VAEXperiment
    torch.manual_seed(4242)        # VanillaVAE seeds the Philox key of its in-kernel latent noise from torch's generator
    m = build()
    start = m.flat_params.clone()
    ddp = GradBucketAllReduce(m, force=True) if use_ddp else None
    exp = VAEXperiment(m, dict(params, hipgraph=True), ddp=ddp)
    if use_ddp:
        assert ddp.active and ddp.world == 1 and ddp.grad_scale == 1.0
    exp.fit(lambda: iter(batches), None, max_epochs=1)
    torch.cuda.synchronize()
    assert exp._graphed and all(g.graph is not None for g in exp._graphed.values()), "the steps were not captured"
    return exp, m, start


@pytest.mark.parametrize("name", ["VanillaVAE", "MCQVAE"])
def test_image_models_step_under_rccl_equals_single_process(dev, one_rank_rccl, name):
    build = (lambda: _vanilla(dev)) if name == "VanillaVAE" else (lambda: _mcq(dev))
    batches = _image_batches(dev, 8, 8)        # 3 eager steps, the capture step, 4 replays
    params = {"LR": 5e-4, "weight_decay": 0.0, "scheduler_gamma": 0.95, "kld_weight": 0.00025}
    finals = {}
    for use_ddp in (False, True):
        exp, m, start = _run(build, batches, params, use_ddp)
        assert exp.global_step == 8
        finals[use_ddp] = m.flat_params.clone()
        assert not torch.equal(finals[use_ddp], start)
    assert torch.isfinite(finals[True]).all()
    assert torch.equal(finals[True], finals[False]), float((finals[True] - finals[False]).abs().max())


def test_ct_modes_step_under_rccl_equals_single_process(dev, one_rank_rccl):
    """CT-MCQ-VAE, one mode per batch, 8 batches per mode (3 eager, capture, 4 replays each), ``update_parameters: ct_layer``:
    bit-equal parameters with and without the exchange, nothing outside ``ct_layer`` moves, and only that range travels."""
    from ctvae_amd.models import causal
    B, A = 4, 12
    batches = []
    for i in range(24):
        x, y, a = filler.synthetic_pairs(100 + i, B, A)
        mode = ["base", "action", "causal"][i % 3]
        opts = {"mode": [mode] * B}
        if mode != "base":
            opts.update(input_y=y.to(dev), action=a.to(dev))
        batches.append((x.to(dev), torch.zeros(B, device=dev), opts))
    params = {"LR": 5e-4, "weight_decay": 0.0, "scheduler_gamma": 0.99, "kld_weight": 0.00025, "update_parameters": "ct_layer"}
    finals = {}
    prev = causal.set_noise_source(_FixedNoise(dev))
    try:
        for use_ddp in (False, True):
            exp, m, start = _run(lambda: build_ct(dev, 5), batches, params, use_ddp)
            assert exp.global_step == 24 and len(exp._graphed) == 3 and all(g.seen == 8 for g in exp._graphed.values())
            sl = m.flat_range("ct_layer")
            if use_ddp:
                assert exp.ddp.range == sl == exp.optimizer.slice
                assert sum(b.numel() for b in exp.ddp.buckets()) == sl.stop - sl.start
            end = m.flat_params.clone()
            assert torch.equal(end[:sl.start], start[:sl.start]) and torch.equal(end[sl.stop:], start[sl.stop:])
            moved = (end[sl] != start[sl]).float().mean().item()
            assert moved > 0.5, f"only {moved:.2%} of the ct_layer parameters moved"
            finals[use_ddp] = end
    finally:
        causal.set_noise_source(prev)
    assert torch.isfinite(finals[True]).all()
    assert torch.equal(finals[True], finals[False]), float((finals[True] - finals[False]).abs().max())
