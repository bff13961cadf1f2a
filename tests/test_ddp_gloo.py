"""CPU, world_size 2, gloo: the gradient-bucket exchange (ddp.py) — parameter broadcast from rank 0, bucketed SUM
all-reduce of the flat gradient buffer, fused scalar reduction.  (RCCL replaces gloo on the GPUs; the call pattern
is identical.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ctvae_amd.ddp import GradBucketAllReduce
        from ctvae_amd.models import vae_models
        torch.manual_seed(100 + rank)                      # different init per rank on purpose
        m = vae_models["VanillaVAE"](in_channels=3, latent_dim=128)
        before = m.flat_params.clone()
        ddp = GradBucketAllReduce(m, bucket_bytes=4 << 20)  # 4 MiB buckets -> 4 buckets for 15.75 MB
        assert len(ddp.buckets()) == 4
        gathered = [torch.empty_like(m.flat_params) for _ in range(world)]
        dist.all_gather(gathered, m.flat_params)
        assert torch.equal(gathered[0], gathered[1]), "parameters not broadcast"
        if rank == 1:
            assert not torch.equal(before, m.flat_params)
        # gradients: rank r holds (r+1) * ramp; after SUM all-reduce both hold 3 * ramp; scale = 1/2
        ramp = torch.arange(m.flat_grads.numel(), dtype=torch.float32) % 97
        m.flat_grads.copy_((rank + 1) * ramp)
        ddp.all_reduce()
        assert torch.equal(m.flat_grads, 3 * ramp)
        assert float(m.fc_mu.bias.grad[0]) == float(3 * ramp[(m.fc_mu.bias.grad.data_ptr() - m.flat_grads.data_ptr()) // 4])
        assert ddp.grad_scale == 0.5
        # range exchange (the overlap path of bench.py: decoder-side range first, encoder-side range later)
        from ctvae_amd.ddp import SplitBackward
        sb = SplitBackward(m)
        assert sb.split == m.flat_range("decoder_input").start and 0 < sb.split < sb.total
        m.flat_grads.copy_((rank + 1) * ramp)
        ddp.wait(ddp.all_reduce_range(sb.split, sb.total))
        assert torch.equal(m.flat_grads[sb.split:], 3 * ramp[sb.split:])
        assert torch.equal(m.flat_grads[:sb.split], (rank + 1) * ramp[:sb.split]), "range exchange touched the other side"
        ddp.wait(ddp.all_reduce_range(0, sb.split))
        assert torch.equal(m.flat_grads, 3 * ramp)
        red = ddp.reduce_scalars({"loss": torch.tensor(float(rank)), "KLD": torch.tensor(2.0 * rank)})
        assert abs(float(red["loss"]) - 0.5) < 1e-7 and abs(float(red["KLD"]) - 1.0) < 1e-7
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == {0: "ok", 1: "ok"}, res
