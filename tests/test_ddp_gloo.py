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


def _adam_cpu(flat_params, flat_grads, exp_avg, exp_avg_sq, state, grad_scale=1.0):
    """TEST DOUBLE of the ctvae_adam_step kernel (csrc/pointwise.hip) so that the harness logic around it can run under
    gloo on the CPU: torch.optim.Adam's rule on the flat buffers, state = [step, lr, b1, b2, eps, wd, b1^t, b2^t]."""
    st = state
    st[0] += 1
    st[6] *= st[2]
    st[7] *= st[3]
    g = flat_grads * grad_scale + st[5] * flat_params
    exp_avg.mul_(st[2]).add_(g, alpha=float(1 - st[2]))
    exp_avg_sq.mul_(st[3]).addcmul_(g, g, value=float(1 - st[3]))
    denom = (exp_avg_sq / (1 - st[7])).sqrt_().add_(st[4])
    flat_params.addcdiv_(exp_avg / (1 - st[6]), denom, value=-float(st[1]))


def _harness_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ctvae_amd import kernels as K
        from ctvae_amd.data import shard_rows
        from ctvae_amd.ddp import GradBucketAllReduce
        from ctvae_amd.experiment import VAEXperiment
        from ctvae_amd.models import vae_models
        K.adam_step = _adam_cpu                               # the product has no CPU Adam; see _adam_cpu
        torch.manual_seed(7 + rank)
        cfg = dict(in_channels=3, embedding_dim=16, hidden_dims=[8, 16], num_embeddings=8, img_size=64, codebooks=1, beta=0.25)
        m = vae_models["MCQVAE"](**cfg)
        ddp = GradBucketAllReduce(m, bucket_bytes=1 << 16)
        exp = VAEXperiment(m, {"LR": 1e-3, "weight_decay": 0.0, "scheduler_gamma": 0.95, "kld_weight": 1.0}, ddp=ddp)
        p0 = m.flat_params.clone()
        n = m.flat_grads.numel()
        # three steps of rank-specific gradients through VAEXperiment.optimizer_step(): all-reduce(SUM) + Adam(grad_scale=1/W)
        single = p0.clone()
        ea, es = torch.zeros(n), torch.zeros(n)
        st = exp.optimizer.state.clone()
        for step in range(3):
            gens = [torch.Generator().manual_seed(1000 * step + r) for r in range(world)]
            grads = [torch.randn(n, generator=g) for g in gens]
            m.flat_grads.copy_(grads[rank])
            exp.optimizer_step()
            _adam_cpu(single, sum(grads) / world, ea, es, st)   # what ONE process sees for the mean gradient
        torch.testing.assert_close(m.flat_params, single, rtol=1e-6, atol=1e-7)
        gathered = [torch.empty_like(m.flat_params) for _ in range(world)]
        dist.all_gather(gathered, m.flat_params)
        assert torch.equal(gathered[0], gathered[1]), "ranks diverged"
        assert exp.global_step == 3 and not torch.equal(m.flat_params, p0)
        # update_parameters (the YAML's second training stage): only that range of the flat buffer travels and is stepped
        ddp2 = GradBucketAllReduce(m, bucket_bytes=1 << 12, broadcast_from=None)
        exp2 = VAEXperiment(m, {"LR": 1e-3, "kld_weight": 1.0, "update_parameters": "decoder"}, ddp=ddp2)
        sl = m.flat_range("decoder")
        lo, hi = sl.start, sl.stop
        assert 0 < lo < hi <= n and ddp2.range == sl and sum(b.numel() for b in ddp2.buckets()) == hi - lo
        before = m.flat_params.clone()
        mine_g = torch.full((n,), float(rank + 1))
        m.flat_grads.copy_(mine_g)
        exp2.optimizer_step()
        assert torch.equal(m.flat_grads[lo:hi], torch.full((hi - lo,), 3.0)), "range not summed over the ranks"
        assert torch.equal(m.flat_grads[:lo], mine_g[:lo]) and torch.equal(m.flat_grads[hi:], mine_g[hi:]), "outside moved"
        assert torch.equal(m.flat_params[:lo], before[:lo]) and torch.equal(m.flat_params[hi:], before[hi:])
        assert not torch.equal(m.flat_params[lo:hi], before[lo:hi])
        # plain datasets: every rank must see the same number of batches (129 rows, 2 ranks, bs 64 -> 65 rows each, 2 batches)
        for n_rows, bs in ((129, 64), (130, 64), (7, 4), (64, 64)):
            order = torch.arange(n_rows)
            mine = shard_rows(order, rank, world)
            nb = torch.tensor([-(-len(mine) // bs)])
            counts = [torch.zeros_like(nb) for _ in range(world)]
            dist.all_gather(counts, nb)
            assert all(int(c) == int(nb) for c in counts), (n_rows, bs, counts)
            for _ in range(int(nb)):                          # one collective per batch, as training / validation do
                dist.all_reduce(torch.ones(1))
            rows = [torch.empty(len(mine), dtype=order.dtype) for _ in range(world)]
            dist.all_gather(rows, mine)
            assert set(torch.cat(rows).tolist()) == set(range(n_rows)), "rows lost by the sharding"
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_harness_step_and_equal_batch_counts():
    """(i) both ranks run VAEXperiment.optimizer_step() on rank-specific flat gradients and end with identical parameters,
    equal to one process stepping on the mean gradient; (ii) run.py's sharding of a plain dataset whose length is not a
    multiple of world * batch_size gives every rank the same number of batches (a rank with one more would hang in RCCL)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_harness_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == {0: "ok", 1: "ok"}, res
