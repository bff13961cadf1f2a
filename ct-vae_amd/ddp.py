"""Data-parallel gradient exchange: all-reduce over the model's flat gradient buffer.

Replaces what the reference gets implicitly from Lightning's ``DDPStrategy`` -> torch
``DistributedDataParallel`` (run.py:99): gradient buckets only, no model sharding,
per-rank BatchNorm statistics (no SyncBN in the reference).  Because parameters and gradients are single
flat fp32 buffers, the exchange is a handful of large RCCL all-reduces (xGMI is point-to-point: few big
messages beat 48 small ones) launched on a side stream as soon as the corresponding part of the backward
pass has been issued; ``1/world`` is folded into the optimizer's ``grad_scale``.

Backend "nccl" is RCCL on ROCm; "gloo" works on CPU tensors for the 2-rank unit tests.
"""
import torch
import torch.distributed as dist


class GradBucketAllReduce:
    def __init__(self, model, process_group=None, bucket_bytes=32 << 20, broadcast_from=0):
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.comm_stream = None
        if self.world > 1 and broadcast_from is not None:
            self.broadcast_parameters(broadcast_from)

    # C3: parameters (and buffers) from rank 0 once at construction
    def broadcast_parameters(self, src=0):
        dist.broadcast(self.model.flat_params, src, group=self.pg)
        for b in self.model.buffers():
            if b.is_floating_point():
                dist.broadcast(b, src, group=self.pg)

    def buckets(self):
        g = self.model.flat_grads
        n = g.numel()
        out, off = [], 0
        while off < n:
            e = min(n, off + self.bucket_elems)
            out.append(g[off:e])
            off = e
        return out

    def all_reduce(self, async_op=False):
        """SUM all-reduce of every bucket (division by world size is the caller's grad_scale)."""
        if self.world == 1:
            return []
        g = self.model.flat_grads
        if g.is_cuda:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=g.device)
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.pg, async_op=True) for b in self.buckets()]
            if not async_op:
                self.wait(works)
            return works
        works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.pg, async_op=True) for b in self.buckets()]
        if not async_op:
            self.wait(works)
        return works

    def wait(self, works):
        for w in works:
            w.wait()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def reduce_scalars(self, values: dict) -> dict:
        """C4: one fused all-reduce(mean) for all logged scalars instead of one per key."""
        keys = sorted(values)
        if self.world == 1 or not keys:
            return dict(values)
        t = torch.stack([values[k].detach().float().reshape(()) for k in keys])
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        t /= self.world
        return {k: t[i] for i, k in enumerate(keys)}
