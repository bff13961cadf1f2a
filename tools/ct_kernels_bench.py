"""Stand-alone timing of the causal-transition layer kernels at the bench shapes (B = 128 action-mode pairs, action_dim 12):
    python tools/ct_kernels_bench.py [--B 128] [--A 12] [--iters 20]            (on the GPU box)
Runs K.GATLayer (hidden layer: 1 + A heads x 100 channels; last layer: 2 head slots x 64) and K.PairScores (hidden 800, two
discoverers) forward + backward on seeded synthetic inputs of the step's shapes and sparsity, and prints the library profiler's
per-kernel average (HIP events per launch).  Fresh process per env setting when sweeping kernel diagnostics."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctvae_amd import kernels as K, native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=128)
    ap.add_argument("--A", type=int, default=12)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--density", type=float, default=0.5)
    ap.add_argument("--cases", default="", help="comma list out of: gat hidden, gat last, pair scores (default all)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(7)
    B, H = args.B, 1 + args.A

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)

    keep = (torch.rand(B, 64, 64, generator=g) < args.density).float()
    adj = (keep * torch.rand(B, 64, 64, generator=g)).to(dev).requires_grad_(True)
    grp = torch.randint(1, H, (B,), generator=g, dtype=torch.int32).to(dev)
    cases = []
    # hidden GATv2 layer: every head, 100 channels, LeakyReLU behind it
    xlr1 = rnd(B, 64, 2 * H * 100, scale=0.5).requires_grad_(True)
    we1, att1, b1 = rnd(H, 100, scale=0.3).requires_grad_(True), rnd(H, 100, scale=0.3).requires_grad_(True), rnd(H * 100, scale=0.1).requires_grad_(True)
    cases.append(("gat hidden", lambda: K.GATLayer.apply(xlr1, adj, we1, att1, b1, None, H, 100, 0.2, K.ACT_LRELU)))
    # last layer: head 0 and head 1 + action, 64 channels
    xlr2 = rnd(B, 64, 2 * 2 * 64, scale=0.5).requires_grad_(True)
    we2, att2, b2 = rnd(H, 64, scale=0.3).requires_grad_(True), rnd(H, 64, scale=0.3).requires_grad_(True), rnd(H * 64, scale=0.1).requires_grad_(True)
    hm = torch.stack([torch.zeros_like(grp), grp], dim=1).contiguous()
    cases.append(("gat last", lambda: K.GATLayer.apply(xlr2, adj, we2, att2, b2, hm, 2, 64, 0.2, K.ACT_NONE)))
    # the discoverers' pair scorers
    uv = rnd(B, 64, 4 * 800, scale=0.3).requires_grad_(True)
    w2, bb2 = rnd(H, 800, scale=0.05).requires_grad_(True), rnd(H, scale=0.1).requires_grad_(True)
    cases.append(("pair scores", lambda: K.PairScores.apply(uv, w2, bb2, grp, 800)))

    if args.cases:
        cases = [c for c in cases if c[0] in args.cases.split(",")]

    def run_all():
        for _, fn in cases:
            out = fn()
            out.backward(torch.ones_like(out) * 0.01)

    for _ in range(3):
        run_all()
    torch.cuda.synchronize()
    native.prof_enable(True)
    native.prof_calibrate(64)
    for _ in range(args.iters):
        run_all()
    torch.cuda.synchronize()
    native.prof_enable(False)
    rep = native.prof_report()
    empty = rep.pop("(empty event pair)", None)
    base = empty["ms"] / empty["count"] if empty else 0.0
    tot = 0.0
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
        us = (v["ms"] / v["count"] - base) * 1e3
        per_iter = us * v["count"] / args.iters
        tot += per_iter
        print(f"{us:9.1f} us x {v['count'] / args.iters:4.1f}  = {per_iter:8.1f} us/iter   {k}")
    print(f"total {tot:.1f} us/iter")


if __name__ == "__main__":
    main()
