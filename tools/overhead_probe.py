"""Fixed cost vs per-chunk cost of the tap-GEMM kernel: same output tile grid (M=16384, N=128 -> 512 workgroups),
growing reduction length.  Times back-to-back launches with HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctvae_amd import kernels as K, native

native.load()
dev = torch.device("cuda")
B, H, Co = int(os.environ.get("PB", 256)), int(os.environ.get("PH", 8)), int(os.environ.get("PCO", 128))
for ci, k in [(32, 1), (64, 1), (32, 3), (64, 3), (128, 3), (256, 3), (512, 3)]:
    spec = K.ConvSpec(K.CONV, ci, Co, k, 1, k // 2, 0, K.ACT_LRELU)
    x = torch.randn(B, H, H, ci, device=dev)
    w = torch.randn(k, k, ci, Co, device=dev) * 0.05
    b = torch.randn(Co, device=dev)
    for _ in range(3):
        K.conv_forward_raw(x, w, b, spec)
    torch.cuda.synchronize()
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        K.conv_forward_raw(x, w, b, spec)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    chunks = k * k * ci // 32
    fl = 2.0 * B * H * H * Co * ci * k * k
    print(f"Ci={ci:4d} k={k} chunks={chunks:4d}  {us:8.2f} us  {fl / us / 1e6:7.1f} TF/s  ({us / chunks:.3f} us/chunk)")
