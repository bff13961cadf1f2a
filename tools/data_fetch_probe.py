"""Throughput of the HBM-resident input pipeline (ctvae_amd.data.HbmImageStore.fetch = ctvae_crop_resize_u8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctvae_amd import data as D, native
native.load()
dev = torch.device("cuda")
for name, (n, H, W) in {"CelebA-sized 218x178": (4096, 218, 178), "3DShapes-sized 64x64": (65536, 64, 64)}.items():
    imgs = torch.randint(0, 256, (n, H, W, 3), dtype=torch.uint8, device=dev)
    store = D.HbmImageStore(imgs, dev)
    for B in (128, 256, 1024):
        rows = torch.randint(0, n, (B,), device=dev)
        for _ in range(5):
            store.fetch(rows)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            store.fetch(rows)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"{name:22s} B={B:5d}: {us:7.1f} us per batch  {B / us:7.2f} M images/s   ({B * 64 * 64 * 3 * 4 / us / 1e3:6.1f} GB/s written)")
