"""Debug aid: ctvae_vq_backward's codebook gradient against a torch computation, per codebook."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctvae_amd import native
dev = torch.device("cuda")
for (B, HW, D, K, C) in [(2, 64, 128, 64, 4), (4, 64, 128, 64, 4), (256, 64, 128, 64, 4), (3, 64, 128, 64, 1)]:
    Dc = D // C
    g = torch.Generator().manual_seed(B)
    lat = torch.randn(B, HW, D, generator=g).to(dev)
    cb = torch.randn(C, K, Dc, generator=g).to(dev)
    inds = torch.randint(0, K, (B, C, HW), generator=g)
    inds[:, 1:, :] = (inds[:, 1:, :] % 3)          # skewed: long runs of the same code
    inds = inds.to(dev)
    gvq = torch.tensor(0.7, device=dev)
    dcb = torch.full_like(cb, 0.125)
    ws = native.workspace(dev)
    native.call("ctvae_vq_backward", None, gvq.data_ptr(), lat.data_ptr(), cb.data_ptr(), inds.data_ptr(), None,
                dcb.data_ptr(), 1, 0.25, B, HW, D, K, C, ws.data_ptr(), ws.numel() * 4)
    torch.cuda.synchronize()
    P = B * HW
    ref = torch.zeros_like(cb)
    x = lat.view(P, D)
    for i in range(C):
        idx = inds[:, i, :].reshape(P)
        cnt = torch.bincount(idx, minlength=K).float()
        sx = torch.zeros(K, Dc, device=dev).index_add_(0, idx, x[:, i:i + Dc])   # codebook i reads columns i .. i+Dc-1
        ref[i] = 0.7 * 2.0 / (P * Dc) * (cnt[:, None] * cb[i] - sx)
    print(B, HW, D, K, C, [f"{float((dcb[i] - 0.125 - ref[i]).abs().max()):.3e}" for i in range(C)], f"ref max {float(ref.abs().max()):.3e}")
