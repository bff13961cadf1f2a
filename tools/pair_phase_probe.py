"""Per-workgroup phase times of conv_bwd_pair_kernel's two roles (diagnostic build, see tools/phase_probe.py):
    CTVAE_TIMING_LIB=tools/_timing/libctvae_timing.so python tools/pair_phase_probe.py B,H,ci,co,k,stride ...
The layer is a Conv2d(ci, co, k, stride) on a B x H x H x ci input; the launch is its backward pass (data + weight gradient)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ctvae_amd import native
native.LIB_PATH = os.environ.get("CTVAE_TIMING_LIB", "/tmp/libctvae_timing.so")
from ctvae_amd import kernels as K

lib = native.load()
for f in (lib.ctvae_debug_phase_read, lib.ctvae_debug_wphase_read):
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda")


def show(title, buf, names, cols):
    t = buf.reshape(8192, 8)[:, cols].astype(np.int64)
    t = t[t[:, 0] > t[:, 0].max() - 3000]
    if (t[:, -2:] < t[:, 0:1]).any():
        t[:, -2:] = t[:, -3:-2]
    t0 = t[:, 0].min()
    rel = (t - t0) / 100.0
    print(f"  {title}: {len(t)} workgroups, first entry -> last mark {rel.max():.1f} us")
    for i, n in enumerate(names):
        d = rel[:, i] - (rel[:, i - 1] if i else 0)
        print(f"    {n:28s} at mean {rel[:, i].mean():7.2f} (min {rel[:, i].min():6.2f} max {rel[:, i].max():6.2f})   phase mean {d.mean():6.2f} us")
    return t0


for arg in sys.argv[1:]:
    B, H, ci, co, k, stride = (int(v) for v in arg.split(","))
    spec = K.ConvSpec(K.CONV, ci, co, k, stride, k // 2, 0, K.ACT_NONE)
    ho, wo = spec.out_hw(H, H)
    x = torch.randn(B, H, H, ci, device=dev)
    dy = torch.randn(B, ho, wo, co, device=dev)
    w = torch.nn.Parameter(torch.randn(k, k, ci, co, device=dev) * 0.05)
    b = torch.nn.Parameter(torch.randn(co, device=dev))
    for _ in range(3):
        K.conv_backward_raw(x, dy, w, b, spec)
    torch.cuda.synchronize()
    print(f"B={B} H={H} Ci={ci} Co={co} k={k} s={stride}")
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    lib.ctvae_debug_phase_read(buf.ctypes.data, buf.size)
    ta = show("data-gradient role", buf, ["entry", "class / tile decoded", "row offsets, sOut", "tap masks (prologue done)", "first chunk in LDS",
                                          "main loop done", "stores issued", "stores retired"], [0, 6, 7, 1, 2, 3, 4, 5])
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    lib.ctvae_debug_wphase_read(buf.ctypes.data, buf.size)
    tb = show("weight-gradient role", buf, ["entry", "constants", "first chunk in LDS", "main loop done", "stores issued", "stores retired"], [0, 1, 2, 3, 4, 5])
    print(f"  weight-gradient role's first entry {(tb - ta) / 100.0:+.2f} us after the data-gradient role's")
