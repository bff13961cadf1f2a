"""Cycle budget of a Winograd weight-gradient launch (wino.hip).  Needs the diagnostic build, see wino_phase_probe.py."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ctvae_amd import native
native.LIB_PATH = os.environ.get("CTVAE_TIMING_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_timing", "libctvae_timing.so"))
from ctvae_amd import kernels as K

lib = native.load()
lib.ctvae_debug_wino_phase_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda")
B, H, C = 256, 8, 256
spec = K.ConvSpec(K.CONV, C, C, 3, 1, 1, 0, K.ACT_NONE)
x = torch.randn(B, H, H, C, device=dev)
dy = torch.randn(B, H, H, C, device=dev)
w = torch.nn.Parameter(torch.randn(3, 3, C, C, device=dev).permute(3, 2, 0, 1) * 0.02)
for _ in range(3):
    K.conv_wgrad_raw(x, dy, w, None, spec)
torch.cuda.synchronize()
buf = np.zeros(8 * 4096, dtype=np.int64)
lib.ctvae_debug_wino_phase_read(buf.ctypes.data, buf.size)
t = buf.reshape(-1, 8)[:1024, :5]
seg = buf[8 * 1024:8 * 2048].reshape(-1, 8)[:1024, :5]
names = ["entry", "prologue done", "main loop done", "last chunk done", "slab stores retired"]
for i, n in enumerate(names):
    d = t[:, i] - (t[:, i - 1] if i else 0)
    print(f"  {n:22s} at mean {t[:, i].mean():9.0f} cycles   phase mean {d.mean():9.0f} (min {d.min():8d} max {d.max():8d})")
print(f"  per chunk (31 chunks in the loop): {(t[:, 2] - t[:, 1]).mean() / 31:.0f} cycles")
names2 = ["steps 0-5 (6 MFMA groups + raw stores + global loads)", "barrier 1", "steps 6-13 (8 groups + transform)", "steps 14-15 (2 groups)", "barrier 2"]
for role, sel in (("X waves", [i for i in range(1024) if i % 4 < 2]), ("dY waves", [i for i in range(1024) if i % 4 >= 2])):
    print(role)
    for i, nme in enumerate(names2):
        print(f"    {nme:55s} {seg[sel, i].mean() / 31:8.0f} cycles per chunk")
