"""Phase timeline of up_bwd_kernel (diagnostic build with -DCTVAE_PHASE_TIMING), row r = 1 of every workgroup's first tile."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ctvae_amd import native
native.LIB_PATH = os.environ.get("CTVAE_TIMING_LIB", "/tmp/libctvae_timing.so")
lib = native.load()
lib.ctvae_debug_upbwd_phase_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda")
B = 256
x = torch.randn(B, 32, 32, 32, device=dev)
w = torch.randn(9 * 32 * 32, device=dev)
ga = torch.randn(B, 64, 64, 32, device=dev)
y = torch.randn(B, 64, 64, 32, device=dev)
coef = torch.rand(5 * 32, device=dev)
gx = torch.empty_like(x)
dw = torch.zeros(9 * 32 * 32, device=dev)
db = torch.zeros(32, device=dev)
ws = native.workspace(dev)
for _ in range(3):
    native.call("ctvae_convt_bn_backward", x.data_ptr(), w.data_ptr(), ga.data_ptr(), y.data_ptr(), coef.data_ptr(), 1, gx.data_ptr(),
                dw.data_ptr(), db.data_ptr(), B, 32, 32, 32, 32, 3, 2, 1, 1, 0, ws.data_ptr(), ws.numel() * 4)
torch.cuda.synchronize()
buf = np.zeros(512 * 64, dtype=np.uint64)
lib.ctvae_debug_upbwd_phase_read(buf.ctypes.data, buf.size)
t = buf.reshape(512, 4, 16).astype(np.int64)
names = ["tile start (patch stored)", "prologue rows loaded", "r1 rows stored", "r1 barrier", "r1 loads issued", "r1 wgrad done", "r1 dgrad done",
         "r1 barrier", "r1 gx stored", "r1 barrier", "first tile done", "kernel end"]
t0 = t[:, :, 0].min()
for wv in range(4):
    print("wave", wv)
    prev = None
    for i, n in enumerate(names):
        v = (t[:, wv, i] - t0) / 100.0
        d = (v - prev) if prev is not None else v
        print(f"  {n:28s} at {v.mean():8.2f} us  (+{d.mean():6.2f})")
        prev = v
