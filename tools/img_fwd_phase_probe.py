"""Phase timeline of img_fwd_kernel (diagnostic build with -DCTVAE_PHASE_TIMING)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ctvae_amd import native
native.LIB_PATH = os.environ.get("CTVAE_TIMING_LIB", "/tmp/libctvae_timing.so")
from ctvae_amd import kernels as K
lib = native.load()
lib.ctvae_debug_img_phase_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda")
B = 256
spec = K.ConvSpec(K.CONV, 32, 3, 3, 1, 1, 0, K.ACT_TANH)
x = torch.randn(B, 64, 64, 32, device=dev)
w = torch.randn(3, 3, 32, 3, device=dev)
b = torch.zeros(3, device=dev)
coef = torch.rand(64, device=dev)
for _ in range(3):
    K.conv_forward_raw(x, w, b, spec, in_coef=coef, in_act=K.ACT_LRELU)
torch.cuda.synchronize()
buf = np.zeros(1024 * 32, dtype=np.uint64)
lib.ctvae_debug_img_phase_read(buf.ctypes.data, buf.size)
t = buf.reshape(1024, 32)[:768].astype(np.int64)
rel = (t - t[:, 0].min()) / 100.0
names = ["start"] + [f"tile{k} {n}" for k in range(5) for n in ("stored", "barrier", "prefetch issued", "MFMA done", "Z written+barrier", "gather done")]
for i, n in enumerate(names[:31]):
    d = rel[:, i] - (rel[:, i - 1] if i else 0)
    print(f"{n:28s} at {rel[:, i].mean():8.2f} us  (+{d.mean():6.2f})")
