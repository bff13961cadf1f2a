"""Where does a Winograd conv launch (wino.hip) spend its cycles?  Needs the diagnostic build:
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCTVAE_PHASE_TIMING -shared ct-vae_amd/csrc/*.hip -o tools/_timing/libctvae_timing.so
Prints the mean over waves of the shader-clock cycles per loop phase (summed over the K chunks)."""
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
spec = K.ConvSpec(K.CONV, C, C, 3, 1, 1, 0, K.ACT_RELU)
x = torch.randn(B, H, H, C, device=dev)
w = torch.randn(3, 3, C, C, device=dev) * 0.02
for _ in range(3):
    K.conv_forward_raw(x, w, None, spec)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
K.conv_forward_raw(x, w, None, spec)
e1.record()
torch.cuda.synchronize()
buf = np.zeros(8 * 4096, dtype=np.int64)
lib.ctvae_debug_wino_phase_read(buf.ctypes.data, buf.size)
t = buf.reshape(-1, 8)[:1024]
names = ["steps 0-7 (8 MFMA groups + 13 LDS stores + 13 global loads)", "barrier 1",
         "steps 8-15 (8 groups + patch reads + transform + V stores)", "barrier 2", "whole loop"]
print(f"event-timed launches (weight transform + conv): {e0.elapsed_time(e1) * 1e3:.1f} us")
for i, n in enumerate(names):
    print(f"  {n:62s} mean {t[:, i].mean():10.0f} cycles   per chunk {t[:, i].mean() / 31:7.0f}")
