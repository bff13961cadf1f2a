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
names = ["store raw+U", "issue global loads", "mfma f0-7", "barrier 1", "patch read + mfma f8-11", "transform + mfma f12-15",
         "barrier 2", "whole loop"]
print(f"event-timed launches (weight transform + conv): {e0.elapsed_time(e1) * 1e3:.1f} us")
for i, n in enumerate(names):
    print(f"  {n:28s} mean {t[:, i].mean():10.0f} cycles  (min {t[:, i].min():8d} max {t[:, i].max():8d})   per chunk {t[:, i].mean() / 32:7.0f}")
