"""Where does a tap-GEMM launch spend its time?  Needs the diagnostic build:
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCTVAE_PHASE_TIMING -shared ct-vae_amd/csrc/*.hip -o /tmp/libctvae_timing.so
Prints, per configuration, the mean over workgroups of each phase (us) and the launch span."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ctvae_amd import native
native.LIB_PATH = os.environ.get("CTVAE_TIMING_LIB", "/tmp/libctvae_timing.so")
from ctvae_amd import kernels as K

lib = native.load()
lib.ctvae_debug_phase_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda")
# shapes "B,H,ci,co,k,stride" on the command line; default: round 1's bs = 256 cases
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(256, 8, 32, 128, 1, 1), (256, 8, 64, 128, 3, 1), (256, 8, 256, 128, 3, 1), (256, 16, 64, 128, 3, 1)]
for B, H, ci, Co, k, stride in shapes:
    spec = K.ConvSpec(K.CONV, ci, Co, k, stride, k // 2, 0, K.ACT_LRELU)
    x = torch.randn(B, H, H, ci, device=dev)
    w = torch.randn(k, k, ci, Co, device=dev) * 0.05
    b = torch.randn(Co, device=dev)
    for _ in range(3):
        K.conv_forward_raw(x, w, b, spec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.conv_forward_raw(x, w, b, spec)
    e1.record()
    torch.cuda.synchronize()
    nwg = 8192
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    lib.ctvae_debug_phase_read(buf.ctypes.data, buf.size)
    t = buf.reshape(8192, 8)[:nwg][:, [0, 6, 7, 1, 2, 3, 4, 5]].astype(np.int64)
    t = t[t[:, 0] > t[:, 0].max() - 3000]          # this launch's workgroups (entries of earlier, larger grids are older than 1 ms)
    nwg = len(t)
    t0 = t[:, 0].min()
    if (t[:, 6:8] < t0).any():                        # split-K workgroups return before the store marks
        t[:, 6:8] = t[:, 5:6]
    rel = (t - t0) / 100.0
    names = ["entry", "class / tile decoded", "row offsets, sOut", "tap masks (prologue done)", "first chunk in LDS", "main loop done", "stores issued", "stores retired"]
    print(f"B={B} H={H} Ci={ci} Co={Co} k={k} s={stride} chunks={k * k * ci // 32} WGs={nwg}  event-timed launch {e0.elapsed_time(e1) * 1e3:.1f} us, "
          f"span first entry -> last retire {rel[:, 7].max():.1f} us")
    if hasattr(lib, "ctvae_debug_loop_read"):
        lib.ctvae_debug_loop_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lb = np.zeros(8192 * 4, dtype=np.int64)
        lib.ctvae_debug_loop_read(lb.ctypes.data, lb.size)
        lb = lb.reshape(8192, 4)
        lb = lb[lb.sum(axis=1) > 0][:nwg]
        nchunk = max(1, (k * k * ci // 32))
        if len(lb):      # only the serial-issue diagnostic build (-DCTVAE_TILE_SERIAL_ISSUE) runs the loop clocks
            print("    pipelined loop, shader cycles per chunk (mean over workgroups; marks cost ~100 cycles each): "
                  f"load wait + LDS store {lb[:, 0].mean() / nchunk:.0f}, issue + 8 MFMA {lb[:, 1].mean() / nchunk:.0f}, "
                  f"barrier {lb[:, 2].mean() / nchunk:.0f}, reads + 8 MFMA {lb[:, 3].mean() / nchunk:.0f}")
    for i, n in enumerate(names):
        d = rel[:, i] - (rel[:, i - 1] if i else 0)
        print(f"    {n:20s} at mean {rel[:, i].mean():7.2f} (min {rel[:, i].min():6.2f} max {rel[:, i].max():6.2f})   phase mean {d.mean():6.2f} us")
