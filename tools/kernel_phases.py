"""Cycles between the phase marks of the causal-transition kernels (csrc/phase.hpp), one sampled workgroup per kernel:
    CTVAE_EXTRA_HIPCC_FLAGS=-DCTVAE_PHASES python -c "from ctvae_amd.build import build; build(force=True)"     (here, no GPU needed)
    gpurun -- 'python tools/kernel_phases.py gat "gat hidden"'        # or: pair "pair scores"
    python -c "from ctvae_amd.build import build; build(force=True)"                                             (back to the product build)
"""
import ctypes
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctvae_amd import native  # noqa: E402

name, cases = sys.argv[1], sys.argv[2]
if cases.startswith("bench:"):     # e.g. "bench:--no-graph --steps 2 --warmup 1 --no-configs --no-cpu-baseline --no-roofline": the kernels of a whole step
    sys.argv = [sys.argv[0]] + cases[6:].split()
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
else:
    sys.argv = [sys.argv[0], "--iters", "3", "--cases", cases]
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ct_kernels_bench.py"), run_name="__main__")
lib = native.load()
buf = (ctypes.c_longlong * (2 * 4 * 8 * 16))()
rc = getattr(lib, "ctvae_debug_phases_" + name)(buf)
print("rc", rc)
for k in range(4):
    rows = []
    for w in range(8):
        v = [buf[(k * 8 + w) * 16 + i] for i in range(16)]
        if not any(v):
            continue
        last = max(i for i in range(16) if v[i])
        wall = [buf[4 * 8 * 16 + (k * 8 + w) * 16 + i] for i in range(16)]
        ns = (wall[last] - wall[0]) * 10.0                      # s_memrealtime: 100 MHz
        rows.append((w, [v[i + 1] - v[i] if v[i + 1] and v[i] else None for i in range(last)], f"{v[last] - v[0]} cycles in {ns:.0f} ns = {(v[last] - v[0]) / max(ns, 1):.2f} GHz"))
    if rows:
        print("kernel slot", k)
        for w, d, tot in rows:
            print("  wave", w, d, "total", tot)
