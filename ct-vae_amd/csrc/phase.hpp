// In-kernel phase clocks (diagnostic, compiled out unless -DCTVAE_PHASES): CTVAE_PH(kernel, slot) stores s_memtime of one
// chosen workgroup's waves; tools/kernel_phases.py builds the library with CTVAE_EXTRA_HIPCC_FLAGS=-DCTVAE_PHASES and prints the
// cycles between consecutive slots.  This is how the serial loops / latency-bound LDS phases of the causal-transition kernels were
// found; the instrumented build is never the one that is benchmarked (s_memtime costs ~10 % wave cycles).
#pragma once
#include <hip/hip_runtime.h>

#ifdef CTVAE_PHASES
#define CTVAE_PHASE_DECL(NAME)                                                                                          \
  __device__ long long g_phase_##NAME[2][4][8][16];                                                                      \
  extern "C" int ctvae_debug_phases_##NAME(long long* out) {                                                            \
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_##NAME), sizeof(long long) * 2 * 4 * 8 * 16);                   \
  }
// the sampled workgroup: blockIdx.y == gridDim.y / 2, blockIdx.x == 0
#define CTVAE_PH(NAME, K_, I_)                                                                                          \
  do {                                                                                                                  \
    if ((threadIdx.x & 63) == 0 && blockIdx.y == gridDim.y / 2 && blockIdx.x == 0 && (threadIdx.x >> 6) < 8)            \
      g_phase_##NAME[0][K_][threadIdx.x >> 6][I_] = clock64(), g_phase_##NAME[1][K_][threadIdx.x >> 6][I_] = wall_clock64();      \
  } while (0)
#else
#define CTVAE_PHASE_DECL(NAME)
#define CTVAE_PH(NAME, K_, I_)
#endif
