// Diagnostic: f32 MFMA issue rate on gfx950 (and the clock it holds) alone and next to VALU / LDS work.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/mfma_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode 0: MFMA only, NACC independent accumulators.  mode 1: + 16 VALU fma per 2 MFMAs in the same wave.
// mode 2: odd waves do VALU only, even waves MFMA only.  mode 3: MFMA + LDS reads.
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* clk, int iters) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) lds[i] = i * 0.001f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int k = 0; k < NACC; ++k)
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  float x = tid * 0.5f, y = 1.0001f, v0 = 0, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[k], 0, 0, 0);
    if constexpr (MODE == 1) {
      v0 = v0 * y + x; v1 = v1 * y + x; v2 = v2 * y + x; v3 = v3 * y + x;
      v4 = v4 * y + x; v5 = v5 * y + x; v6 = v6 * y + x; v7 = v7 * y + x;
      v0 = v0 * y + x; v1 = v1 * y + x; v2 = v2 * y + x; v3 = v3 * y + x;
      v4 = v4 * y + x; v5 = v5 * y + x; v6 = v6 * y + x; v7 = v7 * y + x;
    }
    if constexpr (MODE == 3) {
      v0 += lds[(tid + it) & 4095]; v1 += lds[(tid * 4 + it) & 4095];
      v2 += lds[(tid + 2 * it) & 4095]; v3 += lds[(tid + 3 * it) & 4095];
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int k = 0; k < NACC; ++k)
    for (int i = 0; i < 16; ++i) s += acc[k][i];
  out[blockIdx.x * 256 + tid] = s;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int NACC, int MODE>
void run(float* out, unsigned long long* clk, int wgs_per_cu, int iters) {
  const int mode = MODE;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL((probe<NACC, MODE>), dim3(grid), dim3(256), 0, 0, out, clk, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NACC, MODE>), dim3(grid), dim3(256), 0, 0, out, clk, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / (double)h[1] * 0.1;
  double nm = (double)grid * 4 * iters * NACC;
  double tf = nm * 32 * 32 * 2 * 2 / (ms * 1e-3) / 1e12;
  printf("mode %d acc %d waves/SIMD %d  %.2f ms  %.1f TF/s  clock %.2f GHz  cycles/MFMA/SIMD %.1f\n", mode, NACC, wgs_per_cu, ms, tf,
         ghz, ms * 1e-3 * ghz * 1e9 / (nm / 1024.0));
}

int main() {
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, 256 * 4096 * sizeof(float));
  hipMalloc(&clk, 16);
  const int iters = 20000;
  run<1, 0>(out, clk, 1, iters);
  run<2, 0>(out, clk, 1, iters);
  run<4, 0>(out, clk, 1, iters);
  run<4, 0>(out, clk, 2, iters);
  run<4, 0>(out, clk, 4, iters);
  run<4, 1>(out, clk, 1, iters);
  run<4, 1>(out, clk, 2, iters);
  run<4, 3>(out, clk, 1, iters);
  run<4, 3>(out, clk, 2, iters);
  return 0;
}
