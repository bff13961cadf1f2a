// How does the gfx950 workgroup dispatcher place a grid that is smaller than the chip's resident capacity?
// Every workgroup (256 threads, optional dynamic LDS to cap residency) runs a fixed MFMA loop and records the
// (XCC, SE, CU) it ran on plus start/end timestamps.  Output per grid size: distinct CUs used, max workgroups
// that shared one CU, kernel time.
//   hipcc -O3 --offload-arch=gfx950 tools/dispatch_probe.hip -o /tmp/dispatch_probe && /tmp/dispatch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <map>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void probe(float* out, unsigned* where, unsigned long long* t01, int iters) {
  extern __shared__ float dyn[];
  f32x16 acc = {0};
  const float a = threadIdx.x * 1e-3f, b = 1.0f;
  unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // XCC_ID
    where[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff);
    t01[2 * blockIdx.x] = t0;
    t01[2 * blockIdx.x + 1] = t1;
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 12345.678f) out[0] = s + dyn[0];
}

int main() {
  float* out;
  unsigned* where;
  unsigned long long* t01;
  const int maxg = 4096;
  hipMalloc(&out, 4);
  hipMalloc(&where, maxg * 4);
  hipMalloc(&t01, maxg * 16);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int iters = 40;  // 640 MFMAs x 64 cycles = 41k cycles ~ 17 us per wave when alone on its SIMD
  for (int lds_kb : {0, 64}) {
    for (int grid : {128, 256, 512, 768, 1024, 2048}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds_kb * 1024, 0, out, where, t01, iters);
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds_kb * 1024, 0, out, where, t01, iters);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned> w(grid);
      std::vector<unsigned long long> t(2 * grid);
      hipMemcpy(w.data(), where, grid * 4, hipMemcpyDeviceToHost);
      hipMemcpy(t.data(), t01, grid * 16, hipMemcpyDeviceToHost);
      std::map<unsigned, int> per_cu, per_xcc;
      unsigned long long tmin = ~0ull, tmax = 0, dsum = 0;
      for (int i = 0; i < grid; ++i) {
        const unsigned xcc = w[i] >> 16, hw = w[i] & 0xffff;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
        per_xcc[xcc]++;
        if (t[2 * i] < tmin) tmin = t[2 * i];
        if (t[2 * i + 1] > tmax) tmax = t[2 * i + 1];
        dsum += t[2 * i + 1] - t[2 * i];
      }
      int mx = 0, mn = 1 << 30;
      for (auto& kv : per_cu) {
        if (kv.second > mx) mx = kv.second;
        if (kv.second < mn) mn = kv.second;
      }
      printf("lds %3d KB grid %5d: %.1f us  distinct CUs %3zu  WGs/CU min %d max %d  XCCs %zu  avg WG dur %.1f us (100MHz ticks)  span %.1f us\n",
             lds_kb, grid, ms * 1e3, per_cu.size(), mn, mx, per_xcc.size(), dsum / (double)grid / 100.0, (tmax - tmin) / 100.0);
    }
  }
  return 0;
}
