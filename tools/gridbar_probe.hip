// What does a grid-wide barrier inside ONE launch cost on MI355X, next to a kernel boundary inside a replayed hipGraph?
//   hipcc --offload-arch=gfx950 -O3 -o tools/gridbar_probe tools/gridbar_probe.hip && tools/gridbar_probe
// Decides how the launch-floor-bound small layers of a bs=64 VanillaVAE step are fused.
//   (1) chain of K dependent small kernels replayed as a hipGraph        -> us per kernel boundary
//   (2) one persistent kernel, G workgroups, K barriers: release fetch_add + acquire spin on one counter (agent scope)
//   (3) same with a payload: every workgroup writes 4 KB, barrier, reads the 4 KB of workgroup (id + G/2 + 1) % G -- i.e. of
//       another XCD -- and checks it (the barrier must make plain stores visible across the 8 L2s)
//   (4) two-level: one counter per XCD-group of workgroups (id % 8), the last arrival of a group bumps the global counter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void tiny_kernel(float* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}

__device__ __forceinline__ void grid_barrier(unsigned* cnt, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;   // every wave leaves the loop: a barrier that is not met within ~1 s gives up (the run is then meaningless, not hung)
    while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void bar_kernel(unsigned* cnt, int K, unsigned base) {
  const unsigned G = gridDim.x;
  for (int k = 0; k < K; ++k) grid_barrier(cnt, base + (unsigned)(k + 1) * G);
}

// two-level: 8 group counters (workgroup id % 8 = XCD under round-robin dispatch), then the global one
__global__ __launch_bounds__(256) void bar2_kernel(unsigned* cnt, int K, unsigned base8, unsigned baseg) {
  const unsigned G = gridDim.x, grp = blockIdx.x & 7u, per = (G + 7u - grp) / 8u;
  unsigned* gc = cnt + 32 * (1 + grp);   // own 128-byte line each
  for (int k = 0; k < K; ++k) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == base8 + (unsigned)(k + 1) * per)   // last of its group this round
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < baseg + (unsigned)(k + 1) * 8u && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void bar_payload_kernel(unsigned* cnt, float* buf, int K, unsigned base, int* errs) {
  const unsigned G = gridDim.x, me = blockIdx.x, other = (me + G / 2 + 1) % G;
  int bad = 0;
  for (int k = 0; k < K; ++k) {
    float4* mine = reinterpret_cast<float4*>(buf + ((size_t)(k & 1) * G + me) * 1024);
    const float v = (float)(k * 4096 + (int)me);
    mine[threadIdx.x] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
    grid_barrier(cnt, base + (unsigned)(k + 1) * G);
    const float4* theirs = reinterpret_cast<const float4*>(buf + ((size_t)(k & 1) * G + other) * 1024);
    const float4 t = theirs[threadIdx.x];
    const float w = (float)(k * 4096 + (int)other);
    if (t.x != w || t.w != w + 3.f) ++bad;
  }
  if (bad) atomicAdd(errs, bad);
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float* p;
  CK(hipMalloc(&p, 1 << 24));
  CK(hipMemset(p, 0, 1 << 24));
  unsigned* cnt;
  CK(hipMalloc(&cnt, 4096));
  int* errs;
  CK(hipMalloc(&errs, 4));
  CK(hipMemset(errs, 0, 4));
  float* buf;
  CK(hipMalloc(&buf, (size_t)2 * 2048 * 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms;

  // (1) graph of K dependent tiny kernels
  for (int n : {256 * 64, 256 * 1024}) {
    const int K = 200;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < K; ++k) hipLaunchKernelGGL(tiny_kernel, dim3(n / 256), dim3(256), 0, st, p, n);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph chain: %d tiny kernels of %d threads: %.2f us per kernel\n", K, n, ms * 1e3 / (5 * K));
  }

  // (2)-(4) persistent kernels
  for (int G : {256, 512, 1024}) {
    const int K = 200;
    for (int variant = 0; variant < 3; ++variant) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemsetAsync(cnt, 0, 4096, st));
        CK(hipEventRecord(e0, st));
        if (variant == 0) hipLaunchKernelGGL(bar_kernel, dim3(G), dim3(256), 0, st, cnt, K, 0u);
        else if (variant == 1) hipLaunchKernelGGL(bar_payload_kernel, dim3(G), dim3(256), 0, st, cnt, buf, K, 0u, errs);
        else hipLaunchKernelGGL(bar2_kernel, dim3(G), dim3(256), 0, st, cnt, K, 0u, 0u);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      int h = 0;
      CK(hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost));
      printf("persistent G=%4d %-22s: %.2f us per barrier (%d barriers, %.1f us total)%s\n", G,
             variant == 0 ? "one counter" : variant == 1 ? "one counter + 4KB/wg" : "two-level (id%8)", best * 1e3 / K, K, best * 1e3,
             variant == 1 ? (h ? "  PAYLOAD MISMATCH" : "  payload ok") : "");
    }
  }
  return 0;
}
