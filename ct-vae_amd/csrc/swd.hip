// Sliced Wasserstein distance of SWAE (models/swae.py:150-178): z and prior draws [N][D] are projected on S random unit
// directions, each projection's two sample sets are SORTED over the batch and compared rank by rank:
//     swd = weight * mean_{s, r} (sort_r(z . w_s) - sort_r(prior . w_s))^p
// (torch.sort over a [S, N] matrix, a pow and a mean in the reference).  One workgroup per direction: projections of all N
// samples (dot products over D), bitonic sort of both sets in LDS -- the z side carries its sample index, so the gradient of
// the rank-wise term goes straight back to the sample it came from -- and the direction's partial sum.  A second launch sums
// the partials in a fixed order and forms d swd / d z = (d swd / d projections) . W.  Deterministic, no atomics.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

namespace {

constexpr int kSwdMaxN = 1024;

// base^p as torch.pow(float tensor, python float) gives it for the exponents SWAE uses: p == 2 exactly squares; otherwise
// powf (negative base with a non-integer exponent is NaN there as well)
__device__ __forceinline__ float pow_p(float d, float p) { return p == 2.f ? d * d : powf(d, p); }
__device__ __forceinline__ float dpow_p(float d, float p) { return p == 2.f ? 2.f * d : p * powf(d, p - 1.f); }

__global__ __launch_bounds__(256) void swd_sort_kernel(const float* __restrict__ z, const float* __restrict__ prior,
                                                      const float* __restrict__ proj, int N, int D, int S, int P, float p,
                                                      float* __restrict__ part, float* __restrict__ dA) {
  __shared__ float ka[kSwdMaxN], kc[kSwdMaxN];
  __shared__ int ia[kSwdMaxN];
  __shared__ float sw[512];
  __shared__ float red[4];
  const int tid = threadIdx.x, s = blockIdx.x;
  for (int d = tid; d < D; d += 256) sw[d] = proj[(long)s * D + d];
  __syncthreads();
  for (int b = tid; b < P; b += 256) {
    float a = INFINITY, c = INFINITY;            // padding sorts to the end on both sides
    if (b < N) {
      a = 0.f; c = 0.f;
      const float* zr = z + (long)b * D;
      const float* pr = prior + (long)b * D;
      for (int d = 0; d < D; d += 4) {
        const f32x4 zv = *reinterpret_cast<const f32x4*>(zr + d), pv = *reinterpret_cast<const f32x4*>(pr + d);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(sw + d);
#pragma unroll
        for (int k = 0; k < 4; ++k) { a += zv[k] * wv[k]; c += pv[k] * wv[k]; }
      }
    }
    ka[b] = a; kc[b] = c; ia[b] = b;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const float x0 = ka[i], x1 = ka[l];
          // equal keys keep their places: the order among ties does not change the rank-wise differences
          if ((x0 > x1) == up && x0 != x1) { ka[i] = x1; ka[l] = x0; const int t = ia[i]; ia[i] = ia[l]; ia[l] = t; }
          const float y0 = kc[i], y1 = kc[l];
          if ((y0 > y1) == up && y0 != y1) { kc[i] = y1; kc[l] = y0; }
        }
      }
      __syncthreads();
    }
  }
  float acc = 0.f;
  for (int r = tid; r < N; r += 256) {
    const float d = ka[r] - kc[r];
    acc += pow_p(d, p);
    dA[(long)ia[r] * S + s] = dpow_p(d, p);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) part[s] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = coef * sum_s part[s];  grad_z[b][d] = coef * sum_s dA[b][s] * proj[s][d]
__global__ __launch_bounds__(256) void swd_finish_kernel(const float* __restrict__ part, const float* __restrict__ dA,
                                                        const float* __restrict__ proj, int N, int D, int S, float coef,
                                                        float* __restrict__ out, float* __restrict__ grad_z) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < (long)N * D) {
    const int b = (int)(e / D), d = (int)(e - (long)b * D);
    float g = 0.f;
    for (int s = 0; s < S; ++s) g += dA[(long)b * S + s] * proj[(long)s * D + d];
    grad_z[e] = coef * g;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double t = 0.0;
    for (int s = 0; s < S; ++s) t += (double)part[s];
    out[0] = (float)(t * (double)coef);
  }
}

}  // namespace

int launch_swd_forward(const float* z, const float* prior, const float* proj, int N, int D, int S, float p, float weight, float* out,
                       float* grad_z, float* ws, size_t ws_bytes, hipStream_t st) {
  if (!z || !prior || !proj || !out || !grad_z || !ws || N < 1 || N > kSwdMaxN || D < 4 || D > 512 || D % 4 || S < 1) return kErrBadArg;
  if (ws_bytes / sizeof(float) < (size_t)S + (size_t)N * S) return kErrWorkspace;
  int P = 2;
  while (P < N) P <<= 1;
  float* part = ws;
  float* dA = ws + S;
  {
    ProfScope ps("swd_sort_kernel", st, 4.0 * N * (double)D * S, 4.0 * S * (2.0 * N * D + N));
    hipLaunchKernelGGL(swd_sort_kernel, dim3(S), dim3(256), 0, st, z, prior, proj, N, D, S, P, p, part, dA);
    CTVAE_LAUNCH_CHECK();
  }
  ProfScope ps("swd_finish_kernel", st, 2.0 * N * (double)D * S, 4.0 * N * (S + 2.0 * D));
  hipLaunchKernelGGL(swd_finish_kernel, dim3((unsigned)(((long)N * D + 255) / 256)), dim3(256), 0, st, part, dA, proj, N, D, S,
                     weight / ((float)S * (float)N), out, grad_z);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
