// Categorical latent of CategoricalVAE (models/cat_vae.py): the Gumbel-softmax reparameterisation (:118-132) and the KL
// term between softmax(q) and the uniform categorical prior (:147-168), forward and backward.  Rows are the (sample,
// latent) pairs, Q the number of categories (40 in configs/cat_vae.yaml).  A group of G = min(64, pow2ceil(Q)) lanes owns
// one row at a time (consecutive lanes read consecutive categories: coalesced), row-wise max / sum are xor-shuffle
// reductions inside the group, nothing goes through LDS except the per-workgroup partial of the KL sum.
// HBM-bound: forward 12 B per element (logits, uniform draw, sample), backward 12 B, KL 4 B forward / 8 B backward.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

constexpr int kCatMaxPerLane = 4;     // Q <= 64 * 4
constexpr int kCatBlocks = 1024;

__device__ __forceinline__ float group_max(float v, int G) {
  for (int o = G >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float group_sum(float v, int G) {
  for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// softmax of the values t[k] (element q = gl + k*G of the row; elements beyond Q hold -inf) -> t[k] = probabilities
template <int NPL>
__device__ __forceinline__ void group_softmax(float (&t)[NPL], int G) {
  float m = t[0];
#pragma unroll
  for (int k = 1; k < NPL; ++k) m = fmaxf(m, t[k]);
  m = group_max(m, G);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    t[k] = expf(t[k] - m);          // exp(-inf) = 0 for the padding
    s += t[k];
  }
  s = group_sum(s, G);
  const float inv = 1.f / s;
#pragma unroll
  for (int k = 0; k < NPL; ++k) t[k] *= inv;
}

// s = softmax((z + g) / temp),  g = -log(-log(u + eps) + eps)           (cat_vae.py:125-130)
template <int NPL>
__global__ __launch_bounds__(256) void gumbel_softmax_fwd_kernel(const float* __restrict__ z, const float* __restrict__ u,
                                                                 float* __restrict__ s, long rows, int Q, int G,
                                                                 float inv_temp, float eps) {
  const int gl = threadIdx.x & (G - 1);
  const long gpb = 256 / G;
  for (long row = (long)blockIdx.x * gpb + threadIdx.x / G; row < rows; row += (long)gridDim.x * gpb) {
    float t[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int q = gl + k * G;
      t[k] = -INFINITY;
      if (q < Q) {
        const float g = -logf(-logf(u[row * Q + q] + eps) + eps);
        t[k] = (z[row * Q + q] + g) * inv_temp;
      }
    }
    group_softmax<NPL>(t, G);
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int q = gl + k * G;
      if (q < Q) s[row * Q + q] = t[k];
    }
  }
}

// g_z = s * (g_s - sum_q g_s s) / temp
template <int NPL>
__global__ __launch_bounds__(256) void gumbel_softmax_bwd_kernel(const float* __restrict__ gs, const float* __restrict__ s,
                                                                 float* __restrict__ gz, long rows, int Q, int G, float inv_temp) {
  const int gl = threadIdx.x & (G - 1);
  const long gpb = 256 / G;
  for (long row = (long)blockIdx.x * gpb + threadIdx.x / G; row < rows; row += (long)gridDim.x * gpb) {
    float sv[NPL], gv[NPL];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int q = gl + k * G;
      sv[k] = q < Q ? s[row * Q + q] : 0.f;
      gv[k] = q < Q ? gs[row * Q + q] : 0.f;
      dot += sv[k] * gv[k];
    }
    dot = group_sum(dot, G);
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int q = gl + k * G;
      if (q < Q) gz[row * Q + q] = sv[k] * (gv[k] - dot) * inv_temp;
    }
  }
}

// part[block] = sum over the block's rows of sum_q p (log(p + eps) - c),  p = softmax(q), c = log(1/Q + eps)   (cat_vae.py:147,160-167)
template <int NPL>
__global__ __launch_bounds__(256) void cat_kl_partial_kernel(const float* __restrict__ logits, float* __restrict__ part,
                                                             long rows, int Q, int G, float eps, float c) {
  __shared__ float sm[4];
  const int gl = threadIdx.x & (G - 1);
  const long gpb = 256 / G;
  float acc = 0.f;
  for (long row = (long)blockIdx.x * gpb + threadIdx.x / G; row < rows; row += (long)gridDim.x * gpb) {
    float t[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int q = gl + k * G;
      t[k] = q < Q ? logits[row * Q + q] : -INFINITY;
    }
    group_softmax<NPL>(t, G);
#pragma unroll
    for (int k = 0; k < NPL; ++k)
      if (gl + k * G < Q) acc += t[k] * logf(t[k] + eps) - t[k] * c;
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// out[0] = kld = (sum of the partials) / B
__global__ __launch_bounds__(256) void cat_kl_finish_kernel(const float* __restrict__ part, int nparts, int B, float* __restrict__ out) {
  __shared__ double smd[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += (double)part[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) smd[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((smd[0] + smd[1] + smd[2] + smd[3]) / (double)B);
}

// g_q = go/B * p (t - sum_q p t),  t = d/dp [p log(p+eps) - c p] = log(p+eps) + p/(p+eps) - c
template <int NPL>
__global__ __launch_bounds__(256) void cat_kl_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ go,
                                                         float* __restrict__ gq, long rows, int Q, int G, int B, float eps, float c) {
  const int gl = threadIdx.x & (G - 1);
  const long gpb = 256 / G;
  const float sc = go[0] / (float)B;
  for (long row = (long)blockIdx.x * gpb + threadIdx.x / G; row < rows; row += (long)gridDim.x * gpb) {
    float p[NPL], t[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int q = gl + k * G;
      p[k] = q < Q ? logits[row * Q + q] : -INFINITY;
    }
    group_softmax<NPL>(p, G);
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      t[k] = logf(p[k] + eps) + p[k] / (p[k] + eps) - c;
      dot += p[k] * t[k];           // padding: p = 0 -> contributes 0 (t is finite: log(eps))
    }
    dot = group_sum(dot, G);
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int q = gl + k * G;
      if (q < Q) gq[row * Q + q] = sc * p[k] * (t[k] - dot);
    }
  }
}

static inline int group_of(int Q) {
  int G = 1;
  while (G < Q && G < 64) G <<= 1;
  return G;
}
static inline unsigned cat_grid(long rows, int G, int cap) {
  long b = (rows + 256 / G - 1) / (256 / G);
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

size_t cat_kl_workspace_floats() { return kCatBlocks; }

#define CAT_DISPATCH(KERNEL, GRID, ...)                                                                         \
  do {                                                                                                          \
    if (Q <= G) hipLaunchKernelGGL(KERNEL<1>, dim3(GRID), dim3(256), 0, st, __VA_ARGS__);                       \
    else if (Q <= 2 * G) hipLaunchKernelGGL(KERNEL<2>, dim3(GRID), dim3(256), 0, st, __VA_ARGS__);              \
    else hipLaunchKernelGGL(KERNEL<kCatMaxPerLane>, dim3(GRID), dim3(256), 0, st, __VA_ARGS__);                 \
  } while (0)

int launch_gumbel_softmax_fwd(const float* z, const float* u, float* s, long rows, int Q, float temp, float eps, hipStream_t st) {
  if (Q > 64 * kCatMaxPerLane) return kErrBadArg;
  const int G = group_of(Q);
  ProfScope ps("gumbel_softmax_fwd_kernel", st, 0.0, 12.0 * (double)rows * Q);
  CAT_DISPATCH(gumbel_softmax_fwd_kernel, cat_grid(rows, G, 8192), z, u, s, rows, Q, G, 1.f / temp, eps);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_gumbel_softmax_bwd(const float* gs, const float* s, float* gz, long rows, int Q, float temp, hipStream_t st) {
  if (Q > 64 * kCatMaxPerLane) return kErrBadArg;
  const int G = group_of(Q);
  ProfScope ps("gumbel_softmax_bwd_kernel", st, 0.0, 12.0 * (double)rows * Q);
  CAT_DISPATCH(gumbel_softmax_bwd_kernel, cat_grid(rows, G, 8192), gs, s, gz, rows, Q, G, 1.f / temp);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_cat_kl_fwd(const float* logits, long rows, int Q, int B, float eps, float c, float* out, float* ws, size_t ws_bytes,
                      hipStream_t st) {
  if (Q > 64 * kCatMaxPerLane) return kErrBadArg;
  if (ws_bytes / sizeof(float) < cat_kl_workspace_floats()) return kErrWorkspace;
  const int G = group_of(Q);
  const unsigned blocks = cat_grid(rows, G, kCatBlocks);
  {
    ProfScope ps("cat_kl_partial_kernel", st, 0.0, 4.0 * (double)rows * Q);
    CAT_DISPATCH(cat_kl_partial_kernel, blocks, logits, ws, rows, Q, G, eps, c);
  }
  CTVAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(cat_kl_finish_kernel, dim3(1), dim3(256), 0, st, ws, (int)blocks, B, out);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_cat_kl_bwd(const float* logits, const float* go, float* gq, long rows, int Q, int B, float eps, float c, hipStream_t st) {
  if (Q > 64 * kCatMaxPerLane) return kErrBadArg;
  const int G = group_of(Q);
  ProfScope ps("cat_kl_bwd_kernel", st, 0.0, 8.0 * (double)rows * Q);
  CAT_DISPATCH(cat_kl_bwd_kernel, cat_grid(rows, G, 8192), logits, go, gq, rows, Q, G, B, eps, c);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
