// Importance-weighted objective of IWAE / MIWAE (models/iwae.py:126-155, models/miwae.py:130-163).  The decoder runs on
// R = B * M * S latent samples; row r of `recons` is compared with image r / rep (rep = M * S) of the batch:
//   lp[r]  = mean_n (recons[r] - x[r / rep])^2                     ("log_p_x_z", [B x (M x) S])
//   kld[r] = -0.5 * sum_d (1 + lv[r] - mu[r]^2 - exp(lv[r]))
//   lw     = lp + M_N * kld;  w = softmax(lw) over the S samples of a group;  loss = mean over the R / S groups of sum_s w lw
// The gradient flows through BOTH the weights and the log-weights (the reference does not detach them):
//   d loss / d lw[r] = w[r] * (1 + lw[r] - sum_t w[t] lw[t]) / groups =: coef[r]
// Kernels: one workgroup per row for the reductions (coalesced 16-byte reads of the row and of its image, wavefront
// shuffle + LDS merge), one workgroup for the S-way softmax of all groups and the three scalars, one workgroup per row
// for the backward pass.  HBM-bound: forward 8 B per element of recons (the image re-read hits L2 for rep > 1),
// backward 12 B per element.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

__global__ __launch_bounds__(256) void iw_rows_kernel(const float* __restrict__ recons, const float* __restrict__ x, long n,
                                                      int rep, const float* __restrict__ mu, const float* __restrict__ lv, int L,
                                                      float* __restrict__ lp, float* __restrict__ kld) {
  __shared__ float sm[4];
  const long r = blockIdx.x;
  const f32x4* a = reinterpret_cast<const f32x4*>(recons + r * n);
  const f32x4* b = reinterpret_cast<const f32x4*>(x + (r / rep) * n);
  float s = 0.f;
  const long n4 = n >> 2;
  for (long i = threadIdx.x; i < n4; i += 256) {
    const f32x4 u = a[i], v = b[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = u[k] - v[k];
      s += d * d;
    }
  }
  for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) {
    const float d = recons[r * n + i] - x[(r / rep) * n + i];
    s += d * d;
  }
  float k = 0.f;
  for (int d = threadIdx.x; d < L; d += 256) {
    const float m = mu[r * L + d], l = lv[r * L + d];
    k += 1.f + l - m * m - expf(l);
  }
  s = block_sum_256(s, sm);
  k = block_sum_256(k, sm);
  if (threadIdx.x == 0) {
    lp[r] = s / (float)n;
    kld[r] = -0.5f * k;
  }
}

// out[0] = loss, out[1] = mean lp ('Reconstruction_Loss'), out[2] = mean kld, out[3] = -mean kld ('KLD')
__global__ __launch_bounds__(256) void iw_finish_kernel(const float* __restrict__ lp, const float* __restrict__ kld, int R, int S,
                                                        float M_N, float* __restrict__ coef, float* __restrict__ out) {
  __shared__ double smd[3][4];
  const int groups = R / S;
  double loss = 0.0, sl = 0.0, sk = 0.0;
  for (int g = threadIdx.x; g < groups; g += 256) {
    float m = -INFINITY;
    for (int s = 0; s < S; ++s) m = fmaxf(m, lp[g * S + s] + M_N * kld[g * S + s]);
    float den = 0.f, e = 0.f;
    for (int s = 0; s < S; ++s) {
      const float lw = lp[g * S + s] + M_N * kld[g * S + s];
      const float w = expf(lw - m);
      den += w;
      e += w * lw;
    }
    e /= den;                                   // sum_s w_s lw_s
    for (int s = 0; s < S; ++s) {
      const float lw = lp[g * S + s] + M_N * kld[g * S + s];
      coef[g * S + s] = expf(lw - m) / den * (1.f + lw - e) / (float)groups;
      sl += (double)lp[g * S + s];
      sk += (double)kld[g * S + s];
    }
    loss += (double)e;
  }
  loss = wave_sum_d(loss);
  sl = wave_sum_d(sl);
  sk = wave_sum_d(sk);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { smd[0][w] = loss; smd[1][w] = sl; smd[2][w] = sk; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = (float)((smd[0][0] + smd[0][1] + smd[0][2] + smd[0][3]) / (double)groups);
    out[1] = (float)((smd[1][0] + smd[1][1] + smd[1][2] + smd[1][3]) / (double)R);
    const float k = (float)((smd[2][0] + smd[2][1] + smd[2][2] + smd[2][3]) / (double)R);
    out[2] = k;
    out[3] = -k;
  }
}

// g_recons[r] = go * coef[r] * 2 (recons[r] - x[r/rep]) / n;  g_mu[r] = go * coef[r] * M_N * mu[r];
// g_lv[r] = go * coef[r] * M_N * 0.5 (exp(lv[r]) - 1)
__global__ __launch_bounds__(256) void iw_rows_bwd_kernel(const float* __restrict__ recons, const float* __restrict__ x, long n,
                                                          int rep, const float* __restrict__ mu, const float* __restrict__ lv, int L,
                                                          float M_N, const float* __restrict__ coef, const float* __restrict__ go,
                                                          float* __restrict__ g_recons, float* __restrict__ g_mu,
                                                          float* __restrict__ g_lv) {
  const long r = blockIdx.x;
  const float c = go[0] * coef[r];
  if (g_recons != nullptr) {
    const float sc = c * 2.f / (float)n;
    const f32x4* a = reinterpret_cast<const f32x4*>(recons + r * n);
    const f32x4* b = reinterpret_cast<const f32x4*>(x + (r / rep) * n);
    f32x4* o = reinterpret_cast<f32x4*>(g_recons + r * n);
    const long n4 = n >> 2;
    for (long i = threadIdx.x; i < n4; i += 256) {
      const f32x4 u = a[i], v = b[i];
      f32x4 w;
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = sc * (u[k] - v[k]);
      o[i] = w;
    }
    for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) g_recons[r * n + i] = sc * (recons[r * n + i] - x[(r / rep) * n + i]);
  }
  if (g_mu != nullptr) {
    const float ck = c * M_N;
    for (int d = threadIdx.x; d < L; d += 256) {
      g_mu[r * L + d] = ck * mu[r * L + d];
      g_lv[r * L + d] = ck * 0.5f * (expf(lv[r * L + d]) - 1.f);
    }
  }
}

int launch_iw_loss_forward(const float* recons, const float* x, long n, int R, int rep, const float* mu, const float* lv, int L,
                           int S, float M_N, float* lp, float* kld, float* coef, float* out4, hipStream_t st) {
  {
    ProfScope ps("iw_rows_kernel", st, 0.0, 8.0 * (double)R * n);
    hipLaunchKernelGGL(iw_rows_kernel, dim3((unsigned)R), dim3(256), 0, st, recons, x, n, rep, mu, lv, L, lp, kld);
  }
  CTVAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(iw_finish_kernel, dim3(1), dim3(256), 0, st, lp, kld, R, S, M_N, coef, out4);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_iw_loss_backward(const float* recons, const float* x, long n, int R, int rep, const float* mu, const float* lv, int L,
                            float M_N, const float* coef, const float* go, float* g_recons, float* g_mu, float* g_lv,
                            hipStream_t st) {
  ProfScope ps("iw_rows_bwd_kernel", st, 0.0, 12.0 * (double)R * n);
  hipLaunchKernelGGL(iw_rows_bwd_kernel, dim3((unsigned)R), dim3(256), 0, st, recons, x, n, rep, mu, lv, L, M_N, coef, go,
                     g_recons, g_mu, g_lv);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
