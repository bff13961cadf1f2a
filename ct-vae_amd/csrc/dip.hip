// DIP-VAE II regulariser (models/dip_vae.py:147-159), exactly as the reference writes it -- including its two quirks: mu is
// centred over the LATENT dimension (mean(dim=1)), and the "variance" added for DIP-II is ONE scalar, the mean of the main
// diagonal of the [B][D] matrix exp(2 log_var) (torch.diagonal(..., dim1=0) of a 2-D tensor), broadcast onto every entry
// of the covariance:
//   c = mu - rowmean(mu);  cov = c^T c  [D][D];  v = mean_{i < min(B,D)} exp(2 lv[i][i]);  cz = cov + v
//   dip = l_off * sum_{i != j} cz_ij^2 + l_diag * sum_i (cz_ii - 1)^2
// Backward with G = d dip / d cz (symmetric):  d/dc = 2 c G,  d/dmu = dc - rowmean(dc),  d/dv = sum G,
//   d/dlv[i][i] = dv * 2 exp(2 lv[i][i]) / min(B,D).
// Sizes are latent-sized (B x 128, 128 x 128): three small launches forward (rows, covariance rows, finish), one backward.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

// c[b][:] = mu[b][:] - mean_d mu[b][:];  e[b] = exp(2 lv[b][b]) for b < min(B, D)
__global__ __launch_bounds__(256) void dip_rows_kernel(const float* __restrict__ mu, long mu_rs, const float* __restrict__ lv,
                                                       long lv_rs, float* __restrict__ c, float* __restrict__ e, int B, int D) {
  __shared__ float sm[4];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) s += mu[b * mu_rs + d];
  s = block_sum_256(s, sm);
  const float mean = s / (float)D;
  for (int d = threadIdx.x; d < D; d += 256) c[(long)b * D + d] = mu[b * mu_rs + d] - mean;
  if (threadIdx.x == 0 && b < D) e[b] = expf(2.f * lv[b * lv_rs + b]);
}

// row i of cz and of G; part[i] = row's share of dip, part[D + i] = sum_j G[i][j]
__global__ __launch_bounds__(256) void dip_cov_kernel(const float* __restrict__ c, const float* __restrict__ e, float* __restrict__ G,
                                                      float* __restrict__ part, int B, int D, float l_diag, float l_off) {
  __shared__ float sm[4];
  const int i = blockIdx.x, m = B < D ? B : D;
  float v = 0.f;
  for (int k = threadIdx.x; k < m; k += 256) v += e[k];
  v = block_sum_256(v, sm) / (float)m;
  float loss = 0.f, gsum = 0.f;
  for (int j = threadIdx.x; j < D; j += 256) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += c[(long)b * D + i] * c[(long)b * D + j];
    const float cz = acc + v;
    float g;
    if (j == i) {
      loss += l_diag * (cz - 1.f) * (cz - 1.f);
      g = 2.f * l_diag * (cz - 1.f);
    } else {
      loss += l_off * cz * cz;
      g = 2.f * l_off * cz;
    }
    G[(long)i * D + j] = g;
    gsum += g;
  }
  loss = block_sum_256(loss, sm);
  gsum = block_sum_256(gsum, sm);
  if (threadIdx.x == 0) {
    part[i] = loss;
    part[D + i] = gsum;
  }
}

// out[0] = dip, out[1] = d dip / d v
__global__ __launch_bounds__(256) void dip_finish_kernel(const float* __restrict__ part, int D, float* __restrict__ out) {
  __shared__ double smd[2][4];
  double a = 0.0, g = 0.0;
  for (int i = threadIdx.x; i < D; i += 256) {
    a += (double)part[i];
    g += (double)part[D + i];
  }
  a = wave_sum_d(a);
  g = wave_sum_d(g);
  if ((threadIdx.x & 63) == 0) { smd[0][threadIdx.x >> 6] = a; smd[1][threadIdx.x >> 6] = g; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = (float)(smd[0][0] + smd[0][1] + smd[0][2] + smd[0][3]);
    out[1] = (float)(smd[1][0] + smd[1][1] + smd[1][2] + smd[1][3]);
  }
}

// g_mu[b][:] = go * (dc - mean(dc)), dc[b][j] = 2 sum_i c[b][i] G[i][j];  g_lv[b][j] = (j == b < min(B,D)) go * dv * 2 e[b] / min(B,D)
__global__ __launch_bounds__(256) void dip_bwd_kernel(const float* __restrict__ c, const float* __restrict__ e, const float* __restrict__ G,
                                                      const float* __restrict__ out, const float* __restrict__ go,
                                                      float* __restrict__ g_mu, float* __restrict__ g_lv, int B, int D) {
  __shared__ float sm[4];
  const int b = blockIdx.x, m = B < D ? B : D;
  const float g0 = go[0];
  float dc[4] = {0.f, 0.f, 0.f, 0.f};   // D <= 1024
  float s = 0.f;
  for (int q = 0; q < 4; ++q) {
    const int j = threadIdx.x + 256 * q;
    if (j < D) {
      float acc = 0.f;
      for (int i = 0; i < D; ++i) acc += c[(long)b * D + i] * G[(long)i * D + j];
      dc[q] = 2.f * acc;
      s += dc[q];
    }
  }
  s = block_sum_256(s, sm);
  const float mean = s / (float)D;
  for (int q = 0; q < 4; ++q) {
    const int j = threadIdx.x + 256 * q;
    if (j < D) {
      g_mu[(long)b * D + j] = g0 * (dc[q] - mean);
      g_lv[(long)b * D + j] = (j == b && b < m) ? g0 * out[1] * 2.f * e[b] / (float)m : 0.f;
    }
  }
}

size_t dip_state_floats(int B, int D) { return (size_t)B * D + (size_t)D * D + 3 * (size_t)D + 2; }

// state layout: c[B*D] | G[D*D] | e[D] | part[2D] | out[2]
int launch_dip_forward(const float* mu, long mu_rs, const float* lv, long lv_rs, int B, int D, float l_diag, float l_off,
                       float* state, hipStream_t st) {
  if (D > 1024) return kErrBadArg;
  float* c = state;
  float* G = c + (size_t)B * D;
  float* e = G + (size_t)D * D;
  float* part = e + D;
  float* out = part + 2 * D;
  ProfScope ps("dip_cov_kernel", st, 2.0 * (double)B * D * D, 4.0 * ((double)B * D + (double)D * D));
  hipLaunchKernelGGL(dip_rows_kernel, dim3(B), dim3(256), 0, st, mu, mu_rs, lv, lv_rs, c, e, B, D);
  CTVAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(dip_cov_kernel, dim3(D), dim3(256), 0, st, c, e, G, part, B, D, l_diag, l_off);
  CTVAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(dip_finish_kernel, dim3(1), dim3(256), 0, st, part, D, out);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_dip_backward(const float* state, const float* go, float* g_mu, float* g_lv, int B, int D, hipStream_t st) {
  if (D > 1024) return kErrBadArg;
  const float* c = state;
  const float* G = c + (size_t)B * D;
  const float* e = G + (size_t)D * D;
  const float* out = e + D + 2 * D;
  hipLaunchKernelGGL(dip_bwd_kernel, dim3(B), dim3(256), 0, st, c, e, G, out, go, g_mu, g_lv, B, D);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
