// One rung of the ladder VAE's top-down pass (models/lvae.py:166-204): precision-weighted merge of the bottom-up posterior
// (mu_e, lv_e) with the top-down prediction (mu_t, lv_t), a reparameterised sample of the merged Gaussian, and the KL term the
// reference forms between the merged Gaussian (q) and the bottom-up one (p):
//     p1 = 1 / (e^lv_e + 1e-7),  p2 = 1 / (e^lv_t + 1e-7),  P = p1 + p2,  mu = (mu_e p1 + mu_t p2) / P,  lv = log(1 / P)
//     z  = eps e^{lv / 2} + mu
//     kl[b] = sum_d (lv_e - lv) + (e^lv + (mu - mu_e)^2) / (2 e^lv_e) - 0.5
// As torch ops that is ~20 elementwise launches per rung and as many again in autograd; here one launch each way (a workgroup
// per sample, the row sum by shuffles).
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

namespace {

constexpr float kMergeEps = 1e-7f;

__global__ __launch_bounds__(256) void ladder_fwd_kernel(const float* __restrict__ mu_e, const float* __restrict__ lv_e,
                                                        const float* __restrict__ mu_t, const float* __restrict__ lv_t,
                                                        const float* __restrict__ eps, int D, float* __restrict__ z,
                                                        float* __restrict__ kl) {
  __shared__ float sm[4];
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    const long i = (long)b * D + d;
    const float E = expf(lv_e[i]), p1 = 1.f / (E + kMergeEps), p2 = 1.f / (expf(lv_t[i]) + kMergeEps), P = p1 + p2;
    const float mu = (mu_e[i] * p1 + mu_t[i] * p2) / P, lv = logf(1.f / P);
    z[i] = eps[i] * expf(0.5f * lv) + mu;
    const float dl = mu - mu_e[i];
    acc += (lv_e[i] - lv) + (expf(lv) + dl * dl) / (2.f * E) - 0.5f;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) kl[b] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void ladder_bwd_kernel(const float* __restrict__ gz, const float* __restrict__ gkl,
                                                        const float* __restrict__ mu_e, const float* __restrict__ lv_e,
                                                        const float* __restrict__ mu_t, const float* __restrict__ lv_t,
                                                        const float* __restrict__ eps, long n, int D, float* __restrict__ g_mu_e,
                                                        float* __restrict__ g_lv_e, float* __restrict__ g_mu_t,
                                                        float* __restrict__ g_lv_t) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float gzi = gz != nullptr ? gz[i] : 0.f, gk = gkl != nullptr ? gkl[i / D] : 0.f;
    const float E = expf(lv_e[i]), T = expf(lv_t[i]), v1 = E + kMergeEps, v2 = T + kMergeEps;
    const float p1 = 1.f / v1, p2 = 1.f / v2, P = p1 + p2;
    const float mu = (mu_e[i] * p1 + mu_t[i] * p2) / P, dl = mu - mu_e[i];
    const float Gmu = gzi + gk * dl / E;
    const float GP = gzi * (-0.5f * eps[i] / (P * sqrtf(P))) + gk * (1.f / P - 1.f / (2.f * E * P * P));
    const float Gp1 = Gmu * (mu_e[i] - mu) / P + GP, Gp2 = Gmu * (mu_t[i] - mu) / P + GP;
    g_mu_e[i] = Gmu * p1 / P - gk * dl / E;
    g_mu_t[i] = Gmu * p2 / P;
    g_lv_e[i] = Gp1 * (-E * p1 * p1) + gk * (1.f - (1.f / P + dl * dl) / (2.f * E));
    g_lv_t[i] = Gp2 * (-T * p2 * p2);
  }
}

}  // namespace

int launch_ladder_forward(const float* mu_e, const float* lv_e, const float* mu_t, const float* lv_t, const float* eps, int B, int D,
                          float* z, float* kl, hipStream_t st) {
  if (!mu_e || !lv_e || !mu_t || !lv_t || !eps || !z || !kl || B < 1 || D < 1) return kErrBadArg;
  hipLaunchKernelGGL(ladder_fwd_kernel, dim3(B), dim3(256), 0, st, mu_e, lv_e, mu_t, lv_t, eps, D, z, kl);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ladder_backward(const float* gz, const float* gkl, const float* mu_e, const float* lv_e, const float* mu_t, const float* lv_t,
                           const float* eps, int B, int D, float* g_mu_e, float* g_lv_e, float* g_mu_t, float* g_lv_t, hipStream_t st) {
  if ((!gz && !gkl) || !mu_e || !lv_e || !mu_t || !lv_t || !eps || !g_mu_e || !g_lv_e || !g_mu_t || !g_lv_t || B < 1 || D < 1)
    return kErrBadArg;
  const long n = (long)B * D;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ladder_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gz, gkl, mu_e, lv_e, mu_t, lv_t, eps, n, D, g_mu_e,
                     g_lv_e, g_mu_t, g_lv_t);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
