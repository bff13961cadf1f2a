// VampPrior KL term of VampVAE (models/vampvae.py:140-171): with the posterior sample z [B][D], its parameters mu, lv and the
// K pseudo-input posteriors pmu, plv [K][D] (the encoder applied to the learned pseudo-inputs),
//     E_log_q = mean_b sum_d -0.5 (lv + (z - mu)^2) / exp(lv)
//     s[b,k]  = sum_d -0.5 (plv[k] + (z[b] - pmu[k])^2) / exp(plv[k]) - log K,   E_log_p = mean_b logsumexp_k s[b,k]
//     kld     = -(E_log_p - E_log_q)
// The reference forms the [B,K,D] tensor; here a workgroup owns a sample b (waves walk the components, lanes the latent
// dimensions), keeps s[b,:] in LDS for the logsumexp and leaves the softmax weights for the backward pass, which has one
// producer per output element (samples for z / mu / lv, components for pmu / plv): deterministic, no atomics.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

namespace {

constexpr int kVampMaxK = 512;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void vamp_fwd_kernel(const float* __restrict__ z, const float* __restrict__ mu,
                                                      const float* __restrict__ lv, const float* __restrict__ pmu,
                                                      const float* __restrict__ plv, int D, int K, float* __restrict__ wgt,
                                                      float* __restrict__ row /* [B][2]: lse, q */) {
  __shared__ float sS[kVampMaxK];
  __shared__ float sR[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const float* zb = z + (long)b * D;
  const float logK = __logf((float)K);
  for (int k = wave; k < K; k += 4) {
    float a = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float l = plv[(long)k * D + d], dl = zb[d] - pmu[(long)k * D + d];
      a += -0.5f * (l + dl * dl) * __expf(-l);
    }
    a = wsum(a);
    if (lane == 0) sS[k] = a - logK;
  }
  float q = 0.f;
  for (int d = tid; d < D; d += 256) {
    const float l = lv[(long)b * D + d], dl = zb[d] - mu[(long)b * D + d];
    q += -0.5f * (l + dl * dl) * __expf(-l);
  }
  q = wsum(q);
  if (lane == 0) sR[wave] = q;
  __syncthreads();
  // logsumexp over the K components (K <= 512: two per thread)
  float m = -INFINITY;
  for (int k = tid; k < K; k += 256) m = fmaxf(m, sS[k]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) sR[4 + wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sR[4], sR[5]), fmaxf(sR[6], sR[7]));
  const float qtot = (sR[0] + sR[1]) + (sR[2] + sR[3]);
  __syncthreads();
  float e = 0.f;
  for (int k = tid; k < K; k += 256) e += __expf(sS[k] - m);
  e = wsum(e);
  if (lane == 0) sR[wave] = e;
  __syncthreads();
  const float tot = (sR[0] + sR[1]) + (sR[2] + sR[3]);
  for (int k = tid; k < K; k += 256) wgt[(long)b * K + k] = __expf(sS[k] - m) / tot;
  if (tid == 0) {
    row[2 * b] = m + __logf(tot);
    row[2 * b + 1] = qtot;
  }
}

__global__ __launch_bounds__(256) void vamp_finish_kernel(const float* __restrict__ row, int B, float* __restrict__ out) {
  __shared__ double sm[8];
  double a = 0.0, q = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) { a += (double)row[2 * b]; q += (double)row[2 * b + 1]; }
  a = wave_sum_d(a);
  q = wave_sum_d(q);
  if ((threadIdx.x & 63) == 0) { sm[(threadIdx.x >> 6) * 2] = a; sm[(threadIdx.x >> 6) * 2 + 1] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double ep = ((sm[0] + sm[2]) + (sm[4] + sm[6])) / B, eq = ((sm[1] + sm[3]) + (sm[5] + sm[7])) / B;
    out[0] = (float)(-(ep - eq));      // kld
    out[1] = (float)ep;
    out[2] = (float)eq;
  }
}

// blocks [0, B): sample b -> dz, dmu, dlv;  blocks [B, B+K): component k -> dpmu, dplv.   g = d loss / d kld
__global__ __launch_bounds__(256) void vamp_bwd_kernel(const float* __restrict__ z, const float* __restrict__ mu,
                                                      const float* __restrict__ lv, const float* __restrict__ pmu,
                                                      const float* __restrict__ plv, const float* __restrict__ wgt,
                                                      const float* __restrict__ g, int B, int D, int K, float* __restrict__ dz,
                                                      float* __restrict__ dmu, float* __restrict__ dlv, float* __restrict__ dpmu,
                                                      float* __restrict__ dplv) {
  const float gq = g[0] / (float)B, gp = -g[0] / (float)B;       // d kld / d q_b = 1/B, d kld / d lse_b = -1/B
  if ((int)blockIdx.x < B) {
    const int b = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += 256) {
      const float zz = z[(long)b * D + d];
      const float l = lv[(long)b * D + d], dl = zz - mu[(long)b * D + d], el = __expf(-l);
      float gz = gq * (-dl * el);
      dmu[(long)b * D + d] = gq * (dl * el);
      dlv[(long)b * D + d] = gq * (-0.5f * el * (1.f - l - dl * dl));
      float acc = 0.f;
      for (int k = 0; k < K; ++k) {
        const float pl = plv[(long)k * D + d], pd = zz - pmu[(long)k * D + d];
        acc += wgt[(long)b * K + k] * (-pd * __expf(-pl));
      }
      dz[(long)b * D + d] = gz + gp * acc;
    }
  } else {
    const int k = blockIdx.x - B;
    for (int d = threadIdx.x; d < D; d += 256) {
      const float pl = plv[(long)k * D + d], pm = pmu[(long)k * D + d], el = __expf(-pl);
      float am = 0.f, al = 0.f;
      for (int b = 0; b < B; ++b) {
        const float w = wgt[(long)b * K + k], pd = z[(long)b * D + d] - pm;
        am += w * pd;
        al += w * (1.f - pl - pd * pd);
      }
      dpmu[(long)k * D + d] = gp * am * el;
      dplv[(long)k * D + d] = gp * (-0.5f) * el * al;
    }
  }
}

}  // namespace

int launch_vamp_forward(const float* z, const float* mu, const float* lv, const float* pmu, const float* plv, int B, int D, int K,
                        float* out3, float* wgt, float* ws, size_t ws_bytes, hipStream_t st) {
  if (!z || !mu || !lv || !pmu || !plv || !out3 || !wgt || !ws || B < 1 || D < 1 || K < 1 || K > kVampMaxK) return kErrBadArg;
  if (ws_bytes / sizeof(float) < (size_t)2 * B) return kErrWorkspace;
  {
    ProfScope ps("vamp_fwd_kernel", st, 6.0 * B * (double)K * D, 4.0 * ((double)B * D * 3 + 2.0 * K * D + (double)B * K));
    hipLaunchKernelGGL(vamp_fwd_kernel, dim3(B), dim3(256), 0, st, z, mu, lv, pmu, plv, D, K, wgt, ws);
    CTVAE_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(vamp_finish_kernel, dim3(1), dim3(256), 0, st, ws, B, out3);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_vamp_backward(const float* z, const float* mu, const float* lv, const float* pmu, const float* plv, const float* wgt,
                         const float* g, int B, int D, int K, float* dz, float* dmu, float* dlv, float* dpmu, float* dplv,
                         hipStream_t st) {
  if (!z || !mu || !lv || !pmu || !plv || !wgt || !g || !dz || !dmu || !dlv || !dpmu || !dplv || B < 1 || D < 1 || K < 1)
    return kErrBadArg;
  ProfScope ps("vamp_bwd_kernel", st, 12.0 * B * (double)K * D, 4.0 * ((double)B * D * 6 + 4.0 * K * D + (double)B * K));
  hipLaunchKernelGGL(vamp_bwd_kernel, dim3(B + K), dim3(256), 0, st, z, mu, lv, pmu, plv, wgt, g, B, D, K, dz, dmu, dlv, dpmu, dplv);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
