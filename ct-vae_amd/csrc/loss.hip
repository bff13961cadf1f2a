// Loss reductions: F.mse_loss mean over every element (vanilla_vae.py:140,
// mcq_vae.py:279, ct_mcq_vae.py:611) and the Gaussian KL term mean_b(-0.5*sum_d(1+lv-mu^2-e^lv))
// (vanilla_vae.py:143).  Wavefront shuffle reductions -> one partial per workgroup -> one
// finishing workgroup that adds the partials in double and emits the scalars.  HBM-bound:
// algorithmic bytes = 2 * 4 * n (recons + input) + 2 * 4 * B * L.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

constexpr int kLossBlocks = 1024;

// per-element reconstruction term: MODE 0 (t = r - x): t^2;  MODE 1 (LogCoshVAE, logcosh_vae.py:141-148, written the way the
// reference writes it): alpha t + log(1 + exp(-2 alpha t))   (the constant -log 2 and the factor 1/alpha are applied once, in
// loss_finish_kernel)
template <int MODE>
__device__ __forceinline__ float recon_term(float d, float alpha) {
  if constexpr (MODE == 0) return d * d;
  else if constexpr (MODE == 2) return d * d + fabsf(d);          // F.mse_loss + F.l1_loss (swae.py:121-122) in one pass
  else return alpha * d + logf(1.f + expf(-2.f * alpha * d));
}
// its derivative: 2 t  resp.  alpha (1 - e) / (1 + e), e = exp(-2 alpha t)   (= alpha tanh(alpha t))
template <int MODE>
__device__ __forceinline__ float recon_term_grad(float d, float alpha) {
  if constexpr (MODE == 0) return 2.f * d;
  else if constexpr (MODE == 2) return 2.f * d + (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));      // torch: sign(0) = 0
  else {
    const float e = expf(-2.f * alpha * d);
    return e < 3.0e38f ? alpha * (1.f - e) / (1.f + e) : -alpha;
  }
}

// Optional: the gradients of loss = recon + M_N * kld for a unit upstream gradient, written by the SAME pass that forms the partial
// sums (gr != null): the loss value is not needed for them, so the backward launch and its second read of both pictures
// disappear whenever the loss is the root of the backward pass (kernels.VAELoss).  Same expressions as mse_bwd_body / kl_bwd_body
// with go = 1: the same bits.
struct LossGradOut {
  float* gr;
  float* gmu;
  float* glv;
  float scale;   // 1 / n
  float M_N;
  int ract;
};

template <int MODE>
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ r, const float* __restrict__ x,
                                                          float* __restrict__ part, long n4, long n,
                                                          const float* __restrict__ mu, long mu_rs,
                                                          const float* __restrict__ lv, long lv_rs, int B, int L, float alpha,
                                                          const LossGradOut go) {
  __shared__ float sm[4];
  float s = 0.f;
  const long stride = (long)gridDim.x * 256;
  const float gsc = 1.f * go.scale;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 a = reinterpret_cast<const f32x4*>(r)[i];
    f32x4 b = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) s += recon_term<MODE>(a[k] - b[k], alpha);
    if (go.gr != nullptr) {
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = gsc * recon_term_grad<MODE>(a[k] - b[k], alpha) * act_bwd_from_out(a[k], go.ract);
      reinterpret_cast<f32x4*>(go.gr)[i] = o;
    }
  }
  if (blockIdx.x == 0) {  // tail (n % 4)
    for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) {
      s += recon_term<MODE>(r[i] - x[i], alpha);
      if (go.gr != nullptr) go.gr[i] = gsc * recon_term_grad<MODE>(r[i] - x[i], alpha) * act_bwd_from_out(r[i], go.ract);
    }
  }
  float k = 0.f;          // KL terms 1 + lv - mu^2 - e^lv, spread over the same grid
  if (mu != nullptr) {
    const float ksc = 1.f * go.M_N / (float)B;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)B * L; i += stride) {
      int b = (int)(i / L), d = (int)(i - (long)b * L);
      float m = mu[b * mu_rs + d], l = lv[b * lv_rs + d];
      k += 1.f + l - m * m - expf(l);
      if (go.gmu != nullptr) {
        go.gmu[i] = ksc * m;
        go.glv[i] = ksc * 0.5f * (expf(l) - 1.f);
      }
    }
  }
  s = block_sum_256(s, sm);
  k = block_sum_256(k, sm);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = s;
    part[gridDim.x + blockIdx.x] = k;
  }
}

// out[0] = loss = mse + M_N*kld (+ extra[0] if given), out[1] = mse, out[2] = kld, out[3] = -kld ('KLD' key, vanilla_vae.py:146)
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ part, int nparts, double inv_n, int has_kl,
                                                          int B, float M_N, const float* __restrict__ extra,
                                                          float* __restrict__ out, float post_sub, float post_scale) {
  __shared__ double smd[8];
  double s = 0.0, k = 0.0;
  for (int i0 = 0; i0 < nparts; i0 += 4 * 256) {       // four pairs of loads in flight, added in the order of the plain loop
    float ts[4], tk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + threadIdx.x + 256 * u, ic = i < nparts ? i : nparts - 1;      // clamped index, unconditional loads
      const float vs = part[ic], vk = part[nparts + ic];
      ts[u] = i < nparts ? vs : 0.f;
      tk[u] = i < nparts ? vk : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { s += (double)ts[u]; k += (double)tk[u]; }
  }
  s = wave_sum_d(s);
  k = wave_sum_d(k);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { smd[w] = s; smd[4 + w] = k; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double st = smd[0] + smd[1] + smd[2] + smd[3];
    const double kt = smd[4] + smd[5] + smd[6] + smd[7];
    float mse = ((float)(st * inv_n) - post_sub) * post_scale;   // (0, 1) for the MSE; (log 2, 1/alpha) for log-cosh
    float kld = has_kl ? (float)(-0.5 * kt / (double)B) : 0.f;
    float loss = mse + M_N * kld;
    if (extra != nullptr) loss += extra[0];
    out[0] = loss;
    out[1] = mse;
    out[2] = kld;
    out[3] = -kld;
  }
}

// g_r = go * term'(r - x) * scale      (MSE: 2 (r - x) / n;  log-cosh: tanh(alpha (r - x)) / n)
// ract != ACT_NONE: r is the output of that activation (the Tanh closing final_layer, vanilla_vae.py:74) and the result is
// the gradient w.r.t. its INPUT, g_r * act'(r) -- the producer's activation-backward pass folded into this one.
template <int MODE>
__device__ __forceinline__ void mse_bwd_body(const float* __restrict__ r, const float* __restrict__ x, const float* __restrict__ go,
                                             float* __restrict__ gr, long n4, long n, float scale, float alpha, int blk, int nblk,
                                             int ract) {
  const float sc = go[0] * scale;
  const long stride = (long)nblk * 256;
  for (long i = (long)blk * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 a = reinterpret_cast<const f32x4*>(r)[i];
    f32x4 b = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = sc * recon_term_grad<MODE>(a[k] - b[k], alpha) * act_bwd_from_out(a[k], ract);
    reinterpret_cast<f32x4*>(gr)[i] = o;
  }
  if (blk == 0)
    for (long i = n4 * 4 + threadIdx.x; i < n; i += 256)
      gr[i] = sc * recon_term_grad<MODE>(r[i] - x[i], alpha) * act_bwd_from_out(r[i], ract);
}

template <int MODE>
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ r, const float* __restrict__ x,
                                                      const float* __restrict__ go, float* __restrict__ gr, long n4, long n,
                                                      float scale, float alpha, int ract) {
  mse_bwd_body<MODE>(r, x, go, gr, n4, n, scale, alpha, blockIdx.x, gridDim.x, ract);
}

// g_mu = go*M_N*mu/B ; g_lv = go*M_N*0.5*(e^lv - 1)/B     (dense [B][L] outputs)
__device__ __forceinline__ void kl_bwd_body(const float* __restrict__ mu, long mu_rs, const float* __restrict__ lv, long lv_rs,
                                            const float* __restrict__ go, float* __restrict__ gmu, float* __restrict__ glv, int B,
                                            int L, float M_N, int blk) {
  const int i = blk * 256 + threadIdx.x;
  if (i >= B * L) return;
  const int b = i / L, d = i - b * L;
  const float sc = go[0] * M_N / (float)B;
  gmu[i] = sc * mu[b * mu_rs + d];
  glv[i] = sc * 0.5f * (expf(lv[b * lv_rs + d]) - 1.f);
}

__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ mu, long mu_rs, const float* __restrict__ lv,
                                                     long lv_rs, const float* __restrict__ go, float* __restrict__ gmu,
                                                     float* __restrict__ glv, int B, int L, float M_N) {
  kl_bwd_body(mu, mu_rs, lv, lv_rs, go, gmu, glv, B, L, M_N, blockIdx.x);
}

// both gradients of loss = recon + M_N * kld in ONE launch: blocks [0, nb) the reconstruction term, the rest the KL term
template <int MODE>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ r, const float* __restrict__ x,
                                                       const float* __restrict__ go, float* __restrict__ gr, long n4, long n,
                                                       float scale, float alpha, int nb, const float* __restrict__ mu, long mu_rs,
                                                       const float* __restrict__ lv, long lv_rs, float* __restrict__ gmu,
                                                       float* __restrict__ glv, int B, int L, float M_N, int ract) {
  if ((int)blockIdx.x < nb) mse_bwd_body<MODE>(r, x, go, gr, n4, n, scale, alpha, blockIdx.x, nb, ract);
  else kl_bwd_body(mu, mu_rs, lv, lv_rs, go, gmu, glv, B, L, M_N, (int)blockIdx.x - nb);
}

size_t loss_workspace_floats() { return 2 * kLossBlocks; }

int launch_loss_forward(const float* r, const float* x, long n, const float* mu, long mu_rs, const float* lv, long lv_rs,
                        int B, int L, float M_N, const float* extra, float* out4, float* ws, size_t ws_bytes,
                        hipStream_t st, float logcosh_alpha, float* g_r, float* g_mu, float* g_lv, int ract) {
  if (ws_bytes / sizeof(float) < loss_workspace_floats() || n <= 0) return kErrWorkspace;
  const LossGradOut go{g_r, g_mu, g_lv, (float)(1.0 / ((double)n * (logcosh_alpha > 0.f ? logcosh_alpha : 1.f))), M_N, ract};
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > kLossBlocks) blocks = kLossBlocks;
  if (blocks < 1) blocks = 1;
  {
  ProfScope ps("mse_partial_kernel", st, 0.0, 8.0 * (double)n);
  if (logcosh_alpha > 0.f)
    hipLaunchKernelGGL(mse_partial_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, r, x, ws, n4, n, mu, mu_rs, lv, lv_rs, B, L, logcosh_alpha, go);
  else if (logcosh_alpha < 0.f)      // internal code for the squared + absolute error term (ctvae_l2l1_loss_forward)
    hipLaunchKernelGGL(mse_partial_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, r, x, ws, n4, n, mu, mu_rs, lv, lv_rs, B, L, 0.f, go);
  else
    hipLaunchKernelGGL(mse_partial_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, r, x, ws, n4, n, mu, mu_rs, lv, lv_rs, B, L, 0.f, go);
  }
  CTVAE_LAUNCH_CHECK();
  ProfScope ps2("loss_finish_kernel", st, 0.0, 8.0 * (double)B * L);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, ws, (int)blocks, 1.0 / (double)n, mu != nullptr ? 1 : 0,
                     B, M_N, extra, out4, logcosh_alpha > 0.f ? 0.69314718f : 0.f, logcosh_alpha > 0.f ? 1.f / logcosh_alpha : 1.f);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_mse_backward(const float* r, const float* x, const float* go, float* gr, long n, hipStream_t st, float logcosh_alpha,
                        int ract) {
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  ProfScope ps("mse_bwd_kernel", st, 0.0, 12.0 * (double)n);
  if (logcosh_alpha > 0.f)   // d/dt of (alpha t + log(1 + e^{-2 alpha t}) - log 2) / alpha = tanh(alpha t)
    hipLaunchKernelGGL(mse_bwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, r, x, go, gr, n4, n,
                       (float)(1.0 / ((double)n * logcosh_alpha)), logcosh_alpha, ract);
  else if (logcosh_alpha < 0.f)
    hipLaunchKernelGGL(mse_bwd_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, r, x, go, gr, n4, n, (float)(1.0 / (double)n), 0.f,
                       ract);
  else
    hipLaunchKernelGGL(mse_bwd_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, r, x, go, gr, n4, n, (float)(1.0 / (double)n), 0.f,
                       ract);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_kl_backward(const float* mu, long mu_rs, const float* lv, long lv_rs, const float* go, float* gmu, float* glv,
                       int B, int L, float M_N, hipStream_t st) {
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(ceil_div(B * L, 256)), dim3(256), 0, st, mu, mu_rs, lv, lv_rs, go, gmu, glv, B, L, M_N);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_loss_backward(const float* r, const float* x, const float* go, float* gr, long n, float logcosh_alpha, const float* mu,
                         long mu_rs, const float* lv, long lv_rs, float* gmu, float* glv, int B, int L, float M_N, hipStream_t st,
                         int ract) {
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  const int nb = (int)blocks, nkl = ceil_div(B * L, 256);
  ProfScope ps("loss_bwd_kernel", st, 0.0, 12.0 * (double)n + 16.0 * (double)B * L);
  if (logcosh_alpha > 0.f)
    hipLaunchKernelGGL(loss_bwd_kernel<1>, dim3(nb + nkl), dim3(256), 0, st, r, x, go, gr, n4, n, (float)(1.0 / ((double)n * logcosh_alpha)),
                       logcosh_alpha, nb, mu, mu_rs, lv, lv_rs, gmu, glv, B, L, M_N, ract);
  else
    hipLaunchKernelGGL(loss_bwd_kernel<0>, dim3(nb + nkl), dim3(256), 0, st, r, x, go, gr, n4, n, (float)(1.0 / (double)n), 0.f, nb, mu,
                       mu_rs, lv, lv_rs, gmu, glv, B, L, M_N, ract);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
