// GammaVAE's latent section (models/gamma_vae.py:108-193) and its Sigmoid output.
//   reparameterisation by shape augmentation (:108-149): with the draw zhat ~ Gamma(alpha + Bs, 1) given,
//       a = alpha + Bs,  eps = sqrt(9a - 3) ((zhat / (a - 1/3))^(1/3) - 1),  z = (a - 1/3) (1 + eps / sqrt(9a - 3))^3 / beta
//     (z * beta == zhat up to rounding; the backward pass forms both partial derivatives w.r.t. alpha as autograd does -- they
//     cancel up to rounding -- and d z / d beta = -z / beta)
//   KL between Gamma posteriors and the Gamma prior as the reference writes it (:151-171):
//       I(a,b,c,d) = -c d / a - b log a - lgamma(b) + (b - 1)(digamma(d) + log c),   kld = sum_d I(c,d,c,d) - I(1/alpha, beta, c, d)
//     with c = 1 / prior_alpha, d = prior_beta; mean over the batch.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

namespace {

// psi(x), x > 0: recurrence up to x >= 6, then the asymptotic series
__device__ __forceinline__ float digammaf_pos(float x) {
  float r = 0.f;
  while (x < 6.f) { r -= 1.f / x; x += 1.f; }
  const float i = 1.f / x, i2 = i * i;
  return r + logf(x) - 0.5f * i - i2 * (1.f / 12.f - i2 * (1.f / 120.f - i2 * (1.f / 252.f)));
}

__global__ __launch_bounds__(256) void gamma_reparam_fwd_kernel(const float* __restrict__ alpha, const float* __restrict__ beta,
                                                               const float* __restrict__ zhat, float shape_b, float* __restrict__ z,
                                                               long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float a = alpha[i] + shape_b, s = sqrtf(9.f * a - 3.f), m = a - 1.f / 3.f;
    const float eps = s * (powf(zhat[i] / m, 1.f / 3.f) - 1.f);
    const float t = 1.f + eps / s;
    z[i] = m * t * t * t / beta[i];
  }
}

__global__ __launch_bounds__(256) void gamma_reparam_bwd_kernel(const float* __restrict__ g, const float* __restrict__ alpha,
                                                               const float* __restrict__ beta, const float* __restrict__ zhat,
                                                               float shape_b, float* __restrict__ g_alpha, float* __restrict__ g_beta,
                                                               long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float a = alpha[i] + shape_b, q = 9.f * a - 3.f, s = sqrtf(q), m = a - 1.f / 3.f, zh = zhat[i], b = beta[i];
    const float r = powf(zh / m, 1.f / 3.f);
    const float eps = s * (r - 1.f), t = 1.f + eps / s;
    // eps(a): d/da [ s (r - 1) ] = 4.5 / s (r - 1) + s * (-(1/3) r / m)
    const float deps = 4.5f / s * (r - 1.f) - s * r / (3.f * m);
    // h(a, eps) = m t^3, t = 1 + eps / s:  dh/da|eps = t^3 + 3 m t^2 (-4.5 eps / (q s)),  dh/deps = 3 m t^2 / s
    const float dha = t * t * t - 13.5f * m * t * t * eps / (q * s), dhe = 3.f * m * t * t / s;
    const float zi = m * t * t * t / b;
    g_alpha[i] = g[i] * (dha + dhe * deps) / b;
    g_beta[i] = -g[i] * zi / b;
  }
}

// one workgroup per sample: row[b] = sum_d kld(b, d)
__global__ __launch_bounds__(256) void gamma_kl_fwd_kernel(const float* __restrict__ alpha, const float* __restrict__ beta, int D,
                                                          float c, float d, float* __restrict__ row) {
  __shared__ float sm[4];
  const int b = blockIdx.x;
  const float k0 = digammaf_pos(d) + logf(c);
  const float i_prior = -d - d * logf(c) - lgammaf(d) + (d - 1.f) * k0;      // I(c,d,c,d): -c d / c = -d
  float acc = 0.f;
  for (int j = threadIdx.x; j < D; j += 256) {
    const float al = alpha[(long)b * D + j], be = beta[(long)b * D + j];
    // a = 1 / alpha:  I = -c d alpha + be log(alpha) - lgamma(be) + (be - 1) k0
    acc += i_prior - (-c * d * al + be * logf(al) - lgammaf(be) + (be - 1.f) * k0);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) row[b] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void gamma_kl_finish_kernel(const float* __restrict__ row, int B, float* __restrict__ out) {
  __shared__ double sm[4];
  double a = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) a += (double)row[b];
  a = wave_sum_d(a);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(((sm[0] + sm[1]) + (sm[2] + sm[3])) / B);
}

__global__ __launch_bounds__(256) void gamma_kl_bwd_kernel(const float* __restrict__ g, const float* __restrict__ alpha,
                                                          const float* __restrict__ beta, long n, int B, float c, float d,
                                                          float* __restrict__ g_alpha, float* __restrict__ g_beta) {
  const float k0 = digammaf_pos(d) + logf(c), s = g[0] / (float)B;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float al = alpha[i], be = beta[i];
    g_alpha[i] = s * (c * d - be / al);
    g_beta[i] = s * (-logf(al) + digammaf_pos(be) - k0);
  }
}

__global__ __launch_bounds__(256) void sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n4) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = 1.f / (1.f + __expf(-v[k]));
    reinterpret_cast<f32x4*>(y)[i] = o;
  }
}

__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ gx,
                                                         long n4) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i], yv = reinterpret_cast<const f32x4*>(y)[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = gv[k] * yv[k] * (1.f - yv[k]);
    reinterpret_cast<f32x4*>(gx)[i] = o;
  }
}

int blocks_for(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

int launch_gamma_reparam_forward(const float* alpha, const float* beta, const float* zhat, float shape_b, float* z, long n, hipStream_t st) {
  if (!alpha || !beta || !zhat || !z || n <= 0) return kErrBadArg;
  hipLaunchKernelGGL(gamma_reparam_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, st, alpha, beta, zhat, shape_b, z, n);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_gamma_reparam_backward(const float* g, const float* alpha, const float* beta, const float* zhat, float shape_b, float* ga,
                                  float* gb, long n, hipStream_t st) {
  if (!g || !alpha || !beta || !zhat || !ga || !gb || n <= 0) return kErrBadArg;
  hipLaunchKernelGGL(gamma_reparam_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, st, g, alpha, beta, zhat, shape_b, ga, gb, n);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_gamma_kl_forward(const float* alpha, const float* beta, int B, int D, float prior_alpha, float prior_beta, float* out,
                            float* ws, size_t ws_bytes, hipStream_t st) {
  if (!alpha || !beta || !out || !ws || B < 1 || D < 1 || !(prior_alpha > 0.f) || !(prior_beta > 0.f)) return kErrBadArg;
  if (ws_bytes / sizeof(float) < (size_t)B) return kErrWorkspace;
  ProfScope ps("gamma_kl_fwd_kernel", st, 0.0, 8.0 * B * D);
  hipLaunchKernelGGL(gamma_kl_fwd_kernel, dim3(B), dim3(256), 0, st, alpha, beta, D, 1.f / prior_alpha, prior_beta, ws);
  CTVAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(gamma_kl_finish_kernel, dim3(1), dim3(256), 0, st, ws, B, out);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_gamma_kl_backward(const float* g, const float* alpha, const float* beta, int B, int D, float prior_alpha, float prior_beta,
                             float* ga, float* gb, hipStream_t st) {
  if (!g || !alpha || !beta || !ga || !gb || B < 1 || D < 1) return kErrBadArg;
  const long n = (long)B * D;
  hipLaunchKernelGGL(gamma_kl_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, st, g, alpha, beta, n, B, 1.f / prior_alpha, prior_beta, ga,
                     gb);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_sigmoid(const float* x_or_g, const float* y, float* out, long n, int backward, hipStream_t st) {
  if (!x_or_g || !out || n <= 0 || (n & 3) || (backward && !y)) return kErrBadArg;
  if (backward) hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, st, x_or_g, y, out, n / 4);
  else hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, st, x_or_g, out, n / 4);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
