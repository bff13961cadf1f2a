// Train-mode BatchNorm2d + LeakyReLU on NHWC activations (vanilla_vae.py:30-31,56-57,71-72).
// The activation tensor is a row-major [R = B*H*W][C] matrix, statistics are per column.
//
// forward : stats (shifted sums per block -> Chan merge)  ->  finalize (mean, invstd, scale/shift,
//           running stats: momentum 0.1, unbiased variance)  ->  apply + LeakyReLU
// backward: g_bn = g_a * lrelu'(a);  dgamma = sum g_bn*xhat, dbeta = sum g_bn;
//           g_y = gamma*invstd * (g_bn - dbeta/R - xhat*dgamma/R)
// All of these are HBM-bound streaming kernels (roofline: bytes / 8 TB/s).
#include "common.hpp"

namespace ctvae {

constexpr int kBnMaxBlocks = 1024;

// ---- forward statistics -------------------------------------------------------------------------
// part[blk][c] = (n, mean, M2) of the block's rows for column c
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ y, float* __restrict__ part,
                                                               int R, int C, int rows_per_block) {
  __shared__ float sm[3 * 256];
  const int tid = threadIdx.x;
  const int cpb = C < 256 ? C : 256;  // columns per pass
  const int rl = 256 / cpb;           // row lanes
  const int col_l = tid % cpb, rlane = tid / cpb;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
  for (int c0 = 0; c0 < C; c0 += cpb) {
    const int col = c0 + col_l;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (rlane < rl && col < C && r0 + rlane < r1) {
      const float shift = y[(long)(r0 + rlane) * C + col];
      float s1 = 0.f, s2 = 0.f;
      int cnt = 0;
      for (int r = r0 + rlane; r < r1; r += rl) {
        float d = y[(long)r * C + col] - shift;
        s1 += d;
        s2 += d * d;
        ++cnt;
      }
      n = (float)cnt;
      mean = shift + s1 / n;
      m2 = s2 - s1 * s1 / n;
      if (m2 < 0.f) m2 = 0.f;
    }
    __syncthreads();
    sm[tid] = n;
    sm[256 + tid] = mean;
    sm[512 + tid] = m2;
    __syncthreads();
    if (rlane == 0 && col < C) {
      for (int l = 1; l < rl; ++l) {
        float nb = sm[l * cpb + col_l], mb = sm[256 + l * cpb + col_l], qb = sm[512 + l * cpb + col_l];
        if (nb > 0.f) {
          float nt = n + nb, d = mb - mean;
          mean += d * (nb / nt);
          m2 += qb + d * d * (n * nb / nt);
          n = nt;
        }
      }
      float* p = part + ((long)blockIdx.x * C + col) * 3;
      p[0] = n; p[1] = mean; p[2] = m2;
    }
  }
}

// one thread per channel merges the block partials; writes save_mean/save_invstd/scale/shift and updates running stats
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nblocks, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float momentum, float eps, float* __restrict__ save_mean,
                                                          float* __restrict__ save_invstd, float* __restrict__ scale,
                                                          float* __restrict__ shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int b = 0; b < nblocks; ++b) {
    const float* p = part + ((long)b * C + c) * 3;
    float nb = p[0], mb = p[1], qb = p[2];
    if (nb > 0.f) {
      float nt = n + nb, d = mb - mean;
      mean += d * (nb / nt);
      m2 += qb + d * d * (n * nb / nt);
      n = nt;
    }
  }
  const float var = m2 / n;
  const float invstd = 1.0f / sqrtf(var + eps);
  save_mean[c] = mean;
  save_invstd[c] = invstd;
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
  if (running_mean != nullptr) {
    const float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// eval-mode scale/shift from running statistics
__global__ __launch_bounds__(256) void bn_eval_coeff_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ running_mean,
                                                            const float* __restrict__ running_var, float eps,
                                                            float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(running_var[c] + eps);
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - running_mean[c] * sc;
}

// a = act(y*scale[c] + shift[c]); C % 4 == 0
__global__ __launch_bounds__(256) void bn_apply_act_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float* __restrict__ out,
                                                           long n4, int C, int act) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const int c = (int)((i * 4) % C);
    f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
    f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
    f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = act_fwd(v[k] * sc[k] + sh[k], act);
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

// ---- backward -------------------------------------------------------------------------------------
// part[blk][c] = (sum g_bn, sum g_bn*xhat)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ ga, const float* __restrict__ a_out,
                                                             const float* __restrict__ y, const float* __restrict__ save_mean,
                                                             const float* __restrict__ save_invstd, float* __restrict__ part,
                                                             int R, int C, int rows_per_block, int act) {
  __shared__ float sm[2 * 256];
  const int tid = threadIdx.x;
  const int cpb = C < 256 ? C : 256;
  const int rl = 256 / cpb;
  const int col_l = tid % cpb, rlane = tid / cpb;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
  for (int c0 = 0; c0 < C; c0 += cpb) {
    const int col = c0 + col_l;
    float s1 = 0.f, s2 = 0.f;
    if (rlane < rl && col < C) {
      const float mean = save_mean[col], invstd = save_invstd[col];
      for (int r = r0 + rlane; r < r1; r += rl) {
        const long idx = (long)r * C + col;
        float g = ga[idx] * act_bwd_from_out(a_out[idx], act);
        s1 += g;
        s2 += g * ((y[idx] - mean) * invstd);
      }
    }
    __syncthreads();
    sm[tid] = s1;
    sm[256 + tid] = s2;
    __syncthreads();
    if (rlane == 0 && col < C) {
      for (int l = 1; l < rl; ++l) {
        s1 += sm[l * cpb + col_l];
        s2 += sm[256 + l * cpb + col_l];
      }
      part[((long)blockIdx.x * C + col) * 2 + 0] = s1;
      part[((long)blockIdx.x * C + col) * 2 + 1] = s2;
    }
  }
}

// dgamma/dbeta (accumulate flag) and the three per-channel coefficients of g_y = k1*g_bn + k2*y + k3
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblocks, int C, float R,
                                                              const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                              const float* __restrict__ save_invstd, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate,
                                                              float* __restrict__ coef /* [3][C] */) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int b = 0; b < nblocks; ++b) {
    s1 += (double)part[((long)b * C + c) * 2 + 0];
    s2 += (double)part[((long)b * C + c) * 2 + 1];
  }
  const float db = (float)s1, dg = (float)s2;
  dgamma[c] = (accumulate ? dgamma[c] : 0.f) + dg;
  dbeta[c] = (accumulate ? dbeta[c] : 0.f) + db;
  const float invstd = save_invstd[c], mean = save_mean[c];
  const float k1 = gamma[c] * invstd;
  // g_y = k1 * (g - db/R - xhat*dg/R), xhat = (y-mean)*invstd
  const float k2 = -k1 * dg / R * invstd;
  const float k3 = -k1 * db / R - k2 * mean;
  coef[c] = k1;
  coef[C + c] = k2;
  coef[2 * C + c] = k3;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ ga, const float* __restrict__ a_out,
                                                           const float* __restrict__ y, const float* __restrict__ coef,
                                                           float* __restrict__ gy, long n4, int C, int act) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const int c = (int)((i * 4) % C);
    f32x4 g = reinterpret_cast<const f32x4*>(ga)[i];
    f32x4 ao = reinterpret_cast<const f32x4*>(a_out)[i];
    f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
    f32x4 k1 = *reinterpret_cast<const f32x4*>(coef + c);
    f32x4 k2 = *reinterpret_cast<const f32x4*>(coef + C + c);
    f32x4 k3 = *reinterpret_cast<const f32x4*>(coef + 2 * C + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = k1[k] * (g[k] * act_bwd_from_out(ao[k], act)) + k2[k] * yv[k] + k3[k];
    reinterpret_cast<f32x4*>(gy)[i] = o;
  }
}

static int stat_blocks(int R, int* rows_per_block) {
  int nb = R / 64;
  if (nb < 1) nb = 1;
  if (nb > kBnMaxBlocks) nb = kBnMaxBlocks;
  *rows_per_block = ceil_div(R, nb);
  return ceil_div(R, *rows_per_block);
}

size_t bn_workspace_floats(int C) { return (size_t)kBnMaxBlocks * C * 3 + 3 * (size_t)C; }

int launch_bn_forward(const float* y, int R, int C, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float momentum, float eps, int training, int act, float* out,
                      float* save_mean, float* save_invstd, float* ws, size_t ws_bytes, hipStream_t st) {
  if (C % 4 != 0 || R <= 0) return kErrBadArg;
  if (ws_bytes / sizeof(float) < bn_workspace_floats(C)) return kErrWorkspace;
  float* part = ws;
  float* scale = ws + (size_t)kBnMaxBlocks * C * 3;
  float* shift = scale + C;
  if (training) {
    int rpb;
    const int nb = stat_blocks(R, &rpb);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nb), dim3(256), 0, st, y, part, R, C, rpb);
    CTVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, part, nb, C, gamma, beta,
                       running_mean, running_var, momentum, eps, save_mean, save_invstd, scale, shift);
    CTVAE_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, C, gamma, beta, running_mean,
                       running_var, eps, scale, shift);
    CTVAE_LAUNCH_CHECK();
  }
  const long n4 = (long)R * C / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bn_apply_act_kernel, dim3((unsigned)blocks), dim3(256), 0, st, y, scale, shift, out, n4, C, act);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_bn_backward(const float* ga, const float* a_out, const float* y, int R, int C, const float* gamma,
                       const float* save_mean, const float* save_invstd, int act, float* gy, float* dgamma,
                       float* dbeta, int accumulate, float* ws, size_t ws_bytes, hipStream_t st) {
  if (C % 4 != 0 || R <= 0) return kErrBadArg;
  if (ws_bytes / sizeof(float) < bn_workspace_floats(C)) return kErrWorkspace;
  float* part = ws;
  float* coef = ws + (size_t)kBnMaxBlocks * C * 3;
  int rpb;
  const int nb = stat_blocks(R, &rpb);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nb), dim3(256), 0, st, ga, a_out, y, save_mean, save_invstd, part, R, C,
                     rpb, act);
  CTVAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, part, nb, C, (float)R, gamma,
                     save_mean, save_invstd, dgamma, dbeta, accumulate, coef);
  CTVAE_LAUNCH_CHECK();
  const long n4 = (long)R * C / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ga, a_out, y, coef, gy, n4, C, act);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
