// Train-mode BatchNorm2d + LeakyReLU on NHWC activations (vanilla_vae.py:30-31,56-57,71-72).
// The activation tensor is a row-major [R = B*H*W][C] matrix, statistics are per column.
//
// forward : per-tile (count, mean, M2) partials come for free from the conv epilogue (tapgemm.hip) or from
//           bn_stats_partial_kernel; bn_finalize_kernel merges them (Chan et al., parallel over partials),
//           writes mean / invstd / scale / shift and updates the running statistics (momentum 0.1, unbiased
//           variance); bn_apply_act_kernel streams y -> a = lrelu(y*scale+shift).
// backward: g_bn = g_a * lrelu'(a);  dgamma = sum g_bn*xhat, dbeta = sum g_bn;
//           g_y = gamma*invstd * (g_bn - dbeta/R - xhat*dgamma/R)   (as k1*g_bn + k2*y + k3 per channel)
// All of these are HBM-bound streaming kernels (roofline: bytes / 8 TB/s): 16-B loads, several in flight per lane.
#include "common.hpp"
#include "phase.hpp"
#include "finish.hpp"
#include "prof.hpp"

namespace ctvae {
// d act / d t of a launch-uniform activation at pre-activation t, for element-wise kernels: identity / LeakyReLU / ReLU as one
// select, tanh behind a branch the caller takes once per vector (common.hpp act_slope_bwd)
__device__ __forceinline__ float act_dt(float t, int act, float slope) {
  return act == ACT_TANH ? act_bwd_from_out(act_fwd(t, ACT_TANH), ACT_TANH) : act_slope_bwd(t, slope);
}


constexpr int kBnMaxBlocks = 2048;

__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float qb) {
  if (nb > 0.f) {
    const float nt = n + nb, d = mb - mean;
    mean += d * (nb / nt);
    m2 += qb + d * d * (n * nb / nt);
    n = nt;
  }
}

// ---- forward statistics (stand-alone form; the conv epilogue normally provides the partials) -----------
// thread = (column quad, row lane); part[blk][c] = (n, mean, M2).  Requires (C/4) | 256.
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ y, float* __restrict__ part,
                                                               int R, int C, int rows_per_block) {
  __shared__ float sm[256 * 9];
  const int tid = threadIdx.x;
  const int Q = C / 4;
  const int qpp = Q < 256 ? Q : 256;   // quads per pass
  const int rl = 256 / qpp;
  const int ql = tid % qpp, rlane = tid / qpp;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
  for (int q0 = 0; q0 < Q; q0 += qpp) {
    const int col = 4 * (q0 + ql);
    float n = 0.f;
    f32x4 mean = {0.f, 0.f, 0.f, 0.f}, m2 = {0.f, 0.f, 0.f, 0.f};
    if (r0 + rlane < r1) {
      const f32x4 shift = *reinterpret_cast<const f32x4*>(y + (long)(r0 + rlane) * C + col);
      f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
      int cnt = 0;
#pragma unroll 4
      for (int r = r0 + rlane; r < r1; r += rl) {
        f32x4 v = *reinterpret_cast<const f32x4*>(y + (long)r * C + col);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float d = v[k] - shift[k];
          s1[k] += d;
          s2[k] += d * d;
        }
        ++cnt;
      }
      n = (float)cnt;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        mean[k] = shift[k] + s1[k] / n;
        float q = s2[k] - s1[k] * s1[k] / n;
        m2[k] = q > 0.f ? q : 0.f;
      }
    }
    __syncthreads();
    sm[tid * 9] = n;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sm[tid * 9 + 1 + k] = mean[k];
      sm[tid * 9 + 5 + k] = m2[k];
    }
    __syncthreads();
    if (rlane == 0) {
      for (int l = 1; l < rl; ++l) {
        const float* o = &sm[(l * qpp + ql) * 9];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float nn = n, mk = mean[k], qk = m2[k];
          chan_merge(nn, mk, qk, o[0], o[1 + k], o[5 + k]);
          mean[k] = mk;
          m2[k] = qk;
          if (k == 3) n = nn;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float* p = part + ((long)blockIdx.x * C + col + k) * 3;
        p[0] = n; p[1] = mean[k]; p[2] = m2[k];
      }
    }
  }
}

// one block (256 partial-lanes) per channel: merges `nparts` (n, mean, M2) triples with Chan's formula
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nparts, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float momentum, float eps, float* __restrict__ save_mean,
                                                          float* __restrict__ save_invstd, float* __restrict__ scale,
                                                          float* __restrict__ shift, long long* __restrict__ num_batches_tracked) {
  __shared__ float sm[256 * 3];
  const int tid = threadIdx.x;
  const int c = blockIdx.x;
  if (c == 0 && tid == 0 && num_batches_tracked != nullptr) num_batches_tracked[0] += 1;   // nn.BatchNorm2d bookkeeping
  // what the last lines need, requested with the partial rows: behind the merge they were two more memory round trips
  const float gm = gamma[c], bt = beta[c];
  const float rm0 = running_mean != nullptr ? running_mean[c] : 0.f, rv0 = running_mean != nullptr ? running_var[c] : 0.f;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  // 8 rows per step: the loads go out together, only the (division-carrying) merges are serial
  for (int b0 = tid; b0 < nparts; b0 += 256 * 8) {
    float pn[8], pm[8], pq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + 256 * u;
      const float* p = part + ((long)(b < nparts ? b : b0) * C + c) * 3;
      pn[u] = b < nparts ? p[0] : 0.f;
      pm[u] = p[1];
      pq[u] = p[2];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) chan_merge(n, mean, m2, pn[u], pm[u], pq[u]);
  }
  sm[tid * 3] = n; sm[tid * 3 + 1] = mean; sm[tid * 3 + 2] = m2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      const float* o = &sm[(tid + s) * 3];
      chan_merge(n, mean, m2, o[0], o[1], o[2]);
      sm[tid * 3] = n; sm[tid * 3 + 1] = mean; sm[tid * 3 + 2] = m2;
    }
    __syncthreads();
  }
  if (tid == 0) {
    const float var = m2 / n;
    const float invstd = 1.0f / sqrtf(var + eps);
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    const float sc = gm * invstd;
    scale[c] = sc;
    shift[c] = bt - mean * sc;
    if (running_mean != nullptr) {
      const float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
      running_mean[c] = (1.f - momentum) * rm0 + momentum * mean;
      running_var[c] = (1.f - momentum) * rv0 + momentum * unbiased;
    }
  }
}

// Finalize + apply in ONE launch for layers with few partial rows (<= 128: the small deep layers, whose statistics come from
// bn_stats_partial_kernel).  Workgroup (channel block of 32, row range): it merges the partial rows of ITS 32 channels itself
// (24 KB from L2 at 64 rows -- every row range of a channel block repeats that merge, the first one also writes the statistics
// and updates the running ones), then streams its rows: 128 contiguous bytes per row, eight threads per row.  One dependent
// launch less per layer (~5 us in a replayed graph); the merge order is fixed (row lanes, then lanes): bit-reproducible.
__global__ __launch_bounds__(256) void bn_finalize_apply_kernel(const float* __restrict__ y, const float* __restrict__ part, int nparts,
                                                                int R, int C, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ running_mean,
                                                                float* __restrict__ running_var, float momentum, float eps,
                                                                float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                                float* __restrict__ scale, float* __restrict__ shift,
                                                                long long* __restrict__ num_batches_tracked, float* __restrict__ out,
                                                                int act) {
  __shared__ float sm[256 * 3];
  __shared__ __attribute__((aligned(16))) float sSc[32], sSh[32];
  const int tid = threadIdx.x, cb = blockIdx.x, rs = blockIdx.y;
  const int c0 = cb * 32, ch = tid & 31, rl = tid >> 5;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int b0 = rl; b0 < nparts; b0 += 8 * 4) {   // four rows per lane in flight
    float pn[4], pm[4], pq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = b0 + 8 * u;
      const float* p = part + ((long)(b < nparts ? b : b0) * C + c0 + ch) * 3;
      pn[u] = b < nparts ? p[0] : 0.f;
      pm[u] = p[1];
      pq[u] = p[2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) chan_merge(n, mean, m2, pn[u], pm[u], pq[u]);
  }
  sm[tid * 3] = n; sm[tid * 3 + 1] = mean; sm[tid * 3 + 2] = m2;
  __syncthreads();
  if (rl == 0) {
    for (int l = 1; l < 8; ++l) {
      const float* o = &sm[(l * 32 + ch) * 3];
      chan_merge(n, mean, m2, o[0], o[1], o[2]);
    }
    const int c = c0 + ch;
    const float var = m2 / n;
    const float invstd = 1.0f / sqrtf(var + eps);
    const float sc = gamma[c] * invstd, sh = beta[c] - mean * sc;
    sSc[ch] = sc;
    sSh[ch] = sh;
    if (rs == 0) {
      save_mean[c] = mean;
      save_invstd[c] = invstd;
      scale[c] = sc;
      shift[c] = sh;
      if (running_mean != nullptr) {
        const float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
      }
      if (c == 0 && num_batches_tracked != nullptr) num_batches_tracked[0] += 1;   // nn.BatchNorm2d bookkeeping
    }
  }
  __syncthreads();
  const int q = tid & 7;
  const f32x4 sc4 = *reinterpret_cast<const f32x4*>(&sSc[4 * q]), sh4 = *reinterpret_cast<const f32x4*>(&sSh[4 * q]);
  const int rows_per = (R + gridDim.y - 1) / gridDim.y;
  const int r0 = rs * rows_per, r1 = r0 + rows_per < R ? r0 + rows_per : R;
  for (int r = r0 + (tid >> 3); r < r1; r += 64) {
    const long i0 = ((long)r * C + c0) / 4 + q, i1 = ((long)(r + 32) * C + c0) / 4 + q;
    const bool two = r + 32 < r1;
    const f32x4 v0 = reinterpret_cast<const f32x4*>(y)[i0];
    const f32x4 v1 = two ? reinterpret_cast<const f32x4*>(y)[i1] : v0;
    f32x4 o0, o1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o0[k] = act_fwd(v0[k] * sc4[k] + sh4[k], act);
      o1[k] = act_fwd(v1[k] * sc4[k] + sh4[k], act);
    }
    reinterpret_cast<f32x4*>(out)[i0] = o0;
    if (two) reinterpret_cast<f32x4*>(out)[i1] = o1;
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
  if (1024 % C == 0) {
    // the grid stride (a multiple of 256 float4) is a multiple of C: a thread keeps its channel quad for the whole
    // launch -> coefficients in registers, no 64-bit modulo per element, two float4 in flight per iteration
    const int c = (int)((threadIdx.x * 4) % C);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c);
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
      const f32x4 v0 = reinterpret_cast<const f32x4*>(y)[i], v1 = reinterpret_cast<const f32x4*>(y)[i + stride];
      f32x4 o0, o1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o0[k] = act_fwd(v0[k] * sc[k] + sh[k], act);
        o1[k] = act_fwd(v1[k] * sc[k] + sh[k], act);
      }
      reinterpret_cast<f32x4*>(out)[i] = o0;
      reinterpret_cast<f32x4*>(out)[i + stride] = o1;
    }
    if (i < n4) {
      const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = act_fwd(v[k] * sc[k] + sh[k], act);
      reinterpret_cast<f32x4*>(out)[i] = o;
    }
    return;
  }
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
// part[blk][c] = (sum g_bn, sum g_bn*xhat); thread = (column quad, row lane).  Requires (C/4) | 256.
// The activation derivative is taken from the sign of z = gamma*invstd*(y-mean)+beta (same sign as the saved
// output a, so `a` need not be re-read: 8 instead of 12 bytes per element).
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ ga, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ y, const float* __restrict__ save_mean,
                                                             const float* __restrict__ save_invstd, float* __restrict__ part,
                                                             int R, int C, int rows_per_block, int act) {
  __shared__ float sm[256 * 8];
  const int tid = threadIdx.x;
  const int Q = C / 4;
  const int qpp = Q < 256 ? Q : 256;
  const int rl = 256 / qpp;
  const int ql = tid % qpp, rlane = tid / qpp;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
  for (int q0 = 0; q0 < Q; q0 += qpp) {
    const int col = 4 * (q0 + ql);
    const f32x4 mean = *reinterpret_cast<const f32x4*>(save_mean + col);
    const f32x4 invstd = *reinterpret_cast<const f32x4*>(save_invstd + col);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + col);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + col);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int r = r0 + rlane; r < r1; r += rl) {
      const long idx = (long)r * C + col;
      f32x4 g = *reinterpret_cast<const f32x4*>(ga + idx);
      f32x4 yv = *reinterpret_cast<const f32x4*>(y + idx);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (yv[k] - mean[k]) * invstd[k];
        float gb = g[k] * act_dt(gm[k] * xh + bt[k], act, act_slope(act));
        s1[k] += gb;
        s2[k] += gb * xh;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sm[tid * 8 + k] = s1[k];
      sm[tid * 8 + 4 + k] = s2[k];
    }
    __syncthreads();
    if (rlane == 0) {
      for (int l = 1; l < rl; ++l) {
        const float* o = &sm[(l * qpp + ql) * 8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s1[k] += o[k];
          s2[k] += o[4 + k];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        part[((long)blockIdx.x * C + col + k) * 2 + 0] = s1[k];
        part[((long)blockIdx.x * C + col + k) * 2 + 1] = s2[k];
      }
    }
  }
}

// one block per channel: dgamma/dbeta (accumulate flag) and the per-channel coefficients of g_y = k1*g_bn + k2*y + k3
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblocks, int C, float R,
                                                              const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                              const float* __restrict__ save_invstd, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate,
                                                              const float* __restrict__ beta,
                                                              float* __restrict__ coef /* [5][C]: k1,k2,k3,scale,shift */) {
  __shared__ double sm[8];
  const int tid = threadIdx.x;
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
  for (int b = tid; b < nblocks; b += 256) {
    s1 += (double)part[((long)b * C + c) * 2 + 0];
    s2 += (double)part[((long)b * C + c) * 2 + 1];
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if ((tid & 63) == 0) { sm[(tid >> 6) * 2] = s1; sm[(tid >> 6) * 2 + 1] = s2; }
  __syncthreads();
  if (tid == 0) {
    s1 = (sm[0] + sm[2]) + (sm[4] + sm[6]);
    s2 = (sm[1] + sm[3]) + (sm[5] + sm[7]);
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
    coef[3 * C + c] = k1;                       // forward scale = gamma*invstd
    coef[4 * C + c] = beta[c] - mean * k1;      // forward shift
  }
}

// Backward finalize + apply in ONE launch where the per-tile sums are few rows (the deep layers): workgroup (channel block of
// 32, row range) sums the rows of its channels itself (double, fixed order), forms k1, k2, k3 and the forward scale / shift,
// the first row range commits d gamma / d beta, then every workgroup streams its rows of g_y.  The consumer's finishing launch
// then carries no finalize, and its slab reduction can leave the backward chain (ctvae_defer_*).
__global__ __launch_bounds__(256) void bn_bwd_finalize_apply_kernel(const float* __restrict__ ga, const float* __restrict__ y,
                                                                    const float* __restrict__ part, int nblocks, int R, int C,
                                                                    const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                                    const float* __restrict__ save_invstd, const float* __restrict__ beta,
                                                                    int act, float* __restrict__ gy, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, int accumulate) {
  __shared__ double smd[256 * 2];
  __shared__ __attribute__((aligned(16))) float sK[5][32];
  const int tid = threadIdx.x, cb = blockIdx.x, rs = blockIdx.y;
  const int c0 = cb * 32, ch = tid & 31, rl = tid >> 5;
  double s1 = 0.0, s2 = 0.0;
  for (int b = rl; b < nblocks; b += 8) {
    s1 += (double)part[((long)b * C + c0 + ch) * 2 + 0];
    s2 += (double)part[((long)b * C + c0 + ch) * 2 + 1];
  }
  smd[tid * 2] = s1; smd[tid * 2 + 1] = s2;
  __syncthreads();
  if (rl == 0) {
    for (int l = 1; l < 8; ++l) {
      s1 += smd[(l * 32 + ch) * 2];
      s2 += smd[(l * 32 + ch) * 2 + 1];
    }
    const int c = c0 + ch;
    const float db = (float)s1, dg = (float)s2;
    const float invstd = save_invstd[c], mean = save_mean[c];
    const float k1 = gamma[c] * invstd;
    const float k2 = -k1 * dg / (float)R * invstd;
    const float k3 = -k1 * db / (float)R - k2 * mean;
    sK[0][ch] = k1; sK[1][ch] = k2; sK[2][ch] = k3; sK[3][ch] = k1; sK[4][ch] = beta[c] - mean * k1;
    if (rs == 0) {
      dgamma[c] = (accumulate ? dgamma[c] : 0.f) + dg;
      dbeta[c] = (accumulate ? dbeta[c] : 0.f) + db;
    }
  }
  __syncthreads();
  const int q = tid & 7;
  const f32x4 k1 = *reinterpret_cast<const f32x4*>(&sK[0][4 * q]), k2 = *reinterpret_cast<const f32x4*>(&sK[1][4 * q]);
  const f32x4 k3 = *reinterpret_cast<const f32x4*>(&sK[2][4 * q]), sc = *reinterpret_cast<const f32x4*>(&sK[3][4 * q]);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(&sK[4][4 * q]);
  const int rows_per = (R + gridDim.y - 1) / gridDim.y;
  const int r0 = rs * rows_per, r1 = r0 + rows_per < R ? r0 + rows_per : R;
  for (int r = r0 + (tid >> 3); r < r1; r += 32) {
    const long i = ((long)r * C + c0) / 4 + q;
    const f32x4 g = reinterpret_cast<const f32x4*>(ga)[i], yv = reinterpret_cast<const f32x4*>(y)[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = k1[k] * (g[k] * act_dt(yv[k] * sc[k] + sh[k], act, act_slope(act))) + k2[k] * yv[k] + k3[k];
    reinterpret_cast<f32x4*>(gy)[i] = o;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_job_kernel(BnFinJob j) {
  __shared__ double sm[8];
  bn_bwd_finalize_body(j, blockIdx.x, sm);
}

// dgamma != nullptr: coef carries rows 5, 6 = this pass's d gamma / d beta (finalize rider of ctvae_conv_backward); block 0
// commits them to the parameter gradients
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ ga, const float* __restrict__ y,
                                                           const float* __restrict__ coef, float* __restrict__ gy, long n4,
                                                           int C, int act, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           int accumulate) {
  const long stride = (long)gridDim.x * 256;
  if (dgamma != nullptr && blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      dgamma[c] = (accumulate ? dgamma[c] : 0.f) + coef[5 * C + c];
      dbeta[c] = (accumulate ? dbeta[c] : 0.f) + coef[6 * C + c];
    }
  }
  if (1024 % C == 0) {   // channel quad fixed per thread (see bn_apply_act_kernel)
    const int c = (int)((threadIdx.x * 4) % C);
    const f32x4 k1 = *reinterpret_cast<const f32x4*>(coef + c);
    const f32x4 k2 = *reinterpret_cast<const f32x4*>(coef + C + c);
    const f32x4 k3 = *reinterpret_cast<const f32x4*>(coef + 2 * C + c);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(coef + 3 * C + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(coef + 4 * C + c);
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
      const f32x4 g0 = reinterpret_cast<const f32x4*>(ga)[i], g1 = reinterpret_cast<const f32x4*>(ga)[i + stride];
      const f32x4 y0 = reinterpret_cast<const f32x4*>(y)[i], y1 = reinterpret_cast<const f32x4*>(y)[i + stride];
      f32x4 o0, o1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o0[k] = k1[k] * (g0[k] * act_dt(y0[k] * sc[k] + sh[k], act, act_slope(act))) + k2[k] * y0[k] + k3[k];
        o1[k] = k1[k] * (g1[k] * act_dt(y1[k] * sc[k] + sh[k], act, act_slope(act))) + k2[k] * y1[k] + k3[k];
      }
      reinterpret_cast<f32x4*>(gy)[i] = o0;
      reinterpret_cast<f32x4*>(gy)[i + stride] = o1;
    }
    if (i < n4) {
      const f32x4 g = reinterpret_cast<const f32x4*>(ga)[i];
      const f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        o[k] = k1[k] * (g[k] * act_dt(yv[k] * sc[k] + sh[k], act, act_slope(act))) + k2[k] * yv[k] + k3[k];
      reinterpret_cast<f32x4*>(gy)[i] = o;
    }
    return;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const int c = (int)((i * 4) % C);
    f32x4 g = reinterpret_cast<const f32x4*>(ga)[i];
    f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
    f32x4 k1 = *reinterpret_cast<const f32x4*>(coef + c);
    f32x4 k2 = *reinterpret_cast<const f32x4*>(coef + C + c);
    f32x4 k3 = *reinterpret_cast<const f32x4*>(coef + 2 * C + c);
    f32x4 sc = *reinterpret_cast<const f32x4*>(coef + 3 * C + c);
    f32x4 sh = *reinterpret_cast<const f32x4*>(coef + 4 * C + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = k1[k] * (g[k] * act_dt(yv[k] * sc[k] + sh[k], act, act_slope(act))) + k2[k] * yv[k] + k3[k];
    reinterpret_cast<f32x4*>(gy)[i] = o;
  }
}

// ---- small layers: the whole tail of a split-K convolution in ONE launch (channel owners) ------------------------------
// At small batch the deep layers' GEMMs are split along K to fill the chip, and what followed was a chain of 5 us launches:
// split-K sum -> statistics -> finalize(+apply) forward, split-K sum -> backward sums -> finalize -> apply backward.  Here the
// tile kernel leaves its raw slices CHANNEL-MAJOR (SplitKRaw: part[S][C][R]), and a workgroup that owns 4 channels and ALL R
// rows does the rest: it sums the slices of its channels (contiguous runs along r: coalesced 16-byte loads), holds the result
// in registers (2 or 4 channels x 4 rows per thread), takes the exact two-pass statistics (forward) or the two backward sums
// with block reductions in a fixed order, and writes the NHWC rows of its 4 channels as 16-byte stores.  No partial rows,
// no second pass over memory, bit-reproducible.  The strided 16-byte row accesses touch a line per lane, which is why this
// form is kept to tensors of a few MB (bn_fused_ok): measured on the 64 x 64 x 3, bs = 64 step.
CTVAE_PHASE_DECL(bnf)
#define BNF_PH(I) CTVAE_PH(bnf, (TPB == 1024 ? 0 : (TPB == 256 ? 1 : 2)), I)
#define BNB_PH(I) CTVAE_PH(bnf, 3, I)   /* backward, any size: the last launch wins */

template <int TPB, int NV>
__device__ __forceinline__ void block_sum_n(float (&v)[NV], float* sm /* [TPB/64][NV] */) {
  constexpr int NW = TPB / 64;
#pragma unroll
  for (int c = 0; c < NV; ++c) v[c] = wave_sum_full(v[c]);
  if constexpr (NW > 1) {
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int c = 0; c < NV; ++c) sm[w * NV + c] = v[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < NW; ++i) t += sm[i * NV + c];
      v[c] = t;
    }
  }
}

// v[c] = sum of the S slices of four consecutive rows of channel c (c < CPW): SU slices x CPW channels of loads go out together
// (a slice index beyond S re-reads the last slice with weight 0: the loop body stays branch-free, the summation order fixed)
template <int CPW, int SU>
__device__ __forceinline__ void slice_sum4(f32x4 (&v)[CPW], const float* __restrict__ base, long chan_stride, long slice_stride, int S) {
#pragma unroll
  for (int c = 0; c < CPW; ++c) v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int z0 = 0; z0 < S; z0 += SU) {
    f32x4 t[SU][CPW];
#pragma unroll
    for (int u = 0; u < SU; ++u)
#pragma unroll
      for (int c = 0; c < CPW; ++c)
        t[u][c] = *reinterpret_cast<const f32x4*>(base + (long)(z0 + u < S ? z0 + u : S - 1) * slice_stride + c * chan_stride);
#pragma unroll
    for (int u = 0; u < SU; ++u)
#pragma unroll
      for (int c = 0; c < CPW; ++c) v[c] += z0 + u < S ? t[u][c] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// Channel group of a workgroup.  The groups of one 128-byte line of an NHWC row (32 channels) each write / read 8 or 16 bytes
// of it: dealt to the XCDs in launch order (id % 8) they would leave pieces of every line in eight L2s -- partial-line
// write-backs and eight fills per line.  Whole lines per XCD instead: id = (line % 8) + 8 * (slot + slots * (line / 8)).
__device__ __forceinline__ int fused_group(int id, int C, int cpw) {
  const int slots = 32 / cpw, lines = C / 32;
  if (lines % 8 != 0) return id;
  const int x = id & 7, t = id >> 3, slot = t % slots, jh = t / slots;
  return (jh * 8 + x) * slots + slot;
}

// workgroup = CPW channels x all R rows (R <= 4 * TPB); thread = four consecutive rows of each of its CPW channels
template <int TPB, int CPW, int SU>
__global__ __launch_bounds__(TPB) void bn_fused_fwd_kernel(const BnFusedFwd p) {
  typedef float rowv __attribute__((ext_vector_type(CPW)));
  __shared__ float sm[(TPB / 64) * CPW];
  BNF_PH(0);
  const int tid = threadIdx.x, c0 = fused_group(blockIdx.x, p.C, CPW) * CPW, R = p.R, C = p.C;
  const int r4 = 4 * tid;
  const bool live = r4 < R;
  // everything the tail needs is requested before the slices: nothing below waits for a second memory round trip
  float gmv[CPW], btv[CPW], bsv[CPW], rmv[CPW], rvv[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    gmv[c] = p.gamma[c0 + c];
    btv[c] = p.beta[c0 + c];
    bsv[c] = p.bias != nullptr ? p.bias[c0 + c] : 0.f;
    rmv[c] = p.running_mean != nullptr ? p.running_mean[c0 + c] : 0.f;
    rvv[c] = p.running_mean != nullptr ? p.running_var[c0 + c] : 0.f;
  }
  const long long nbt0 = (blockIdx.x == 0 && p.nbt != nullptr) ? p.nbt[0] : 0;
  f32x4 v[CPW];
  BNF_PH(1);
  slice_sum4<CPW, SU>(v, p.part + (long)c0 * R + (live ? r4 : 0), R, (long)C * R, p.S);
#pragma unroll
  for (int c = 0; c < CPW; ++c) v[c] = live ? v[c] + bsv[c] : f32x4{0.f, 0.f, 0.f, 0.f};
  BNF_PH(2);
  float s[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) s[c] = (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);   // rows beyond R hold zeros
  block_sum_n<TPB, CPW>(s, sm);
  BNF_PH(3);
  float mean[CPW], q[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    mean[c] = s[c] / (float)R;
    float t = 0.f;
    if (live) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[c][e] - mean[c];
        t += d * d;
      }
    }
    q[c] = t;
  }
  block_sum_n<TPB, CPW>(q, sm);
  BNF_PH(4);
  float sc[CPW], sh[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const float var = q[c] / (float)R;
    const float invstd = 1.0f / sqrtf(var + p.eps);
    sc[c] = gmv[c] * invstd;
    sh[c] = btv[c] - mean[c] * sc[c];
    if (tid == 0) {
      p.save_mean[c0 + c] = mean[c];
      p.save_invstd[c0 + c] = invstd;
      if (p.scale_shift != nullptr) {
        p.scale_shift[c0 + c] = sc[c];
        p.scale_shift[C + c0 + c] = sh[c];
      }
      if (p.running_mean != nullptr) {
        const float unbiased = R > 1 ? q[c] / (float)(R - 1) : var;
        p.running_mean[c0 + c] = (1.f - p.momentum) * rmv[c] + p.momentum * mean[c];
        p.running_var[c0 + c] = (1.f - p.momentum) * rvv[c] + p.momentum * unbiased;
      }
    }
  }
  if (tid == 0 && blockIdx.x == 0 && p.nbt != nullptr) p.nbt[0] = nbt0 + 1;   // nn.BatchNorm2d bookkeeping
  if (live) {
    const bool atanh_ = p.act == ACT_TANH;
    const float asl = act_slope(p.act);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      rowv yr, ar;
#pragma unroll
      for (int c = 0; c < CPW; ++c) yr[c] = v[c][e];
      if (atanh_) {
#pragma unroll
        for (int c = 0; c < CPW; ++c) ar[c] = act_fwd(yr[c] * sc[c] + sh[c], ACT_TANH);
      } else {
#pragma unroll
        for (int c = 0; c < CPW; ++c) ar[c] = act_slope_fwd(yr[c] * sc[c] + sh[c], asl);
      }
      const long o = (long)row_map_pixel(p.rows, r4 + e) * C + c0;
      *reinterpret_cast<rowv*>(p.y + o) = yr;
      if (p.a != nullptr) *reinterpret_cast<rowv*>(p.a + o) = ar;
    }
  }
#ifdef CTVAE_PHASES
  BNF_PH(5);
  __builtin_amdgcn_s_waitcnt(0);
  BNF_PH(6);
#endif
}

template <int TPB, int CPW, int SU>
__global__ __launch_bounds__(TPB) void bn_fused_bwd_kernel(const BnFusedBwd p) {
  typedef float rowv __attribute__((ext_vector_type(CPW)));
  __shared__ float sm[(TPB / 64) * 2 * CPW];
  BNB_PH(0);
  const int tid = threadIdx.x, c0 = fused_group(blockIdx.x, p.C, CPW) * CPW, R = p.R, C = p.C;
  const int r4 = 4 * tid;
  const bool live = r4 < R;
  rowv yv[4];        // [row] along channels
  f32x4 g[CPW];      // [channel] along rows: g_a, then g' = g_a * act'
  long po[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) po[e] = live ? (long)row_map_pixel(p.rows, r4 + e) * C + c0 : c0;
  // UNCONDITIONAL loads (threads beyond R read row 0; their g is zeroed below, so the value never counts): under `if (live)` each
  // load is a branch and the compiler waits for it before the next one -- four memory round trips in a row at the top of the kernel
#pragma unroll
  for (int e = 0; e < 4; ++e) yv[e] = *reinterpret_cast<const rowv*>(p.y + po[e]);
  float mean[CPW], invstd[CPW], gm[CPW], bt[CPW], s12[2 * CPW], dg0[CPW], db0[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    mean[c] = p.mean[c0 + c]; invstd[c] = p.invstd[c0 + c]; gm[c] = p.gamma[c0 + c]; bt[c] = p.beta[c0 + c];
    dg0[c] = p.accumulate ? p.dgamma[c0 + c] : 0.f;   // requested with the rest: the commit below does not wait for memory
    db0[c] = p.accumulate ? p.dbeta[c0 + c] : 0.f;
  }
  BNB_PH(1);
  slice_sum4<CPW, SU>(g, p.part + (long)c0 * R + (live ? r4 : 0), R, (long)C * R, p.S);
  BNB_PH(2);
  const bool btanh_ = p.act == ACT_TANH;
  const float bsl_ = act_slope(p.act);
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    if (!live) g[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (yv[e][c] - mean[c]) * invstd[c];
      const float t = gm[c] * xh + bt[c];
      const float gp = g[c][e] * (btanh_ ? act_bwd_from_out(act_fwd(t, ACT_TANH), ACT_TANH) : act_slope_bwd(t, bsl_));   // rows beyond R: g = 0
      g[c][e] = gp;
      s1 += gp;
      s2 += gp * xh;
    }
    s12[c] = s1;
    s12[CPW + c] = s2;
  }
  BNB_PH(3);
  block_sum_n<TPB, 2 * CPW>(s12, sm);
  BNB_PH(4);
  float k1[CPW], k2[CPW], k3[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const float db = s12[c], dg = s12[CPW + c];
    k1[c] = gm[c] * invstd[c];
    k2[c] = -k1[c] * dg / (float)R * invstd[c];
    k3[c] = -k1[c] * db / (float)R - k2[c] * mean[c];
    if (tid == 0) {
      p.dgamma[c0 + c] = dg0[c] + dg;
      p.dbeta[c0 + c] = db0[c] + db;
    }
  }
  if (live) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      rowv o;
#pragma unroll
      for (int c = 0; c < CPW; ++c) o[c] = k1[c] * g[c][e] + k2[c] * yv[e][c] + k3[c];
      *reinterpret_cast<rowv*>(p.gy + po[e]) = o;
    }
  }
#ifdef CTVAE_PHASES
  BNB_PH(5);
  __builtin_amdgcn_s_waitcnt(0);
  BNB_PH(6);
#endif
}

// rows / channels the channel-owner kernels take (and the tensor size up to which their strided row accesses pay)
bool bn_fused_ok(int R, int C) {
  static const long max_elems = [] { const char* e = getenv("CTVAE_BN_FUSED_MAX"); return e ? atol(e) : (1L << 20); }();   // 0 = off
  return R > 0 && R % 4 == 0 && C % 4 == 0 && R <= 4096 && (long)R * C <= max_elems;
}

// threads = R / 4 rounded up to 64 / 256 / 1024; two channels per workgroup where four would leave most of the chip idle

static int fused_two_below() {
  static const int v = [] { const char* e = getenv("CTVAE_BN_FUSED_TWO"); return e ? atoi(e) : 256; }();   // diagnostic
  return v;
}

#define CTVAE_BN_FUSED_LAUNCH(KERNEL, P, ST)                                                                      \
  do {                                                                                                            \
    const bool two = (P).C < fused_two_below();                                                                   \
    const dim3 grid((P).C / (two ? 2 : 4));                                                                       \
    if ((P).R <= 256) {                                                                                           \
      if (two) hipLaunchKernelGGL((KERNEL<64, 2, 8>), grid, dim3(64), 0, ST, P);                                  \
      else hipLaunchKernelGGL((KERNEL<64, 4, 8>), grid, dim3(64), 0, ST, P);                                      \
    } else if ((P).R <= 1024) {                                                                                   \
      if (two) hipLaunchKernelGGL((KERNEL<256, 2, 8>), grid, dim3(256), 0, ST, P);                                \
      else hipLaunchKernelGGL((KERNEL<256, 4, 8>), grid, dim3(256), 0, ST, P);                                    \
    } else {                                                                                                      \
      if (two) hipLaunchKernelGGL((KERNEL<1024, 2, 4>), grid, dim3(1024), 0, ST, P);                              \
      else hipLaunchKernelGGL((KERNEL<1024, 4, 4>), grid, dim3(1024), 0, ST, P);                                  \
    }                                                                                                             \
  } while (0)

int launch_bn_fused_forward(const BnFusedFwd& p, hipStream_t st) {
  if (!bn_fused_ok(p.R, p.C) || p.S < 1) return kErrBadArg;
  ProfScope ps("bn_fused_fwd_kernel", st, 0.0, 4.0 * (double)p.R * p.C * (p.S + (p.a != nullptr ? 2 : 1)));
  CTVAE_BN_FUSED_LAUNCH(bn_fused_fwd_kernel, p, st);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_bn_fused_backward(const BnFusedBwd& p, hipStream_t st) {
  if (!bn_fused_ok(p.R, p.C) || p.S < 1) return kErrBadArg;
  ProfScope ps("bn_fused_bwd_kernel", st, 0.0, 4.0 * (double)p.R * p.C * (p.S + 2));
  CTVAE_BN_FUSED_LAUNCH(bn_fused_bwd_kernel, p, st);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

static bool bn_shape_ok(int R, int C) {
  if (C % 4 != 0 || R <= 0) return false;
  const int Q = C / 4;
  return Q >= 256 ? (Q % 256 == 0) : (256 % Q == 0);
}

static int stat_blocks(int R, int C, int* rows_per_block) {
  const int Q = C / 4;
  const int rl = Q >= 256 ? 1 : 256 / Q;
  int nb = R / (rl * 8);             // >= 8 vector loads per lane
  if (nb < 1) nb = 1;
  if (nb > kBnMaxBlocks) nb = kBnMaxBlocks;
  *rows_per_block = ceil_div(R, nb);
  return ceil_div(R, *rows_per_block);
}

// workspace layout: [part: max(kBnMaxBlocks, conv tiles) * C * 3][scale C][shift C][spare C]
size_t bn_workspace_floats(int C, int nparts) {
  const size_t parts = nparts > kBnMaxBlocks ? (size_t)nparts : (size_t)kBnMaxBlocks;
  return parts * C * 3 + 5 * (size_t)C;
}

// finalize (from `nparts` partial triples already in ws) or eval coefficients, then apply + activation
int launch_bn_finish_forward(const float* y, int R, int C, int nparts, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps, int training, int act,
                             float* out, float* save_mean, float* save_invstd, float* ws, long long* nbt, hipStream_t st,
                             float* coef_out) {
  const size_t parts = nparts > kBnMaxBlocks ? (size_t)nparts : (size_t)kBnMaxBlocks;
  float* scale = coef_out != nullptr ? coef_out : ws + parts * C * 3;   // coef_out [2][C]: kept by the caller (lazy apply)
  float* shift = scale + C;
  if (training && out != nullptr && nparts >= 1 && nparts <= 128 && C % 32 == 0) {
    static const int on = [] { const char* e = getenv("CTVAE_BN_FIN_APPLY"); return e ? atoi(e) : 1; }();   // diagnostic: 0 = two launches
    if (on) {
      int rsn = 512 / (C / 32);
      if (rsn > R / 32) rsn = R / 32;
      if (rsn < 1) rsn = 1;
      ProfScope ps("bn_finalize_apply_kernel", st, 0.0, 8.0 * (double)R * C + 12.0 * (double)nparts * C * rsn);
      hipLaunchKernelGGL(bn_finalize_apply_kernel, dim3(C / 32, rsn), dim3(256), 0, st, y, ws, nparts, R, C, gamma, beta, running_mean,
                         running_var, momentum, eps, save_mean, save_invstd, scale, shift, nbt, out, act);
      CTVAE_LAUNCH_CHECK();
      return 0;
    }
  }
  if (training) {
    ProfScope ps("bn_finalize_kernel", st, 0.0, 12.0 * (double)nparts * C);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, st, ws, nparts, C, gamma, beta, running_mean,
                       running_var, momentum, eps, save_mean, save_invstd, scale, shift, nbt);
    CTVAE_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, C, gamma, beta, running_mean,
                       running_var, eps, scale, shift);
    CTVAE_LAUNCH_CHECK();
  }
  if (out == nullptr) return 0;   // the consumer applies scale/shift + activation while loading y
  const long n4 = (long)R * C / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  ProfScope ps("bn_apply_act_kernel", st, 0.0, 8.0 * (double)R * C);
  hipLaunchKernelGGL(bn_apply_act_kernel, dim3((unsigned)blocks), dim3(256), 0, st, y, scale, shift, out, n4, C, act);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_bn_forward(const float* y, int R, int C, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float momentum, float eps, int training, int act, float* out,
                      float* save_mean, float* save_invstd, float* ws, size_t ws_bytes, long long* nbt, hipStream_t st,
                      float* coef_out) {
  if (!bn_shape_ok(R, C)) return kErrBadArg;
  if (ws_bytes / sizeof(float) < bn_workspace_floats(C, 0)) return kErrWorkspace;
  int nb = 0;
  if (training) {
    int rpb;
    nb = stat_blocks(R, C, &rpb);
    ProfScope ps("bn_stats_partial_kernel", st, 0.0, 4.0 * (double)R * C);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nb), dim3(256), 0, st, y, ws, R, C, rpb);
    CTVAE_LAUNCH_CHECK();
  }
  return launch_bn_finish_forward(y, R, C, nb, gamma, beta, running_mean, running_var, momentum, eps, training, act, out,
                                  save_mean, save_invstd, ws, nbt, st, coef_out);
}

int launch_bn_backward(const float* ga, const float* beta, const float* y, int R, int C, const float* gamma,
                       const float* save_mean, const float* save_invstd, int act, float* gy, float* dgamma,
                       float* dbeta, int accumulate, float* ws, size_t ws_bytes, const float* part_in, int part_rows,
                       hipStream_t st, float* coef_out, const float* coef_in) {
  if (!bn_shape_ok(R, C)) return kErrBadArg;
  if (coef_in != nullptr) {   // finalized already (ctvae_conv_backward bn_coef_out): apply + commit d gamma / d beta
    if (gy == nullptr) return kErrBadArg;
    ProfScope ps("bn_bwd_apply_kernel", st, 0.0, 12.0 * (double)R * C);
    const long n4 = (long)R * C / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ga, y, coef_in, gy, n4, C, act, dgamma,
                       dbeta, accumulate);
    CTVAE_LAUNCH_CHECK();
    return 0;
  }
  if (ws_bytes / sizeof(float) < bn_workspace_floats(C, 0)) return kErrWorkspace;
  const float* part = ws;
  float* coef = coef_out != nullptr ? coef_out : ws + (size_t)kBnMaxBlocks * C * 3;   // [5][C]
  int rpb;
  int nb = stat_blocks(R, C, &rpb);
  if (part_in != nullptr && part_rows > 0) {   // sums already emitted per tile by the dgrad that produced ga
    part = part_in;
    nb = part_rows;
  } else {
    ProfScope ps("bn_bwd_partial_kernel", st, 0.0, 8.0 * (double)R * C);
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nb), dim3(256), 0, st, ga, gamma, beta, y, save_mean, save_invstd, ws, R,
                       C, rpb, act);
  }
  CTVAE_LAUNCH_CHECK();
  if (part_in != nullptr && gy != nullptr && coef_out == nullptr && nb <= 256 && C % 32 == 0) {
    static const int on = [] { const char* e = getenv("CTVAE_BN_BWD_FIN_APPLY"); return e ? atoi(e) : 1; }();   // diagnostic
    if (on) {
      int rsn = 512 / (C / 32);
      if (rsn > R / 32) rsn = R / 32;
      if (rsn < 1) rsn = 1;
      ProfScope ps("bn_bwd_finalize_apply_kernel", st, 0.0, 12.0 * (double)R * C + 8.0 * (double)nb * C * rsn);
      hipLaunchKernelGGL(bn_bwd_finalize_apply_kernel, dim3(C / 32, rsn), dim3(256), 0, st, ga, y, part, nb, R, C, gamma, save_mean,
                         save_invstd, beta, act, gy, dgamma, dbeta, accumulate);
      CTVAE_LAUNCH_CHECK();
      return 0;
    }
  }
  {
    ProfScope ps("bn_bwd_finalize_kernel", st, 0.0, 8.0 * (double)nb * C);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, part, nb, C, (float)R, gamma,
                       save_mean, save_invstd, dgamma, dbeta, accumulate, beta, coef);
  }
  CTVAE_LAUNCH_CHECK();
  if (gy == nullptr) return 0;   // the consumer (weight-gradient kernel) applies the coefficients on load
  ProfScope ps("bn_bwd_apply_kernel", st, 0.0, 12.0 * (double)R * C);
  const long n4 = (long)R * C / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ga, y, coef, gy, n4, C, act, (float*)nullptr,
                     (float*)nullptr, 0);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_bn_bwd_finalize_job(const BnFinJob& j, hipStream_t st) {
  ProfScope ps("bn_bwd_finalize_kernel", st, 0.0, 8.0 * (double)j.nblocks * j.C);
  hipLaunchKernelGGL(bn_bwd_finalize_job_kernel, dim3(j.C), dim3(256), 0, st, j);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
