// Finishing passes behind the GEMM launches: the split-K sum + epilogue of the tile kernels (tapgemm.hip) and the slab
// reduction of the weight-gradient kernels (wgrad.hip).  Shared here so that ONE launch can serve both when a layer's
// backward pass was issued as a pair (ctvae_conv_backward): reduce_partials_kernel takes the split-K job as extra blocks.
#pragma once
#include "common.hpp"

namespace ctvae {

struct ReduceJob {
  const float* part;
  float* dst;
  long n;
  int S;
  long stride;
  int small;   // block shape
  int vec;     // float4 lanes (n % 4 == 0 and stride % 4 == 0)
};

// S[i] = epi(sum_z part[z][i]) for the split-K path (N % 4 == 0); nblk blocks of 256 threads walk the n4 float4 elements
struct SplitKJob {
  const float* part;
  int splitk;
  long stride;
  const float* bias;
  const float* add;
  const float* mask;
  int mask_act, act;
  float* S;
  long n4;
  int N;
  int nblk;
};

__device__ __forceinline__ void splitk_finish_body(const SplitKJob& k, int blk) {
  const long gs = (long)k.nblk * 256;
  for (long i = (long)blk * 256 + threadIdx.x; i < k.n4; i += gs) {
    f32x4 v = reinterpret_cast<const f32x4*>(k.part)[i];
    for (int z = 1; z < k.splitk; ++z) {
      f32x4 t = reinterpret_cast<const f32x4*>(k.part + z * k.stride)[i];
      v += t;
    }
    const int col = (int)((i * 4) % k.N);
    if (k.bias != nullptr) v += *reinterpret_cast<const f32x4*>(k.bias + col);
    if (k.add != nullptr) v += reinterpret_cast<const f32x4*>(k.add)[i];
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = k.act == ACT_TANH ? act_fwd(v[q], ACT_TANH) : act_slope_fwd(v[q], act_slope(k.act));
    if (k.mask != nullptr) {
      f32x4 m = reinterpret_cast<const f32x4*>(k.mask)[i];
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] *= k.mask_act == ACT_TANH ? 1.f - m[q] * m[q] : (m[q] > 0.f ? 1.f : act_slope(k.mask_act));
    }
    reinterpret_cast<f32x4*>(k.S)[i] = o;
  }
}

// BatchNorm-backward finalize as a rider of the finishing launch (ctvae_conv_backward with bn_coef_out): the data gradient's
// tile epilogues left part[nblocks][C][2] = (sum g', sum g'*xhat); block c turns channel c's column into
//   coef[0..4][C] = k1, k2, k3 of g_y = k1*g' + k2*y + k3 and the forward scale / shift,
//   coef[5][C], coef[6][C] = d gamma, d beta of this pass (committed to the parameter gradients by the apply launch of
//   ctvae_bn_backward(coef_in), which is also where it is decided that these sums belong to the gradient that arrived).
struct BnFinJob {
  const float* part;
  int nblocks, C;
  float R;
  const float *gamma, *mean, *invstd, *beta;
  float* coef;   // [7][C]
  float *dgamma, *dbeta;   // non-null: commit here as well (no apply launch follows), under `accumulate`
  int accumulate;
};

__device__ __forceinline__ void bn_bwd_finalize_body(const BnFinJob& j, int c, double* sm /* [8] */) {
  const int tid = threadIdx.x, C = j.C;
  // the operands of the last lines, requested with the partial rows (behind the sums they were two more dependent round trips)
  const float invstd = j.invstd[c], mean = j.mean[c], gm = j.gamma[c], bt = j.beta[c];
  const float dg0 = (j.dgamma != nullptr && j.accumulate) ? j.dgamma[c] : 0.f, db0 = (j.dgamma != nullptr && j.accumulate) ? j.dbeta[c] : 0.f;
  double s1 = 0.0, s2 = 0.0;
  // eight rows of a thread in flight (a loop over `b += 256` with a tid-dependent trip count is one round trip per iteration);
  // rows beyond the table re-read the thread's first row with weight 0: same summation order as before
  for (int b0 = tid; b0 < j.nblocks; b0 += 256 * 8) {
    f32x2 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + 256 * u;
      t[u] = *reinterpret_cast<const f32x2*>(j.part + ((long)(b < j.nblocks ? b : b0) * C + c) * 2);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (b0 + 256 * u < j.nblocks) { s1 += (double)t[u][0]; s2 += (double)t[u][1]; }
    }
  }
  s1 = wave_sum_d_full(s1);
  s2 = wave_sum_d_full(s2);
  if ((tid & 63) == 0) { sm[(tid >> 6) * 2] = s1; sm[(tid >> 6) * 2 + 1] = s2; }
  __syncthreads();
  if (tid == 0) {
    s1 = (sm[0] + sm[2]) + (sm[4] + sm[6]);
    s2 = (sm[1] + sm[3]) + (sm[5] + sm[7]);
    const float db = (float)s1, dg = (float)s2;
    const float k1 = gm * invstd;
    const float k2 = -k1 * dg / j.R * invstd;
    const float k3 = -k1 * db / j.R - k2 * mean;
    j.coef[c] = k1;
    j.coef[C + c] = k2;
    j.coef[2 * C + c] = k3;
    j.coef[3 * C + c] = k1;
    j.coef[4 * C + c] = bt - mean * k1;
    j.coef[5 * C + c] = dg;
    j.coef[6 * C + c] = db;
    if (j.dgamma != nullptr) {
      j.dgamma[c] = dg0 + dg;
      j.dbeta[c] = db0 + db;
    }
  }
}

// Channel-owner tails of a split-K convolution next to a train-mode BatchNorm (bn.hip bn_fused_fwd_kernel / bn_fused_bwd_kernel):
// the slices arrive channel-major (SplitKRaw), everything behind the GEMM is one launch.
struct BnFusedFwd {
  const float* part;   // [S][C][R], rows in the producing launch's class-major order
  RowMap rows;         // ... -> pixel of y / a
  int S, R, C;
  const float *bias, *gamma, *beta;
  float *running_mean, *running_var;
  float momentum, eps;
  float *save_mean, *save_invstd, *scale_shift;   // scale_shift [2][C] may be null
  long long* nbt;
  float* y;            // [R][C] pre-BatchNorm output (the backward pass reads it)
  float* a;            // [R][C] act(BN(y)); null when the consumer applies scale/shift on load
  int act;
};


struct BnFusedBwd {
  const float* part;   // [S][C][R] slices of g_a, the gradient w.r.t. act(BN(y)), rows in the data gradient's class-major order
  RowMap rows;         // ... -> pixel of y / g_y
  int S, R, C;
  const float *y, *gamma, *beta, *mean, *invstd;
  int act;
  float* gy;           // [R][C]
  float *dgamma, *dbeta;
  int accumulate;
};


bool bn_fused_ok(int R, int C);
int launch_bn_fused_forward(const BnFusedFwd& p, hipStream_t st);
int launch_bn_fused_backward(const BnFusedBwd& p, hipStream_t st);

}  // namespace ctvae
