// Row / sample-wise pieces of CausalTransition (ct_mcq_vae.py) that were chains of 5-15 tiny torch launches each, forward and
// again in autograd's backward.  All are HBM / launch bound (a few MB per launch); the point is one launch instead of ten.
//
//   ct_reg_fwd/bwd_kernel        forward_action's regulariser (ct_mcq_vae.py:275, 314-323), one workgroup per sample:
//                                  beta * KL(softmax(u) || softmax(adj))  (adjacency_KL_loss: kl_div(log_softmax(adj), target,
//                                  'batchmean') with target = softmax(rand))
//                                + delta * mean_b ||graph_b||_F            (graph_size_loss)
//                                + epsilon * mean_b || prod_j (1 - adj_b[i,j]) ||_2   (positive_trial_loss)
//                                the product's gradient through exclusive prefix / suffix products in the wave (exact with
//                                zeros, no division)
//   ct_blend_softmax_fwd/bwd     _compute_y's tail (:226-228): softmax_d( y0 * (1 - mask) + y1 * mask ), one wave per node
//   ct_latent_ce_fwd/bwd         latent_CrossEntropy_loss (:306-311): cross_entropy(log(clamp(p, 1e-4)), target), one wave per node
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float wave_prod(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v *= __shfl_xor(v, o, 64);
  return v;
}

// block-wide (256 threads) reductions of two values at once; results valid in every thread
__device__ __forceinline__ void block_max2(float& a, float& b, float* sm /* >= 8 */) {
  a = wave_max(a);
  b = wave_max(b);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sm[w] = a; sm[4 + w] = b; }
  __syncthreads();
  a = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  b = fmaxf(fmaxf(sm[4], sm[5]), fmaxf(sm[6], sm[7]));
}

__device__ __forceinline__ void block_sum2(float& a, float& b, float* sm /* >= 8 */) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sm[w] = a; sm[4 + w] = b; }
  __syncthreads();
  a = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  b = (sm[4] + sm[5]) + (sm[6] + sm[7]);
}

constexpr int RN = 64;            // nodes
constexpr int RE = RN * RN / 256; // elements per thread (16): element e = tid + 256*i -> row (tid >> 6) + 4*i, column tid & 63

struct RegStats {                 // what the backward needs again (recomputed, not stored)
  float ma, lsa, mu, lsu;         // max / log-sum-exp of adj and of the uniform draws
};

__device__ __forceinline__ RegStats reg_softmax_stats(const float (&a)[RE], const float (&u)[RE], float* sm) {
  float ma = -INFINITY, mu = -INFINITY;
#pragma unroll
  for (int i = 0; i < RE; ++i) { ma = fmaxf(ma, a[i]); mu = fmaxf(mu, u[i]); }
  block_max2(ma, mu, sm);
  float sa = 0.f, su = 0.f;
#pragma unroll
  for (int i = 0; i < RE; ++i) { sa += __expf(a[i] - ma); su += __expf(u[i] - mu); }
  block_sum2(sa, su, sm);
  return RegStats{ma, __logf(sa), mu, __logf(su)};
}

// part [B][4] = {kl_b, ||graph_b||_F, ||prod||_2, ckl*kl_b + cgs*||graph_b||_F + cpt*||prod||_2}
__global__ __launch_bounds__(256) void ct_reg_fwd_kernel(const float* __restrict__ adj, const float* __restrict__ graph,
                                                        const float* __restrict__ uni, float* __restrict__ part, float ckl,
                                                        float cgs, float cpt) {
  __shared__ float sm[8];
  const int tid = threadIdx.x, b = blockIdx.x;
  const long o = (long)b * RN * RN;
  float a[RE], u[RE], g2 = 0.f, pp = 0.f;
#pragma unroll
  for (int i = 0; i < RE; ++i) {
    a[i] = adj[o + tid + 256 * i];
    u[i] = uni[o + tid + 256 * i];
    const float g = graph[o + tid + 256 * i];
    g2 += g * g;
    const float p = wave_prod(1.f - a[i]);     // the wave holds one whole row of the adjacency
    pp += p * p;                                // (identical in every lane; counted once below)
  }
  const RegStats st = reg_softmax_stats(a, u, sm);
  float kl = 0.f;
#pragma unroll
  for (int i = 0; i < RE; ++i) {
    const float lt = u[i] - st.mu - st.lsu;    // log target
    kl += __expf(lt) * (lt - (a[i] - st.ma - st.lsa));
  }
  if ((tid & 63) != 0) pp = 0.f;
  block_sum2(kl, g2, sm);
  float zero = 0.f;
  block_sum2(pp, zero, sm);
  if (tid == 0) {
    part[4 * b] = kl;
    part[4 * b + 1] = sqrtf(g2);
    part[4 * b + 2] = sqrtf(pp);
    part[4 * b + 3] = ckl * kl + cgs * sqrtf(g2) + cpt * sqrtf(pp);
  }
}

// d adj = ckl * (softmax(adj) - target) - cpt / pt_b * P_i * prod_{j' != j}(1 - adj[i,j']);  d graph = cgs * graph / gs_b
__global__ __launch_bounds__(256) void ct_reg_bwd_kernel(const float* __restrict__ adj, const float* __restrict__ graph,
                                                        const float* __restrict__ uni, const float* __restrict__ part,
                                                        const float* __restrict__ g_loss, float ckl, float cgs, float cpt,
                                                        float* __restrict__ d_adj, float* __restrict__ d_graph) {
  __shared__ float sm[8];
  const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
  const long o = (long)b * RN * RN;
  const float gl = g_loss[0];
  float a[RE], u[RE];
#pragma unroll
  for (int i = 0; i < RE; ++i) {
    a[i] = adj[o + tid + 256 * i];
    u[i] = uni[o + tid + 256 * i];
  }
  const RegStats st = reg_softmax_stats(a, u, sm);
  const float gs = part[4 * b + 1], pt = part[4 * b + 2];
  const float kgs = gs > 0.f ? gl * cgs / gs : 0.f, kpt = pt > 0.f ? gl * cpt / pt : 0.f, kkl = gl * ckl;
#pragma unroll
  for (int i = 0; i < RE; ++i) {
    const float q = 1.f - a[i];
    // exclusive prefix and suffix products along the row (the wave), log-step scans
    float pre = q, suf = q;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float up = __shfl_up(pre, d, 64), dn = __shfl_down(suf, d, 64);
      if (lane >= d) pre *= up;
      if (lane + d < 64) suf *= dn;
    }
    const float row = __shfl(pre, 63, 64);                         // P_i
    float ex = __shfl_up(pre, 1, 64), sx = __shfl_down(suf, 1, 64);
    if (lane == 0) ex = 1.f;
    if (lane == 63) sx = 1.f;
    const float others = ex * sx;                                  // prod over j' != j
    const float dkl = __expf(a[i] - st.ma - st.lsa) - __expf(u[i] - st.mu - st.lsu);
    d_adj[o + tid + 256 * i] = kkl * dkl - kpt * row * others;
    d_graph[o + tid + 256 * i] = kgs * graph[o + tid + 256 * i];
  }
}

// one wave per row r < R; y [R][Hs][D], mask [R] (Hs == 2), D <= 64
__global__ __launch_bounds__(256) void ct_blend_softmax_fwd_kernel(const float* __restrict__ y, const float* __restrict__ mask,
                                                                  float* __restrict__ probs, long R, int Hs, int D) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const bool ok = lane < D;
  float v = ok ? y[(r * Hs) * D + lane] : -INFINITY;
  if (Hs == 2) {
    const float m = mask[r];
    const float v1 = ok ? y[(r * Hs + 1) * D + lane] : 0.f;
    v = ok ? v * (1.f - m) + v1 * m : -INFINITY;
  }
  const float mx = wave_max(v);
  const float e = ok ? __expf(v - mx) : 0.f;
  const float s = wave_sum(e);
  if (ok) probs[r * D + lane] = e / s;
}

__global__ __launch_bounds__(256) void ct_blend_softmax_bwd_kernel(const float* __restrict__ g, const float* __restrict__ probs,
                                                                  const float* __restrict__ y, const float* __restrict__ mask,
                                                                  float* __restrict__ dy, float* __restrict__ dmask, long R, int Hs,
                                                                  int D) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const bool ok = lane < D;
  const float p = ok ? probs[r * D + lane] : 0.f, gv = ok ? g[r * D + lane] : 0.f;
  const float dot = wave_sum(p * gv);
  const float dv = p * (gv - dot);
  if (Hs == 2) {
    const float m = mask[r];
    float dm = 0.f;
    if (ok) {
      dy[(r * 2) * D + lane] = dv * (1.f - m);
      dy[(r * 2 + 1) * D + lane] = dv * m;
      dm = dv * (y[(r * 2 + 1) * D + lane] - y[(r * 2) * D + lane]);
    }
    dm = wave_sum(dm);
    if (lane == 0 && dmask != nullptr) dmask[r] = dm;
  } else if (ok) {
    dy[r * D + lane] = dv;
  }
}

// row_loss[r] = logsumexp_d(lp) - lp[target[r]], lp = log(max(p, 1e-4))
__global__ __launch_bounds__(256) void ct_latent_ce_fwd_kernel(const float* __restrict__ probs, const long long* __restrict__ target,
                                                              float* __restrict__ row_loss, long R, int D) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const bool ok = lane < D;
  const float lp = ok ? __logf(fmaxf(probs[r * D + lane], 1e-4f)) : -INFINITY;
  const float mx = wave_max(lp);
  const float s = wave_sum(ok ? __expf(lp - mx) : 0.f);
  const float lt = __shfl(lp, (int)target[r], 64);
  if (lane == 0) row_loss[r] = mx + __logf(s) - lt;
}

__global__ __launch_bounds__(256) void ct_latent_ce_bwd_kernel(const float* __restrict__ probs, const long long* __restrict__ target,
                                                              const float* __restrict__ g_loss, float* __restrict__ d_probs, long R,
                                                              int D) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const bool ok = lane < D;
  const float p = ok ? probs[r * D + lane] : 1.f;
  const float lp = ok ? __logf(fmaxf(p, 1e-4f)) : -INFINITY;
  const float mx = wave_max(lp);
  const float e = ok ? __expf(lp - mx) : 0.f;
  const float s = wave_sum(e);
  const float dlp = (e / s - (lane == (int)target[r] ? 1.f : 0.f)) * (g_loss[0] / (float)R);
  if (ok) d_probs[r * D + lane] = p > 1e-4f ? dlp / p : 0.f;      // clamp(min=1e-4) passes no gradient below the bound
}

// ---- CausalTransition._compute_mask (ct_mcq_vae.py:117-127) ------------------------------------------------------------
//   pos = dropout(pe)                (PositionalEncoding of zeros, :119; keep = the dropout keep mask or null in eval mode)
//   inter = sigmoid(W [action ; pos] + b),  p = sum_d x[b,s,d] * inter[b,s,d],  sample = straight-through Bernoulli(p)
// One workgroup per sample; W^T ([A+D][D]) staged in LDS; thread = (node s, a quarter of the D outputs).  D == S == 64.
constexpr int MD = 64;

__global__ __launch_bounds__(256) void ct_mask_fwd_kernel(const float* __restrict__ x, const float* __restrict__ action,
                                                         const float* __restrict__ pe, const float* __restrict__ keep, float scale,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         const float* __restrict__ expo, int A, float* __restrict__ inter,
                                                         float* __restrict__ p_out, float* __restrict__ sample,
                                                         float* __restrict__ soft) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem;                       // [A+D][D]  (W^T)
  float* sPos = sW + (A + MD) * MD;       // [S][D+1]
  float* sAct = sPos + MD * (MD + 1);     // [D]  bias + W[:, :A] action
  const int tid = threadIdx.x, b = blockIdx.x, In = A + MD;
  // staging in batches of eight loads per thread (as `for (e = tid; ...; e += 256) s[e] = g[e]` these were 35 memory round trips in a
  // row -- most of this kernel's 23 us)
  for (int e0 = 0; e0 < MD * In; e0 += 8 * 256) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + tid + 256 * u;
      t[u] = W[e < MD * In ? e : MD * In - 1];          // clamped index: a predicated load is a branch and is waited for on the spot
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + tid + 256 * u;
      if (e < MD * In) {
        const int d = e / In, i = e - d * In;          // W is [D][A+D] row-major
        sW[i * MD + d] = t[u];
      }
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float tp[8], tk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + 256 * (8 * h + u);
      tp[u] = pe[e];
      tk[u] = keep ? keep[(long)b * MD * MD + e] * scale : 1.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + 256 * (8 * h + u);
      sPos[(e >> 6) * (MD + 1) + (e & 63)] = tp[u] * tk[u];
    }
  }
  float* sA = sAct + MD;                  // [A] the sample's action row (a serial loop of global loads per output otherwise)
  if (tid < A) sA[tid] = action[(long)b * A + tid];
  const float bias_v = tid < MD ? bias[tid] : 0.f;
  __syncthreads();
  if (tid < MD) {
    float z = bias_v;
    for (int a = 0; a < A; ++a) z += sW[a * MD + tid] * sA[a];
    sAct[tid] = z;
  }
  __syncthreads();
  const int s = tid >> 2, d0 = (tid & 3) * 16;
  float z[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) z[j] = sAct[d0 + j];
  for (int e = 0; e < MD; ++e) {
    const float pv = sPos[s * (MD + 1) + e];
    const float* w = sW + (A + e) * MD + d0;
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(w + 4 * j4);
#pragma unroll
      for (int j = 0; j < 4; ++j) z[4 * j4 + j] += w4[j] * pv;
    }
  }
  const long row = ((long)b * MD + s) * MD + d0;
  float p = 0.f;
#pragma unroll
  for (int j4 = 0; j4 < 4; ++j4) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + row + 4 * j4);
    f32x4 sv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sv[j] = 1.f / (1.f + __expf(-z[4 * j4 + j]));
      p += xv[j] * sv[j];
    }
    *reinterpret_cast<f32x4*>(inter + row + 4 * j4) = sv;
  }
  p += __shfl_xor(p, 1, 64);
  p += __shfl_xor(p, 2, 64);
  if ((tid & 3) == 0) {
    const long r = (long)b * MD + s;
    const float a0 = __logf(fmaxf(1.f - p, 1e-4f)) - __logf(expo[2 * r]);        // Gumbel noise = -log(exponential draw)
    const float a1 = __logf(fmaxf(p, 1e-4f)) - __logf(expo[2 * r + 1]);
    const float y1 = 1.f / (1.f + __expf(a0 - a1));
    const float hard = a1 > a0 ? 1.f : 0.f;
    p_out[r] = p;
    sample[r] = (hard - y1) + y1;
    soft[r] = y1;
  }
}

// dWp [B][A+D][D] (W^T layout), dbp [B][D]: per-sample partials, summed by the caller
__global__ __launch_bounds__(256) void ct_mask_bwd_kernel(const float* __restrict__ x, const float* __restrict__ action,
                                                         const float* __restrict__ pe, const float* __restrict__ keep, float scale,
                                                         const float* __restrict__ inter, const float* __restrict__ p_in,
                                                         const float* __restrict__ soft, const float* __restrict__ g, int A,
                                                         float* __restrict__ dWp, float* __restrict__ dbp) {
  __shared__ float sDz[MD * (MD + 1)];    // [s][d]
  __shared__ float sPos[MD * (MD + 1)];   // [s][e]
  __shared__ float sCol[MD];
  const int tid = threadIdx.x, b = blockIdx.x;
#pragma unroll
  for (int h = 0; h < 4; ++h) {                                     // four elements' loads in flight per thread (16 round trips in a row before)
    float pi[4], y1[4], gg[4], sg[4], xv[4], kv[4], pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * (4 * h + u);
      const long r = (long)b * MD + (e >> 6);
      pi[u] = p_in[r]; y1[u] = soft[r]; gg[u] = g[r];
      sg[u] = inter[(long)b * MD * MD + e];
      xv[u] = x[(long)b * MD * MD + e];
      kv[u] = keep ? keep[(long)b * MD * MD + e] * scale : 1.f;
      pv[u] = pe[e];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * (4 * h + u);
      const int s = e >> 6, d = e & 63;
      const float dd = (pi[u] > 1e-4f ? 1.f / pi[u] : 0.f) + ((1.f - pi[u]) > 1e-4f ? 1.f / (1.f - pi[u]) : 0.f);
      const float dp = gg[u] * y1[u] * (1.f - y1[u]) * dd;          // straight-through estimator (gumbel_st_bwd_kernel)
      sDz[s * (MD + 1) + d] = dp * xv[u] * sg[u] * (1.f - sg[u]);
      sPos[s * (MD + 1) + d] = pv[u] * kv[u];
    }
  }
  __syncthreads();
  if (tid < MD) {
    float c = 0.f;
    for (int s = 0; s < MD; ++s) c += sDz[s * (MD + 1) + tid];
    sCol[tid] = c;
    dbp[(long)b * MD + tid] = c;
  }
  __syncthreads();
  float* out = dWp + (long)b * (A + MD) * MD;
  for (int e = tid; e < A * MD; e += 256) {                         // action rows: act[a] * sum_s dz[s][d]
    const int a = e >> 6, d = e & 63;
    out[e] = action[(long)b * A + a] * sCol[d];
  }
  const int ei = tid >> 2, d0 = (tid & 3) * 16;                     // position rows: sum_s pos[s][e] * dz[s][d]
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int s = 0; s < MD; ++s) {
    const float pv = sPos[s * (MD + 1) + ei];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] += pv * sDz[s * (MD + 1) + d0 + j];
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) out[(A + ei) * MD + d0 + j] = acc[j];
}

// Straight-through Bernoulli(p) with the exponential draws given (Gumbel = -log E), optionally also w = p * sample
// (weighted_graph = adjacency_coeffs * causal_graph, ct_mcq_vae.py:241-244,268-271)
__global__ __launch_bounds__(256) void ct_sample_fwd_kernel(const float* __restrict__ p, const float* __restrict__ expo,
                                                           float* __restrict__ out, float* __restrict__ soft,
                                                           float* __restrict__ weighted, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float pi = p[i];
    const f32x2 e = *reinterpret_cast<const f32x2*>(expo + 2 * i);
    const float a0 = __logf(fmaxf(1.f - pi, 1e-4f)) - __logf(e[0]);
    const float a1 = __logf(fmaxf(pi, 1e-4f)) - __logf(e[1]);
    const float y1 = 1.f / (1.f + __expf(a0 - a1));
    const float hard = a1 > a0 ? 1.f : 0.f;
    const float sm = (hard - y1) + y1;
    out[i] = sm;
    soft[i] = y1;
    if (weighted) weighted[i] = pi * sm;
  }
}

// gp = (g_sample + g_w * p) * d sample / d p + g_w * sample
__global__ __launch_bounds__(256) void ct_sample_bwd_kernel(const float* __restrict__ gs, const float* __restrict__ gw,
                                                           const float* __restrict__ p, const float* __restrict__ soft,
                                                           const float* __restrict__ sample, float* __restrict__ gp, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float pi = p[i], y1 = soft[i];
    const float d = (pi > 1e-4f ? 1.f / pi : 0.f) + ((1.f - pi) > 1e-4f ? 1.f / (1.f - pi) : 0.f);
    const float g_s = (gs ? gs[i] : 0.f) + (gw ? gw[i] * pi : 0.f);
    gp[i] = g_s * y1 * (1.f - y1) * d + (gw ? gw[i] * sample[i] : 0.f);
  }
}

// adjacency = s0 * (1 - m) + s1 * m with the intervention mask m per (sample, source node) row of 64 targets
// (CausalTransition._compute_adj, ct_mcq_vae.py:153): one launch instead of rsub, mul, mul, add -- and one instead of their
// five autograd mirrors on the way back (a wave owns a row, so d m is a shuffle reduction)
__global__ __launch_bounds__(256) void ct_blend_fwd_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                                                          const float* __restrict__ mask, float* __restrict__ out, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float m = mask[i >> 6];
    out[i] = s0[i] * (1.f - m) + s1[i] * m;
  }
}

__global__ __launch_bounds__(256) void ct_blend_bwd_kernel(const float* __restrict__ g, const float* __restrict__ s0,
                                                          const float* __restrict__ s1, const float* __restrict__ mask,
                                                          float* __restrict__ g0, float* __restrict__ g1, float* __restrict__ gm,
                                                          long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {      // n % 64 == 0: a wave walks whole rows
    const float m = mask[i >> 6], gi = g[i];
    g0[i] = gi * (1.f - m);
    g1[i] = gi * m;
    float d = gi * (s1[i] - s0[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if ((threadIdx.x & 63) == 0) gm[i >> 6] = d;
  }
}

// PositionalEncoding.forward (ct_mcq_vae.py:33-38): out = (x + pe[s]) * keep * scale, keep the dropout mask (NULL: eval /
// p = 0) -- one launch instead of add, mul, mul; backward g * keep * scale
__global__ __launch_bounds__(256) void ct_posenc_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                           const float* __restrict__ keep, float scale, float* __restrict__ out,
                                                           long n4, int sd4) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i] + reinterpret_cast<const f32x4*>(pe)[i % sd4];
    if (keep != nullptr) v = v * reinterpret_cast<const f32x4*>(keep)[i] * scale;
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
}

__global__ __launch_bounds__(256) void ct_posenc_bwd_kernel(const float* __restrict__ g, const float* __restrict__ keep, float scale,
                                                           float* __restrict__ gx, long n4) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
    reinterpret_cast<f32x4*>(gx)[i] = reinterpret_cast<const f32x4*>(g)[i] * reinterpret_cast<const f32x4*>(keep)[i] * scale;
}

// F.one_hot(inds, N).float() (CTMCQVAE.ct_preprocess, ct_mcq_vae.py:472-480) in one launch (fill, scatter, cast as torch ops)
__global__ __launch_bounds__(256) void one_hot_kernel(const long long* __restrict__ inds, long n, int N4, float* __restrict__ out) {
  const long stride = (long)gridDim.x * 256, tot = n * N4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += stride) {
    const long r = i / N4;
    const int c = (int)(i - r * N4) * 4, k = (int)inds[r];
    reinterpret_cast<f32x4*>(out)[i] = f32x4{k == c ? 1.f : 0.f, k == c + 1 ? 1.f : 0.f, k == c + 2 ? 1.f : 0.f, k == c + 3 ? 1.f : 0.f};
  }
}

// out[z][g][c] (+)= sum over the rows r (in order) with grp[r] == g of parts[z][r][c]: per-sample partial gradients gathered
// into the rows of a parameter bank (scorer rows of the discoverers, per-head vectors of the last GATv2 layer) without
// atomics -- four lanes own an output element, walk the rows in a fixed interleave and meet in a fixed order.  grp == NULL: every
// row belongs to group 0.
__global__ __launch_bounds__(256) void group_rowsum_kernel(const float* __restrict__ parts, long mat_stride, int rows, int C, int ld,
                                                          const int* __restrict__ grp, int G, float* __restrict__ out,
                                                          int accumulate) {
  // 64 columns x 4 row lanes per workgroup: lane q walks rows q, q+4, ... with eight loads in flight (every row is loaded,
  // rows of other groups are dropped by a select: no divergent branch around the load), the four lanes meet in a fixed order
  __shared__ float sm[4][64];
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, g = blockIdx.y, z = blockIdx.z;
  const bool cok = c < C;
  const float* p = parts + (long)z * mat_stride + (cok ? c : 0);
  float acc = 0.f;
  if (grp != nullptr || g == 0) {
    int r = q;
    for (; r + 28 < rows; r += 32) {
      float v[8];
      int k[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = p[(long)(r + 4 * u) * ld];
        k[u] = grp != nullptr ? grp[r + 4 * u] : 0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += k[u] == g ? v[u] : 0.f;
    }
    for (; r < rows; r += 4) acc += (grp != nullptr ? grp[r] : 0) == g ? p[(long)r * ld] : 0.f;
  }
  sm[q][cl] = acc;
  __syncthreads();
  if (q == 0 && cok) {
    const float t = ((sm[0][cl] + sm[1][cl]) + sm[2][cl]) + sm[3][cl];
    float* o = out + ((long)z * G + g) * C + c;
    *o = accumulate ? *o + t : t;
  }
}

}  // namespace

int launch_ct_blend_forward(const float* s0, const float* s1, const float* mask, float* out, long rows, hipStream_t st) {
  if (!s0 || !s1 || !mask || !out || rows <= 0) return kErrBadArg;
  const long n = rows * 64;
  ProfScope ps("ct_blend_fwd_kernel", st, 0.0, 4.0 * n * 3.0);
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ct_blend_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, s0, s1, mask, out, n);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ct_blend_backward(const float* g, const float* s0, const float* s1, const float* mask, float* g0, float* g1, float* gm,
                             long rows, hipStream_t st) {
  if (!g || !s0 || !s1 || !mask || !g0 || !g1 || !gm || rows <= 0) return kErrBadArg;
  const long n = rows * 64;
  ProfScope ps("ct_blend_bwd_kernel", st, 0.0, 4.0 * n * 5.0);
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ct_blend_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, s0, s1, mask, g0, g1, gm, n);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ct_posenc_forward(const float* x, const float* pe, const float* keep, float scale, float* out, long n, int sd,
                             hipStream_t st) {
  if (!x || !pe || !out || n <= 0 || sd <= 0 || (n & 3) || (sd & 3) || n % sd) return kErrBadArg;
  ProfScope ps("ct_posenc_fwd_kernel", st, 0.0, 4.0 * n * (keep ? 3.0 : 2.0));
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ct_posenc_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, pe, keep, scale, out, n / 4, sd / 4);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ct_posenc_backward(const float* g, const float* keep, float scale, float* gx, long n, hipStream_t st) {
  if (!g || !keep || !gx || n <= 0 || (n & 3)) return kErrBadArg;
  ProfScope ps("ct_posenc_bwd_kernel", st, 0.0, 4.0 * n * 3.0);
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ct_posenc_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, keep, scale, gx, n / 4);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_one_hot(const long long* inds, long n, int N, float* out, hipStream_t st) {
  if (!inds || !out || n <= 0 || N <= 0 || (N & 3)) return kErrBadArg;
  ProfScope ps("one_hot_kernel", st, 0.0, 8.0 * n + 4.0 * n * N);
  long blocks = (n * (N / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(one_hot_kernel, dim3((unsigned)blocks), dim3(256), 0, st, inds, n, N / 4, out);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_group_rowsum(const float* parts, long mat_stride, int nmat, int rows, int C, int ld, const int* grp, int G, float* out,
                        int accumulate, hipStream_t st) {
  if (!parts || !out || nmat <= 0 || rows <= 0 || C <= 0 || ld < 1 || G <= 0) return kErrBadArg;
  ProfScope ps("group_rowsum_kernel", st, 0.0, 4.0 * nmat * ((double)rows * C + (double)G * C));
  hipLaunchKernelGGL(group_rowsum_kernel, dim3((C + 63) / 64, G, nmat), dim3(256), 0, st, parts, mat_stride, rows, C, ld, grp, G, out,
                     accumulate);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ct_reg_forward(const float* adj, const float* graph, const float* uni, float* part, float ckl, float cgs, float cpt,
                          int B, int N, hipStream_t st) {
  if (!adj || !graph || !uni || !part || B <= 0 || N != RN) return kErrBadArg;
  ProfScope ps("ct_reg_fwd_kernel", st, 0.0, 4.0 * B * 3.0 * N * N);
  hipLaunchKernelGGL(ct_reg_fwd_kernel, dim3(B), dim3(256), 0, st, adj, graph, uni, part, ckl, cgs, cpt);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ct_reg_backward(const float* adj, const float* graph, const float* uni, const float* part, const float* g_loss,
                           float ckl, float cgs, float cpt, float* d_adj, float* d_graph, int B, int N, hipStream_t st) {
  if (!adj || !graph || !uni || !part || !g_loss || !d_adj || !d_graph || B <= 0 || N != RN) return kErrBadArg;
  ProfScope ps("ct_reg_bwd_kernel", st, 0.0, 4.0 * B * 5.0 * N * N);
  hipLaunchKernelGGL(ct_reg_bwd_kernel, dim3(B), dim3(256), 0, st, adj, graph, uni, part, g_loss, ckl, cgs, cpt, d_adj, d_graph);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ct_blend_softmax_forward(const float* y, const float* mask, float* probs, long R, int Hs, int D, hipStream_t st) {
  if (!y || !probs || R <= 0 || D <= 0 || D > 64 || (Hs != 1 && Hs != 2) || (Hs == 2 && !mask)) return kErrBadArg;
  ProfScope ps("ct_blend_softmax_fwd_kernel", st, 0.0, 4.0 * R * (Hs + 1.0) * D);
  hipLaunchKernelGGL(ct_blend_softmax_fwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, y, mask, probs, R, Hs, D);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ct_blend_softmax_backward(const float* g, const float* probs, const float* y, const float* mask, float* dy, float* dmask,
                                     long R, int Hs, int D, hipStream_t st) {
  if (!g || !probs || !y || !dy || R <= 0 || D <= 0 || D > 64 || (Hs != 1 && Hs != 2) || (Hs == 2 && !mask)) return kErrBadArg;
  ProfScope ps("ct_blend_softmax_bwd_kernel", st, 0.0, 4.0 * R * (2.0 * Hs + 2.0) * D);
  hipLaunchKernelGGL(ct_blend_softmax_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, g, probs, y, mask, dy, dmask, R, Hs, D);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ct_latent_ce_forward(const float* probs, const long long* target, float* row_loss, long R, int D, hipStream_t st) {
  if (!probs || !target || !row_loss || R <= 0 || D <= 0 || D > 64) return kErrBadArg;
  ProfScope ps("ct_latent_ce_fwd_kernel", st, 0.0, 4.0 * R * D);
  hipLaunchKernelGGL(ct_latent_ce_fwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, probs, target, row_loss, R, D);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ct_latent_ce_backward(const float* probs, const long long* target, const float* g_loss, float* d_probs, long R, int D,
                                 hipStream_t st) {
  if (!probs || !target || !g_loss || !d_probs || R <= 0 || D <= 0 || D > 64) return kErrBadArg;
  ProfScope ps("ct_latent_ce_bwd_kernel", st, 0.0, 8.0 * R * D);
  hipLaunchKernelGGL(ct_latent_ce_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, probs, target, g_loss, d_probs, R, D);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ct_mask_forward(const float* x, const float* action, const float* pe, const float* keep, float scale, const float* W,
                           const float* bias, const float* expo, int B, int S, int D, int A, float* inter, float* p, float* sample,
                           float* soft, hipStream_t st) {
  if (!x || !action || !pe || !W || !bias || !expo || !inter || !p || !sample || !soft || B <= 0 || S != MD || D != MD || A <= 0 ||
      A > 256)
    return kErrBadArg;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ct_mask_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const size_t smem = ((size_t)(A + MD) * MD + MD * (MD + 1) + MD + A) * sizeof(float);
  ProfScope ps("ct_mask_fwd_kernel", st, 2.0 * B * S * D * (A + D), 4.0 * B * S * D * 3.0);
  hipLaunchKernelGGL(ct_mask_fwd_kernel, dim3(B), dim3(256), smem, st, x, action, pe, keep, scale, W, bias, expo, A, inter, p, sample,
                     soft);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ct_mask_backward(const float* x, const float* action, const float* pe, const float* keep, float scale,
                            const float* inter, const float* p, const float* soft, const float* g, int B, int S, int D, int A,
                            float* dWp, float* dbp, hipStream_t st) {
  if (!x || !action || !pe || !inter || !p || !soft || !g || !dWp || !dbp || B <= 0 || S != MD || D != MD || A <= 0) return kErrBadArg;
  ProfScope ps("ct_mask_bwd_kernel", st, 2.0 * B * S * D * D, 4.0 * B * (S * D * 3.0 + (A + D) * D));
  hipLaunchKernelGGL(ct_mask_bwd_kernel, dim3(B), dim3(256), 0, st, x, action, pe, keep, scale, inter, p, soft, g, A, dWp, dbp);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ct_sample_forward(const float* p, const float* expo, float* out, float* soft, float* weighted, long n, hipStream_t st) {
  if (!p || !expo || !out || !soft || n <= 0) return kErrBadArg;
  ProfScope ps("ct_sample_fwd_kernel", st, 0.0, 4.0 * n * (weighted ? 6.0 : 5.0));
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ct_sample_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, expo, out, soft, weighted, n);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ct_sample_backward(const float* gs, const float* gw, const float* p, const float* soft, const float* sample, float* gp,
                              long n, hipStream_t st) {
  if ((!gs && !gw) || !p || !soft || !sample || !gp || n <= 0) return kErrBadArg;
  ProfScope ps("ct_sample_bwd_kernel", st, 0.0, 4.0 * n * 6.0);
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ct_sample_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gs, gw, p, soft, sample, gp, n);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
