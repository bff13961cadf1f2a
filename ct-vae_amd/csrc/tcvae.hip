// Total-correlation decomposition of BetaTCVAE's KL term (models/betatc_vae.py:128-199): with z, mu, lv [B][D] and the
// log importance weights liw [B][B] of minibatch stratified sampling,
//     M[i,j,d]     = log N(z_i[d]; mu_j[d], exp(lv_j[d])) + liw[i,j]
//     log_q_z[i]   = logsumexp_j sum_d M[i,j,d]        (liw enters D times, as the reference writes it, :185-187)
//     log_prod[i]  = sum_d logsumexp_j M[i,j,d]
//     log_q_zx[i]  = sum_d log N(z_i[d]; mu_i[d], exp(lv_i[d])),   log_p_z[i] = sum_d log N(z_i[d]; 0, 1)
//     mi = mean(log_q_zx - log_q_z),  tc = mean(log_q_z - log_prod),  kld = mean(log_prod - log_p_z)
// The reference materialises M as a [B,B,D] tensor.  Here a workgroup owns sample i, its threads walk the j with running
// (max, sum) pairs for the D + 1 logsumexps, merged by shuffles and once through LDS; the backward pass recomputes M and has
// one producer per output element (blocks [0,B): d/dz_i; blocks [B,2B): d/dmu_j, d/dlv_j).  D <= 32, B <= 4096.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

namespace {

constexpr int kTcMaxD = 32;
constexpr float kLog2Pi = 1.8378770664093453f;

__device__ __forceinline__ float tc_wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ void lse_merge(float& m, float& l, float m2, float l2) {
  const float mm = fmaxf(m, m2);
  if (mm == -INFINITY) { m = mm; l = 0.f; return; }
  l = l * __expf(m - mm) + l2 * __expf(m2 - mm);
  m = mm;
}

template <int DP>   // D padded to a compile-time bound (registers)
__global__ __launch_bounds__(256) void tc_fwd_kernel(const float* __restrict__ z, const float* __restrict__ mu,
                                                    const float* __restrict__ lv, const float* __restrict__ liw, int B, int D,
                                                    float* __restrict__ row /* [B][4] */, float* __restrict__ lse_s,
                                                    float* __restrict__ lse_d /* [B][D] */) {
  __shared__ float sz[DP];
  __shared__ float sM[4][DP + 1], sL[4][DP + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = blockIdx.x;
  if (tid < D) sz[tid] = z[(long)i * D + tid];
  __syncthreads();
  float m[DP + 1], l[DP + 1];
#pragma unroll
  for (int d = 0; d <= DP; ++d) { m[d] = -INFINITY; l[d] = 0.f; }
  for (int j = tid; j < B; j += 256) {
    const float w = liw[(long)i * B + j];
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      if (d < D) {
        const float lj = lv[(long)j * D + d], dl = sz[d] - mu[(long)j * D + d];
        const float v = -0.5f * (kLog2Pi + lj) - 0.5f * dl * dl * __expf(-lj) + w;
        s += v;
        lse_merge(m[d], l[d], v, 1.f);
      }
    }
    lse_merge(m[DP], l[DP], s, 1.f);
  }
#pragma unroll
  for (int d = 0; d <= DP; ++d) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float m2 = __shfl_xor(m[d], o, 64), l2 = __shfl_xor(l[d], o, 64);
      lse_merge(m[d], l[d], m2, l2);
    }
    if (lane == 0) { sM[wave][d] = m[d]; sL[wave][d] = l[d]; }
  }
  __syncthreads();
  if (tid <= DP && (tid < D || tid == DP)) {
    float mm = sM[0][tid], ll = sL[0][tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) lse_merge(mm, ll, sM[w][tid], sL[w][tid]);
    const float r = mm + __logf(ll);
    if (tid == DP) lse_s[i] = r;
    else lse_d[(long)i * D + tid] = r;
    sM[0][tid] = r;
  }
  __syncthreads();
  if (tid == 0) {
    float prod = 0.f, qzx = 0.f, pz = 0.f;
    for (int d = 0; d < D; ++d) {
      prod += sM[0][d];
      const float li = lv[(long)i * D + d], dl = sz[d] - mu[(long)i * D + d];
      qzx += -0.5f * (kLog2Pi + li) - 0.5f * dl * dl * __expf(-li);
      pz += -0.5f * kLog2Pi - 0.5f * sz[d] * sz[d];
    }
    row[4 * i] = qzx; row[4 * i + 1] = sM[0][DP]; row[4 * i + 2] = prod; row[4 * i + 3] = pz;
  }
}

__global__ __launch_bounds__(256) void tc_finish_kernel(const float* __restrict__ row, int B, float* __restrict__ out) {
  __shared__ double sm[4][4];
  double a[4] = {0, 0, 0, 0};
  for (int b = threadIdx.x; b < B; b += 256)
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] += (double)row[4 * b + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = wave_sum_d(a[k]);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][k] = a[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[4];
    for (int k = 0; k < 4; ++k) t[k] = ((sm[0][k] + sm[1][k]) + (sm[2][k] + sm[3][k])) / B;
    out[0] = (float)(t[0] - t[1]);      // mi
    out[1] = (float)(t[1] - t[2]);      // tc
    out[2] = (float)(t[2] - t[3]);      // kld
  }
}

// g3 = {d loss/d mi, d loss/d tc, d loss/d kld} on the device
template <int DP>
__global__ __launch_bounds__(256) void tc_bwd_kernel(const float* __restrict__ z, const float* __restrict__ mu,
                                                    const float* __restrict__ lv, const float* __restrict__ liw,
                                                    const float* __restrict__ lse_s, const float* __restrict__ lse_d,
                                                    const float* __restrict__ g3, int B, int D, float* __restrict__ dz,
                                                    float* __restrict__ dmu, float* __restrict__ dlv) {
  __shared__ float sv[3][DP];
  __shared__ float sRed[4][2 * DP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float invB = 1.f / (float)B;
  const float ca = g3[0] * invB;                     // on log_q_zx
  const float cs = (-g3[0] + g3[1]) * invB;          // on log_q_z
  const float cp = (-g3[1] + g3[2]) * invB;          // on log_prod
  const float cz = -g3[2] * invB;                    // on log_p_z
  const bool roleA = (int)blockIdx.x < B;
  const int me = roleA ? blockIdx.x : blockIdx.x - B;
  if (tid < D) {
    sv[0][tid] = z[(long)me * D + tid];
    sv[1][tid] = mu[(long)me * D + tid];
    sv[2][tid] = lv[(long)me * D + tid];
  }
  __syncthreads();
  float a0[DP], a1[DP], r0[DP], r1[DP], r2[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    a0[d] = 0.f; a1[d] = 0.f;
    r0[d] = d < D ? sv[0][d] : 0.f; r1[d] = d < D ? sv[1][d] : 0.f; r2[d] = d < D ? sv[2][d] : 0.f;
  }
  if (roleA) {          // my sample is i: walk the j
    for (int j = tid; j < B; j += 256) {
      const float w = liw[(long)me * B + j];
      float md[DP], dl[DP], el[DP], s = 0.f;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        md[d] = 0.f; dl[d] = 0.f; el[d] = 0.f;
        if (d < D) {
          const float lj = lv[(long)j * D + d];
          dl[d] = r0[d] - mu[(long)j * D + d];
          el[d] = __expf(-lj);
          md[d] = -0.5f * (kLog2Pi + lj) - 0.5f * dl[d] * dl[d] * el[d] + w;
          s += md[d];
        }
      }
      const float ws = cs * __expf(s - lse_s[me]);
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D) a0[d] += (ws + cp * __expf(md[d] - lse_d[(long)me * D + d])) * (-dl[d] * el[d]);
    }
  } else {              // my sample is j: walk the i
    for (int i = tid; i < B; i += 256) {
      const float w = liw[(long)i * B + me];
      float md[DP], dl[DP], s = 0.f;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        md[d] = 0.f; dl[d] = 0.f;
        if (d < D) {
          dl[d] = z[(long)i * D + d] - r1[d];
          md[d] = -0.5f * (kLog2Pi + r2[d]) - 0.5f * dl[d] * dl[d] * __expf(-r2[d]) + w;
          s += md[d];
        }
      }
      const float ws = cs * __expf(s - lse_s[i]);
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D) {
          const float el = __expf(-r2[d]);
          const float wd = ws + cp * __expf(md[d] - lse_d[(long)i * D + d]);
          a0[d] += wd * (dl[d] * el);
          a1[d] += wd * (-0.5f + 0.5f * dl[d] * dl[d] * el);
        }
    }
  }
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    a0[d] = tc_wsum(a0[d]);
    a1[d] = tc_wsum(a1[d]);
    if (lane == 0) { sRed[wave][d] = a0[d]; sRed[wave][DP + d] = a1[d]; }
  }
  __syncthreads();
  if (tid < D) {
    const float s0 = (sRed[0][tid] + sRed[1][tid]) + (sRed[2][tid] + sRed[3][tid]);
    const float s1 = (sRed[0][DP + tid] + sRed[1][DP + tid]) + (sRed[2][DP + tid] + sRed[3][DP + tid]);
    const float zz = sv[0][tid], dl = zz - sv[1][tid], el = __expf(-sv[2][tid]);
    if (roleA) dz[(long)me * D + tid] = s0 + ca * (-dl * el) + cz * (-zz);
    else {
      dmu[(long)me * D + tid] = s0 + ca * (dl * el);
      dlv[(long)me * D + tid] = s1 + ca * (-0.5f + 0.5f * dl * dl * el);
    }
  }
}

}  // namespace

int launch_tc_forward(const float* z, const float* mu, const float* lv, const float* liw, int B, int D, float* out3, float* lse_s,
                      float* lse_d, float* ws, size_t ws_bytes, hipStream_t st) {
  if (!z || !mu || !lv || !liw || !out3 || !lse_s || !lse_d || !ws || B < 2 || B > 4096 || D < 1 || D > kTcMaxD) return kErrBadArg;
  if (ws_bytes / sizeof(float) < (size_t)4 * B) return kErrWorkspace;
  {
    ProfScope ps("tc_fwd_kernel", st, 8.0 * B * (double)B * D, 4.0 * ((double)B * B + 3.0 * B * D));
    if (D <= 16) hipLaunchKernelGGL(tc_fwd_kernel<16>, dim3(B), dim3(256), 0, st, z, mu, lv, liw, B, D, ws, lse_s, lse_d);
    else hipLaunchKernelGGL(tc_fwd_kernel<32>, dim3(B), dim3(256), 0, st, z, mu, lv, liw, B, D, ws, lse_s, lse_d);
    CTVAE_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(tc_finish_kernel, dim3(1), dim3(256), 0, st, ws, B, out3);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_tc_backward(const float* z, const float* mu, const float* lv, const float* liw, const float* lse_s, const float* lse_d,
                       const float* g3, int B, int D, float* dz, float* dmu, float* dlv, hipStream_t st) {
  if (!z || !mu || !lv || !liw || !lse_s || !lse_d || !g3 || !dz || !dmu || !dlv || B < 2 || B > 4096 || D < 1 || D > kTcMaxD)
    return kErrBadArg;
  ProfScope ps("tc_bwd_kernel", st, 24.0 * B * (double)B * D, 8.0 * ((double)B * B + 6.0 * B * D));
  if (D <= 16)
    hipLaunchKernelGGL(tc_bwd_kernel<16>, dim3(2 * B), dim3(256), 0, st, z, mu, lv, liw, lse_s, lse_d, g3, B, D, dz, dmu, dlv);
  else
    hipLaunchKernelGGL(tc_bwd_kernel<32>, dim3(2 * B), dim3(256), 0, st, z, mu, lv, liw, lse_s, lse_d, g3, B, D, dz, dmu, dlv);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
