// Maximum-mean-discrepancy regulariser of WAE_MMD / InfoVAE (models/wae_mmd.py:120-203, models/info_vae.py:150-229):
//   mmd = w_pp * K(p, p) + w_zz * K(z, z) - 2 * w_pz * K(p, z),   z = latent codes [N][D], p = N(0,1) draws [N][D]
//   imq:  K(a, b) = sum_{i != j} C / (eps + C + |a_i - b_j|^2)           (the reference removes the diagonal of ALL three
//                                                                          terms, the cross term's included)
//   rbf:  K(a, b) = mean_{i,j} exp(-mean_d (a_i - b_j)^2 / sigma)
// The reference materialises three [N][N][D] tensors; here one workgroup owns row i of z and of p, keeps both rows in
// registers (lane l holds dimensions l, l+64, ...), its four waves walk the other rows j (one 256-byte coalesced read
// per row and wave for D = 64..128, everything after the first touch comes from L2: N*D*4 bytes = 128 KB at N = 256),
// squared distances are xor-shuffle sums.  Row i also accumulates d mmd / d z_i (p carries no gradient) in the same
// pass, so the backward pass is a scaling by the incoming gradient.  Partial sums are combined in a fixed order:
// bit-reproducible.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

constexpr int kMmdMaxPerLane = 8;   // D <= 512

template <int NPL, int KIND>   // KIND 0: imq, 1: rbf
__global__ __launch_bounds__(256) void mmd_rows_kernel(const float* __restrict__ z, const float* __restrict__ p, int N, int D,
                                                       float c, float eps, float w_zz, float w_pz, float* __restrict__ part,
                                                       float* __restrict__ grad) {
  __shared__ float sS[4][3];
  __shared__ float sG[4][64 * NPL];
  const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float zi[NPL], pi[NPL], g[NPL];
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    const int d = lane + 64 * k;
    zi[k] = d < D ? z[(long)i * D + d] : 0.f;
    pi[k] = d < D ? p[(long)i * D + d] : 0.f;
    g[k] = 0.f;
  }
  float s_pp = 0.f, s_zz = 0.f, s_pz = 0.f;   // row i of K(p,p), K(z,z) and COLUMN i of K(p,z) (z_i against every p_j)
  for (int j = wave; j < N; j += 4) {
    float zj[NPL], pj[NPL];
    float dzz = 0.f, dpp = 0.f, dpz = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int d = lane + 64 * k;
      zj[k] = d < D ? z[(long)j * D + d] : 0.f;
      pj[k] = d < D ? p[(long)j * D + d] : 0.f;
      const float a = zi[k] - zj[k], b = pi[k] - pj[k], e = zi[k] - pj[k];
      dzz += a * a;
      dpp += b * b;
      dpz += e * e;
    }
    dzz = wave_sum(dzz);
    dpp = wave_sum(dpp);
    dpz = wave_sum(dpz);
    float kzz, kpp, kpz, czz, cpz;   // kernel values and the factors of (z_i - other) in d k / d z_i
    if constexpr (KIND == 0) {
      const float rzz = 1.f / (eps + c + dzz), rpp = 1.f / (eps + c + dpp), rpz = 1.f / (eps + c + dpz);
      kzz = c * rzz; kpp = c * rpp; kpz = c * rpz;
      czz = -2.f * c * rzz * rzz;
      cpz = -2.f * c * rpz * rpz;
      if (j == i) { kzz = kpp = kpz = 0.f; czz = cpz = 0.f; }   // diagonal excluded (wae_mmd.py:189)
    } else {
      const float invD = 1.f / (float)D;
      kzz = expf(-(dzz * invD) / c); kpp = expf(-(dpp * invD) / c); kpz = expf(-(dpz * invD) / c);
      czz = kzz * (-2.f * invD / c);
      cpz = kpz * (-2.f * invD / c);
    }
    s_zz += kzz; s_pp += kpp; s_pz += kpz;
    // K(z,z) holds the pair (i,j) twice (as (i,j) and (j,i)); K(p,z) holds z_i once per p_j
    const float fz = 2.f * w_zz * czz, fp = -2.f * w_pz * cpz;
#pragma unroll
    for (int k = 0; k < NPL; ++k) g[k] += fz * (zi[k] - zj[k]) + fp * (zi[k] - pj[k]);
  }
  if (lane == 0) { sS[wave][0] = s_pp; sS[wave][1] = s_zz; sS[wave][2] = s_pz; }
#pragma unroll
  for (int k = 0; k < NPL; ++k) sG[wave][lane + 64 * k] = g[k];
  __syncthreads();
  if (threadIdx.x < 3) part[(long)i * 3 + threadIdx.x] = (sS[0][threadIdx.x] + sS[1][threadIdx.x]) + (sS[2][threadIdx.x] + sS[3][threadIdx.x]);
  for (int d = threadIdx.x; d < D; d += 256) grad[(long)i * D + d] = (sG[0][d] + sG[1][d]) + (sG[2][d] + sG[3][d]);
}

// out4 = {mmd, K(p,p), K(z,z), K(p,z)};  norm = 1 (imq: plain sums) or 1/N^2 (rbf: means); grad *= norm
__global__ __launch_bounds__(256) void mmd_finish_kernel(const float* __restrict__ part, int N, int D, float norm, float w_pp,
                                                         float w_zz, float w_pz, float* __restrict__ grad, float* __restrict__ out) {
  __shared__ double smd[3][4];
  if (blockIdx.x == 0) {
    double s[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < N; i += 256)
      for (int t = 0; t < 3; ++t) s[t] += (double)part[(long)i * 3 + t];
    for (int t = 0; t < 3; ++t) {
      s[t] = wave_sum_d(s[t]);
      if ((threadIdx.x & 63) == 0) smd[t][threadIdx.x >> 6] = s[t];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float k[3];
      for (int t = 0; t < 3; ++t) k[t] = (float)((smd[t][0] + smd[t][1] + smd[t][2] + smd[t][3]) * (double)norm);
      out[0] = w_pp * k[0] + w_zz * k[1] - 2.f * w_pz * k[2];
      out[1] = k[0];
      out[2] = k[1];
      out[3] = k[2];
    }
  }
  if (norm != 1.f) {
    const long n = (long)N * D;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) grad[e] *= norm;
  }
}

size_t mmd_workspace_floats(int N) { return (size_t)N * 3; }

int launch_mmd_forward(const float* z, const float* p, int N, int D, int kind, float c, float eps, float w_pp, float w_zz,
                       float w_pz, float* out4, float* grad, float* ws, size_t ws_bytes, hipStream_t st) {
  if (D > 64 * kMmdMaxPerLane) return kErrBadArg;
  if (ws_bytes / sizeof(float) < mmd_workspace_floats(N)) return kErrWorkspace;
  {
    ProfScope ps("mmd_rows_kernel", st, 9.0 * (double)N * N * D, 8.0 * (double)N * D);
#define MMD_LAUNCH(NPL_)                                                                                                   \
  do {                                                                                                                     \
    if (kind == 0) hipLaunchKernelGGL((mmd_rows_kernel<NPL_, 0>), dim3(N), dim3(256), 0, st, z, p, N, D, c, eps, w_zz, w_pz, ws, grad); \
    else hipLaunchKernelGGL((mmd_rows_kernel<NPL_, 1>), dim3(N), dim3(256), 0, st, z, p, N, D, c, eps, w_zz, w_pz, ws, grad);           \
  } while (0)
    if (D <= 64) MMD_LAUNCH(1);
    else if (D <= 128) MMD_LAUNCH(2);
    else if (D <= 256) MMD_LAUNCH(4);
    else MMD_LAUNCH(kMmdMaxPerLane);
#undef MMD_LAUNCH
  }
  CTVAE_LAUNCH_CHECK();
  const float norm = kind == 0 ? 1.f : 1.f / ((float)N * (float)N);
  long blocks = kind == 0 ? 1 : ((long)N * D + 255) / 256;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(mmd_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ws, N, D, norm, w_pp, w_zz, w_pz, grad, out4);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
