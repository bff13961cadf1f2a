// Multi-codebook vector quantiser (mcq_vae.py:7-137) on NHWC latents [P = B*H*W][D].
//
//  * index search (mcq_vae.py:26-39): dist = (|x|^2 + |e_k|^2) - 2 x.e_k in the reference's expanded
//    form, first-min arg-min.  Codebook staged in LDS (odd row stride -> conflict-free), one wavefront
//    per latent row, lane == code, wave shuffle arg-min on (dist, k).
//  * lookup + loss (mcq_vae.py:41-64): q = E[idx]; out = x + (q - x) (same rounding as the reference's
//    straight-through expression); per-codebook sum (q-x)^2 -> vq_loss = sum_i (beta*mse_i + mse_i).
//  * backward: straight-through pass of g_q plus beta*2(x-q)/N into the latents; embedding-loss
//    gradient 2(q-x)/N scattered per code, computed without atomics (one workgroup per code).
//  * the reference's slice quirk is reproduced: codebook i reads channels [i, i+D/C) (mcq_vae.py:104,117).
// All HBM/L2-bound; algorithmic bytes = latents read + quantized written + indices.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

// grid (row blocks, C); block 256 = 4 waves, each wave walks rows
__global__ __launch_bounds__(256) void vq_inds_kernel(const float* __restrict__ lat, const float* __restrict__ cb,
                                                      long long* __restrict__ inds, int P, int D, int K, int Dc, int C,
                                                      int HW, int rows_per_block) {
  extern __shared__ float smem[];
  const int ld = Dc | 1;                 // odd stride
  float* sE = smem;                      // [K][ld]
  float* sEE = sE + (size_t)K * ld;      // [K]
  float* sX = sEE + K;                   // [4][Dc]
  const int cbi = blockIdx.y;
  const float* E = cb + (size_t)cbi * K * Dc;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < K * Dc; i += 256) sE[(i / Dc) * ld + (i % Dc)] = E[i];
  __syncthreads();
  for (int k = tid; k < K; k += 256) {
    float s = 0.f;
    for (int d = 0; d < Dc; ++d) s += sE[k * ld + d] * sE[k * ld + d];
    sEE[k] = s;
  }
  __syncthreads();
  float* myX = sX + wave * Dc;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  for (int p = r0 + wave; p < r1; p += 4) {
    const float* x = lat + (size_t)p * D + cbi;  // slice offset i, not i*Dc (reference quirk)
    for (int d = lane; d < Dc; d += 64) myX[d] = x[d];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float xx = 0.f;
    for (int d = 0; d < Dc; ++d) xx += myX[d] * myX[d];
    float best = 3.4e38f;
    int bestk = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
      float dot = 0.f;
      for (int d = 0; d < Dc; ++d) dot += myX[d] * sE[k * ld + d];
      float dist = (xx + sEE[k]) - 2.f * dot;
      if (dist < best) { best = dist; bestk = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ob = __shfl_xor(best, o, 64);
      int ok = __shfl_xor(bestk, o, 64);
      if (ob < best || (ob == best && ok < bestk)) { best = ob; bestk = ok; }
    }
    if (lane == 0) {
      int b = p / HW, hw = p - b * HW;
      inds[((size_t)b * C + cbi) * HW + hw] = (long long)bestk;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Register-blocked form for Dc in {32, 64, 128} and K <= 64: lane k keeps ITS code row E[k][:] in registers (the generic
// kernel re-reads it from LDS for every latent row: two LDS reads per multiply-add), the latent row is broadcast from LDS
// four elements at a time.  Same arithmetic and summation order per (row, code) as the generic kernel.
template <int DC>
__global__ __launch_bounds__(256) void vq_inds_reg_kernel(const float* __restrict__ lat, const float* __restrict__ cb,
                                                          long long* __restrict__ inds, int P, int D, int K, int C, int HW,
                                                          int rows_per_block) {
  // The code rows reach the lanes through LDS: the workgroup reads the codebook coalesced (K * DC / 256 loads per thread) and lane k
  // takes row k with conflict-free reads (row stride DC + 1).  Loaded directly, `e[d] = E[k * DC + d]` is DC load instructions of 64
  // different cache lines each -- 8192 line requests per wave for a 32 KB codebook, most of this kernel's time at 4 rows per wave.
  __shared__ float sE[64 * (DC + 1)];
  __shared__ __attribute__((aligned(16))) float sX[4][2][DC];
  const int cbi = blockIdx.y;
  const float* E = cb + (size_t)cbi * K * DC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    constexpr int NL = 64 * DC / 256;
    float t[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int e = tid + 256 * j;
      t[j] = E[e < K * DC ? e : K * DC - 1];               // clamped index, unconditional loads (codebooks sit at 4-byte-aligned offsets)
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int e = tid + 256 * j;
      sE[(e / DC) * (DC + 1) + (e % DC)] = t[j];
    }
  }
  float* myX = sX[wave][0];
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  // the latent row of the NEXT position is loaded while the current one is scored (two wave-private LDS rows)
  constexpr int XL = (DC + 63) / 64;
  float xn[XL];
  auto load_row = [&](int p) {
    const float* x = lat + (size_t)(p < r1 ? p : r1 - 1) * D + cbi;      // slice offset i, not i*Dc (reference quirk)
#pragma unroll
    for (int j = 0; j < XL; ++j) {
      const int d = lane + 64 * j;
      xn[j] = x[d < DC ? d : DC - 1];
    }
  };
  if (r0 + wave < r1) load_row(r0 + wave);
  __syncthreads();
  float e[DC];
  float ee = 0.f;
  {
    const float* row = sE + (lane < K ? lane : 0) * (DC + 1);
#pragma unroll
    for (int d = 0; d < DC; ++d) e[d] = row[d];
#pragma unroll
    for (int d = 0; d < DC; ++d) ee += e[d] * e[d];
  }
  int buf = 0;
  for (int p = r0 + wave; p < r1; p += 4) {
    float* cx = myX + buf * DC;
#pragma unroll
    for (int j = 0; j < XL; ++j) {
      const int d = lane + 64 * j;
      if (d < DC) cx[d] = xn[j];
    }
    load_row(p + 4);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float xx = 0.f, dot = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < DC; d4 += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(cx + d4);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xx += v[u] * v[u];
        dot += v[u] * e[d4 + u];
      }
    }
    float best = lane < K ? (xx + ee) - 2.f * dot : 3.4e38f;
    int bestk = lane < K ? lane : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ob = __shfl_xor(best, o, 64);
      int ok = __shfl_xor(bestk, o, 64);
      if (ob < best || (ob == best && ok < bestk)) { best = ob; bestk = ok; }
    }
    {                                                      // every lane holds the winner and stores it (same 8 bytes: one request); under
      int b = p / HW, hw = p - b * HW;                       // `if (lane == 0)` the store is a branch and the next row's load waits for it
      inds[((size_t)b * C + cbi) * HW + hw] = (long long)bestk;
    }
    buf ^= 1;
  }
}

// out[p][i*Dc+d] = x + (E_i[idx][d] - x),  x = lat[p][i+d];  part[blk][i] = sum (q-x)^2
__global__ __launch_bounds__(256) void vq_lookup_kernel(const float* __restrict__ lat, const float* __restrict__ cb,
                                                        const long long* __restrict__ inds, float* __restrict__ out,
                                                        float* __restrict__ part, int P, int D, int K, int Dc, int C, int HW) {
  __shared__ float sm[4];
  const long n = (long)P * D;
  const long stride = (long)gridDim.x * 256;
  // every thread owns fixed columns when stride % D == 0 is not guaranteed -> accumulate per codebook generally
  float acc[kMaxCls * 2];  // up to 8 codebooks
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
    const int col = (int)(e % D);
    const long p = e / D;
    const int i = col / Dc, d = col - i * Dc;
    const int b = (int)(p / HW), hw = (int)(p - (long)b * HW);
    const long long idx = inds[((size_t)b * C + i) * HW + hw];
    const float x = lat[p * D + i + d];
    const float q = cb[((size_t)i * K + (size_t)idx) * Dc + d];
    const float df = q - x;
    out[e] = x + df;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j == i) acc[j] += df * df;
  }
  for (int i = 0; i < C; ++i) {
    float s = block_sum_256(acc[i], sm);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * C + i] = s;
  }
}

// vq_loss = sum_i ( mse_i*beta + mse_i ), mse_i = sum_i/(P*Dc)
__global__ __launch_bounds__(64) void vq_loss_finish_kernel(const float* __restrict__ part, int nblocks, int C, double inv_n,
                                                            float beta, float* __restrict__ out) {
  float total = 0.f;
  for (int i = 0; i < C; ++i) {
    double s = 0.0;
    for (int b0 = 0; b0 < nblocks; b0 += 8 * 64) {        // eight loads in flight, added in the order of the plain loop
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + threadIdx.x + 64 * u;
        const float v = part[(size_t)(b < nblocks ? b : nblocks - 1) * C + i];      // unconditional load of a clamped index: a predicated
        t[u] = b < nblocks ? v : 0.f;                                               // load is a branch and is waited for on the spot
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)t[u];
    }
    s = wave_sum_d(s);
    float mse = (float)(s * inv_n);
    total = total + (mse * beta + mse);
  }
  if (threadIdx.x == 0) out[0] = total;
}

// g_lat[p][j] = sum_{i: 0<=j-i<Dc} ( g_q[p][i*Dc + j-i] + g_vq*beta*2*(x_j - E_i[idx][j-i])/(P*Dc) )
__global__ __launch_bounds__(256) void vq_bwd_latents_kernel(const float* __restrict__ gq, const float* __restrict__ gvq,
                                                             const float* __restrict__ lat, const float* __restrict__ cb,
                                                             const long long* __restrict__ inds, float* __restrict__ glat,
                                                             int P, int D, int K, int Dc, int C, int HW, float beta) {
  const long n = (long)P * D;
  const long stride = (long)gridDim.x * 256;
  const float sc = (gvq != nullptr ? gvq[0] : 0.f) * beta * 2.f / ((float)P * (float)Dc);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
    const int j = (int)(e % D);
    const long p = e / D;
    const int b = (int)(p / HW), hw = (int)(p - (long)b * HW);
    const float x = lat[e];
    float g = 0.f;
    int ilo = j - Dc + 1;
    if (ilo < 0) ilo = 0;
    int ihi = j < C - 1 ? j : C - 1;
    for (int i = ilo; i <= ihi; ++i) {
      const int d = j - i;
      const long long idx = inds[((size_t)b * C + i) * HW + hw];
      const float q = cb[((size_t)i * K + (size_t)idx) * Dc + d];
      g += (gq != nullptr ? gq[p * D + i * Dc + d] : 0.f) + sc * (x - q);
    }
    glat[e] = g;
  }
}

// dE_i[k][d] (+)= g_vq * 2 * (cnt*E_i[k][d] - sum_{p: idx=k} x[p][i+d]) / (P*Dc)      grid (K, C, S), block 256
// S > 1: slice z covers positions [z*Ps, (z+1)*Ps) and writes its (linear) share to part[z][C*K*Dc]; the shares are
// summed in a fixed order by vq_cb_reduce_kernel (64 codes x 1 codebook alone would use 64 of the 256 CUs).
__global__ __launch_bounds__(256) void vq_bwd_codebook_kernel(const float* __restrict__ gvq, const float* __restrict__ lat,
                                                              const float* __restrict__ cb, const long long* __restrict__ inds,
                                                              float* __restrict__ dcb, int P, int D, int K, int Dc, int C,
                                                              int HW, int accumulate, float* __restrict__ part, int Ps) {
  __shared__ float sAcc[4][256];
  __shared__ int sCnt[4];
  const int k = blockIdx.x, i = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};  // d = lane + 64*u, Dc <= 256
  int cnt = 0;
  const int p_lo = part != nullptr ? blockIdx.z * Ps : 0;
  const int p_hi = part != nullptr ? (p_lo + Ps < P ? p_lo + Ps : P) : P;
  for (int base = p_lo + wave * 64; base < p_hi; base += 256) {
    const int p = base + lane;
    bool hit = false;
    if (p < p_hi) {
      int b = p / HW, hw = p - b * HW;
      hit = inds[((size_t)b * C + i) * HW + hw] == (long long)k;
    }
    unsigned long long mask = __ballot(hit);
    cnt += __popcll(mask);
    while (mask) {
      int l = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const float* x = lat + (size_t)(base + l) * D + i;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int d = lane + 64 * u;
        if (d < Dc) acc[u] += x[d];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) sAcc[wave][lane + 64 * u] = acc[u];
  if (lane == 0) sCnt[wave] = cnt;
  __syncthreads();
  if (tid < Dc) {
    const float sx = ((sAcc[0][tid] + sAcc[1][tid]) + sAcc[2][tid]) + sAcc[3][tid];
    const float c = (float)(sCnt[0] + sCnt[1] + sCnt[2] + sCnt[3]);
    const size_t o = ((size_t)i * K + k) * Dc + tid;
    const float sc = (gvq != nullptr ? gvq[0] : 0.f) * 2.f / ((float)P * (float)Dc);
    const float g = sc * (c * cb[o] - sx);
    if (part != nullptr) part[(size_t)blockIdx.z * ((size_t)C * K * Dc) + o] = g;
    else dcb[o] = (accumulate ? dcb[o] : 0.f) + g;
  }
}

// Same result, organised by POSITIONS instead of by codes: workgroup = (slice of positions, codebook i); a half-wave owns
// one position per step (lane = d < 32) and adds its latent row into ITS OWN table sAcc[wave][half][k][d] in LDS -- a
// plain read-modify-write, no atomics: one owner per table, positions in a fixed order, so the sums are bit-reproducible.
// The scan-for-my-code form above walks every index once per code and then gathers its hits one dependent 128-byte
// load at a time (377 us for P = 16384, K = 64, C = 4); here every index and every latent element is read exactly once.
// Requires Dc <= 32 and K <= 128 (8 tables of K x 32 floats + counts in LDS).  grid (S, C), part[z][C*K*Dc] as above.
__global__ __launch_bounds__(256) void vq_bwd_codebook_pos_kernel(const float* __restrict__ gvq, const float* __restrict__ lat,
                                                                  const float* __restrict__ cb,
                                                                  const long long* __restrict__ inds, float* __restrict__ part,
                                                                  int P, int D, int K, int Dc, int C, int HW, int Ps) {
  extern __shared__ float vq_smem[];
  float* sAcc = vq_smem;                               // [8][K][32]
  float* sCnt = sAcc + 8 * K * 32;                     // [8][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, d = lane & 31;
  const int i = blockIdx.y;
  for (int e = tid; e < 8 * K * 32 + 8 * K; e += 256) vq_smem[e] = 0.f;
  __syncthreads();
  float* tAcc = sAcc + (size_t)(wave * 2 + half) * K * 32;
  float* tCnt = sCnt + (wave * 2 + half) * K;
  const int p_lo = blockIdx.x * Ps;
  const int p_hi = p_lo + Ps < P ? p_lo + Ps : P;
  constexpr int U = 4;                                 // positions in flight per half-wave
  for (int base = p_lo + wave * 2 + half; base < p_hi; base += 8 * U) {
    int kk[U];
    float vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = base + 8 * u;
      kk[u] = -1;
      vv[u] = 0.f;
      if (p < p_hi) {
        const int b = p / HW, hw = p - b * HW;
        kk[u] = (int)inds[((size_t)b * C + i) * HW + hw];
        if (d < Dc) vv[u] = lat[(size_t)p * D + i + d];    // codebook i reads latent columns i .. i+Dc-1 (reference slicing)
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (kk[u] >= 0) {                                // uniform over the half-wave
        tAcc[kk[u] * 32 + d] += vv[u];
        if (d == 0) tCnt[kk[u]] += 1.f;
      }
    }
  }
  __syncthreads();
  const float sc = (gvq != nullptr ? gvq[0] : 0.f) * 2.f / ((float)P * (float)Dc);
  for (int e = tid; e < K * Dc; e += 256) {
    const int k = e / Dc, dd = e - k * Dc;
    float sx = 0.f, c = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      sx += sAcc[((size_t)t * K + k) * 32 + dd];
      c += sCnt[t * K + k];
    }
    const size_t o = ((size_t)i * K + k) * Dc + dd;
    part[(size_t)blockIdx.x * ((size_t)C * K * Dc) + o] = sc * (c * cb[o] - sx);
  }
}

// The same for wide codebook slices (32 < Dc <= 128: CT-MCQ-VAE's single 128-wide codebook): a whole wave owns a position per
// step (lane = d and d + 64) and a table per wave, sAcc[wave][k][128] -- 4 x K x 128 floats, 128 KB at K = 64.  The
// scan-per-code kernel above took 82 us on that model's 8192 positions.
__global__ __launch_bounds__(256) void vq_bwd_codebook_posw_kernel(const float* __restrict__ gvq, const float* __restrict__ lat,
                                                                   const float* __restrict__ cb,
                                                                   const long long* __restrict__ inds, float* __restrict__ part,
                                                                   int P, int D, int K, int Dc, int C, int HW, int Ps) {
  extern __shared__ float vq_smem[];
  float* sAcc = vq_smem;                               // [4][K][128]
  float* sCnt = sAcc + 4 * K * 128;                    // [4][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.y;
  for (int e = tid; e < 4 * K * 128 + 4 * K; e += 256) vq_smem[e] = 0.f;
  __syncthreads();
  float* tAcc = sAcc + (size_t)wave * K * 128;
  float* tCnt = sCnt + wave * K;
  const int p_lo = blockIdx.x * Ps;
  const int p_hi = p_lo + Ps < P ? p_lo + Ps : P;
  constexpr int U = 4;                                 // positions in flight per wave
  for (int base = p_lo + wave; base < p_hi; base += 4 * U) {
    int kk[U];
    float v0[U], v1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = base + 4 * u;
      kk[u] = -1;
      v0[u] = v1[u] = 0.f;
      if (p < p_hi) {
        const int b = p / HW, hw = p - b * HW;
        kk[u] = (int)inds[((size_t)b * C + i) * HW + hw];
        if (lane < Dc) v0[u] = lat[(size_t)p * D + i + lane];
        if (lane + 64 < Dc) v1[u] = lat[(size_t)p * D + i + lane + 64];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (kk[u] >= 0) {                                // uniform over the wave
        tAcc[kk[u] * 128 + lane] += v0[u];
        tAcc[kk[u] * 128 + 64 + lane] += v1[u];
        if (lane == 0) tCnt[kk[u]] += 1.f;
      }
    }
  }
  __syncthreads();
  const float sc = (gvq != nullptr ? gvq[0] : 0.f) * 2.f / ((float)P * (float)Dc);
  for (int e = tid; e < K * Dc; e += 256) {
    const int k = e / Dc, dd = e - k * Dc;
    float sx = 0.f, c = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      sx += sAcc[((size_t)t * K + k) * 128 + dd];
      c += sCnt[t * K + k];
    }
    const size_t o = ((size_t)i * K + k) * Dc + dd;
    part[(size_t)blockIdx.x * ((size_t)C * K * Dc) + o] = sc * (c * cb[o] - sx);
  }
}

__global__ __launch_bounds__(256) void vq_cb_reduce_kernel(const float* __restrict__ part, float* __restrict__ dcb, int n, int S,
                                                          int accumulate) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  float v = accumulate ? dcb[o] : 0.f;
  int z = 0;
  for (; z + 8 <= S; z += 8) {        // eight slabs' loads in flight, added in slab order (the same sum as one by one; as a plain loop
    float t[8];                       // this was S memory round trips in a row per thread: 31 us for 32 workgroups at S = 128)
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = part[(size_t)(z + u) * n + o];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; z < S; ++z) v += part[(size_t)z * n + o];
  dcb[o] = v;
}

size_t vq_workspace_floats(int C) { return (size_t)1024 * C; }

int launch_vq_inds(const float* lat, const float* cb, long long* inds, int B, int HW, int D, int K, int C, hipStream_t st) {
  if (D % C != 0 || C > 8) return kErrBadArg;
  const int Dc = D / C, P = B * HW;
  const size_t smem = ((size_t)K * (Dc | 1) + K + 4 * Dc) * sizeof(float);
  if (smem > 150 * 1024) return kErrBadArg;
  int blocks = ceil_div(P, 64);
  if (blocks > 512) blocks = 512;
  const int rpb = ceil_div(P, blocks);
  blocks = ceil_div(P, rpb);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vq_inds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  ProfScope ps("vq_inds_kernel", st, 2.0 * (double)P * K * D, 4.0 * (double)P * D + 8.0 * (double)P * C);
  if (K <= 64 && (Dc == 32 || Dc == 64 || Dc == 128)) {
    // rows per workgroup: the codebook is staged per workgroup (measured at 16 / 32 / 64 rows: 40.8 / 38.2 / 36.9 us for Dc = 128,
    // 25.2 / 24.7 / 25.8 us for Dc = 32)
    static const int rpw_env = [] { const char* e = getenv("CTVAE_VQ_ROWS"); return e ? atoi(e) : 0; }();   // diagnostic override
    const int rpw = rpw_env ? rpw_env : (Dc == 128 ? 64 : 16);
    int nb = ceil_div(P, rpw);
    if (nb > 2048) nb = 2048;
    const int rp = ceil_div(P, nb);
    nb = ceil_div(P, rp);
    if (Dc == 32) hipLaunchKernelGGL(vq_inds_reg_kernel<32>, dim3(nb, C), dim3(256), 0, st, lat, cb, inds, P, D, K, C, HW, rp);
    else if (Dc == 64) hipLaunchKernelGGL(vq_inds_reg_kernel<64>, dim3(nb, C), dim3(256), 0, st, lat, cb, inds, P, D, K, C, HW, rp);
    else hipLaunchKernelGGL(vq_inds_reg_kernel<128>, dim3(nb, C), dim3(256), 0, st, lat, cb, inds, P, D, K, C, HW, rp);
    CTVAE_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(vq_inds_kernel, dim3(blocks, C), dim3(256), smem, st, lat, cb, inds, P, D, K, Dc, C, HW, rpb);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_vq_lookup(const float* lat, const float* cb, const long long* inds, float* out, float* vq_loss, float beta, int B,
                     int HW, int D, int K, int C, float* ws, size_t ws_bytes, hipStream_t st) {
  if (D % C != 0 || C > 8) return kErrBadArg;
  if (ws_bytes / sizeof(float) < vq_workspace_floats(C)) return kErrWorkspace;
  const int Dc = D / C, P = B * HW;
  long blocks = ((long)P * D + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  ProfScope ps("vq_lookup+loss_finish", st, 0.0, 8.0 * (double)P * D + 8.0 * (double)P * C + 4.0 * (double)K * D);
  hipLaunchKernelGGL(vq_lookup_kernel, dim3((unsigned)blocks), dim3(256), 0, st, lat, cb, inds, out, ws, P, D, K, Dc, C, HW);
  CTVAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(vq_loss_finish_kernel, dim3(1), dim3(64), 0, st, ws, (int)blocks, C, 1.0 / ((double)P * Dc), beta, vq_loss);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_vq_backward(const float* gq, const float* gvq, const float* lat, const float* cb, const long long* inds,
                       float* glat, float* dcb, int accumulate, float beta, int B, int HW, int D, int K, int C, float* ws,
                       size_t ws_bytes, hipStream_t st) {
  if (D % C != 0 || C > 8 || D / C > 256) return kErrBadArg;
  const int Dc = D / C, P = B * HW;
  if (glat != nullptr) {
    long blocks = ((long)P * D + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(vq_bwd_latents_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gq, gvq, lat, cb, inds, glat, P, D, K,
                       Dc, C, HW, beta);
    CTVAE_LAUNCH_CHECK();
  }
  if (dcb != nullptr) {
    const size_t n_all = (size_t)C * K * Dc;
    if (Dc <= 32 && K <= 128 && ws != nullptr) {
      // position-major kernel: ~256 positions per workgroup
      int S = ceil_div(P, 256);
      if (S > 512) S = 512;
      while (S > 1 && (size_t)S * n_all > ws_bytes / sizeof(float)) --S;
      if ((size_t)S * n_all <= ws_bytes / sizeof(float)) {
        const int Ps = ceil_div(P, S);
        S = ceil_div(P, Ps);
        const size_t smem = ((size_t)8 * K * 32 + 8 * K) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vq_bwd_codebook_pos_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          attr_set = true;
        }
        ProfScope ps("vq_bwd_codebook_pos_kernel", st, 0.0, 4.0 * (double)P * D + 8.0 * (double)P * C + 4.0 * S * n_all);
        hipLaunchKernelGGL(vq_bwd_codebook_pos_kernel, dim3(S, C), dim3(256), smem, st, gvq, lat, cb, inds, ws, P, D, K, Dc, C,
                           HW, Ps);
        CTVAE_LAUNCH_CHECK();
        hipLaunchKernelGGL(vq_cb_reduce_kernel, dim3(ceil_div((int)n_all, 256)), dim3(256), 0, st, ws, dcb, (int)n_all, S,
                           accumulate);
        CTVAE_LAUNCH_CHECK();
        return 0;
      }
    }
    if (Dc <= 128 && (size_t)(4 * K * 128 + 4 * K) * sizeof(float) <= 150 * 1024 && ws != nullptr) {
      // wide slices: a wave per position, a table per wave; ~64 positions per workgroup
      int S = ceil_div(P, 64);
      if (S > 512) S = 512;
      while (S > 1 && (size_t)S * n_all > ws_bytes / sizeof(float)) --S;
      if ((size_t)S * n_all <= ws_bytes / sizeof(float)) {
        const int Ps = ceil_div(P, S);
        S = ceil_div(P, Ps);
        const size_t smem = ((size_t)4 * K * 128 + 4 * K) * sizeof(float);
        static bool attr_set_w = false;
        if (!attr_set_w) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vq_bwd_codebook_posw_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          attr_set_w = true;
        }
        ProfScope ps("vq_bwd_codebook_posw_kernel", st, 0.0, 4.0 * (double)P * D + 8.0 * (double)P * C + 4.0 * S * n_all);
        hipLaunchKernelGGL(vq_bwd_codebook_posw_kernel, dim3(S, C), dim3(256), smem, st, gvq, lat, cb, inds, ws, P, D, K, Dc, C,
                           HW, Ps);
        CTVAE_LAUNCH_CHECK();
        hipLaunchKernelGGL(vq_cb_reduce_kernel, dim3(ceil_div((int)n_all, 256)), dim3(256), 0, st, ws, dcb, (int)n_all, S,
                           accumulate);
        CTVAE_LAUNCH_CHECK();
        return 0;
      }
    }
    // enough slices of the position range for ~1024 workgroups (each slice >= 512 positions), if scratch is available
    int S = 1024 / (K * C);
    if (S > P / 512) S = P / 512;
    const size_t n = (size_t)C * K * Dc;
    if (ws == nullptr || S < 2 || ws_bytes / sizeof(float) < (size_t)S * n) S = 1;
    const int Ps = ceil_div(P, S);
    ProfScope ps("vq_bwd_codebook_kernel", st, 0.0, 4.0 * (double)P * D + 8.0 * (double)P * C * (double)K / S);
    hipLaunchKernelGGL(vq_bwd_codebook_kernel, dim3(K, C, S), dim3(256), 0, st, gvq, lat, cb, inds, dcb, P, D, K, Dc, C, HW,
                       accumulate, S > 1 ? ws : nullptr, Ps);
    CTVAE_LAUNCH_CHECK();
    if (S > 1) {
      hipLaunchKernelGGL(vq_cb_reduce_kernel, dim3(ceil_div((int)n, 256)), dim3(256), 0, st, ws, dcb, (int)n, S, accumulate);
      CTVAE_LAUNCH_CHECK();
    }
  }
  return 0;
}

}  // namespace ctvae
