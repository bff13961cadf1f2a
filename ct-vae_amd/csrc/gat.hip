// Attention scores of the dense GATv2 layers inside CausalTransition.graph_transitioner (ct_mcq_vae.py:103-114,
// `gnn.GATv2Conv(..., heads, edge_dim=1)` on B x (N+1)-node dense weighted graphs):
//
//     S[b,h,r,c] = sum_k att[h,k] * leaky_relu(xl[b,r,h,k] + xr[b,c,h,k] + attr[b,r,c] * we[h,k], 0.2)
//
// for every source r / target c / head h.  As torch ops this materialises a [B,N,N,C] tensor per head several times
// (216 MB per head at B = 128) and was 70 % of a CT-MCQ-VAE step; here nothing of size N*N*C touches memory.
//   gat_score_kernel<0>  thread = (r,c) pairs of one (b,h), xl/xr of that head staged transposed in LDS, k-sum in-thread
//   gat_score_kernel<1>  same walk, returns T = sum_k att*we*lrelu'(.) so that d attr = sum_h g*T (the caller's product)
//   gat_score_bwd_kernel thread = (channel k, half of the sources) of one (b,h): every d xl[b,r,h,k] has one producer, the
//                        two halves of d xr[b,c,h,k] meet in a shuffle (no atomics, bit-reproducible); d att / d we
//                        leave as per-sample partials.
// Masked softmax over sources and the alpha-weighted aggregation stay as small torch ops on [B,H,N,N].
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

namespace {

constexpr int PMAX = 17;    // (r,c) pairs per thread: N*N <= 17*256 -> N <= 65 (64 latent nodes + the action node)
constexpr int NB = 68;      // backward: rows padded to a multiple of 4 (b128 broadcast reads)

template <int MODE, int NPC>   // NPC: compile-time row length N|1 (65 for the model's 64 + 1 nodes) or 0 = runtime
__global__ __launch_bounds__(256) void gat_score_kernel(const float* __restrict__ xl, const float* __restrict__ xr,
                                                       const float* __restrict__ attr, const float* __restrict__ we,
                                                       const float* __restrict__ att, float* __restrict__ out, int N, int H,
                                                       int C, float slope) {
  extern __shared__ float smem[];
  const int NP = NPC ? NPC : (N | 1);   // odd row length: conflict-free for consecutive c, broadcast for equal r
  float* sL = smem;                     // [C][NP]
  float* sR = smem + C * NP;            // [C][NP]
  float* sWe = sR + C * NP;             // [C]
  float* sAt = sWe + C;                 // [C]
  const int tid = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
  for (int e = tid; e < N * C; e += 256) {          // coalesced along k, transposed into LDS
    const int n = e / C, k = e - n * C;
    const long src = (((long)b * N + n) * H + h) * C + k;
    sL[k * NP + n] = xl[src];
    sR[k * NP + n] = xr[src];
  }
  for (int k = tid; k < C; k += 256) {
    sWe[k] = we[h * C + k];
    sAt[k] = att[h * C + k];
  }
  // pairs are processed two at a time in packed f32x2 registers (v_pk_add/fma/mul_f32: two lanes of arithmetic per
  // instruction; only the max has no packed form) -- this loop is pure VALU work
  constexpr int PH = (PMAX + 1) / 2;
  int rr[2 * PH], cc[2 * PH];
  f32x2 at[PH], acc[PH];
#pragma unroll
  for (int q = 0; q < 2 * PH; ++q) {
    const int p = tid + 256 * q;
    const bool ok = q < PMAX && p < N * N;
    const int r = ok ? p / N : 0, c = ok ? p - r * N : 0;
    rr[q] = r;
    cc[q] = c;
    at[q >> 1][q & 1] = ok ? attr[((long)b * N + r) * N + c] : 0.f;
    acc[q >> 1][q & 1] = 0.f;
  }
  __syncthreads();
  // per-pair LDS addresses advance by one row (NP floats) per k: kept in registers and bumped once per 4 k (the row
  // stride of the 4 unrolled steps is an immediate offset when N = 65, the model's 64 latent nodes + 1 action node)
  for (int k0 = 0; k0 < C; k0 += 4) {
#pragma unroll
   for (int kk = 0; kk < 4; ++kk) {
    const int k = k0 + kk;
    if (k < C) {
    const float wk = sWe[k], ak = sAt[k];
    const int ro = kk * NP;                       // an immediate LDS offset when NPC is given
    const f32x2 wk2 = {wk, wk}, sl2 = {slope, slope};
    const f32x2 ak2 = MODE == 0 ? f32x2{ak, ak} : f32x2{ak * wk, ak * wk};
#pragma unroll
    for (int h = 0; h < PH; ++h) {
      const f32x2 lv = {sL[rr[2 * h] + ro], sL[rr[2 * h + 1] + ro]};
      const f32x2 rv = {sR[cc[2 * h] + ro], sR[cc[2 * h + 1] + ro]};
      const f32x2 m = (lv + rv) + at[h] * wk2;
      if (MODE == 0) {
        const f32x2 ms = m * sl2;
        const f32x2 lr = {fmaxf(m[0], ms[0]), fmaxf(m[1], ms[1])};       // slope < 1
        acc[h] += ak2 * lr;
      } else {
        const f32x2 d = {m[0] > 0.f ? 1.f : slope, m[1] > 0.f ? 1.f : slope};
        acc[h] += ak2 * d;
      }
    }
    }
   }
#pragma unroll
    for (int q = 0; q < 2 * PH; ++q) { rr[q] += 4 * NP; cc[q] += 4 * NP; }     // row index -> running LDS element index
  }
#pragma unroll
  for (int q = 0; q < PMAX; ++q) {
    const int p = tid + 256 * q;
    if (p < N * N) out[(((long)b * H + h) * N) * N + p] = acc[q >> 1][q & 1];
  }
}

// grid (H, B), 256 threads: thread = (channel k < C, half of the source range).  Two lanes of a pair share k and split
// the sources r, which halves the registers holding xl[r] / d xl[r] (34 + 34 instead of 68 + 68: three waves per SIMD
// instead of one); their partial sums over r meet through one shuffle.  g [B,H,N,N] (r,c).  d*_part [B][H][C].
constexpr int NH = NB / 2;   // 34 sources per lane

__global__ __launch_bounds__(256, 2) void gat_score_bwd_kernel(const float* __restrict__ xl, const float* __restrict__ xr,
                                                           const float* __restrict__ attr, const float* __restrict__ we,
                                                           const float* __restrict__ att, const float* __restrict__ g,
                                                           float* __restrict__ dxl, float* __restrict__ dxr,
                                                           float* __restrict__ datt_part, float* __restrict__ dwe_part, int N,
                                                           int H, int C, float slope) {
  __shared__ __attribute__((aligned(16))) float sG[NB][NB];   // [c][r]; read as float2 (r0 = 34 is 8-B aligned)
  __shared__ __attribute__((aligned(16))) float sA[NB][NB];
  const int tid = threadIdx.x, h = blockIdx.x, b = blockIdx.y, k = tid >> 1, half = tid & 1;
  const bool kok = k < C;
  const int r0 = half * NH;
  for (int e = tid; e < NB * NB; e += 256) {
    const int r = e / NB, c = e - r * NB;     // reads coalesced along c
    float gv = 0.f, av = 0.f;
    if (r < N && c < N) {
      gv = g[(((long)b * H + h) * N + r) * N + c];
      av = attr[((long)b * N + r) * N + c];
    }
    sG[c][r] = gv;
    sA[c][r] = av;
  }
  // sources are processed two at a time in packed f32x2 registers (NH = 34 is even): v_pk_add/mul/fma_f32
  f32x2 xlr[NH / 2], dl[NH / 2];
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int r = r0 + i;
    xlr[i >> 1][i & 1] = (kok && r < N) ? xl[(((long)b * N + r) * H + h) * C + k] : 0.f;
    dl[i >> 1][i & 1] = 0.f;
  }
  const float wk = kok ? we[h * C + k] : 0.f, ak = kok ? att[h * C + k] : 0.f;
  const f32x2 wk2 = {wk, wk}, ak2 = {ak, ak};
  f32x2 datt2 = {0.f, 0.f}, dwe2 = {0.f, 0.f};
  __syncthreads();
  float xr_next = kok ? xr[(((long)b * N) * H + h) * C + k] : 0.f;        // one target ahead: the load is off the chain
  for (int c = 0; c < N; ++c) {
    const float xrc = xr_next;
    if (c + 1 < N) xr_next = kok ? xr[(((long)b * N + c + 1) * H + h) * C + k] : 0.f;
    const f32x2 xrc2 = {xrc, xrc};
    f32x2 dr2 = {0.f, 0.f};
#pragma unroll
    for (int i2 = 0; i2 < NH / 2; ++i2) {
      const f32x2 g2 = *reinterpret_cast<const f32x2*>(&sG[c][r0 + 2 * i2]);   // two addresses per wave: broadcast reads
      const f32x2 a2 = *reinterpret_cast<const f32x2*>(&sA[c][r0 + 2 * i2]);
      const f32x2 m = (xlr[i2] + xrc2) + a2 * wk2;
      const f32x2 sl = {m[0] > 0.f ? 1.f : slope, m[1] > 0.f ? 1.f : slope};
      const f32x2 gs = g2 * sl;            // g * lrelu'(m)
      datt2 += gs * m;                     // g * lrelu(m)
      const f32x2 gd = gs * ak2;           // d m
      dl[i2] += gd;
      dr2 += gd;
      dwe2 += gd * a2;
    }
    float dr = dr2[0] + dr2[1];
    dr += __shfl_xor(dr, 1, 64);
    if (kok && half == 0) dxr[(((long)b * N + c) * H + h) * C + k] = dr;
  }
  float datt = datt2[0] + datt2[1], dwe = dwe2[0] + dwe2[1];
  datt += __shfl_xor(datt, 1, 64);
  dwe += __shfl_xor(dwe, 1, 64);
  if (kok) {
#pragma unroll
    for (int i = 0; i < NH; ++i)
      if (r0 + i < N) dxl[(((long)b * N + r0 + i) * H + h) * C + k] = dl[i >> 1][i & 1];
    if (half == 0) {
      datt_part[((long)b * H + h) * C + k] = datt;
      dwe_part[((long)b * H + h) * C + k] = dwe;
    }
  }
}

}  // namespace

int launch_gat_score(int mode, const float* xl, const float* xr, const float* attr, const float* we, const float* att, float* out,
                     int B, int N, int H, int C, float slope, hipStream_t st) {
  if (B <= 0 || N <= 0 || H <= 0 || C <= 0 || N * N > PMAX * 256 || slope >= 1.f) return kErrBadArg;
  const size_t smem = ((size_t)2 * C * (N | 1) + 2 * C) * 4;
  if (smem > 160 * 1024) return kErrBadArg;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gat_score_kernel<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gat_score_kernel<1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gat_score_kernel<0, 65>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gat_score_kernel<1, 65>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  ProfScope ps(mode == 0 ? "gat_score_kernel<0>" : "gat_score_kernel<1>", st, 4.0 * B * H * (double)N * N * C,
               4.0 * B * H * (2.0 * N * C + (double)N * N));
  const bool n65 = (N | 1) == 65;
  if (mode == 0 && n65) hipLaunchKernelGGL((gat_score_kernel<0, 65>), dim3(H, B), dim3(256), smem, st, xl, xr, attr, we, att, out, N, H, C, slope);
  else if (mode == 0) hipLaunchKernelGGL((gat_score_kernel<0, 0>), dim3(H, B), dim3(256), smem, st, xl, xr, attr, we, att, out, N, H, C, slope);
  else if (n65) hipLaunchKernelGGL((gat_score_kernel<1, 65>), dim3(H, B), dim3(256), smem, st, xl, xr, attr, we, att, out, N, H, C, slope);
  else hipLaunchKernelGGL((gat_score_kernel<1, 0>), dim3(H, B), dim3(256), smem, st, xl, xr, attr, we, att, out, N, H, C, slope);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_gat_score_backward(const float* xl, const float* xr, const float* attr, const float* we, const float* att,
                              const float* g, float* dxl, float* dxr, float* datt_part, float* dwe_part, int B, int N, int H,
                              int C, float slope, hipStream_t st) {
  if (B <= 0 || N <= 0 || N > NB - 3 || H <= 0 || C <= 0 || C > 128) return kErrBadArg;
  ProfScope ps("gat_score_bwd_kernel", st, 10.0 * B * H * (double)N * N * C, 4.0 * B * H * (4.0 * N * C + (double)N * N));
  hipLaunchKernelGGL(gat_score_bwd_kernel, dim3(H, B), dim3(256), 0, st, xl, xr, attr, we, att, g, dxl, dxr, datt_part, dwe_part,
                     N, H, C, slope);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
