// Grouped Linear on the causal-transition layer's node tensors: every sample is a block of 64 rows (the 64 latent nodes)
// and chooses, per output segment, WHICH weight matrix of a stacked bank multiplies it:
//
//     y[b, m, s*N + n] = sum_k x[b, m, k] * W_s[group_s[b]][n][k] + bias_s[group_s[b]][n]        m < 64, n < N, s < nseg
//
// This is what the reference reaches with one nn.Linear per group and Python-side splitting / merging of the batch
// (ct_mcq_vae.py:129-154: graph_discovers[0] for everybody and graph_discovers[1 + argmax(action)] per sample; :224-226: of the
// last GATv2Conv only head 0 and head 1 + argmax(action) are read, i.e. the rows of lin_l / lin_r of two heads).  As tensor ops
// that needs a gathered copy of the weights per sample ([B,800,128] = 52 MB at B = 128) and an index_put backward; here the
// group id is one scalar load per workgroup and the weights are read where they lie.  group_s == NULL: everybody uses
// matrix 0 (plain Linear: the first GATv2 layer's lin_l | lin_r, discoverer 0).
//
// All three directions are 64x64-tile f32 MFMA GEMMs (v_mfma_f32_32x32x2_f32, 4 waves = 2x2 blocks of 32x32, 32-deep
// chunks through LDS with k-slots permuted inside each 8-group so a lane's operand for four MFMAs is ONE 16-byte read):
//   glinear_fwd_kernel    tile = (sample, 64 output columns of one segment)
//   glinear_dgrad_kernel  tile = (sample, 64 input columns); reduction over every segment's columns
//   glinear_wgrad_kernel  tile = (group, 64 x 64 block of dW, slice of the batch): walks the samples of its slice that belong
//                         to the group, rows are the reduction; slices land in slabs that glinear_reduce_kernel sums in a
//                         fixed order (no atomics: bit-reproducible); the bias gradient rides along as column sums.
#include "common.hpp"
#include "glinear.hpp"
#include "prof.hpp"

namespace ctvae {

namespace {

constexpr int LP = 36;   // LDS row stride of a [64][32] operand tile: 16-byte fragment reads are conflict-free
constexpr int LQ = 68;   // LDS row stride of a [32][64] operand tile

__device__ __forceinline__ f32x4 ld4(const float* p, bool ok) { return ok ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f}; }

__global__ __launch_bounds__(256) void glinear_fwd_kernel(GLinArgs a) {
  __shared__ __attribute__((aligned(16))) float As[64 * LP];
  __shared__ __attribute__((aligned(16))) float Bs[64 * LP];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;
  const int ntiles = (a.N + 63) >> 6;
  const int seg = blockIdx.x / ntiles, n0 = (blockIdx.x - seg * ntiles) * 64, b = blockIdx.y;
  const int g = a.group[seg] ? a.group[seg][b] : 0;
  const float* Wg = a.W[seg] + (long)g * a.wgs[seg];
  const int ldw = a.ldw[seg];
  const int lrow = tid >> 2, lk = (tid & 3) * 8;
  const float* xrow = a.x + ((long)b * 64 + lrow) * a.ldx;
  const bool wok = n0 + lrow < a.N;
  const float* wrow = Wg + (long)(n0 + lrow) * ldw;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // the next chunk's global loads are in flight while the MFMAs of the current one run (register double buffer)
  f32x4 a0 = ld4(xrow + lk, lk < a.K), a1 = ld4(xrow + lk + 4, lk + 4 < a.K);
  f32x4 b0 = ld4(wrow + lk, wok && lk < a.K), b1 = ld4(wrow + lk + 4, wok && lk + 4 < a.K);
  for (int k0 = 0; k0 < a.K; k0 += 32) {
    __syncthreads();
    *reinterpret_cast<f32x4*>(&As[lrow * LP + lk]) = a0;
    *reinterpret_cast<f32x4*>(&As[lrow * LP + lk + 4]) = a1;
    *reinterpret_cast<f32x4*>(&Bs[lrow * LP + lk]) = b0;
    *reinterpret_cast<f32x4*>(&Bs[lrow * LP + lk + 4]) = b1;
    __syncthreads();
    const int k = k0 + 32 + lk;
    if (k0 + 32 < a.K) {
      a0 = ld4(xrow + k, k < a.K); a1 = ld4(xrow + k + 4, k + 4 < a.K);
      b0 = ld4(wrow + k, wok && k < a.K); b1 = ld4(wrow + k + 4, wok && k + 4 < a.K);
    }
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(&As[(wm * 32 + li) * LP + kg * 8 + 4 * lh]);
      const f32x4 bf = *reinterpret_cast<const f32x4*>(&Bs[(wn * 32 + li) * LP + kg * 8 + 4 * lh]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf[j], acc, 0, 0, 0);
    }
  }
  const int col = n0 + wn * 32 + li;
  if (col < a.N) {
    const float bv = a.bias[seg] ? a.bias[seg][(long)g * a.bgs[seg] + col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = wm * 32 + 8 * (i >> 2) + 4 * lh + (i & 3);
      a.y[((long)b * 64 + row) * a.ldy + seg * a.N + col] = acc[i] + bv;
    }
  }
}

// dx[b, m, k] = sum_s sum_n dy[b, m, s*N + n] * W_s[g_s][n][k]     (a.y = dy, a.x unused; dx [B*64][ldx])
__global__ __launch_bounds__(256) void glinear_dgrad_kernel(GLinArgs a, float* __restrict__ dx) {
  __shared__ __attribute__((aligned(16))) float As[64 * LP];   // dy[m][n chunk]
  __shared__ __attribute__((aligned(16))) float Bs[32 * LQ];   // W[n chunk][k tile]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;
  const int kt0 = blockIdx.x * 64, b = blockIdx.y;
  const int lrow = tid >> 2, lk = (tid & 3) * 8;      // dy tile: 64 rows x 32 columns
  const int brow = tid >> 3, bk = (tid & 7) * 8;      // W tile: 32 rows (n) x 64 columns (k)
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int cps = (a.N + 31) >> 5, nch = cps * a.nseg;       // chunks per segment, chunks in all
  f32x4 a0, a1, b0, b1;
  auto fetch = [&](int ch) {
    const int seg = ch / cps, n0 = (ch - seg * cps) * 32;
    const int g = a.group[seg] ? a.group[seg][b] : 0;
    const float* yrow = a.y + ((long)b * 64 + lrow) * a.ldy + seg * a.N;
    const int n = n0 + lk;
    a0 = ld4(yrow + n, n < a.N);
    a1 = ld4(yrow + n + 4, n + 4 < a.N);
    const bool nok = n0 + brow < a.N;
    const float* wr = a.W[seg] + (long)g * a.wgs[seg] + (long)(n0 + brow) * a.ldw[seg] + kt0 + bk;
    b0 = ld4(wr, nok && kt0 + bk < a.K);
    b1 = ld4(wr + 4, nok && kt0 + bk + 4 < a.K);
  };
  fetch(0);
  for (int ch = 0; ch < nch; ++ch) {
    __syncthreads();
    *reinterpret_cast<f32x4*>(&As[lrow * LP + lk]) = a0;
    *reinterpret_cast<f32x4*>(&As[lrow * LP + lk + 4]) = a1;
    *reinterpret_cast<f32x4*>(&Bs[brow * LQ + bk]) = b0;
    *reinterpret_cast<f32x4*>(&Bs[brow * LQ + bk + 4]) = b1;
    __syncthreads();
    if (ch + 1 < nch) fetch(ch + 1);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(&As[(wm * 32 + li) * LP + kg * 8 + 4 * lh]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], Bs[(kg * 8 + 4 * lh + j) * LQ + wn * 32 + li], acc, 0, 0, 0);
    }
  }
  const int col = kt0 + wn * 32 + li;
  if (col < a.K) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = wm * 32 + 8 * (i >> 2) + 4 * lh + (i & 3);
      dx[((long)b * 64 + row) * a.ldx + col] = acc[i];
    }
  }
}

struct GLinWgArgs {
  const float* x; int ldx, K;
  const float* dy; int ldy, col0, N;     // this segment's columns col0 .. col0+N-1 of dy
  const int* group;                      // [B] or null (everybody in group 0)
  int G, B, S;
  float* slab;                           // [S][G][N][K] weights, then [S][G][N] bias column sums
};

// grid (ntiles*ktiles, G, S)
__global__ __launch_bounds__(256) void glinear_wgrad_kernel(GLinWgArgs a) {
  __shared__ __attribute__((aligned(16))) float Ds[32 * LQ];   // dy[m chunk][n tile]
  __shared__ __attribute__((aligned(16))) float Xs[32 * LQ];   // x [m chunk][k tile]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;
  const int ktiles = (a.K + 63) >> 6;
  const int nt = blockIdx.x / ktiles, kt = blockIdx.x - nt * ktiles, g = blockIdx.y, s = blockIdx.z;
  const int n0 = nt * 64, k0 = kt * 64;
  const int per = (a.B + a.S - 1) / a.S, b_lo = s * per, b_hi = min(a.B, b_lo + per);
  const int row = tid >> 3, cq = (tid & 7) * 8;       // 32 rows x 64 columns per chunk
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float bsum = 0.f;                                     // column sum of dy for column n0 + tid (tid < 64), k-tile 0 only
  __shared__ int sList[64];                             // the samples of this slice that belong to group g (<= 64 per slice)
  __shared__ int sCount;
  if (tid < 64) {                                       // one wave compacts the slice (<= 64 samples): ballot + prefix popcount
    const int b = b_lo + tid;
    const bool mine = b < b_hi && (a.group ? a.group[b] : 0) == g;
    const unsigned long long m = __ballot(mine);
    if (mine) sList[__popcll(m & ((1ull << tid) - 1ull))] = b;
    if (tid == 0) sCount = __popcll(m);
  }
  __syncthreads();
  const int nch = 2 * sCount;                           // two 32-row chunks per sample
  f32x4 d0, d1, x0, x1;
  auto fetch = [&](int ch) {
    const int b = sList[ch >> 1], m0 = (ch & 1) * 32;
    const float* dr = a.dy + ((long)b * 64 + m0 + row) * a.ldy + a.col0 + n0 + cq;
    const float* xr = a.x + ((long)b * 64 + m0 + row) * a.ldx + k0 + cq;
    d0 = ld4(dr, n0 + cq < a.N);
    d1 = ld4(dr + 4, n0 + cq + 4 < a.N);
    x0 = ld4(xr, k0 + cq < a.K);
    x1 = ld4(xr + 4, k0 + cq + 4 < a.K);
  };
  if (nch > 0) fetch(0);
  for (int ch = 0; ch < nch; ++ch) {
    __syncthreads();
    *reinterpret_cast<f32x4*>(&Ds[row * LQ + cq]) = d0;
    *reinterpret_cast<f32x4*>(&Ds[row * LQ + cq + 4]) = d1;
    *reinterpret_cast<f32x4*>(&Xs[row * LQ + cq]) = x0;
    *reinterpret_cast<f32x4*>(&Xs[row * LQ + cq + 4]) = x1;
    __syncthreads();
    if (ch + 1 < nch) fetch(ch + 1);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = kg * 8 + 4 * lh + j;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ds[m * LQ + wm * 32 + li], Xs[m * LQ + wn * 32 + li], acc, 0, 0, 0);
      }
    if (kt == 0 && tid < 64) {
#pragma unroll
      for (int m = 0; m < 32; ++m) bsum += Ds[m * LQ + tid];
    }
  }
  float* slab = a.slab + ((long)s * a.G + g) * a.N * a.K;
  const int col = k0 + wn * 32 + li;
  if (col < a.K) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + wm * 32 + 8 * (i >> 2) + 4 * lh + (i & 3);
      if (n < a.N) slab[(long)n * a.K + col] = acc[i];
    }
  }
  if (kt == 0 && tid < 64 && n0 + tid < a.N)
    a.slab[(long)a.S * a.G * a.N * a.K + ((long)s * a.G + g) * a.N + n0 + tid] = bsum;
}

// dW[(g*N + n)*ldo + k] (+)= sum_s slab[s][g][n][k];  dbias[g*N + n] (+)= sum_s of the column sums behind the weight slabs
__global__ __launch_bounds__(256) void glinear_reduce_kernel(const float* __restrict__ slab, long rows, int K, int S,
                                                            float* __restrict__ dW, int ldo, float* __restrict__ dbias,
                                                            int accumulate) {
  const long nw = rows * K, k4 = K >> 2, nw4 = rows * k4;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nw4 + rows; i += stride) {
    if (i < nw4) {
      const long r = i / k4, c = (i - r * k4) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < S; ++s) v += *reinterpret_cast<const f32x4*>(slab + (long)s * nw + r * K + c);
      f32x4* d = reinterpret_cast<f32x4*>(dW + r * ldo + c);
      *d = accumulate ? *d + v : v;
    } else if (dbias != nullptr) {
      const long r = i - nw4;
      float v = 0.f;
      for (int s = 0; s < S; ++s) v += slab[(long)S * nw + (long)s * rows + r];
      dbias[r] = accumulate ? dbias[r] + v : v;
    }
  }
}

bool args_ok(const GLinArgs& a, bool need_x) {
  if ((need_x && !a.x) || !a.y || a.B <= 0 || a.K <= 0 || a.K % 4 || a.N <= 0 || a.N % 4 || a.nseg < 1 || a.nseg > kGLinMaxSeg ||
      a.ldx < a.K || a.ldx % 4 || a.ldy < a.nseg * a.N || a.ldy % 4)
    return false;
  for (int s = 0; s < a.nseg; ++s)
    if (!a.W[s] || a.ldw[s] < a.K || a.ldw[s] % 4 || a.wgs[s] % 4) return false;
  return true;
}

}  // namespace

int launch_glinear_forward(const GLinArgs& a, hipStream_t st) {
  if (!args_ok(a, true)) return kErrBadArg;
  char name[96];
  snprintf(name, sizeof name, "glinear_fwd_kernel");
  if (prof_detailed()) snprintf(name, sizeof name, "glinear_fwd_kernel K=%d N=%d nseg=%d", a.K, a.N, a.nseg);
  ProfScope ps(name, st, 2.0 * a.B * 64.0 * a.K * a.N * a.nseg,
               4.0 * a.B * 64.0 * (a.K + (double)a.nseg * a.N));
  hipLaunchKernelGGL(glinear_fwd_kernel, dim3(((a.N + 63) / 64) * a.nseg, a.B), dim3(256), 0, st, a);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_glinear_dgrad(const GLinArgs& a, float* dx, hipStream_t st) {
  if (!args_ok(a, false) || !dx) return kErrBadArg;
  ProfScope ps("glinear_dgrad_kernel", st, 2.0 * a.B * 64.0 * a.K * a.N * a.nseg,
               4.0 * a.B * 64.0 * (a.K + (double)a.nseg * a.N));
  hipLaunchKernelGGL(glinear_dgrad_kernel, dim3((a.K + 63) / 64, a.B), dim3(256), 0, st, a, dx);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

size_t glinear_wgrad_ws_floats(int G, int N, int K, int S) { return (size_t)S * G * N * ((size_t)K + 1); }

int launch_glinear_wgrad(const float* x, int ldx, int K, const float* dy, int ldy, int col0, int N, const int* group, int G,
                         int B, float* dW, int ldo, float* dbias, int accumulate, float* ws, size_t ws_floats, hipStream_t st) {
  if (!x || !dy || !dW || !ws || B <= 0 || K <= 0 || K % 4 || N <= 0 || N % 4 || G <= 0 || ldx % 4 || ldy % 4 || col0 % 4 ||
      ldo < K || ldo % 4)
    return kErrBadArg;
  // slices of the batch: no group ids -> one group holds every sample and its 64x64 tiles are few (13-42): many slices fill
  // the chip; group ids -> a group holds B/G samples on average and there are G times the tiles: few slices (a slab is
  // written per (slice, group) whether or not the slice holds a sample of the group)
  int S = group == nullptr ? B / 8 : B / (4 * G);
  S = S < 1 ? 1 : (S > 16 ? 16 : S);
  while ((B + S - 1) / S > 64) ++S;      // a slice's list of samples has 64 entries
  if (glinear_wgrad_ws_floats(G, N, K, S) > ws_floats) return kErrWorkspace;
  GLinWgArgs a{x, ldx, K, dy, ldy, col0, N, group, G, B, S, ws};
  {
    char name[96];
    snprintf(name, sizeof name, "glinear_wgrad_kernel");
    if (prof_detailed()) snprintf(name, sizeof name, "glinear_wgrad_kernel K=%d N=%d G=%d S=%d grp=%d", K, N, G, S, group != nullptr);
    ProfScope ps(name, st, 2.0 * B * 64.0 * K * N, 4.0 * (B * 64.0 * (K + N) + (double)S * G * N * K));
    hipLaunchKernelGGL(glinear_wgrad_kernel, dim3(((N + 63) / 64) * ((K + 63) / 64), G, S), dim3(256), 0, st, a);
    CTVAE_LAUNCH_CHECK();
  }
  {
    const long rows = (long)G * N, work = rows * (K / 4) + rows;
    ProfScope ps("glinear_reduce_kernel", st, 0.0, 4.0 * (S + 1.0) * rows * (K + 1.0));
    long blocks = (work + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(glinear_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ws, rows, K, S, dW, ldo, dbias, accumulate);
    CTVAE_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace ctvae
