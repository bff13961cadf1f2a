// Weight-gradient kernel: autograd's wgrad of every Conv2d / ConvTranspose2d / Linear on the hot path
// (SURVEY.md K20).  In the layer's FORWARD tap-GEMM geometry (geom.hpp)
//
//     dW[wtap(t)][c][n] = sum_m  X[gpix(m,t)][c] * dY[spix(m)][n]        (+ dbias[n] = sum_m dY[spix(m)][n])
//
// i.e. a GEMM whose output is small (taps*Ci x Co) and whose reduction runs over every output
// pixel of the batch.  One workgroup owns a KT x NT tile of dW for one slice of the m range
// ("split-M"), streams 32-pixel chunks of im2col(X) and dY through LDS and accumulates with
// v_mfma_f32_32x32x2_f32; slices are combined by a second, deterministic pass (no float atomics:
// bitwise reproducible, and atomics would be bound at ~1.3 TB/s on gfx950).
#include "common.hpp"
#include "prof.hpp"
#include "finish.hpp"
#include "pair.hpp"
#include "wgrad_fast.hpp"

namespace ctvae {


template <int WK, int WN, bool XVEC, bool DVEC>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
  kernarg_warm<sizeof(WgradArgs)>();
  constexpr int KT = WK * 32, NT = WN * 32;
  static_assert(WK * WN == 4, "4 waves");
  // single-buffered on purpose: the split-M grid is sized for ~4 workgroups per CU and occupancy hides the
  // load latency better than a second LDS buffer (which would halve the resident workgroups)
  __shared__ __attribute__((aligned(16))) float sXbuf[MC * KT];
  __shared__ __attribute__((aligned(16))) float sDbuf[MC * NT];
  __shared__ int sRowPix[2][MC];
  __shared__ int sRowYX[2][MC];
  __shared__ int sRowOut[2][MC];

  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware order (see wgrad_fast_kernel): the output tiles of one pixel slice share one L2
  int split = blockIdx.y, xtile = blockIdx.x;
  {
    const int T = gridDim.x, L = blockIdx.y * T + blockIdx.x, full = (gridDim.y >> 3) * 8 * T;
    if (L < full) {
      const int grp = L / (8 * T), r = L - grp * 8 * T;
      split = grp * 8 + (r & 7);
      xtile = r >> 3;
    }
  }
  const int ktg = xtile / a.ntiles, nt = xtile - ktg * a.ntiles;
  int cls = 0;
#pragma unroll
  for (int c = 1; c < kMaxCls; ++c)
    if (c < g.ncls && ktg >= a.ktile_start[c]) cls = c;
  const int kt0 = (ktg - a.ktile_start[cls]) * KT;
  const int n0 = nt * NT;
  const int gC = g.gC;
  const int Ktot = g.ntaps[cls] * gC;
  const int N = a.N;

  const int cbeg = split * a.chunks_per_split;
  const int nchunks_all = (a.Mc + MC - 1) / MC;
  int cend = cbeg + a.chunks_per_split;
  if (cend > nchunks_all) cend = nchunks_all;
  const int nch = cend - cbeg;

  auto rowinfo = [&](int c, int buf) {
    if (tid < MC) {
      int m = (cbeg + c) * MC + tid;
      if (m < a.Mc) {
        int b, qy, qx;
        decode_m(g, m, b, qy, qx);
        sRowPix[buf][tid] = (b * g.gH + qy * g.is) * g.gW + qx * g.is;
        sRowYX[buf][tid] = ((qy * g.is) << 16) | (qx * g.is);
        sRowOut[buf][tid] = scatter_pix(g, cls, b, qy, qx);
      } else {
        sRowPix[buf][tid] = -1;
        sRowYX[buf][tid] = 0;
        sRowOut[buf][tid] = -1;
      }
    }
  };

  // per-thread fixed k column(s) of the X tile
  constexpr int XQ = KT / 4;            // float4 per row
  constexpr int X_V = (MC * XQ) / 256;  // float4 per thread
  constexpr int X_S = (MC * KT) / 256;  // scalars per thread
  constexpr int DQ = NT / 4;
  constexpr int D_V = (MC * DQ) / 256 > 0 ? (MC * DQ) / 256 : 1;
  constexpr int D_S = (MC * NT) / 256;

  int x_dy = 0, x_dx = 0, x_c = 0;
  bool x_kok = false;
  {
    int kcol = XVEC ? 4 * (tid % XQ) : (tid % KT);
    int k = kt0 + kcol;
    x_kok = k < Ktot;
    int t = x_kok ? k / gC : 0;
    x_c = k - t * gC;
    Tap tp = g.taps[cls][t];
    x_dy = tp.dy;
    x_dx = tp.dx;
  }

  const bool xf_on = XVEC && a.xf_scale != nullptr && x_kok;
  const f32x4 xf_sc = xf_on ? *reinterpret_cast<const f32x4*>(a.xf_scale + x_c) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 xf_sh = xf_on ? *reinterpret_cast<const f32x4*>(a.xf_shift + x_c) : f32x4{0.f, 0.f, 0.f, 0.f};
  const float xf_ns = a.xf_act == ACT_LRELU ? kLeaky : (a.xf_act == ACT_RELU ? 0.f : 1.f);
  f32x4 rx[XVEC ? X_V : 1];
  [[maybe_unused]] float rx_in[XVEC ? X_V : 1];      // 1 where the loaded pixel lies inside the image (the shift applies), else 0
  float rxs[XVEC ? 1 : X_S];
  f32x4 rd[DVEC ? D_V : 1];
  [[maybe_unused]] bool rd_ok[DVEC ? D_V : 1];
  float rds[DVEC ? 1 : D_S];

  auto load_chunk = [&](int buf) {
    if constexpr (XVEC) {
      // table reads first, then UNCONDITIONAL loads at a clamped address (the tensor's first pixel for a tap outside the image; the
      // zero is selected when the chunk is stored): `if (ok) v = load` is a branch per load, behind which the compiler waits
      // vmcnt(0) -- the chunk's loads went out one memory round trip at a time
      int pixv[X_V], yxv[X_V];
#pragma unroll
      for (int j = 0; j < X_V; ++j) {
        const int r = tid / XQ + (256 / XQ) * j;
        pixv[j] = sRowPix[buf][r];
        yxv[j] = sRowYX[buf][r];
      }
#pragma unroll
      for (int j = 0; j < X_V; ++j) {
        const int iy = (yxv[j] >> 16) + x_dy, ix = (yxv[j] & 0xffff) + x_dx;
        const bool ok = x_kok && pixv[j] >= 0 && (unsigned)iy < (unsigned)g.gH && (unsigned)ix < (unsigned)g.gW;
        const long off = ok ? (long)(pixv[j] + x_dy * g.gW + x_dx) * gC + x_c : 0L;
        rx[j] = *reinterpret_cast<const f32x4*>(a.X + off);
        rx_in[j] = ok ? 1.f : 0.f;      // zero / lazy BatchNorm apply when the chunk is STORED (applied here it consumed every load at once)
      }
    } else {
#pragma unroll
      for (int j = 0; j < X_S; ++j) {
        int r = tid / KT + (256 / KT) * j;
        int pix = sRowPix[buf][r], yx = sRowYX[buf][r];
        int iy = (yx >> 16) + x_dy, ix = (yx & 0xffff) + x_dx;
        bool ok = x_kok && pix >= 0 && (unsigned)iy < (unsigned)g.gH && (unsigned)ix < (unsigned)g.gW;
        rxs[j] = ok ? a.X[(long)(pix + x_dy * g.gW + x_dx) * gC + x_c] : 0.f;
      }
    }
    if constexpr (DVEC) {
      int spv[D_V];
#pragma unroll
      for (int j = 0; j < D_V; ++j) {
        const int r = (tid + 256 * j) / DQ;
        spv[j] = sRowOut[buf][r < MC ? r : 0];
      }
#pragma unroll
      for (int j = 0; j < D_V; ++j) {
        const int f = tid + 256 * j;
        const int r = f / DQ, nq = f - r * DQ;
        const int n = n0 + 4 * nq;
        const bool ok = r < MC && spv[j] >= 0 && n < N;
        rd[j] = *reinterpret_cast<const f32x4*>(a.dY + (ok ? (long)spv[j] * N + n : 0L));     // unconditional, clamped (see above)
        rd_ok[j] = ok;
      }
    } else {
#pragma unroll
      for (int j = 0; j < D_S; ++j) {
        int e = tid + 256 * j;
        int r = e / NT, nn = e - r * NT;
        int sp = sRowOut[buf][r];
        int n = n0 + nn;
        rds[j] = (sp >= 0 && n < N) ? a.dY[(long)sp * N + n] : 0.f;
      }
    }
  };

  auto store_chunk = [&]() {
    float* sX = sXbuf;
    float* sD = sDbuf;
    if constexpr (XVEC) {
#pragma unroll
      for (int j = 0; j < X_V; ++j)
        if (rx_in[j] == 0.f) rx[j] = f32x4{0.f, 0.f, 0.f, 0.f};       // tap outside the image / row beyond the tensor (clamped load)
      if (xf_on) {   // lazy BatchNorm apply of the previous block (InXform); padding stays 0
#pragma unroll
        for (int j = 0; j < X_V; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = rx[j][e] * xf_sc[e] + xf_sh[e] * rx_in[j];
            rx[j][e] = fmaxf(t, t * xf_ns);
          }
      }
#pragma unroll
      for (int j = 0; j < X_V; ++j) {
        int r = tid / XQ + (256 / XQ) * j;
        *reinterpret_cast<f32x4*>(&sX[r * KT + 4 * (tid % XQ)]) = rx[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < X_S; ++j) sX[(tid / KT + (256 / KT) * j) * KT + (tid % KT)] = rxs[j];
    }
    if constexpr (DVEC) {
#pragma unroll
      for (int j = 0; j < D_V; ++j) {
        int f = tid + 256 * j;
        int r = f / DQ, nq = f - r * DQ;
        if (r < MC) *reinterpret_cast<f32x4*>(&sD[r * NT + 4 * nq]) = rd_ok[j] ? rd[j] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int j = 0; j < D_S; ++j) sD[tid + 256 * j] = rds[j];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  const bool do_bias = (a.pbias != nullptr) && (kt0 == 0) && (tid < NT);

  if (nch > 0) {
    rowinfo(0, 0);
    __syncthreads();
    load_chunk(0);
    if (nch > 1) rowinfo(1, 1);
    store_chunk();
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
      const float* sX = sXbuf;
      const float* sD = sDbuf;
      if (c + 1 < nch) load_chunk((c + 1) & 1);
      if (c + 2 < nch) rowinfo(c + 2, c & 1);
      {   // fragments one group of four steps ahead, MFMA / LDS-read order pinned (wgrad_fast.hpp)
        constexpr int GS = 4, NG = (MC / 2) / GS;
        float fa[2][GS], fb[2][GS];
        auto rdg = [&](int gi, int set) {
#pragma unroll
          for (int q = 0; q < GS; ++q) {
            const int s = gi * GS + q;
            fa[set][q] = sX[(2 * s + lh) * KT + wk * 32 + li];
            fb[set][q] = sD[(2 * s + lh) * NT + wn * 32 + li];
          }
        };
        rdg(0, 0);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
          if (gi + 1 < NG) rdg(gi + 1, (gi + 1) & 1);
#pragma unroll
          for (int q = 0; q < GS; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[gi & 1][q], fb[gi & 1][q], acc, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < GS; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
        }
      }
      if (do_bias) {
#pragma unroll 8
        for (int r = 0; r < MC; ++r) bsum += sD[r * NT + tid];
      }
      __syncthreads();
      if (c + 1 < nch) {
        store_chunk();
        __syncthreads();
      }
    }
  }

  // ---- write partial tile ---------------------------------------------------------------------
  const int col = n0 + wn * 32 + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int k = kt0 + wk * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
    if (k < Ktot && col < N) {
      int t = k / gC, c = k - t * gC;
      int wrow = g.taps[cls][t].wtap * (g.wts / g.wCo) + c * (g.wrs / g.wCo);   // row of the [rows][N] weight block (geom.hpp wts / wrs)
      a.part[((long)split * a.rows_total + wrow) * N + col] = acc[r];
    }
  }
  if (do_bias && n0 + tid < N) a.pbias[(long)(split * g.ncls + cls) * N + n0 + tid] = bsum;
}

// ---- lean variant: body in wgrad_fast.hpp (shared with the paired backward launch of tapgemm_fast.hip) --------
template <int WK, int WN, int TK, int TN, bool XF = false>
__global__ __launch_bounds__(256) void wgrad_fast_kernel(const WgradArgs a, int lgQw, int lgQhw, int lgC) {
  kernarg_warm<sizeof(WgradArgs) + 12>();
  constexpr int KT = WK * TK * 32, NT = WN * TN * 32;
  __shared__ __attribute__((aligned(16))) float sX[MC * KT];
  __shared__ __attribute__((aligned(16))) float sD[MC * NT];
  __shared__ unsigned sPix[2][MC];   // byte offset of the pixel base in X, or kOOBw
  __shared__ unsigned sMsk[2][MC];   // y/x validity bits (see tapgemm_fast.hip)
  __shared__ unsigned sOutB[2][MC];  // byte offset of the scatter pixel in dY, or kOOBw
  wgrad_fast_body<WK, WN, TK, TN, XF>(a, lgQw, lgQhw, lgC, sX, sD, sPix, sMsk, sOutB, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}

// dst[i] = (accumulate ? dst[i] : 0) + sum_s part[s*stride + i], for up to two jobs (weights, bias) in ONE launch:
// blocks [0, nb1) serve job 1, the rest job 2.  Two block shapes, both deterministic (fixed summation order):
//  * few slices (S <= 8): one thread per float4 / float, a plain pass over the slices;
//  * many slices: 16 split-lanes x 16 element-lanes; lane j sums s = j, j+16, ... , then a fixed shuffle tree
//    and a 4-wave LDS combine.  With VEC the element lane covers a float4 (64 elements per block): the most workgroups and
//    the shortest dependent chain -- the shape for the small jobs that sit in the backward chain;
//  * many slices of a LARGE gradient (>= kReduceWideItems elements per slice; `small` == 2): 4 split-lanes (the waves) x 64
//    element-lanes, wave w sums s = w, w+4, ... with eight loads in flight, 4-wave LDS combine in a fixed order -- 1 KB of every
//    slab per workgroup instead of 256 B (MCQ-VAE's 3x3 layers: 600 MB of slabs per step, 3.5 -> 5.6 TB/s).

template <typename T>
__device__ __forceinline__ T rzero();
template <>
__device__ __forceinline__ float rzero<float>() { return 0.f; }
template <>
__device__ __forceinline__ f32x4 rzero<f32x4>() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

template <typename T>
__device__ __forceinline__ void reduce_small(const ReduceJob& j, int blk, int accumulate) {
  const long n = j.n / (long)(sizeof(T) / 4), i = (long)blk * 256 + threadIdx.x;
  if (i >= n) return;
  T v = accumulate ? reinterpret_cast<const T*>(j.dst)[i] : rzero<T>();
  for (int s = 0; s < j.S; ++s) v += reinterpret_cast<const T*>(j.part + (long)s * j.stride)[i];
  reinterpret_cast<T*>(j.dst)[i] = v;
}

template <typename T>
__device__ __forceinline__ void reduce_split(const ReduceJob& j, int blk, int accumulate, T (*sm)[64]) {
  const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long n = j.n / (long)(sizeof(T) / 4), i = (long)blk * 16 + el;
  T v = rzero<T>();
  if (i < n) {
#pragma unroll 4
    for (int s = sl; s < j.S; s += 16) v += reinterpret_cast<const T*>(j.part + (long)s * j.stride)[i];
  }
  // lanes with equal `el` sit 16 apart: xor 16 and 32 inside a wave, then 4 waves through LDS
  if constexpr (sizeof(T) == 4) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] += __shfl_xor(v[k], 16, 64);
      v[k] += __shfl_xor(v[k], 32, 64);
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane < 16) sm[wave][lane] = v;
  __syncthreads();
  if (threadIdx.x < 16 && i < n) {
    T t = ((sm[0][el] + sm[1][el]) + sm[2][el]) + sm[3][el];
    T* d = reinterpret_cast<T*>(j.dst) + i;
    *d = accumulate ? (*d + t) : t;
  }
}

template <typename T>
__device__ __forceinline__ void reduce_wide(const ReduceJob& j, int blk, int accumulate, T (*sm)[64]) {
  // wave = split lane: wave w sums the slabs w, w+4, ... for 64 consecutive elements (1 KB of every slab per workgroup with
  // float4 elements), eight loads in flight; the four waves' sums are combined in a fixed order
  const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long n = j.n / (long)(sizeof(T) / 4), i = (long)blk * 64 + el;
  T v = rzero<T>();
  if (i < n) {
#pragma unroll 8
    for (int s = sl; s < j.S; s += 4) v += reinterpret_cast<const T*>(j.part + (long)s * j.stride)[i];
  }
  sm[sl][el] = v;
  __syncthreads();
  if (sl == 0 && i < n) {
    T t = ((sm[0][el] + sm[1][el]) + sm[2][el]) + sm[3][el];
    T* d = reinterpret_cast<T*>(j.dst) + i;
    *d = accumulate ? (*d + t) : t;
  }
}

// blocks [0, nb1) job 1, [nb1, nb12) job 2, [nb12, nb123) the split-K finish of the paired data gradient (sk.nblk blocks, may be
// 0), [nb123, ...) the BatchNorm-backward finalize of the layer below (bf.C blocks, may be 0)
__global__ __launch_bounds__(256) void reduce_partials_kernel(ReduceJob j1, ReduceJob j2, int nb1, int accumulate, SplitKJob sk, int nb12,
                                                              BnFinJob bf, int nb123) {
  __shared__ f32x4 sm[4][64];
  if ((int)blockIdx.x >= nb123) {
    bn_bwd_finalize_body(bf, (int)blockIdx.x - nb123, reinterpret_cast<double*>(&sm[0][0]));
    return;
  }
  if ((int)blockIdx.x >= nb12) {
    splitk_finish_body(sk, (int)blockIdx.x - nb12);
    return;
  }
  const bool second = (int)blockIdx.x >= nb1;
  const ReduceJob j = second ? j2 : j1;
  const int blk = second ? blockIdx.x - nb1 : blockIdx.x;
  if (j.small == 1) {
    if (j.vec) reduce_small<f32x4>(j, blk, accumulate);
    else reduce_small<float>(j, blk, accumulate);
  } else if (j.small == 2) {
    if (j.vec) reduce_wide<f32x4>(j, blk, accumulate, sm);
    else reduce_wide<float>(j, blk, accumulate, reinterpret_cast<float(*)[64]>(sm));
  } else {
    if (j.vec) reduce_split<f32x4>(j, blk, accumulate, sm);
    else reduce_split<float>(j, blk, accumulate, reinterpret_cast<float(*)[64]>(sm));
  }
}

static int reduce_job_blocks(ReduceJob& j) {
  if (j.n <= 0 || j.S <= 0) return 0;
  j.vec = ((j.n & 3) == 0 && (j.stride & 3) == 0 && (((uintptr_t)j.part | (uintptr_t)j.dst) & 15) == 0) ? 1 : 0;
  const long items = j.vec ? j.n / 4 : j.n;
  static const long wide_items = [] { const char* e = getenv("CTVAE_REDUCE_WIDE_ITEMS"); return e ? atol(e) : 65536L; }();   // diagnostic
  j.small = j.S <= 8 ? 1 : (items >= wide_items ? 2 : 0);
  return (int)(j.small == 1 ? (items + 255) / 256 : (j.small == 2 ? (items + 63) / 64 : (items + 15) / 16));
}

static void launch_reduce2(const float* p1, float* d1, long n1, int S1, long st1, const float* p2, float* d2, long n2, int S2,
                           long st2, int accumulate, hipStream_t st) {
  ReduceJob j1{p1, d1, n1, S1, st1, 0, 0}, j2{p2, d2, n2, S2, st2, 0, 0};
  const int nb1 = reduce_job_blocks(j1), nb2 = reduce_job_blocks(j2);
  if (nb1 + nb2 == 0) return;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(nb1 + nb2), dim3(256), 0, st, j1, j2, nb1, accumulate, SplitKJob{}, nb1 + nb2,
                     BnFinJob{}, nb1 + nb2);
}

static void launch_reduce(const float* part, float* dst, long n, int S, long stride, int accumulate, hipStream_t st) {
  launch_reduce2(part, dst, n, S, stride, nullptr, nullptr, 0, 0, 0, accumulate, st);
}

// ---- deferred slab reductions (ctvae_defer_begin / ctvae_defer_flush) -------------------------------------------------------
// A parameter gradient is not read before the optimizer step, so the slab reduction behind a weight-gradient kernel need not sit
// in the backward chain: between ctvae_defer_begin and ctvae_defer_flush the weight-gradient kernels write their slabs into the
// caller's arena (each call behind the previous one's slabs) and the reductions are only recorded; the flush runs all of them
// in one or two launches.  MCQ-VAE / CT-MCQ-VAE: 33 finishing launches of ~19 MB each per step become one pass over 630 MB.
// Calls whose finishing launch carries a BatchNorm-backward finalize keep it (VanillaVAE's: that launch is in the chain anyway).
constexpr int kDeferJobsPerLaunch = 24;
struct DeferTable {
  ReduceJob j[kDeferJobsPerLaunch];
  int acc[kDeferJobsPerLaunch];
  int blk0[kDeferJobsPerLaunch + 1];
  int n;
};
struct DeferJob {
  ReduceJob j;
  int acc, nb;
};
struct DeferCtx {
  bool active = false;
  float* base = nullptr;
  size_t floats = 0, off = 0;
  float* cur = nullptr;       // arena workspace handed to the weight-gradient call that is running (null: not deferring this call)
  size_t cur_floats = 0;
  std::vector<DeferJob> jobs;
  double bytes = 0;
};
// process-wide, not thread-local: loss.backward() runs the gradient kernels' launchers on autograd's device thread while the
// thread that called ctvae_defer_begin waits for it -- one deferral at a time per process (one GPU per process, one stream)
static DeferCtx& defer_ctx() {
  static DeferCtx c;
  return c;
}

__global__ __launch_bounds__(256) void reduce_multi_kernel(const DeferTable t) {
  kernarg_warm<sizeof(DeferTable)>();
  __shared__ f32x4 sm[4][64];
  int k = 0;
  while (k + 1 < t.n && (int)blockIdx.x >= t.blk0[k + 1]) ++k;   // workgroup-uniform
  const ReduceJob j = t.j[k];
  const int blk = blockIdx.x - t.blk0[k], accumulate = t.acc[k];
  if (j.small == 1) {
    if (j.vec) reduce_small<f32x4>(j, blk, accumulate);
    else reduce_small<float>(j, blk, accumulate);
  } else if (j.small == 2) {
    if (j.vec) reduce_wide<f32x4>(j, blk, accumulate, sm);
    else reduce_wide<float>(j, blk, accumulate, reinterpret_cast<float(*)[64]>(sm));
  } else {
    if (j.vec) reduce_split<f32x4>(j, blk, accumulate, sm);
    else reduce_split<float>(j, blk, accumulate, reinterpret_cast<float(*)[64]>(sm));
  }
}

// workspace of the weight-gradient call that follows: a slice of the arena while deferring (and there is room), else ws
float* defer_wgrad_ws(float* ws, size_t ws_floats) {
  DeferCtx& d = defer_ctx();
  d.cur = nullptr;
  if (!d.active || d.floats - d.off < ws_floats) return ws;
  d.cur = d.base + d.off;
  d.cur_floats = ws_floats;
  return d.cur;
}
void defer_wgrad_done() { defer_ctx().cur = nullptr; }

int defer_begin(float* arena, size_t arena_floats) {
  DeferCtx& d = defer_ctx();
  if (d.active || arena == nullptr) return kErrBadArg;
  d.active = true;
  d.base = arena; d.floats = arena_floats; d.off = 0; d.cur = nullptr;
  d.jobs.clear();
  d.bytes = 0;
  return 0;
}

static int defer_run_pending(DeferCtx& d, hipStream_t st);

int defer_flush(hipStream_t st) {
  DeferCtx& d = defer_ctx();
  if (!d.active) return kErrBadArg;
  d.active = false;
  d.cur = nullptr;
  return defer_run_pending(d, st);
}

static int defer_run_pending(DeferCtx& d, hipStream_t st) {
  for (size_t i0 = 0; i0 < d.jobs.size(); i0 += kDeferJobsPerLaunch) {
    DeferTable t{};
    int nb = 0;
    double bytes = 0;
    t.n = (int)(d.jobs.size() - i0 < (size_t)kDeferJobsPerLaunch ? d.jobs.size() - i0 : (size_t)kDeferJobsPerLaunch);
    for (int k = 0; k < t.n; ++k) {
      const DeferJob& dj = d.jobs[i0 + k];
      t.j[k] = dj.j; t.acc[k] = dj.acc; t.blk0[k] = nb;
      nb += dj.nb;
      bytes += 4.0 * (double)(dj.j.S + 1) * dj.j.n;
    }
    t.blk0[t.n] = nb;
    if (nb == 0) continue;
    ProfScope ps("reduce_multi_kernel", st, 0.0, bytes);
    hipLaunchKernelGGL(reduce_multi_kernel, dim3(nb), dim3(256), 0, st, t);
    CTVAE_LAUNCH_CHECK();
  }
  d.jobs.clear();
  return 0;
}

// record one call's reductions instead of running them; false: this call is not being deferred (or must not be)
static bool defer_record(const float* p1, float* d1, long n1, int S1, long st1, const float* p2, float* d2, long n2, int S2, long st2,
                         int accumulate, hipStream_t st) {
  DeferCtx& d = defer_ctx();
  if (!d.active) return false;
  // a second writer of a gradient that has a recorded reduction (a layer used twice in one pass): keep their order -- the
  // recorded ones run now, this one after them as usual
  for (const DeferJob& dj : d.jobs)
    if (dj.j.dst == d1 || (d2 != nullptr && dj.j.dst == d2)) {
      defer_run_pending(d, st);
      return false;
    }
  if (d.cur == nullptr) return false;
  const float* lo = d.cur;
  const float* hi = d.cur + d.cur_floats;
  if (p1 < lo || p1 + (size_t)S1 * st1 > hi) return false;                       // slabs are not in the arena slice
  if (p2 != nullptr && d2 != nullptr && (p2 < lo || p2 + (size_t)S2 * st2 > hi)) return false;
  const float* end = p1 + (size_t)S1 * st1;
  ReduceJob j1{p1, d1, n1, S1, st1, 0, 0};
  const int nb1 = reduce_job_blocks(j1);
  if (nb1 > 0) d.jobs.push_back(DeferJob{j1, accumulate, nb1});
  if (p2 != nullptr && d2 != nullptr) {
    ReduceJob j2{p2, d2, n2, S2, st2, 0, 0};
    const int nb2 = reduce_job_blocks(j2);
    if (nb2 > 0) d.jobs.push_back(DeferJob{j2, accumulate, nb2});
    if (p2 + (size_t)S2 * st2 > end) end = p2 + (size_t)S2 * st2;
  }
  d.off = ((size_t)(end - d.base) + 63) & ~(size_t)63;                          // the next call's slice starts behind these slabs
  d.cur = nullptr;
  return true;
}

// Slab reduction behind a weight-gradient kernel: issued now, or -- inside ctvae_conv_backward -- recorded so that it shares
// the call's ONE finishing launch with the data gradient's split-K sum and the BatchNorm-backward finalize (pair.hpp)
static int finish_reduce(const float* p1, float* d1, long n1, int S1, long st1, const float* p2, float* d2, long n2, int S2, long st2,
                         int accumulate, hipStream_t st) {
  if (defer_record(p1, d1, n1, S1, st1, p2, d2, n2, S2, st2, accumulate, st)) return 0;
  if (PairCtx* pc = pair_ctx(); pc != nullptr && !pc->haveRed) {
    pc->j1 = ReduceJob{p1, d1, n1, S1, st1, 0, 0};
    pc->j2 = (p2 != nullptr && d2 != nullptr) ? ReduceJob{p2, d2, n2, S2, st2, 0, 0} : ReduceJob{nullptr, nullptr, 0, 0, 0, 0, 0};
    pc->nb1 = reduce_job_blocks(pc->j1);
    pc->nb2 = reduce_job_blocks(pc->j2);
    pc->accumulate = accumulate;
    pc->bytesRed = 4.0 * (double)(S1 + 1) * n1;
    pc->haveRed = pc->nb1 + pc->nb2 > 0;
    return 0;
  }
  ProfScope ps("reduce_partials_kernel", st, 0.0, 4.0 * (double)(S1 + 1) * n1);
  if (p2 != nullptr && d2 != nullptr) launch_reduce2(p1, d1, n1, S1, st1, p2, d2, n2, S2, st2, accumulate, st);
  else launch_reduce(p1, d1, n1, S1, st1, accumulate, st);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

// for a kernel outside this file that produced weight-gradient slabs (image.hip img_bwd_fused_kernel)
int wgrad_finish_slabs(const float* part, float* dW, long n, int nparts, const float* pbias, float* dbias, long nb, int accumulate,
                       hipStream_t st) {
  return finish_reduce(part, dW, n, nparts, n, pbias, dbias, nb, nparts, nb, accumulate, st);
}

size_t wgrad_workspace_floats(const ConvGeom& g, int S) {
  int taps_total = 0;
  for (int c = 0; c < g.ncls; ++c) taps_total += g.ntaps[c];
  size_t rows = (size_t)taps_total * g.gC;
  return (size_t)S * rows * g.sC + (size_t)S * g.ncls * g.sC;
}

// choose the number of m-splits so that the grid has ~1024 workgroups
static int choose_splits(const ConvGeom& g, int KT, int NT, size_t ws_floats, int target_wgs = 1024) {
  int ktiles = 0;
  for (int c = 0; c < g.ncls; ++c) ktiles += ceil_div(g.ntaps[c] * g.gC, KT);
  int tiles = ktiles * ceil_div(g.sC, NT);
  int Mc = g.B * g.Qh * g.Qw;
  int nchunks = ceil_div(Mc, MC);
  int S = target_wgs / (tiles > 0 ? tiles : 1);
  if (S < 1) S = 1;
  // at least 8 chunks of 32 pixels per split (4 for pixel ranges below 32 chunks: the Linear layers' rows).  Round 3, after the
  // kernels' prologues / epilogues got cheaper, the small-batch step prefers half as many, twice as long weight-gradient workgroups
  // (CTVAE_WGRAD_MIN_CHUNKS 4 / 6 / 8 / 12 / 16 at bs = 64: 0.7355 / 0.7305 / 0.7226 / 0.7304 / 0.7554 ms; bs = 256 and the
  // Winograd models are bounded by the workgroup target, not by this)
  static const int min_chunks_env = [] { const char* e = getenv("CTVAE_WGRAD_MIN_CHUNKS"); return e ? atoi(e) : 0; }();   // diagnostic
  const int min_chunks = min_chunks_env > 0 ? min_chunks_env : (nchunks >= 32 ? 8 : 4);
  int maxS = ceil_div(nchunks, min_chunks);
  if (maxS < 1) maxS = 1;
  if (S > maxS) S = maxS;
  while (S > 1 && wgrad_workspace_floats(g, S) > ws_floats) --S;
  return S;
}

bool thin_wgrad_supported(const ConvGeom& g);
size_t thin_wgrad_workspace_floats(const ConvGeom& g);
int launch_thin_wgrad(const ConvGeom& g, const float* X, const float* dY, float* ws, float** part_out, float** pbias_out,
                      int* nparts_w, int* nparts_b, bool want_bias, hipStream_t st, const InXform* xf);

bool upconv_wgrad_supported(const ConvGeom& g);
int launch_upconv_wgrad(const ConvGeom& g, const float* X, const float* dY, float* ws, float** part_out, float** pbias_out,
                        int* nparts, bool want_bias, hipStream_t st, const DyXform* dyx, const InXform* xf);
bool img_enc_supported(const ConvGeom& g);
int launch_img_enc_wgrad(const ConvGeom& g, const float* X, const float* dY, float* ws, float** part_out, float** pbias_out,
                         int* nparts, bool want_bias, hipStream_t st, const DyXform* dyx);

bool wino_wgrad_supported(const ConvGeom& g, size_t ws_floats, int* splits);
int launch_wino_wgrad(const ConvGeom& g, const float* X, const float* dY, float* ws, size_t ws_floats, int* nparts,
                      bool want_bias, float** pbias_out, hipStream_t st);

bool wino_enabled();

int launch_wgrad(const ConvGeom& g, const float* X, const float* dY, float* dW, float* dbias, float* ws,
                 size_t ws_bytes, int accumulate, hipStream_t st, const InXform* xf, const DyXform* dyx) {
  // 3x3 / stride 1 / same-padding layers: Winograd F(3x3,2x2), see wino.hip
  if ((xf == nullptr || xf->scale == nullptr) && (dyx == nullptr || dyx->y == nullptr) && wino_enabled() &&
      wino_wgrad_supported(g, ws_bytes / sizeof(float), nullptr)) {
    int np = 0;
    float* pb = nullptr;
    int rc = launch_wino_wgrad(g, X, dY, ws, ws_bytes / sizeof(float), &np, dbias != nullptr, &pb, st);
    if (rc) return rc;
    const long n = 9L * g.gC * g.sC;
    return finish_reduce(ws, dW, n, np, n, pb, dbias, (long)g.sC, np, (long)g.sC, accumulate, st);
  }
  const bool up = upconv_wgrad_supported(g) && ws_bytes / sizeof(float) >= (size_t)512 * (9 * 32 * 32 + 32);
  const bool enc = img_enc_supported(g) && (xf == nullptr || xf->scale == nullptr) &&
                   ws_bytes / sizeof(float) >= (size_t)512 * (27 * 32 + 32);
  // BN-backward on load: the transposed-conv kernel (writes g_y for the data gradient) and encoder.0's (no data gradient, no g_y)
  if (dyx != nullptr && dyx->y != nullptr && !up && !(enc && dyx->gy_out == nullptr)) return kErrBadArg;
  if (dyx != nullptr && dyx->y != nullptr && up && dyx->gy_out == nullptr) return kErrBadArg;
  if (up) {
    float *part = nullptr, *pb = nullptr;
    int np = 0;
    int rc = launch_upconv_wgrad(g, X, dY, ws, &part, &pb, &np, dbias != nullptr, st, dyx, xf);
    if (rc) return rc;
    const long n = 9L * 32 * 32;
    return finish_reduce(part, dW, n, np, n, pb, dbias, 32L, np, 32L, accumulate, st);
  }
  if (enc) {
    float *part = nullptr, *pb = nullptr;
    int np = 0;
    int rc = launch_img_enc_wgrad(g, X, dY, ws, &part, &pb, &np, dbias != nullptr, st, dyx);
    if (rc) return rc;
    const long n = 27L * 32;
    return finish_reduce(part, dW, n, np, n, pb, dbias, 32L, np, 32L, accumulate, st);
  }
  const bool thin = thin_wgrad_supported(g) && thin_wgrad_workspace_floats(g) <= ws_bytes / sizeof(float);
  const bool has_xf = xf != nullptr && xf->scale != nullptr;
  // transform on load: the thin kernels and the lean 64 x 64 kernel (wgrad_fast.hpp XF; also inside the paired launch)
  // ... and the plain-load kernel of the narrow layers (N <= 32: per-thread coefficients as well)
  if (has_xf && !thin && !((g.gC % 4) == 0 && (g.sC % 4) == 0 && g.wT == 0)) return kErrBadArg;
  if (thin) {
    float *part = nullptr, *pb = nullptr;
    int nw = 0, nb = 0;
    int rc = launch_thin_wgrad(g, X, dY, ws, &part, &pb, &nw, &nb, dbias != nullptr, st, xf);
    if (rc) return rc;
    int taps = 0;
    for (int c = 0; c < g.ncls; ++c) taps += g.ntaps[c];
    const long n = (long)taps * g.gC * g.sC;
    return finish_reduce(part, dW, n, nw, n, pb, dbias, (long)g.sC, nb, (long)g.sC, accumulate, st);
  }
  WgradArgs a{};
  a.g = g;
  a.X = X; a.dY = dY;
  a.Mc = g.B * g.Qh * g.Qw;
  a.N = g.sC;
  if (g.wT != 0 || a.Mc <= 0) return kErrBadArg;
  int taps_total = 0;
  for (int c = 0; c < g.ncls; ++c) taps_total += g.ntaps[c];
  a.rows_total = taps_total * g.gC;
  const bool xvec = (g.gC % 4) == 0, dvec = (a.N % 4) == 0;
  const bool narrow = a.N <= 32;
  // 128x128 tiles for wide layers whose pixel range still leaves >= 8 chunks per workgroup at ~1024 workgroups
  static const int no_big = [] { const char* e = getenv("CTVAE_WGRAD_NO_BIG"); return e ? atoi(e) : 0; }();   // diagnostic
  bool big = !no_big && !has_xf && xvec && dvec && (a.N % 128) == 0 && a.Mc >= 8192;
  if (has_xf) { a.xf_scale = xf->scale; a.xf_shift = xf->shift; a.xf_act = xf->act; }
  for (int c = 0; c < g.ncls && big; ++c) big = (g.ntaps[c] * g.gC) % 128 == 0;
  if (big) {
    int tiles128 = 0;
    for (int c = 0; c < g.ncls; ++c) tiles128 += g.ntaps[c] * g.gC / 128;
    tiles128 *= a.N / 128;
    big = tiles128 >= 16 && (long)ceil_div(a.Mc, MC) * tiles128 >= 8L * 1024;
  }
  const int KT = big ? 128 : (narrow ? 128 : 64), NT = big ? 128 : (narrow ? 32 : 64);
  const size_t ws_floats = ws_bytes / sizeof(float);
  // 128x128 tiles hold two workgroups per CU (64 accumulator + 134 other registers): one resident round, and half the
  // partial slabs for the reduce pass
  // grid-size target of the pixel split: 1024 workgroups; 768 for a weight gradient that shares its launch (ctvae_conv_backward)
  // with a data gradient of fewer than 1024 LONG workgroups (>= 12 K chunks: VanillaVAE's stride-2 3x3 layers) -- together
  // about one resident round, the coarser slices end with the data gradient and write fewer slabs.  Next to short
  // data-gradient workgroups (the 1x1 convs of the residual stacks, 8 chunks) or a data gradient that is a round or more by
  // itself (the 4x4 convs of MCQ / CT-MCQ-VAE) the finer split wins.  One target everywhere, 768 / 1024: VanillaVAE bs=256
  // 1.719 / 1.723 ms, MCQVAE bs=256 7.65 / 7.53 ms, CT-MCQ-VAE 128 pairs 7.19 / 7.15 ms.
  static const int tgt_env = [] { const char* e = getenv("CTVAE_WGRAD_WGS"); return e ? atoi(e) : 0; }();   // diagnostic override
  const PairCtx* pcw = pair_ctx();
  const bool will_pair = pcw != nullptr && !pcw->haveB && xvec && dvec && !narrow && !big;
  const bool coarse = will_pair && pcw->dgrad_wgs > 0 && pcw->dgrad_wgs < 1024 && pcw->dgrad_chunks >= 12;
  const int tgt_small = tgt_env ? tgt_env : (coarse ? 768 : 1024);
  int S = choose_splits(g, KT, NT, ws_floats, big ? 512 : tgt_small);
  // paired launch: a multiple of 8 slices lets the pair kernel give each XCD one contiguous eighth of the pixels in BOTH roles
  // (conv_bwd_pair_kernel, role-aware placement); rounded down, never below 8 nor above what the workspace / chunk count allow
  static const int role_xcd = [] { const char* e = getenv("CTVAE_PAIR_ROLE_XCD"); return e ? atoi(e) : 0; }();
  if (will_pair && role_xcd && S >= 8) S &= ~7;
  if (wgrad_workspace_floats(g, S) > ws_floats) return kErrWorkspace;
  a.S = S;
  const int nchunks = ceil_div(a.Mc, MC);
  a.chunks_per_split = ceil_div(nchunks, S);
  a.part = ws;
  a.pbias = dbias ? ws + (size_t)S * a.rows_total * a.N : nullptr;
  int kt = 0;
  for (int c = 0; c < g.ncls; ++c) {
    a.ktile_start[c] = kt;
    kt += ceil_div(g.ntaps[c] * g.gC, KT);
  }
  a.ktile_start[g.ncls] = kt;
  a.ntiles = ceil_div(a.N, NT);
  dim3 grid(kt * a.ntiles, S), block(256);
  bool recorded = false;
  if (PairCtx* pc = pair_ctx(); pc != nullptr && !pc->haveB && xvec && dvec && !narrow && !big) {
    // ctvae_conv_backward: the lean 64x64 kernel shares a launch with the data-gradient kernel (pair_flush())
    auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    const int lw = lg2(g.Qw), lh2 = lg2(g.Qh);
    pc->haveB = true;
    pc->B = a;
    pc->lgQw = (lw >= 0 && lh2 >= 0) ? lw : -1;
    pc->lgQhw = (lw >= 0 && lh2 >= 0) ? lw + lh2 : -1;
    pc->lgC = lg2(g.gC);
    pc->gxB = grid.x; pc->gyB = grid.y;
    pc->flopsB = 2.0 * (double)a.Mc * a.N * a.rows_total;
    pc->bytesB = 4.0 * ((double)g.B * g.gH * g.gW * g.gC + (double)g.B * g.sH * g.sW * g.sC);
    recorded = true;
  }
  if (!recorded) {
  char name[160];
  if (big) snprintf(name, sizeof name, "wgrad_fast_kernel<2,2,2,2>");
  else if (xvec && dvec && !narrow) snprintf(name, sizeof name, "wgrad_fast_kernel<%d,%d,1,1>", narrow ? 4 : 2, narrow ? 1 : 2);
  else snprintf(name, sizeof name, "wgrad_kernel<%d,%d,%s,%s>", narrow ? 4 : 2, narrow ? 1 : 2, xvec ? "true" : "false",
                dvec ? "true" : "false");
  const double macs = (double)a.Mc * a.N * a.rows_total;
  if (prof_detailed()) {
    size_t l = strlen(name);
    snprintf(name + l, sizeof name - l, " M=%dx%d N=%d rows=%d S=%d", g.ncls, a.Mc, a.N, a.rows_total, S);
  }
  const double bytes = 4.0 * ((double)g.B * g.gH * g.gW * g.gC + (double)g.B * g.sH * g.sW * g.sC);
  ProfScope ps(name, st, 2.0 * macs, bytes);
#define CTVAE_WG(WK_, WN_)                                                                              \
  do {                                                                                                  \
    if (xvec && dvec) hipLaunchKernelGGL((wgrad_kernel<WK_, WN_, true, true>), grid, block, 0, st, a);  \
    else if (!xvec && dvec) hipLaunchKernelGGL((wgrad_kernel<WK_, WN_, false, true>), grid, block, 0, st, a); \
    else if (xvec && !dvec) hipLaunchKernelGGL((wgrad_kernel<WK_, WN_, true, false>), grid, block, 0, st, a); \
    else hipLaunchKernelGGL((wgrad_kernel<WK_, WN_, false, false>), grid, block, 0, st, a);             \
  } while (0)
  if (xvec && dvec && !narrow) {   // (measured: the 128x32 tile is faster with the plain-load kernel)
    auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    const int lw = lg2(g.Qw), lh2 = lg2(g.Qh);
    const int lgQw = (lw >= 0 && lh2 >= 0) ? lw : -1, lgQhw = (lw >= 0 && lh2 >= 0) ? lw + lh2 : -1;
    const int lgC = lg2(g.gC);
    if (big) hipLaunchKernelGGL((wgrad_fast_kernel<2, 2, 2, 2>), grid, block, 0, st, a, lgQw, lgQhw, lgC);
    else if (has_xf) hipLaunchKernelGGL((wgrad_fast_kernel<2, 2, 1, 1, true>), grid, block, 0, st, a, lgQw, lgQhw, lgC);
    else hipLaunchKernelGGL((wgrad_fast_kernel<2, 2, 1, 1>), grid, block, 0, st, a, lgQw, lgQhw, lgC);
  } else if (narrow) CTVAE_WG(4, 1);
  else CTVAE_WG(2, 2);
#undef CTVAE_WG
  }
  CTVAE_LAUNCH_CHECK();
  const long n = (long)a.rows_total * a.N;
  const float* part = a.part;
  const float* pbias = a.pbias;
  const int ncls = g.ncls, N_ = a.N;
  // deferred (ctvae_defer_*), recorded for the call's finishing launch (ctvae_conv_backward), or issued now
  return finish_reduce(part, dW, n, S, n, dbias ? pbias : nullptr, dbias, (long)N_, S * ncls, (long)N_, accumulate, st);
}

int launch_finish_recorded(const PairCtx& c, hipStream_t st) {
  const int nsk = c.haveSK ? c.sk.nblk : 0, nbf = c.haveBF ? c.bf.C : 0;
  ProfScope ps(nsk ? "reduce_partials_kernel (+ split-K finish)" : (nbf ? "reduce_partials_kernel (+ BN-backward finalize)" : "reduce_partials_kernel"),
               st, 0.0, c.bytesRed + (nsk ? c.bytesSK : 0.0) + (nbf ? 8.0 * c.bf.nblocks * c.bf.C : 0.0));
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(c.nb1 + c.nb2 + nsk + nbf), dim3(256), 0, st, c.j1, c.j2, c.nb1, c.accumulate,
                     nsk ? c.sk : SplitKJob{}, c.nb1 + c.nb2, nbf ? c.bf : BnFinJob{}, c.nb1 + c.nb2 + nsk);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_wgrad_fast_recorded(const PairCtx& c, hipStream_t st) {
  ProfScope ps("wgrad_fast_kernel<2,2,1,1>", st, c.flopsB, c.bytesB);
  if (c.B.xf_scale != nullptr)
    hipLaunchKernelGGL((wgrad_fast_kernel<2, 2, 1, 1, true>), dim3(c.gxB, c.gyB), dim3(256), 0, st, c.B, c.lgQw, c.lgQhw, c.lgC);
  else
    hipLaunchKernelGGL((wgrad_fast_kernel<2, 2, 1, 1>), dim3(c.gxB, c.gyB), dim3(256), 0, st, c.B, c.lgQw, c.lgQhw, c.lgC);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
