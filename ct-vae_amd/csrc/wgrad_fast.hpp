// Lean weight-gradient kernel body (moved out of wgrad.hip so that tapgemm_fast.hip can pair it with a data-gradient
// tile kernel in ONE launch, see conv_bwd_pair_kernel).  wgrad.hip documents the algorithm.
#pragma once
#include "common.hpp"

namespace ctvae {

struct WgradArgs {
  ConvGeom g;
  const float* X;
  const float* dY;
  float* part;   // [S][rows_total][N]
  float* pbias;  // [S*ncls][N] or null
  int Mc, N, S, chunks_per_split;
  int rows_total;          // taps_total * gC
  int ktile_start[kMaxCls + 1];
  int ntiles;
  // lazy BatchNorm apply (InXform, lean kernel): the X operand is act(X*scale[c] + shift[c]) of the tensor bound as X (the raw
  // BatchNorm input of the previous block), formed between the global load and the LDS store; padding stays 0
  const float* xf_scale;
  const float* xf_shift;
  int xf_act;
};

constexpr int MC = 32;

#ifdef CTVAE_PHASE_TIMING
// diagnostic build only (tools/pair_phase_probe.py): per-workgroup timestamps of the weight-gradient role, 100 MHz clock
static __device__ unsigned long long g_wphase[8 * 8192];
#define WPHASE(i) do { if (threadIdx.x == 0) g_wphase[((vby * vgx + vbx) & 8191) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define WPHASE(i) do {} while (0)
#endif

// ---- lean variant (gC % 4 == 0, N % 4 == 0): same tiling, minimal VALU around the MFMAs ---------------------
// f32 MFMA and VALU share the SIMD's vector datapath on gfx950 (tools/mfma_probe.hip), so address arithmetic is
// hoisted: a thread's tap / channel offset is constant for the launch, the per-pixel part (byte offset, y/x validity
// masks, scatter offset) is decoded once per 32-pixel chunk by 32 lanes with shifts, loads go through buffer
// resources (32-bit offsets, out-of-range -> 0, no zero-fill selects).
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
constexpr unsigned kOOBw = 0x80000000u;

// TK x TN: 32x32 MFMA tiles per wave.  <2,2,1,1> = 64x64 output tile; <2,2,2,2> = 128x128 for the wide layers of
// MCQ / CT-MCQ-VAE (rows and N multiples of 128): per MFMA half the LDS reads and half the L2->LDS bytes per FLOP
// (a 64x64 tile streams 1/16 B/FLOP, i.e. ~6 TB/s of L2 traffic at 95 TFLOP/s -- that, not the MFMA pipe, capped it).

// sX[MC*KT], sD[MC*NT], sPix / sMsk / sOutB [2][MC]: the workgroup's LDS, five DISTINCT arrays of the calling kernel (carving them
// from one buffer costs the tile kernels their no-alias information: 83 -> 116 VGPRs, measured on the tap-GEMM body).
// vbx, vby / vgx, vgy: the workgroup's position in / the size of the kernel's own (tile, slice) grid.
// PLACED: (vbx, vby) = (output tile, pixel slice) as they are -- the caller has done the XCD placement (conv_bwd_pair_kernel)
template <int WK, int WN, int TK, int TN, bool XF = false, bool PLACED = false>
__device__ __forceinline__ void wgrad_fast_body(const WgradArgs& a, int lgQw, int lgQhw, int lgC, float* sX, float* sD,
                                                unsigned (*sPix)[MC], unsigned (*sMsk)[MC], unsigned (*sOutB)[MC], int vbx,
                                                int vby, int vgx, int vgy) {
  constexpr int KT = WK * TK * 32, NT = WN * TN * 32;
  static_assert(WK * WN == 4, "4 waves");

  WPHASE(0);
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware order: every output tile of one pixel slice reads the same X / dY rows, and workgroups are dealt
  // round-robin to the 8 XCDs (one L2 each) in linear order -> give each group of 8 slices one XCD per slice, so a
  // slice's rows are filled into ONE L2 instead of eight.  The last (gridDim.y % 8) slices keep the plain order.
  int split = vby, xtile = vbx;
  if constexpr (!PLACED) {
    const int T = vgx, L = vby * T + vbx, full = (vgy >> 3) * 8 * T;
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
  const int gC = g.gC, N = a.N;
  const int Ktot = g.ntaps[cls] * gC;
  const int cbeg = split * a.chunks_per_split;
  const int nchunks_all = (a.Mc + MC - 1) / MC;
  int cend = cbeg + a.chunks_per_split;
  if (cend > nchunks_all) cend = nchunks_all;
  const int nch = cend - cbeg;

  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)((long)g.B * g.gH * g.gW * gC * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dY), 0, (int)((long)g.B * g.sH * g.sW * N * 4), 0x00020000);

  const int sc_py = g.py[cls], sc_px = g.px[cls];
  auto rowinfo = [&](int c, int buf) {
    if (tid < MC) {
      const int m = (cbeg + c) * MC + tid;
      unsigned pixb = kOOBw, msk = 0, outb = kOOBw;
      if (m < a.Mc) {
        int b, qy, qx;
        if (lgQw >= 0) {
          b = m >> lgQhw;
          const int rr = m & ((1 << lgQhw) - 1);
          qy = rr >> lgQw;
          qx = rr & ((1 << lgQw) - 1);
        } else {
          decode_m(g, m, b, qy, qx);
        }
        const int iy0 = qy * g.is, ix0 = qx * g.is;
        pixb = (unsigned)(((b * g.gH + iy0) * g.gW + ix0) * gC) * 4u;
        // bit (d+3): dy = d allowed, bit (d+11): dx = d allowed, d = -3 .. 4, as two bit ranges (tapgemm_fast_body.inc); the class's
        // parity offsets are read once per workgroup (sc_py / sc_px), not per chunk: this runs on HALF A WAVE while the other
        // fifteen half-waves of the workgroup wait for it at the barrier
        auto range_bits = [](int i0, int n) {
          const int lo = max(0, 3 - i0), hi = min(7, n + 2 - i0);
          return hi >= lo ? (2u << hi) - (1u << lo) : 0u;
        };
        msk = range_bits(iy0, g.gH) | (range_bits(ix0, g.gW) << 8);
        outb = (unsigned)(((b * g.sH + qy * g.os + sc_py) * g.sW + qx * g.os + sc_px) * N) * 4u;
      }
      sPix[buf][tid] = pixb; sMsk[buf][tid] = msk; sOutB[buf][tid] = outb;
    }
  };

  constexpr int XQ = KT / 4, X_V = (MC * XQ) / 256;
  constexpr int DQ = NT / 4, D_V = (MC * DQ) / 256;
  // this thread's fixed k column: tap and channel
  unsigned x_const = 0;
  int x_sy = 0, x_sx = 0;
  bool x_kok;
  [[maybe_unused]] f32x4 xsc = {0.f, 0.f, 0.f, 0.f}, xsh = {0.f, 0.f, 0.f, 0.f};   // XF: this thread's four channels, fixed for the launch
  {
    const int k = kt0 + 4 * (tid % XQ);
    x_kok = k < Ktot;
    int t = 0, c = 0;
    if (x_kok) {
      t = lgC >= 0 ? (k >> lgC) : (k / gC);
      c = k - t * gC;
    }
    if constexpr (XF) {
      if (x_kok) {
        xsc = *reinterpret_cast<const f32x4*>(a.xf_scale + c);
        xsh = *reinterpret_cast<const f32x4*>(a.xf_shift + c);
      }
    }
    const Tap tp = g.taps[cls][t];
    x_const = (unsigned)(((tp.dy * g.gW + tp.dx) * gC + c) * 4);
    x_sy = tp.dy + 3;
    x_sx = tp.dx + 11;
  }
  const unsigned d_const = (n0 + 4 * (tid % DQ) < N) ? (unsigned)(n0 + 4 * (tid % DQ)) * 4u : kOOBw;

  f32x4 rx[X_V], rd[D_V];
  [[maybe_unused]] float rx_in[X_V];   // XF: 1 where the loaded pixel lies inside the image (the shift applies), else 0
  auto load_chunk = [&](int buf) {
    // the row table is read first, for every row of the thread: as `ok ? sPix[r] + c : OOB` the table reads sat behind branches,
    // each with its own wait -- two dependent LDS round trips per load in front of the chunk's MFMAs
    unsigned mskv[X_V], pixv[X_V], outv[D_V];
#pragma unroll
    for (int j = 0; j < X_V; ++j) {
      const int r = tid / XQ + (256 / XQ) * j;
      mskv[j] = sMsk[buf][r];
      pixv[j] = sPix[buf][r];
    }
#pragma unroll
    for (int j = 0; j < D_V; ++j) outv[j] = sOutB[buf][tid / DQ + (256 / DQ) * j];
#pragma unroll
    for (int j = 0; j < X_V; ++j) {
      const bool ok = x_kok && ((mskv[j] >> x_sy) & (mskv[j] >> x_sx) & 1u) != 0;
      rx[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, (int)(ok ? pixv[j] + x_const : kOOBw), 0, 0));
      if constexpr (XF) rx_in[j] = (ok && pixv[j] != kOOBw) ? 1.f : 0.f;
    }
#pragma unroll
    for (int j = 0; j < D_V; ++j)
      rd[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rD, (int)((outv[j] | d_const) >= kOOBw ? kOOBw : outv[j] + d_const), 0, 0));
  };
  auto store_chunk = [&]() {
    if constexpr (XF) {
      const float nslope = a.xf_act == ACT_LRELU ? kLeaky : (a.xf_act == ACT_RELU ? 0.f : 1.f);   // act(t) = max(t, t * nslope)
#pragma unroll
      for (int j = 0; j < X_V; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = rx[j][e] * xsc[e] + xsh[e] * rx_in[j];
          rx[j][e] = fmaxf(t, t * nslope);
        }
    }
#pragma unroll
    for (int j = 0; j < X_V; ++j) *reinterpret_cast<f32x4*>(&sX[(tid / XQ + (256 / XQ) * j) * KT + 4 * (tid % XQ)]) = rx[j];
#pragma unroll
    for (int j = 0; j < D_V; ++j) *reinterpret_cast<f32x4*>(&sD[(tid / DQ + (256 / DQ) * j) * NT + 4 * (tid % DQ)]) = rd[j];
  };

  // Where this lane's accumulator rows go in the slab, worked out BEFORE the main loop: behind it the tap lookups (a load from
  // the argument block per row group) and two integer divisions were a dependent tail of every workgroup of the role
  // that finishes a paired launch.  4 consecutive k rows: same tap when gC % 4 == 0.
  const int wcr = g.wrs / g.wCo, wtc = g.wts / g.wCo;          // rows between consecutive channels / taps (geom.hpp wts / wrs)
  int wrow0[TK][4];
#pragma unroll
  for (int i = 0; i < TK; ++i)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int k0 = kt0 + (wk * TK + i) * 32 + 8 * q4 + 4 * lh;
      const int t = k0 < Ktot ? (lgC >= 0 ? (k0 >> lgC) : (k0 / gC)) : 0;
      const int r0 = g.taps[cls][t].wtap * wtc + (k0 - t * gC) * wcr;
      wrow0[i][q4] = k0 < Ktot ? r0 : -1;
    }
  f32x16 acc[TK][TN];
#pragma unroll
  for (int i = 0; i < TK; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum = 0.f;
  const bool do_bias = (a.pbias != nullptr) && (kt0 == 0) && (tid < NT);

  WPHASE(1);
  if (nch > 0) {
    rowinfo(0, 0);
    __syncthreads();
    load_chunk(0);
    if (nch > 1) rowinfo(1, 1);
    store_chunk();
    __syncthreads();
    WPHASE(2);
    for (int c = 0; c < nch; ++c) {
      if (c + 1 < nch) load_chunk((c + 1) & 1);
      if (c + 2 < nch) rowinfo(c + 2, c & 1);
      // The 16 steps of a chunk in groups of four, fragments one group ahead: as `read; mfma` per step the compiler put every
      // pair of MFMAs behind the wait for its own LDS reads (~130 cycles of an idle pipe per pair).  The order inside a group is
      // pinned: an MFMA, then the next group's reads that can issue while it runs.
      {
        constexpr int GS = 4, NG = (MC / 2) / GS;
        float fa[2][GS][TK], fb[2][GS][TN];
        auto rdg = [&](int g, int set) {
#pragma unroll
          for (int q = 0; q < GS; ++q) {
            const int s = g * GS + q;
#pragma unroll
            for (int i = 0; i < TK; ++i) fa[set][q][i] = sX[(2 * s + lh) * KT + (wk * TK + i) * 32 + li];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[set][q][j] = sD[(2 * s + lh) * NT + (wn * TN + j) * 32 + li];
          }
        };
        rdg(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (g + 1 < NG) rdg(g + 1, (g + 1) & 1);
#pragma unroll
          for (int q = 0; q < GS; ++q)
#pragma unroll
            for (int i = 0; i < TK; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][q][i], fb[g & 1][q][j], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < GS; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, TK * TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TK + TN, 0);
          }
        }
      }
      if (do_bias) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
        for (int r = 0; r < MC; r += 4) {
          t0 += sD[r * NT + tid]; t1 += sD[(r + 1) * NT + tid]; t2 += sD[(r + 2) * NT + tid]; t3 += sD[(r + 3) * NT + tid];
        }
        bsum += (t0 + t1) + (t2 + t3);
      }
      __syncthreads();
      if (c + 1 < nch) {
        store_chunk();
        __syncthreads();
      }
    }
  }

  WPHASE(3);
#pragma unroll
  for (int i = 0; i < TK; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        if (wrow0[i][q4] >= 0 && col < N) {
          float* dst = a.part + ((long)split * a.rows_total + wrow0[i][q4]) * N + col;
#pragma unroll
          for (int q = 0; q < 4; ++q) dst[(long)q * wcr * N] = acc[i][j][4 * q4 + q];
        }
      }
    }
  if (do_bias && n0 + tid < N) a.pbias[(long)(split * g.ncls + cls) * N + n0 + tid] = bsum;
#ifdef CTVAE_PHASE_TIMING
  WPHASE(4);
  __builtin_amdgcn_s_waitcnt(0);
  WPHASE(5);
#endif
}

}  // namespace ctvae
