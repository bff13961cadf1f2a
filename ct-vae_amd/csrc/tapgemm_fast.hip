// Tap-GEMM main kernel (vector path: gathered channels % 32 == 0, N % 4 == 0).  See tapgemm.hip for the op
// list and geom.hpp for the geometry.
//
// Design rule measured on gfx950 (tools/mfma_probe.hip): v_mfma_f32_32x32x2_f32 runs on the SIMD's vector
// datapath, so VALU instructions of ANY wave on that SIMD do not overlap with it -- SIMD time = MFMA time +
// VALU time.  Hence everything around the 64-cycle MFMAs is written to issue as few VALU instructions as
// possible:
//   * global loads go through buffer resources: 32-bit byte offsets, the wave-uniform part of an address rides
//     in the scalar offset, and an out-of-range offset returns 0 -> zero padding / tile edges cost one select;
//   * per-row offsets and the separable y/x validity masks of the im2col gather are computed once per tile;
//   * the epilogue uses 32-bit indices, drops stores by sending them out of range (no branches), and takes
//     per-row scatter indices from arithmetic (dense scatter) or one 16-byte LDS read per 4 rows (parity classes);
//   * fragments of the next 8-deep k-group are read from LDS before the MFMAs of the current one;
//   * per-tap constants sit in lanes and are fetched with v_readlane (no scalar loads, no division in the loop).
// Workgroup = 4 waves, tile BM x BN = (WM*TM*32) x (WN*TN*32), K chunks of 32; grids with <= 4 workgroups per CU run
// the double-buffered, explicitly software-pipelined loop (PF = 3), larger grids the single-buffer loop (PF = 0).
#include <type_traits>

#include "prof.hpp"
#include "tapgemm.hpp"

namespace ctvae {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0x80000000u;  // byte offset beyond any buffer we bind (tensors are < 2 GiB)

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned voff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

#ifdef CTVAE_PHASE_TIMING
// diagnostic build only (tools/phase_probe.py): per-workgroup timestamps of the kernel phases, 100 MHz clock
__device__ unsigned long long g_phase[8 * 8192];
#define PHASE(i) do { if (threadIdx.x == 0 && blockIdx.z == 0) g_phase[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 8 + (i)] = wall_clock64(); } while (0)
extern "C" int ctvae_debug_phase_read(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), (size_t)n * 8);
}
#else
#define PHASE(i) do {} while (0)
#endif

template <int WM, int WN, int TM, int TN, bool WT, int PF>
__global__ __launch_bounds__(256) void tapgemm_fast_kernel(const TapGemmArgs a) {
  static_assert(PF == 0 || PF == 3, "PF: 0 single LDS buffer, 3 double buffer + explicit software pipeline");
  constexpr bool DB = PF == 3;
  constexpr int NSET = 1;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int SA = BM * LDK, SB = WT ? BN * LDK : KC * BN;
  constexpr int NBUF = DB ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float sAbuf[NBUF * SA];
  __shared__ __attribute__((aligned(16))) float sBbuf[NBUF * SB];
  __shared__ __attribute__((aligned(16))) int sOut[BM];

  PHASE(0);
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  int cls_i = a.cls_rot == 1 ? ((blockIdx.y + blockIdx.z) & 3) : blockIdx.y, bx = blockIdx.x;
  if (a.cls_rot == 2) {   // classes interleaved in groups of 8 workgroups (one per XCD): L = ((t_hi * 4 + class) * 8 + t_lo)
    const int L = blockIdx.y * gridDim.x + blockIdx.x;
    cls_i = (L >> 3) & 3;
    bx = ((L >> 5) << 3) | (L & 7);
  }
  const int cls = a.cls_order[cls_i];
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the n-tiles of one
  // m-tile (same gathered pixels) would land in 8 different L2s.  Remap so that each XCD walks a contiguous range of
  // the (m-tile, n-tile) space; the remainder (grid.x % 8) keeps its place.
  int tile = bx;
  {
    const int per = gridDim.x >> 3;
    if (tile < per * 8) tile = (tile & 7) * per + (tile >> 3);
  }
  const int mt = tile / a.ntiles, nt = tile - mt * a.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int ntaps = g.ntaps[cls];
  const int gC = g.gC, N = a.N;
  const int nch = ntaps * gC / KC;
  const bool dense = (g.os == 1);   // scatter index == m

  // The gathered operand is addressed as  (G - BIAS) + a_off[row] + (tap offset + BIAS + channel offset):  the second
  // term is the per-lane VGPR offset, the third is wave-uniform and rides in the load's scalar-offset operand -- no VALU
  // add per row and chunk.  BIAS (>= the most negative tap offset) keeps the scalar part non-negative.
  const unsigned gbias = (unsigned)((3 * g.gW + 3) * gC) * 4u;
  const __amdgpu_buffer_rsrc_t rG = make_rsrc(reinterpret_cast<const char*>(a.G) - gbias,
                                              (long)g.B * g.gH * g.gW * gC * 4 + gbias);
  long wtaps = 0;
  for (int c = 0; c < g.ncls; ++c) wtaps += g.ntaps[c];
  const __amdgpu_buffer_rsrc_t rW = make_rsrc(a.W, wtaps * g.wCi * g.wCo * 4);

  // ---- per-thread constants of the tile ---------------------------------------------------------------
  constexpr int A_V = BM / 32;  // 16-B loads per thread and chunk (gathered operand)
  constexpr int B_V = BN / 32;  // 16-B loads per thread and chunk (weights)
  unsigned a_off[A_V], a_ok[A_V];
#pragma unroll
  for (int j = 0; j < A_V; ++j) {
    const int m = m0 + (tid >> 3) + 32 * j;
    unsigned msk = 0;
    int pix = 0;
    if (m < a.Mc) {
      int b, qy, qx;
      decode_m_fast(a, m, b, qy, qx);
      const int iy0 = qy * g.is, ix0 = qx * g.is;
      pix = (b * g.gH + iy0) * g.gW + ix0;
#pragma unroll
      for (int d = -3; d <= 4; ++d) {  // bit (d+3): dy = d allowed; bit (d+11): dx = d allowed
        if ((unsigned)(iy0 + d) < (unsigned)g.gH) msk |= 1u << (d + 3);
        if ((unsigned)(ix0 + d) < (unsigned)g.gW) msk |= 1u << (d + 11);
      }
    }
    a_ok[j] = msk;
    a_off[j] = (unsigned)(pix * gC + 4 * (tid & 7)) * 4u;
  }
  unsigned b_off[B_V];
#pragma unroll
  for (int j = 0; j < B_V; ++j) {
    if constexpr (!WT) {   // Wmat[t][c][n] = W[t][c][n]: rows k, 16 B along n
      const int f = tid + 256 * j;
      const int kr = f / (BN / 4), nq = f - kr * (BN / 4);
      const int n = n0 + 4 * nq;
      b_off[j] = n < N ? (unsigned)(kr * g.wCo + n) * 4u : kOOB;
    } else {               // Wmat[t][c][n] = W[t][n][c]: rows n, 16 B along c
      const int n = n0 + (tid >> 3) + 32 * j;
      b_off[j] = n < N ? (unsigned)(n * g.wCo + 4 * (tid & 7)) * 4u : kOOB;
    }
  }
  if (!dense) {
    for (int r = tid; r < BM; r += 256) {
      const int m = m0 + r;
      int sp = -1;
      if (m < a.Mc) {
        int b, qy, qx;
        decode_m_fast(a, m, b, qy, qx);
        sp = scatter_pix(g, cls, b, qy, qx);
      }
      sOut[r] = sp;
    }
  }

  f32x4 ra[NSET][A_V], rb[NSET][B_V];
  auto store_chunk = [&](int buf, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    float* sA = sAbuf + buf * SA;
    float* sB = sBbuf + buf * SB;
#pragma unroll
    for (int j = 0; j < A_V; ++j) *reinterpret_cast<f32x4*>(&sA[((tid >> 3) + 32 * j) * LDK + 4 * (tid & 7)]) = ra[set][j];
#pragma unroll
    for (int j = 0; j < B_V; ++j) {
      if constexpr (!WT) {
        const int f = tid + 256 * j;
        const int kr = f / (BN / 4), nq = f - kr * (BN / 4);
        *reinterpret_cast<f32x4*>(&sB[kr * BN + 4 * nq]) = rb[set][j];
      } else {
        *reinterpret_cast<f32x4*>(&sB[((tid >> 3) + 32 * j) * LDK + 4 * (tid & 7)]) = rb[set][j];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // split-K: this workgroup owns chunks [c0, c1) of the class's K range
  int c0 = 0, c1 = nch;
  if (a.splitk > 1) {
    const int cps = (nch + a.splitk - 1) / a.splitk;
    c0 = blockIdx.z * cps;
    c1 = c0 + cps < nch ? c0 + cps : nch;
  }
  using S0 = std::integral_constant<int, 0>;
  auto compute_chunk = [&](int cur) {
    const float* sA = sAbuf + cur * SA;
    const float* sB = sBbuf + cur * SB;
    f32x4 af[2][TM];
    float bf[2][TN][4];
    auto read_frags = [&](int kg, int slot) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[slot][i] = *reinterpret_cast<const f32x4*>(&sA[((wm * TM + i) * 32 + li) * LDK + kg * 8 + 4 * lh]);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (WT) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(&sB[((wn * TN + j) * 32 + li) * LDK + kg * 8 + 4 * lh]);
          bf[slot][j][0] = t4[0]; bf[slot][j][1] = t4[1]; bf[slot][j][2] = t4[2]; bf[slot][j][3] = t4[3];
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) bf[slot][j][s] = sB[(kg * 8 + 4 * lh + s) * BN + (wn * TN + j) * 32 + li];
        }
      }
    };
    read_frags(0, 0);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      if (kg + 1 < 4) read_frags(kg + 1, (kg + 1) & 1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg & 1][i][s], bf[kg & 1][j][s], acc[i][j], 0, 0, 0);
    }
  };

  // Tap constants live in lanes (lane t = tap t) and are fetched with v_readlane: no scalar loads in the loop (an
  // s_load forces an lgkmcnt(0) drain that also waits for every LDS read in flight) and no integer division.
  unsigned tv_off = 0, tv_sh = 0, tv_w = 0;
  if (lane < ntaps) {
    const Tap tp = g.taps[cls][lane];
    tv_off = (unsigned)(((tp.dy * g.gW + tp.dx) * gC) * 4);
    tv_sh = (unsigned)(tp.dy + 3) | ((unsigned)(tp.dx + 11) << 8);
    tv_w = WT ? (unsigned)(tp.wtap * g.wCi * g.wCo) * 4u : (unsigned)(tp.wtap * g.wCi * g.wCo) * 4u;
  }
  // per row: bit t = tap t reads inside the image (<= 16 taps); one bit-field extract per row and chunk then decides
  // between the row offset and the out-of-range offset (top bit set) -- no compare, no select, no add in the loop
  unsigned a_tm[A_V];
#pragma unroll
  for (int j = 0; j < A_V; ++j) a_tm[j] = 0;
  for (int t = 0; t < ntaps; ++t) {
    const unsigned sh = (unsigned)__builtin_amdgcn_readlane((int)tv_sh, t);
    const unsigned sy = sh & 0xff, sx = sh >> 8;
#pragma unroll
    for (int j = 0; j < A_V; ++j) a_tm[j] |= ((a_ok[j] >> sy) & (a_ok[j] >> sx) & 1u) << t;
  }
  int lt = (c0 * KC) / gC, lci = c0 * KC - lt * gC;    // tap / channel offset of the next chunk to load
  auto load_next = [&]() {
    const unsigned tapoff = (unsigned)__builtin_amdgcn_readlane((int)tv_off, lt) + gbias + (unsigned)lci * 4u;
    const unsigned wsoff = (unsigned)__builtin_amdgcn_readlane((int)tv_w, lt) + (WT ? (unsigned)lci * 4u : (unsigned)(lci * g.wCo) * 4u);
#pragma unroll
    for (int j = 0; j < A_V; ++j) {
      const unsigned out = ((a_tm[j] >> lt) & 1u) ^ 1u;            // 1 = this tap falls outside for this row
      ra[0][j] = buf_load4(rG, a_off[j] | (out << 31), tapoff);
    }
#pragma unroll
    for (int j = 0; j < B_V; ++j) rb[0][j] = buf_load4(rW, b_off[j], wsoff);
    lci += KC;
    if (lci == gC) { lci = 0; ++lt; }
  };
  if constexpr (PF == 3) {
    // ---- software-pipelined loop: one barrier per chunk, no exposed latency ---------------------------------
    // iteration c:  A) chunk c+1 registers -> other LDS buffer, global loads of chunk c+2, LDS reads of k-groups
    //                  2,3 of chunk c, MFMAs of k-groups 0,1 (fragments read during the previous iteration)
    //               B) barrier: chunk c+1 is complete in LDS, nobody reads this buffer's k-groups any more
    //               C) LDS reads of k-groups 0,1 of chunk c+1, MFMAs of k-groups 2,3 of chunk c
    // sched_barrier keeps the compiler from sinking the reads next to their uses (it otherwise serialises
    // read -> wait -> MFMA and leaves the matrix pipe idle for an LDS latency 6x per chunk).
    // Tap constants live in lanes (lane t = tap t) and are fetched with v_readlane: no scalar loads, hence no
    // lgkmcnt(0) drains, and no integer division in the loop.
    f32x4 af[4][TM];
    float bf[4][TN][4];
    auto read_kg = [&](int buf, auto kg_c) {
      constexpr int kg = decltype(kg_c)::value;
      const float* sA = sAbuf + buf * SA;
      const float* sB = sBbuf + buf * SB;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[kg][i] = *reinterpret_cast<const f32x4*>(&sA[((wm * TM + i) * 32 + li) * LDK + kg * 8 + 4 * lh]);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (WT) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(&sB[((wn * TN + j) * 32 + li) * LDK + kg * 8 + 4 * lh]);
          bf[kg][j][0] = t4[0]; bf[kg][j][1] = t4[1]; bf[kg][j][2] = t4[2]; bf[kg][j][3] = t4[3];
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) bf[kg][j][q] = sB[(kg * 8 + 4 * lh + q) * BN + (wn * TN + j) * 32 + li];
        }
      }
    };
    auto mfma_kg = [&](auto kg_c) {
      constexpr int kg = decltype(kg_c)::value;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg][i][q], bf[kg][j][q], acc[i][j], 0, 0, 0);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    PHASE(1);
    if (c0 < c1) {
      load_next();
      store_chunk(0, S0{});
    }
    __syncthreads();
    PHASE(2);
    if (c0 < c1) {
      read_kg(0, K0{});
      read_kg(0, K1{});
      if (c0 + 1 < c1) load_next();
    }
    for (int c = c0; c < c1; ++c) {
      const int cur = (c - c0) & 1;
      const bool more = c + 1 < c1;
      if (more) store_chunk(cur ^ 1, S0{});
      if (c + 2 < c1) load_next();
      read_kg(cur, K2{});
      read_kg(cur, K3{});
      __builtin_amdgcn_sched_barrier(0);
      mfma_kg(K0{});
      mfma_kg(K1{});
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        read_kg(cur ^ 1, K0{});
        read_kg(cur ^ 1, K1{});
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_kg(K2{});
      mfma_kg(K3{});
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // single LDS buffer: grids with many workgroups per CU hide the latency by occupancy (a second buffer would cost it)
    if (c0 < c1) {
      load_next();
      store_chunk(0, S0{});
    }
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
      if (c + 1 < c1) load_next();
      compute_chunk(0);
      __syncthreads();
      if (c + 1 < c1) {
        store_chunk(0, S0{});
        __syncthreads();
      }
    }
  }   // PF != 3

  // ---- epilogue -----------------------------------------------------------------------------------------
  PHASE(3);
  const long sbytes = (long)g.B * g.sH * g.sW * N * 4;
  if (a.splitk > 1) {  // raw partial sums; bias / activation happen in splitk_finish_kernel
    const __amdgpu_buffer_rsrc_t rP = make_rsrc(a.part + (long)blockIdx.z * (sbytes / 4), sbytes);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int row0 = (wm * TM + i) * 32 + 8 * q4 + 4 * lh;
          int sp4[4];
          if (dense) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sp4[q] = (m0 + row0 + q < a.Mc) ? m0 + row0 + q : -1;
          } else {
            const int4 t = *reinterpret_cast<const int4*>(&sOut[row0]);
            sp4[0] = t.x; sp4[1] = t.y; sp4[2] = t.z; sp4[3] = t.w;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            buf_store1(rP, (sp4[q] >= 0 && col < N) ? ((unsigned)sp4[q] * (unsigned)N + (unsigned)col) * 4u : kOOB,
                       acc[i][j][4 * q4 + q]);
        }
    }
    return;
  }

  const __amdgpu_buffer_rsrc_t rS = make_rsrc(a.S, sbytes);
  const __amdgpu_buffer_rsrc_t rAdd = make_rsrc(a.add, a.add != nullptr ? sbytes : 0);
  const __amdgpu_buffer_rsrc_t rMask = make_rsrc(a.mask, a.mask != nullptr ? sbytes : 0);
  const bool bnb = a.bnb_part != nullptr;
  const __amdgpu_buffer_rsrc_t rY = make_rsrc(a.bnb_y, bnb ? sbytes : 0);
  const bool full_m = (m0 + BM <= a.Mc);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + li;
    const bool cok = col < N;
    const float bv = (a.bias != nullptr && cok) ? a.bias[col] : 0.f;
    float cnt = 0.f, s1 = 0.f;     // BN statistics of this lane's column (valid rows only)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int row0 = (wm * TM + i) * 32 + 8 * q4 + 4 * lh;
        unsigned voff[4];
        if (dense) {
          const unsigned base = ((unsigned)(m0 + row0) * (unsigned)N + (unsigned)col) * 4u;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool ok = cok && (full_m || m0 + row0 + q < a.Mc);
            voff[q] = ok ? base + (unsigned)(q * N) * 4u : kOOB;
          }
        } else {
          const int4 t = *reinterpret_cast<const int4*>(&sOut[row0]);
          const int sp4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
          for (int q = 0; q < 4; ++q)
            voff[q] = (cok && sp4[q] >= 0) ? ((unsigned)sp4[q] * (unsigned)N + (unsigned)col) * 4u : kOOB;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = acc[i][j][4 * q4 + q] + bv;
          acc[i][j][4 * q4 + q] = v;                      // kept for the variance pass
          if (a.bn_part != nullptr && voff[q] != kOOB) { cnt += 1.f; s1 += v; }
          if (a.add != nullptr) v += buf_load1(rAdd, voff[q]);
          if (a.act == ACT_LRELU) v = fmaxf(v, v * kLeaky);
          else if (a.act == ACT_RELU) v = fmaxf(v, 0.f);
          else if (a.act == ACT_TANH) v = act_fwd(v, ACT_TANH);
          if (a.mask != nullptr) {
            const float mk = buf_load1(rMask, voff[q]);
            if (a.mask_act == ACT_LRELU) v = mk > 0.f ? v : v * kLeaky;
            else if (a.mask_act == ACT_RELU) v = mk > 0.f ? v : 0.f;
            else if (a.mask_act == ACT_TANH) v *= 1.f - mk * mk;
          }
          buf_store1(rS, voff[q], v);
          if (bnb) acc[i][j][4 * q4 + q] = v;            // the stored gradient, for the BN-backward sums below
        }
      }
    }
    // ---- fused BatchNorm-backward sums: S is d/d(act(BN(y))) of the producing layer -------------------------
    if (bnb) {
      float yv[TM][16];
      bool okv[TM][16];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int row0 = (wm * TM + i) * 32 + 8 * q4 + 4 * lh;
          unsigned voff[4];
          if (dense) {
            const unsigned base = ((unsigned)(m0 + row0) * (unsigned)N + (unsigned)col) * 4u;
#pragma unroll
            for (int q = 0; q < 4; ++q) voff[q] = (cok && (full_m || m0 + row0 + q < a.Mc)) ? base + (unsigned)(q * N) * 4u : kOOB;
          } else {
            const int4 t = *reinterpret_cast<const int4*>(&sOut[row0]);
            const int sp4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
              voff[q] = (cok && sp4[q] >= 0) ? ((unsigned)sp4[q] * (unsigned)N + (unsigned)col) * 4u : kOOB;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            yv[i][4 * q4 + q] = buf_load1(rY, voff[q]);
            okv[i][4 * q4 + q] = voff[q] != kOOB;
          }
        }
      const float bmean = cok ? a.bnb_mean[col] : 0.f, binv = cok ? a.bnb_invstd[col] : 0.f;
      const float bgm = cok ? a.bnb_gamma[col] : 0.f, bbt = cok ? a.bnb_beta[col] : 0.f;
      float s1b = 0.f, s2b = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float xh = (yv[i][r] - bmean) * binv;
          const float gp = okv[i][r] ? acc[i][j][r] * act_bwd_from_out(act_fwd(bgm * xh + bbt, a.bnb_act), a.bnb_act) : 0.f;
          s1b += gp;
          s2b += gp * xh;
        }
      s1b += __shfl_xor(s1b, 32, 64);
      s2b += __shfl_xor(s2b, 32, 64);
      if (lh == 0) {  // the A buffer is free after the main loop
        float* st = &sAbuf[(wm * BN + (wn * TN + j) * 32 + li) * 2];
        st[0] = s1b; st[1] = s2b;
      }
    }
    // ---- fused BatchNorm statistics of this tile (train-mode BN follows the conv: vanilla_vae.py:28-31) ----
    if (a.bn_part != nullptr) {
      const float mean0 = cnt > 0.f ? s1 / cnt : 0.f;
      float m2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int row0 = (wm * TM + i) * 32 + 8 * q4 + 4 * lh;
          bool ok4[4];
          if (dense) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ok4[q] = cok && (full_m || m0 + row0 + q < a.Mc);
          } else {
            const int4 t = *reinterpret_cast<const int4*>(&sOut[row0]);
            ok4[0] = cok && t.x >= 0; ok4[1] = cok && t.y >= 0; ok4[2] = cok && t.z >= 0; ok4[3] = cok && t.w >= 0;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float d = acc[i][j][4 * q4 + q] - mean0;
            m2 += ok4[q] ? d * d : 0.f;
          }
        }
      // merge the two lane halves (Chan et al.)
      float mean = mean0;
      const float ocnt = __shfl_xor(cnt, 32, 64), omean = __shfl_xor(mean, 32, 64), om2 = __shfl_xor(m2, 32, 64);
      const float ntot = cnt + ocnt;
      if (ntot > 0.f) {
        const float d = omean - mean;
        m2 = m2 + om2 + d * d * (cnt * ocnt / ntot);
        mean = mean + d * (ocnt / ntot);
      }
      if (lh == 0) {  // the A buffer is free after the main loop (its last iteration ended with a barrier)
        float* st = &sAbuf[(wm * BN + (wn * TN + j) * 32 + li) * 3];
        st[0] = ntot; st[1] = mean; st[2] = m2;
      }
    }
  }
  if (a.bn_part != nullptr) {
    __syncthreads();
    if (tid < BN && n0 + tid < N) {
      float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        const float nb = sAbuf[(w * BN + tid) * 3], mb = sAbuf[(w * BN + tid) * 3 + 1], qb = sAbuf[(w * BN + tid) * 3 + 2];
        if (nb > 0.f) {
          const float nt2 = n + nb, d = mb - mean;
          mean += d * (nb / nt2);
          m2 += qb + d * d * (n * nb / nt2);
          n = nt2;
        }
      }
      float* p = a.bn_part + ((long)(cls * a.mtiles + mt) * N + n0 + tid) * 3;
      p[0] = n; p[1] = mean; p[2] = m2;
    }
  }
  if (bnb) {
    __syncthreads();
    if (tid < BN && n0 + tid < N) {
      float s1b = 0.f, s2b = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {   // fixed order -> deterministic
        s1b += sAbuf[(w * BN + tid) * 2];
        s2b += sAbuf[(w * BN + tid) * 2 + 1];
      }
      float* p = a.bnb_part + ((long)(cls * a.mtiles + mt) * N + n0 + tid) * 2;
      p[0] = s1b; p[1] = s2b;
    }
  }
#ifdef CTVAE_PHASE_TIMING
  PHASE(4);
  __builtin_amdgcn_s_waitcnt(0);   // all counters zero: the stores have retired
  PHASE(5);
#endif
}

template <int WM, int WN, int TM, int TN>
static int launch_fast_cfg(const TapGemmArgs& a, int pf, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  TapGemmArgs args = a;
  args.mtiles = ceil_div(a.Mc, BM);
  args.ntiles = ceil_div(a.N, BN);
  const bool wt = a.g.wT != 0;
  dim3 grid(args.mtiles * args.ntiles, a.g.ncls, a.splitk > 1 ? a.splitk : 1), block(256);
  // Parity classes of a stride-2 3x3 layer reduce over 1, 2, 2 and 4 taps: workgroups of one class are 4x longer than those
  // of another.  Workgroups are dealt to the CUs in linear order (x, then y = class row, then z = K slice), so with the
  // plain order the CUs that got the 4-tap rows finish last while the others idle (resident grids), or the longest
  // workgroups start last (larger grids).  Resident grids (<= 1024 workgroups): class rows in the order
  // [fewest, middle, most, middle] taps, so that the rows 256 workgroups apart -- which share CUs when a class row is half
  // a round -- pair up as (1 + 4) and (2 + 2), and rotated by the K slice when the launch is split (a CU then sees three
  // different classes instead of three times the same).  VanillaVAE bs=256: 1.882 -> 1.849 ms per step (M=4x1024 sk=3 launches
  // 43 -> 33 us, M=4x4096 42 -> 38 us).  Larger grids keep the plain order: longest classes first 1.853 ms, classes
  // interleaved in groups of 8 workgroups 1.880 ms against 1.849 / 1.857 ms (same box).
  for (int c = 0; c < kMaxCls; ++c) args.cls_order[c] = c;
  args.cls_rot = 0;
  static const int no_cls_order = [] { const char* e = getenv("CTVAE_NO_CLS_ORDER"); return e ? atoi(e) : 0; }();   // diagnostic
  if (a.g.ncls == 4 && !no_cls_order) {
    int idx[4] = {0, 1, 2, 3};
    for (int i = 1; i < 4; ++i)
      for (int j = i; j > 0 && a.g.ntaps[idx[j]] < a.g.ntaps[idx[j - 1]]; --j) { const int t = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = t; }
    const long total = (long)grid.x * grid.y * grid.z;
    if (total <= 1024) {
      args.cls_order[0] = idx[0]; args.cls_order[1] = idx[1]; args.cls_order[2] = idx[3]; args.cls_order[3] = idx[2];
      args.cls_rot = grid.z > 1 ? 1 : 0;
    } else {
      static const int big = [] { const char* e = getenv("CTVAE_CLS_BIG"); return e ? atoi(e) : 0; }();   // diagnostic: 1 longest first, 2 interleaved -- both measured slower (below)
      if (big == 1) for (int c = 0; c < 4; ++c) args.cls_order[c] = idx[3 - c];
      if (big == 2 && grid.x % 8 == 0 && grid.z == 1) args.cls_rot = 2;
    }
  }
  char name[160];
  snprintf(name, sizeof name, "tapgemm_fast_kernel<%d,%d,%d,%d,%s,%d>", WM, WN, TM, TN, wt ? "true" : "false", pf);
  double macs = 0;
  for (int c = 0; c < a.g.ncls; ++c) macs += (double)a.Mc * a.N * a.g.ntaps[c] * a.g.gC;
  if (prof_detailed()) {
    size_t l = strlen(name);
    snprintf(name + l, sizeof name - l, " M=%dx%d N=%d C=%d taps=%d sk=%d", a.g.ncls, a.Mc, a.N, a.g.gC, a.g.ntaps[a.g.ncls - 1],
             a.splitk);
  }
  const double bytes = 4.0 * ((double)a.g.B * a.g.gH * a.g.gW * a.g.gC + (double)a.g.B * a.g.sH * a.g.sW * a.g.sC);
  ProfScope ps(name, st, 2.0 * macs, bytes);
  if (!wt) {
    // PF 1 / 2 (plain double buffer, two-chunk register prefetch) were measured and lost to 3: not instantiated
    if (pf == 3) hipLaunchKernelGGL((tapgemm_fast_kernel<WM, WN, TM, TN, false, 3>), grid, block, 0, st, args);
    else hipLaunchKernelGGL((tapgemm_fast_kernel<WM, WN, TM, TN, false, 0>), grid, block, 0, st, args);
  } else {
    if (pf == 3) hipLaunchKernelGGL((tapgemm_fast_kernel<WM, WN, TM, TN, true, 3>), grid, block, 0, st, args);
    else hipLaunchKernelGGL((tapgemm_fast_kernel<WM, WN, TM, TN, true, 0>), grid, block, 0, st, args);
  }
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_tapgemm_fast(const TapGemmArgs& a, const TapGemmPlan& plan, int pf, hipStream_t st) {
  if (plan.BN == 32) return launch_fast_cfg<4, 1, 1, 1>(a, pf, st);
  if (plan.BM == 128) return launch_fast_cfg<2, 2, 2, 1>(a, pf, st);
  return launch_fast_cfg<2, 2, 1, 1>(a, pf, st);
}

}  // namespace ctvae
