// Tap-GEMM: the one f32-MFMA implicit-GEMM kernel behind every Conv2d / ConvTranspose2d / Linear
// forward and data-gradient on the hot path (see geom.hpp for the geometry model).
//
//   S[spix(m)][n] = epi( sum_t sum_c G[gpix(m,t)][c] * Wmat[t][c][n] )
//
// Replaces (reference call sites): nn.Conv2d k3s2 (vanilla_vae.py:28-29), nn.ConvTranspose2d k3s2
// (vanilla_vae.py:50-55,65-70), nn.Conv2d k3s1 (vanilla_vae.py:73-74, mcq_vae.py:178-179, vq_vae.py:63-64),
// nn.Conv2d k4s2 (mcq_vae.py:170-171), k1 (vq_vae.py:66-67, mcq_vae.py:189-190), nn.ConvTranspose2d k4s2
// (mcq_vae.py:223-236), nn.Linear (vanilla_vae.py:36-37,43) and autograd's dgrad of each.
//
// Design (CDNA4): one 256-thread workgroup = 4 waves computes a BM x BN tile of S with
// v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma chain -> 1e-4 parity is safe).  Per 32-deep K
// chunk the gathered operand is im2col'ed on the fly from NHWC global memory into an LDS tile
// [BM][32+4] (16-B vector loads along channels, rows padded so ds_read_b128 is conflict-free), the
// weight slab goes to LDS as [32][BN] (or [BN][32+4] when the per-tap transpose is needed for
// dgrad).  Next-chunk global loads are issued before the MFMA block of the current chunk
// (register prefetch) and occupancy (small accumulators) hides the rest.  Epilogue fuses bias,
// residual add, forward activation, or the multiplication by the previous layer's activation
// derivative (backward).
#include "prof.hpp"
#include "pair.hpp"
#include "tapgemm.hpp"

namespace ctvae {

// Masked / scalar-staging variant (any channel count): used for the 3-channel-input side of the nets.
template <int WM, int WN, int TM, int TN, bool WT, bool AVEC, bool BVEC>
__global__ __launch_bounds__(256) void tapgemm_masked_kernel(const TapGemmArgs a) {
  kernarg_warm<sizeof(TapGemmArgs)>();
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int SA = BM * LDK, SB = WT ? BN * LDK : KC * BN;
  constexpr bool DB = false;
  constexpr int NBUF = 1;
  __shared__ __attribute__((aligned(16))) float sAbuf[NBUF * SA];
  __shared__ __attribute__((aligned(16))) float sBbuf[NBUF * SB];
  __shared__ int sRowPix[BM];
  __shared__ int sRowYX[BM];
  __shared__ int sOut[BM];
  __shared__ int sTab[AVEC ? 1 : 64 * 4];  // per flattened k: dy, dx, c, wtap

  const ConvGeom& g = a.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int cls = blockIdx.y;
  const int tile = blockIdx.x;
  const int mt = tile / a.ntiles, nt = tile - mt * a.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int ntaps = g.ntaps[cls];
  const int gC = g.gC;
  const int Ktot = ntaps * gC;

  // ---- per-tile row tables --------------------------------------------------------------
  for (int r = tid; r < BM; r += 256) {
    int m = m0 + r;
    if (m < a.Mc) {
      int b, qy, qx;
      if (a.lgQw >= 0) {          // power-of-two grids (every layer of the three models): shifts instead of divisions
        b = m >> a.lgQhw;
        const int rr = m & ((1 << a.lgQhw) - 1);
        qy = rr >> a.lgQw;
        qx = rr & ((1 << a.lgQw) - 1);
      } else {
        decode_m(g, m, b, qy, qx);
      }
      sRowPix[r] = (b * g.gH + qy * g.is) * g.gW + qx * g.is;  // may lie outside the image; range-checked per tap
      sRowYX[r] = ((qy * g.is) << 16) | (qx * g.is);
      sOut[r] = scatter_pix(g, cls, b, qy, qx);
    } else {
      sRowPix[r] = -1;
      sRowYX[r] = 0;
      sOut[r] = -1;
    }
  }
  if constexpr (!AVEC) {
    for (int k = tid; k < 64; k += 256) {
      int t = k / gC, c = k - t * gC;
      bool ok = k < Ktot;
      Tap tp = g.taps[cls][ok ? t : 0];
      sTab[k * 4 + 0] = tp.dy;
      sTab[k * 4 + 1] = tp.dx;
      sTab[k * 4 + 2] = c;
      sTab[k * 4 + 3] = ok ? tp.wtap : -1;
    }
  }
  __syncthreads();

  const int nch = AVEC ? (Ktot / KC) : ((Ktot + KC - 1) / KC);

  // ---- staging registers ------------------------------------------------------------------
  constexpr int A_V = BM / 32;      // float4 per thread (vector path)
  constexpr int A_S = BM / 8;       // floats per thread (scalar path)
  constexpr int B_V = BN / 32;      // float4 per thread
  constexpr int B_S = BN / 8;       // floats per thread
  f32x4 ra[AVEC ? A_V : 1];
  float rs[AVEC ? 1 : A_S];
  f32x4 rb[BVEC ? B_V : 1];
  float rbs[BVEC ? 1 : B_S];

  // vector path: per-row element offset and per-row validity masks are fixed for the tile, so a chunk's address
  // is one add and its bounds check two bit tests (all loads are issued unconditionally).  Tap offsets lie in
  // [-3, 4] (k <= 4): bit (d+3) of the y / x byte says whether row r may read at dy = d / dx = d.
  int a_off[AVEC ? A_V : 1];
  unsigned a_ok[AVEC ? A_V : 1];
  if constexpr (AVEC) {
#pragma unroll
    for (int j = 0; j < A_V; ++j) {
      const int r = (tid >> 3) + 32 * j;
      const int pix = sRowPix[r], yx = sRowYX[r];
      unsigned m = 0;
      if (pix >= 0) {
        const int iy0 = yx >> 16, ix0 = yx & 0xffff;
#pragma unroll
        for (int d = -3; d <= 4; ++d) {
          if ((unsigned)(iy0 + d) < (unsigned)g.gH) m |= 1u << (d + 3);
          if ((unsigned)(ix0 + d) < (unsigned)g.gW) m |= 1u << (d + 11);
        }
      }
      a_ok[j] = m;
      a_off[j] = pix * gC + 4 * (tid & 7);
    }
  }

  auto load_chunk = [&](int c) {
    // tap for this chunk (vector path: chunk lies inside one tap)
    int t = 0, ci0 = 0;
    Tap tp{0, 0, 0};
    if constexpr (AVEC) {
      int k0 = c * KC;
      t = k0 / gC;
      ci0 = k0 - t * gC;
      tp = g.taps[cls][t];
    }
    // ---- A ----
    if constexpr (AVEC) {
      const int tapoff = (tp.dy * g.gW + tp.dx) * gC + ci0;   // wave-uniform
#pragma unroll
      for (int j = 0; j < A_V; ++j) {
        const bool ok = ((a_ok[j] >> (tp.dy + 3)) & (a_ok[j] >> (tp.dx + 11)) & 1u) != 0;
        const unsigned off = ok ? (unsigned)(a_off[j] + tapoff) : 0u;
        f32x4 v = *reinterpret_cast<const f32x4*>(a.G + off);
        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
        ra[j] = v;
      }
    } else {
      const int kk = tid & 31;
      const int k = c * KC + kk;
      const int dy = sTab[(k & 63) * 4 + 0], dx = sTab[(k & 63) * 4 + 1], cc = sTab[(k & 63) * 4 + 2];
      const bool kok = (k < Ktot);
#pragma unroll
      for (int j = 0; j < A_S; ++j) {
        int r = (tid >> 5) + 8 * j;
        int pix = sRowPix[r], yx = sRowYX[r];
        int iy = (yx >> 16) + dy, ix = (yx & 0xffff) + dx;
        bool ok = kok && (pix >= 0) && ((unsigned)iy < (unsigned)g.gH) && ((unsigned)ix < (unsigned)g.gW);
        float v = 0.f;
        if (ok) v = a.G[(long)(pix + dy * g.gW + dx) * gC + cc];
        rs[j] = v;
      }
    }
    // ---- B ----
    if constexpr (!WT) {
      if constexpr (BVEC) {
#pragma unroll
        for (int j = 0; j < B_V; ++j) {
          int f = tid + 256 * j;
          int kr = f / (BN / 4), nq = f - kr * (BN / 4);
          int n = n0 + 4 * nq;
          bool ok = n < a.N;
          if constexpr (AVEC) {
            const unsigned wbase = (unsigned)(tp.wtap * g.wts + ci0 * g.wrs);   // wave-uniform
            const unsigned off = ok ? wbase + (unsigned)(kr * g.wrs + n) : 0u;
            f32x4 v = *reinterpret_cast<const f32x4*>(a.W + off);
            if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            rb[j] = v;
          } else {
            int k = c * KC + kr;
            int wt = sTab[(k & 63) * 4 + 3];
            ok = ok && (k < Ktot);
            long row = (long)wt * g.wts + (long)sTab[(k & 63) * 4 + 2] * g.wrs;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(a.W + row + n);
            rb[j] = v;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < B_S; ++j) {
          int e = tid + 256 * j;
          int kr = e / BN, nn = e - kr * BN;
          int n = n0 + nn;
          long row;
          bool ok = n < a.N;
          if constexpr (AVEC) {
            row = (long)tp.wtap * g.wts + (long)(ci0 + kr) * g.wrs;
          } else {
            int k = c * KC + kr;
            int wt = sTab[(k & 63) * 4 + 3];
            ok = ok && (k < Ktot);
            row = (long)wt * g.wts + (long)sTab[(k & 63) * 4 + 2] * g.wrs;
          }
          rbs[j] = ok ? a.W[row + n] : 0.f;
        }
      }
    } else {
      // Wmat[t][c][n] = W[t][n][c]: rows n, contiguous along c
      if constexpr (BVEC) {
        const int kq = tid & 7;
#pragma unroll
        for (int j = 0; j < B_V; ++j) {
          int nr = (tid >> 3) + 32 * j;
          int n = n0 + nr;
          const bool ok = n < a.N;
          const unsigned wbase = (unsigned)(tp.wtap * g.wts + ci0);      // wave-uniform
          const unsigned off = ok ? wbase + (unsigned)(n * g.wrs + 4 * kq) : 0u;
          f32x4 v = *reinterpret_cast<const f32x4*>(a.W + off);
          if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
          rb[j] = v;
        }
      } else {
        const int kk = tid & 31;
        const int k = c * KC + kk;
        const int cc = sTab[(k & 63) * 4 + 2], wt = sTab[(k & 63) * 4 + 3];
        const bool kok = (k < Ktot);
#pragma unroll
        for (int j = 0; j < B_S; ++j) {
          int nr = (tid >> 5) + 8 * j;
          int n = n0 + nr;
          rbs[j] = (kok && n < a.N) ? a.W[(long)wt * g.wts + (long)n * g.wrs + cc] : 0.f;
        }
      }
    }
  };

  auto store_chunk = [&](int buf) {
    float* sA = sAbuf + buf * SA;
    float* sB = sBbuf + buf * SB;
    if constexpr (AVEC) {
      const int kq = tid & 7;
#pragma unroll
      for (int j = 0; j < A_V; ++j) {
        int r = (tid >> 3) + 32 * j;
        *reinterpret_cast<f32x4*>(&sA[r * LDK + 4 * kq]) = ra[j];
      }
    } else {
      const int kk = tid & 31;
#pragma unroll
      for (int j = 0; j < A_S; ++j) sA[((tid >> 5) + 8 * j) * LDK + kk] = rs[j];
    }
    if constexpr (!WT) {
      if constexpr (BVEC) {
#pragma unroll
        for (int j = 0; j < B_V; ++j) {
          int f = tid + 256 * j;
          int kr = f / (BN / 4), nq = f - kr * (BN / 4);
          *reinterpret_cast<f32x4*>(&sB[kr * BN + 4 * nq]) = rb[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < B_S; ++j) sB[tid + 256 * j] = rbs[j];
      }
    } else {
      if constexpr (BVEC) {
        const int kq = tid & 7;
#pragma unroll
        for (int j = 0; j < B_V; ++j) *reinterpret_cast<f32x4*>(&sB[((tid >> 3) + 32 * j) * LDK + 4 * kq]) = rb[j];
      } else {
        const int kk = tid & 31;
#pragma unroll
        for (int j = 0; j < B_S; ++j) sB[((tid >> 5) + 8 * j) * LDK + kk] = rbs[j];
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

  const int li = lane & 31, lh = lane >> 5;

  // split-K: this workgroup owns chunks [c0, c1) of the class's K range
  int c0 = 0, c1 = nch;
  if (a.splitk > 1) {
    const int cps = (nch + a.splitk - 1) / a.splitk;
    c0 = blockIdx.z * cps;
    c1 = c0 + cps < nch ? c0 + cps : nch;
  }
  if (c0 < c1) {
    load_chunk(c0);
    store_chunk(0);
  }
  __syncthreads();
  for (int c = c0; c < c1; ++c) {
    const int cur = DB ? ((c - c0) & 1) : 0;
    const float* sA = sAbuf + cur * SA;
    const float* sB = sBbuf + cur * SB;
    if (c + 1 < c1) load_chunk(c + 1);
    // fragments of k-group kg+1 are read from LDS before the MFMAs of k-group kg are issued (register double
    // buffering), so the LDS latency hides under the 64-cycle MFMAs instead of stalling in front of them
    f32x4 af[2][TM];
    float bf[2][TN][4];
    auto read_frags = [&](int kg, int slot) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[slot][i] = *reinterpret_cast<const f32x4*>(&sA[((wm * TM + i) * 32 + li) * LDK + kg * 8 + 4 * lh]);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (WT) {
          f32x4 t4 = *reinterpret_cast<const f32x4*>(&sB[((wn * TN + j) * 32 + li) * LDK + kg * 8 + 4 * lh]);
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
      {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg & 1][i][s], bf[kg & 1][j][s], acc[i][j], 0, 0, 0);
      }
    }
    if constexpr (DB) {
      // the other buffer was last read in iteration c-1, which every wave left through the barrier below
      if (c + 1 < c1) store_chunk(cur ^ 1);
      __syncthreads();
    } else {
      __syncthreads();
      if (c + 1 < c1) {
        store_chunk(0);
        __syncthreads();
      }
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------
  const int N = a.N;
  if (a.splitk > 1) {  // raw partial sums; bias / activation / BN statistics happen in splitk_finish_kernel
    float* dst = a.part + (long)blockIdx.z * ((long)g.B * g.sH * g.sW) * N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
          const int sp = sOut[row];
          if (sp >= 0 && col < N) dst[(long)sp * N + col] = acc[i][j][r];
        }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + li;
    const bool cok = col < N;
    const float bv = (a.bias != nullptr && cok) ? a.bias[col] : 0.f;
    const bool plain = (a.add == nullptr) && (a.mask == nullptr);
    // launch-uniform activations decided once (common.hpp act_slope_fwd): identity / LeakyReLU / ReLU are one select per value
    const bool o_tanh = a.act == ACT_TANH, m_tanh = a.mask_act == ACT_TANH, b_tanh = a.bnb_act == ACT_TANH;
    const float o_slope = act_slope(a.act), m_slope = act_slope(a.mask_act), b_slope = act_slope(a.bnb_act);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        // rows 8*q4 + 4*lh + {0,1,2,3} of this 32-row tile: their scatter indices are 4 consecutive ints in LDS
        const int row0 = (wm * TM + i) * 32 + 8 * q4 + 4 * lh;
        const int sp4[4] = {sOut[row0], sOut[row0 + 1], sOut[row0 + 2], sOut[row0 + 3]};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int sp = sp4[q];
          if (sp >= 0 && cok) {
            const unsigned idx = (unsigned)sp * (unsigned)N + (unsigned)col;
            float v = acc[i][j][4 * q4 + q] + bv;
            if (plain) {
              v = o_tanh ? act_fwd(v, ACT_TANH) : act_slope_fwd(v, o_slope);
            } else {
              if (a.add != nullptr) v += a.add[idx];
              v = o_tanh ? act_fwd(v, ACT_TANH) : act_slope_fwd(v, o_slope);
              if (a.mask != nullptr) {
                const float mk = a.mask[idx];
                v *= m_tanh ? 1.f - mk * mk : (mk > 0.f ? 1.f : m_slope);
              }
            }
            a.S[idx] = v;
            if (a.bnb_part != nullptr) acc[i][j][4 * q4 + q] = v;   // the stored gradient, for the BN-backward sums
          }
        }
      }
    }
    // ---- fused BatchNorm-backward sums (see TapGemmArgs::bnb_*) ----
    if (a.bnb_part != nullptr) {
      const float bmean = cok ? a.bnb_mean[col] : 0.f, binv = cok ? a.bnb_invstd[col] : 0.f;
      const float bgm = cok ? a.bnb_gamma[col] : 0.f, bbt = cok ? a.bnb_beta[col] : 0.f;
      float s1b = 0.f, s2b = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float yv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int sp = sOut[(wm * TM + i) * 32 + 8 * (r >> 2) + 4 * lh + (r & 3)];
          yv[r] = (sp >= 0 && cok) ? a.bnb_y[(unsigned)sp * (unsigned)N + (unsigned)col] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int sp = sOut[(wm * TM + i) * 32 + 8 * (r >> 2) + 4 * lh + (r & 3)];
          const float xh = (yv[r] - bmean) * binv;
          const float tb = bgm * xh + bbt;
          const float gp = (sp >= 0 && cok) ? acc[i][j][r] * (b_tanh ? act_bwd_from_out(act_fwd(tb, ACT_TANH), ACT_TANH) : act_slope_bwd(tb, b_slope)) : 0.f;
          s1b += gp;
          s2b += gp * xh;
        }
      }
      s1b += __shfl_xor(s1b, 32, 64);
      s2b += __shfl_xor(s2b, 32, 64);
      if (lh == 0) {
        float* st = &sAbuf[(wm * BN + (wn * TN + j) * 32 + li) * 2];
        st[0] = s1b; st[1] = s2b;
      }
    }
    // ---- fused BatchNorm statistics of this tile (train-mode BN follows the conv: vanilla_vae.py:28-31) ----
    if (a.bn_part != nullptr) {
      float cnt = 0.f, s1 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
          if (sOut[row] >= 0) { cnt += 1.f; s1 += acc[i][j][r] + bv; }
        }
      float mean = cnt > 0.f ? s1 / cnt : 0.f, m2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
          if (sOut[row] >= 0) { float d = (acc[i][j][r] + bv) - mean; m2 += d * d; }
        }
      // merge the two lane halves (Chan et al.)
      const float ocnt = __shfl_xor(cnt, 32, 64), omean = __shfl_xor(mean, 32, 64), om2 = __shfl_xor(m2, 32, 64);
      const float nt = cnt + ocnt;
      if (nt > 0.f) {
        const float d = omean - mean;
        m2 = m2 + om2 + d * d * (cnt * ocnt / nt);
        mean = mean + d * (ocnt / nt);
      }
      if (lh == 0) {  // sA is free after the main loop (last iteration ended with a barrier)
        float* st = &sAbuf[(wm * BN + (wn * TN + j) * 32 + li) * 3];
        st[0] = nt; st[1] = mean; st[2] = m2;
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
          const float nt = n + nb, d = mb - mean;
          mean += d * (nb / nt);
          m2 += qb + d * d * (n * nb / nt);
          n = nt;
        }
      }
      float* p = a.bn_part + ((long)(cls * a.mtiles + mt) * N + n0 + tid) * 3;
      p[0] = n; p[1] = mean; p[2] = m2;
    }
  }
  if (a.bnb_part != nullptr) {
    __syncthreads();
    if (tid < BN && n0 + tid < N) {
      float s1b = 0.f, s2b = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        s1b += sAbuf[(w * BN + tid) * 2];
        s2b += sAbuf[(w * BN + tid) * 2 + 1];
      }
      float* p = a.bnb_part + ((long)(cls * a.mtiles + mt) * N + n0 + tid) * 2;
      p[0] = s1b; p[1] = s2b;
    }
  }
}

// split-K finish: body in finish.hpp (shared with the paired backward's combined finishing launch in wgrad.hip)
__global__ __launch_bounds__(256) void splitk_finish_kernel(const SplitKJob k) { splitk_finish_body(k, blockIdx.x); }

// ---- host side --------------------------------------------------------------------------------
// masked variants: one tile shape (128 x 32), single-buffered
static int launch_masked(const TapGemmArgs& a, bool wt, bool avec, bool bvec, hipStream_t st) {
  constexpr int BM = 128, BN = 32;
  TapGemmArgs args = a;
  args.mtiles = ceil_div(a.Mc, BM);
  args.ntiles = ceil_div(a.N, BN);
  dim3 grid(args.mtiles * args.ntiles, a.g.ncls, 1), block(256);
  char name[160];
  snprintf(name, sizeof name, "tapgemm_masked_kernel<%s,%s,%s>", wt ? "true" : "false", avec ? "true" : "false",
           bvec ? "true" : "false");
  double macs = 0;
  for (int c = 0; c < a.g.ncls; ++c) macs += (double)a.Mc * a.N * a.g.ntaps[c] * a.g.gC;
  if (prof_detailed()) {
    size_t l = strlen(name);
    snprintf(name + l, sizeof name - l, " M=%dx%d N=%d C=%d taps=%d", a.g.ncls, a.Mc, a.N, a.g.gC, a.g.ntaps[a.g.ncls - 1]);
  }
  const double bytes = 4.0 * ((double)a.g.B * a.g.gH * a.g.gW * a.g.gC + (double)a.g.B * a.g.sH * a.g.sW * a.g.sC);
  ProfScope ps(name, st, 2.0 * macs, bytes);
#define CTVAE_TG(WT_, AV_, BV_) hipLaunchKernelGGL((tapgemm_masked_kernel<4, 1, 1, 1, WT_, AV_, BV_>), grid, block, 0, st, args)
  if (!wt) {
    if (avec && bvec) CTVAE_TG(false, true, true);
    else if (!avec && bvec) CTVAE_TG(false, false, true);
    else if (avec && !bvec) CTVAE_TG(false, true, false);
    else CTVAE_TG(false, false, false);
  } else {
    if (avec) CTVAE_TG(true, true, true);
    else CTVAE_TG(true, false, false);
  }
#undef CTVAE_TG
  CTVAE_LAUNCH_CHECK();
  return 0;
}

bool upconv_supported(const ConvGeom& g);
int upconv_rows(const ConvGeom& g);
int launch_upconv_forward(const ConvGeom& g, const float* X, const float* W, const float* bias, float* S, int act,
                          float* bn_part, hipStream_t st, const InXform* xf);
bool img_enc_supported(const ConvGeom& g);
int img_enc_rows(const ConvGeom& g);
int launch_img_enc_forward(const ConvGeom& g, const float* X, const float* W, const float* bias, float* S, int act,
                           float* bn_part, hipStream_t st);
bool thin_forward_supported(const ConvGeom& g);
int thin_bn_parts(const ConvGeom& g);
int launch_thin_forward(const ConvGeom& g, const float* G, const float* W, const float* bias, const float* add,
                        const float* mask, int mask_act, float* S, int act, float* bn_part, hipStream_t st,
                        const InXform* xf);

// tile shape and split-K factor a launch will use (mirrored by the BN-statistics consumers)
void tapgemm_plan(const ConvGeom& g, size_t ws_floats, TapGemmPlan& p) {
  const int Mc = g.B * g.Qh * g.Qw, N = g.sC;
  p.thin = 0;
  if (img_enc_supported(g)) {   // picture-side stride-2 conv: dedicated kernel, one statistics row per workgroup
    p.BM = p.BN = 0; p.mtiles = p.ntiles = 0; p.splitk = 1;
    p.bn_parts = img_enc_rows(g);
    return;
  }
  if (upconv_supported(g)) {    // 32 -> 32 transposed conv on the big decoder activations: persistent LDS-tile kernel
    p.BM = p.BN = 0; p.mtiles = p.ntiles = 0; p.splitk = 1;
    p.bn_parts = upconv_rows(g);
    return;
  }
  if (thin_forward_supported(g)) {
    p.thin = 1; p.BM = p.BN = 0; p.mtiles = p.ntiles = 0; p.splitk = 1;
    p.bn_parts = thin_bn_parts(g);
    return;
  }
  const bool avec = (g.gC % KC) == 0;
  const bool bvec = g.wT ? avec : ((N % 4) == 0);
  p.BM = 128; p.BN = 32; p.splitk = 1;
  if (!(N <= 32 || !avec || !bvec)) {
    const long tiles128 = (long)ceil_div(Mc, 128) * ceil_div(N, 64) * g.ncls;
    // 128-row tiles only for long reductions: with <= 72 K-chunks per workgroup the launch is dominated by its prologue /
    // epilogue phases and twice as many 64-row workgroups hide them better (measured, MI355X: VanillaVAE bs=256 step
    // 1.913 -> 1.879 ms, MCQVAE 8.15 -> 7.93 ms; CTVAE_SHORT_K overrides the threshold for experiments)
    static const int short_k = [] { const char* e = getenv("CTVAE_SHORT_K"); return e ? atoi(e) : 72; }();
    int kch = 1 << 30;
    for (int c = 0; c < g.ncls; ++c) kch = g.ntaps[c] * g.gC / KC < kch ? g.ntaps[c] * g.gC / KC : kch;
    p.BM = (tiles128 >= 512 && kch > short_k) ? 128 : 64;
    static const int force_bm = [] { const char* e = getenv("CTVAE_FORCE_BM"); return e ? atoi(e) : 0; }();   // diagnostic
    if (force_bm == 64 || force_bm == 128) p.BM = force_bm;
    p.BN = 64;
  }
  p.mtiles = ceil_div(Mc, p.BM);
  p.ntiles = ceil_div(N, p.BN);
  const long wgs = (long)p.mtiles * p.ntiles * g.ncls;
  int nch_min = 1 << 30;
  for (int c = 0; c < g.ncls; ++c) {
    const int nch = avec ? g.ntaps[c] * g.gC / KC : ceil_div(g.ntaps[c] * g.gC, KC);
    if (nch < nch_min) nch_min = nch;
  }
  static const int sk_target = [] { const char* e = getenv("CTVAE_SK_TARGET"); return e ? atoi(e) : 768; }();   // diagnostic
  // a launch that already has a workgroup for every CU is not split (round 3: 384 -> 256; the bs = 64 step's encoder.1 -- 256
  // tiles of 9 chunks -- keeps its BatchNorm statistics in the epilogue instead of three launches behind two slices: -1.4 %
  // there, +-0.3 % on the other configurations)
  static const int sk_maxwgs = [] { const char* e = getenv("CTVAE_SK_MAXWGS"); return e ? atoi(e) : 256; }();
  // a data gradient that shares its launch with the weight gradient (ctvae_conv_backward) does not have to fill the chip
  // on its own: the ~1000 weight-gradient workgroups do.  It is split only as far as its workgroups would otherwise be the
  // launch's long pole -- fewer partial sums to write and finish, and an unsplit launch keeps the fused BatchNorm-backward
  // sums in its epilogue (no bn_bwd_partial pass)
  static const int pair_target = [] { const char* e = getenv("CTVAE_PAIR_SK_TARGET"); return e ? atoi(e) : 256; }();   // sweep, ms per step: 768 1.804, 512 1.798, 384 1.800, 256 1.796, 128 1.813, no split 1.847
  static const int pair_maxwgs = [] { const char* e = getenv("CTVAE_PAIR_SK_MAXWGS"); return e ? atoi(e) : 384; }();
  const bool paired = pair_ctx() != nullptr && g.wT != 0;
  // small problems (few scattered pixels in all): every slice is another copy of the output to write, flush at the kernel
  // boundary and sum again -- a lower target (fewer, longer slices) wins there
  static const int sk_small_m = [] { const char* e = getenv("CTVAE_SK_SMALL_M"); return e ? atoi(e) : 0; }();
  static const int sk_small_target = [] { const char* e = getenv("CTVAE_SK_SMALL_TARGET"); return e ? atoi(e) : 256; }();
  const int tgt = paired ? pair_target : ((long)Mc * g.ncls <= sk_small_m ? sk_small_target : sk_target), maxw = paired ? pair_maxwgs : sk_maxwgs;
  // paired with a reduction of fewer than 16 chunks (K < 512: the Linear heads' data gradient): splitting saves ~1 us of a
  // launch the weight gradient fills anyway and would cost the fused BatchNorm-backward sums of the layer below
  static const int pair_minch = [] { const char* e = getenv("CTVAE_PAIR_SK_MINCH"); return e ? atoi(e) : 8; }();   // round 3: 16 -> 8 (the slices go to the BatchNorm's channel-owner launch: bs = 64 -0.9 %, bs = 256 -0.3 %)
  if (avec && bvec && (N % 4) == 0 && wgs < maxw && nch_min >= (paired ? pair_minch : 8)) {
    int sk = (int)((tgt + wgs - 1) / wgs);
    if (sk > nch_min / 4) sk = nch_min / 4;
    static const int sk_max = [] { const char* e = getenv("CTVAE_SK_MAX"); return e ? atoi(e) : 16; }();   // diagnostic
    if (sk > sk_max) sk = sk_max;
    const size_t per = (size_t)g.B * g.sH * g.sW * N;
    while (sk > 1 && per * sk > ws_floats) --sk;
    if (sk > 1) p.splitk = sk;
  }
  p.bn_parts = p.mtiles * g.ncls;
}

bool img_dgrad_supported(const ConvGeom& g);
int img_dgrad_rows(const ConvGeom& g);
int launch_img_dgrad(const ConvGeom& g, const float* dY, const float* W, float* dX, const BnBwdFuse* bnb, hipStream_t st);

// rows of the fused BN-backward partial table a dgrad launch of this geometry writes (0: this configuration cannot
// fuse -- split-K keeps its epilogue in splitk_finish_kernel, the thin path has none)
int tapgemm_bnb_rows(const ConvGeom& g, size_t ws_floats) {
  if (img_dgrad_supported(g)) return img_dgrad_rows(g);
  TapGemmPlan plan;
  tapgemm_plan(g, ws_floats, plan);
  if (plan.thin || plan.splitk > 1) return 0;
  return plan.bn_parts;
}

bool wino_supported(const ConvGeom& g, size_t ws_floats);
int launch_wino_conv(const ConvGeom& g, const float* X, const float* Wp, const float* bias, float* Y, int act, float* ws,
                     size_t ws_floats, hipStream_t st, const float* filters_ready, float* bwd_out, const float* add);

bool wino_enabled();

// a 1 x 1 layer on a 1 x 1 image (an nn.Linear) that the vector tile kernel runs unsplit: the only launch that honours out_pix
bool out_pix_supported(const ConvGeom& g, size_t ws_floats, int out_pix) {
  if (out_pix <= 1 || g.os != 1 || g.ncls != 1 || g.ntaps[0] != 1 || g.sH * g.sW != 1 || g.sC % out_pix != 0 || g.wT != 0) return false;
  if ((g.gC % KC) != 0 || (g.sC % 4) != 0) return false;
  TapGemmPlan plan;
  tapgemm_plan(g, ws_floats, plan);
  return !plan.thin && plan.BM != 0 && plan.splitk <= 1;
}

int launch_tapgemm(const ConvGeom& g, const float* G, const float* W, const float* bias, const float* add,
                   const float* mask, int mask_act, float* S, int act, float* bn_part, float* ws, size_t ws_floats,
                   hipStream_t st, const BnBwdFuse* bnb, const InXform* xf, const WinoFilters* wf, SplitKRaw* raw, int out_pix) {
  if (raw != nullptr) raw->splitk = 0;
  if (out_pix > 1) {   // TapGemmArgs::out_pix: the vector tile kernel's plain epilogue only (see out_pix_supported)
    if (!out_pix_supported(g, ws != nullptr ? ws_floats : 0, out_pix) || add != nullptr || mask != nullptr || bn_part != nullptr ||
        (bnb != nullptr && bnb->part != nullptr) || raw != nullptr)
      return kErrBadArg;
  }
  // 3x3 / stride 1 / same-padding layers with plain epilogues: Winograd F(2x2,3x3), see wino.hip
  // (the skip operand `add` is taken by the Winograd epilogue; a mask is not)
  if (mask == nullptr && bn_part == nullptr && (bnb == nullptr || bnb->part == nullptr) &&
      (xf == nullptr || xf->scale == nullptr) && ws != nullptr && wino_enabled() && wino_supported(g, ws_floats))
    return launch_wino_conv(g, G, W, bias, S, act, ws, ws_floats, st, wf != nullptr ? wf->ready : nullptr,
                            wf != nullptr ? wf->bwd_out : nullptr, add);
  // data gradient of the 3x3 image-side conv (3 gathered channels): dedicated kernel, see image.hip
  if (img_dgrad_supported(g) && bias == nullptr && add == nullptr && mask == nullptr && act == ACT_NONE && bn_part == nullptr &&
      (xf == nullptr || xf->scale == nullptr))
    return launch_img_dgrad(g, G, W, S, bnb, st);
  if (img_enc_supported(g) && act != ACT_TANH && add == nullptr && mask == nullptr && (bnb == nullptr || bnb->part == nullptr) &&
      (xf == nullptr || xf->scale == nullptr))
    return launch_img_enc_forward(g, G, W, bias, S, act, bn_part, st);
  if (upconv_supported(g) && add == nullptr && mask == nullptr && (bnb == nullptr || bnb->part == nullptr))
    return launch_upconv_forward(g, G, W, bias, S, act, bn_part, st, xf);
  TapGemmArgs a{};
  a.bn_part = bn_part;
  if (bnb != nullptr && bnb->part != nullptr) {
    a.bnb_y = bnb->y; a.bnb_mean = bnb->mean; a.bnb_invstd = bnb->invstd; a.bnb_gamma = bnb->gamma; a.bnb_beta = bnb->beta;
    a.bnb_act = bnb->act; a.bnb_part = bnb->part;
  }
  a.g = g;
  a.G = G; a.W = W; a.bias = bias; a.add = add; a.mask = mask; a.S = S;
  a.act = act; a.mask_act = mask_act;
  a.out_pix = out_pix;
  a.Mc = g.B * g.Qh * g.Qw;
  a.N = g.sC;
  if (a.Mc <= 0 || a.N <= 0) return kErrBadArg;
  // buffer resources take 31-bit byte counts: every tensor bound here must stay below 2 GiB
  // (the fast kernel binds the gathered tensor with a bias of (3*gW+3)*gC elements in front of it: keep that inside 2 GiB too)
  if ((long)g.B * g.sH * g.sW * g.sC >= (1L << 29) ||
      (long)g.B * g.gH * g.gW * g.gC + (long)(3 * g.gW + 3) * g.gC >= (1L << 29))
    return kErrBadArg;
  for (int c = 0; c < g.ncls; ++c)
    for (int t = 0; t < g.ntaps[c]; ++t) {
      const Tap& tp = g.taps[c][t];
      if (tp.dy < -3 || tp.dy > 4 || tp.dx < -3 || tp.dx > 4) return kErrBadArg;
    }
  {
    auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    const int lw = lg2(g.Qw), lh2 = lg2(g.Qh);
    a.lgQw = (lw >= 0 && lh2 >= 0) ? lw : -1;
    a.lgQhw = (lw >= 0 && lh2 >= 0) ? lw + lh2 : -1;
  }
  const bool wt = g.wT != 0;
  const bool avec = (g.gC % KC) == 0;
  const bool bvec = wt ? avec : ((a.N % 4) == 0);
  if (!avec && !thin_forward_supported(g)) {
    for (int c = 0; c < g.ncls; ++c)
      if (g.ntaps[c] * g.gC > 64) return kErrBadArg;  // scalar path keeps its k-table in LDS
  }
  // tile choice: N<=32 -> 128x32 (4x1 waves); big problems -> 128x64; small M -> 64x64 (+ split-K when the grid is small)
  TapGemmPlan plan;
  tapgemm_plan(g, ws != nullptr ? ws_floats : 0, plan);
  if (a.bnb_part != nullptr && (plan.thin || plan.splitk > 1)) return kErrBadArg;   // see tapgemm_bnb_rows()
  // transform on load: the thin kernels and the vector tile kernel in forward orientation (tapgemm_fast_body.inc XF)
  if (xf != nullptr && xf->scale != nullptr && !plan.thin && !(avec && bvec && !wt)) return kErrBadArg;
  if (xf != nullptr && xf->scale != nullptr && !plan.thin) { a.xf_scale = xf->scale; a.xf_shift = xf->shift; a.xf_act = xf->act; }
  if (plan.thin) return launch_thin_forward(g, G, W, bias, add, mask, mask_act, S, act, bn_part, st, xf);
  a.splitk = plan.splitk;
  a.part = ws;
  // the consumer sums the slices itself: channel-major raw partials in its buffer, no finishing launch
  // (bias / activation are then the consumer's job as well: they are NOT applied to the raw slices)
  const bool raw_T = raw != nullptr && raw->part != nullptr && plan.splitk > 1 && avec && bvec && (raw->pixel_major || a.Mc % 4 == 0) &&
                     add == nullptr && mask == nullptr;
  if (raw_T) {
    a.part = raw->part;
    a.part_T = raw->pixel_major ? 0 : 1;
  }

  if (plan.splitk > 1 && bn_part != nullptr) return kErrBadArg;  // caller must take BN statistics from S instead
  int rc;
  if (!avec || !bvec) {
    rc = launch_masked(a, wt, avec, bvec, st);
  } else {
    // few workgroups per CU -> latency must be hidden inside the workgroup (double-buffered LDS); many -> by occupancy
    static const long pf_wgs = [] { const char* e = getenv("CTVAE_PF_WGS"); return e ? atol(e) : 1024L; }();   // diagnostic
    static const int force_pf = [] { const char* e = getenv("CTVAE_FORCE_PF"); return e ? atoi(e) : -1; }();   // diagnostic
    bool db = (long)plan.mtiles * plan.ntiles * g.ncls * plan.splitk <= pf_wgs;
    if (force_pf == 0) db = false;
    if (force_pf == 3) db = true;
    // measured (bench.py, VanillaVAE bs=256): pipelined double-buffer loop 2.35 ms vs 2.37 (plain double buffer) vs 2.42
    // when the large grids use it too (LDS doubling costs them occupancy)
    rc = launch_tapgemm_fast(a, plan, db ? 3 : 0, st);
  }
  if (rc || plan.splitk <= 1) return rc;
  if (raw_T) {
    raw->splitk = plan.splitk;
    return 0;
  }
  const long n = (long)g.B * g.sH * g.sW * a.N, n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const SplitKJob job{ws, plan.splitk, n, bias, add, mask, mask_act, act, S, n4, a.N, (int)blocks};
  if (PairCtx* pc = pair_ctx()) {   // ctvae_conv_backward: behind the paired main launch (pair_flush)
    pc->haveSK = true;
    pc->sk = job;
    pc->bytesSK = 4.0 * (double)(plan.splitk + 1) * n;
    return 0;
  }
  ProfScope ps("splitk_finish_kernel", st, 0.0, 4.0 * (double)(plan.splitk + 1) * n);
  hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, st, job);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_splitk_recorded(const PairCtx& c, hipStream_t st) {
  ProfScope ps("splitk_finish_kernel", st, 0.0, c.bytesSK);
  hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)c.sk.nblk), dim3(256), 0, st, c.sk);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
