// ConvTranspose2d(32 -> 32, k=3, stride 2, pad 1, output_padding 1) on the largest activations of the decoder
// (vanilla_vae.py:65-70, final_layer.0: [B,32,32,32] -> [B,64,64,32]), forward (+ BatchNorm statistics) and weight
// gradient.
//
// As a tap-GEMM (tapgemm_fast.hip) this layer is 8192 short workgroups: each output-parity class has only 1..4 taps
// (K = 32..128), so a workgroup spends its life in prologue / first-load latency / epilogue.  Here one persistent
// workgroup owns an 8 x 32 tile of INPUT pixels at a time:
//   * the input tile (+1 halo row/column) is staged once into LDS (each input element leaves L2 once instead of
//     2.25 times) and all nine 32x32 tap matrices stay in LDS for the whole launch;
//   * forward: for each of the four output-parity classes a wave accumulates its taps for one 32-pixel row in a
//     32x32 MFMA tile, adds the bias, merges the (count, mean, M2) statistics of BatchNorm and stores 128 B per pixel;
//   * wgrad: nine 32x32 accumulators (one per tap) live in AGPRs for the whole launch; dy streams from global memory
//     straight into the MFMA B operand (128 B per pixel) and is reused by every tap of its class, x comes from LDS.
// Fixed-order merges everywhere: bit-reproducible.
#include "common.hpp"
#include "phase.hpp"
#include <type_traits>

#include "prof.hpp"

namespace ctvae {

namespace {

constexpr int TH = 8, TW = 32, PH = 9, PW = 33, NP = PH * PW /*297*/;
constexpr int LDA = 36, C = 32, NT = 9, NCLS = 4, MAXT = 4;
constexpr int NLD = (NP * 8 + 255) / 256;   // 10 float4 per thread
constexpr unsigned kOOBu = 0x80000000u;

struct UpArgs {
  const float* X;     // [B,H,W,32]
  const float* Wt;    // [9][32 ci][32 co]
  const float* bias;
  const float* dY;    // wgrad: [B,2H,2W,32]
  float* out;         // forward: y [B,2H,2W,32]; wgrad: slabs [nwg][9][32][32]
  float* pbias;       // wgrad: [nwg][32]
  float* bn_part;     // forward: [nwg][32][3], may be null
  const float* dy_y;  // wgrad: dY is g_a; g_y = k1*g_a*act'(y*scale+shift) + k2*y + k3 is formed on load and written to gy_out
  const float* dy_coef;
  float* gy_out;
  int dy_act;
  // lazy BatchNorm apply (InXform): X is the raw BatchNorm input of the previous block; the patch is staged as
  // max(t, t*slope), t = X*scale[c] + shift[c] (LeakyReLU / ReLU / none), pixels beyond the image stay 0
  const float* xf_scale;
  const float* xf_shift;
  int xf_act;
  int act;
  int B, H, W, tiles_y, tiles_x, ntiles;
  int ntaps[NCLS], cpy[NCLS], cpx[NCLS];
  int tdy[NCLS][MAXT], tdx[NCLS][MAXT], twt[NCLS][MAXT];
};

__device__ __forceinline__ f32x4 ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
__device__ __forceinline__ float ld1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ void st1(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

struct TileXY {
  int b, y0, x0;
};
__device__ __forceinline__ TileXY tile_xy(const UpArgs& a, int tile) {
  const int per = a.tiles_y * a.tiles_x;
  const int b = tile / per, r = tile - b * per;
  const int ty = r / a.tiles_x;
  return TileXY{b, ty * TH, (r - ty * a.tiles_x) * TW};
}

struct Patch {
  f32x4 v[NLD];
};
// patch pixel (py,px) <-> input pixel (y0 + py, x0 + px); rows/columns beyond the image read as 0
__device__ __forceinline__ void patch_load(const UpArgs& a, __amdgpu_buffer_rsrc_t rX, const TileXY& t, Patch& p) {
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int e = threadIdx.x + 256 * j;
    const int pp = e >> 3, c4 = e & 7;
    const int py = (pp * 1986) >> 16, px = pp - py * PW;   // pp / 33 for pp < 330
    const int iy = t.y0 + py, ix = t.x0 + px;
    const bool ok = e < NP * 8 && iy < a.H && ix < a.W;
    p.v[j] = ld4(rX, ok ? (unsigned)(((t.b * a.H + iy) * a.W + ix) * C + 4 * c4) * 4u : kOOBu);
  }
}
__device__ __forceinline__ void patch_store(const Patch& p, float* sA) {
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int e = threadIdx.x + 256 * j;
    if (e < NP * 8) *reinterpret_cast<f32x4*>(&sA[(e >> 3) * LDA + 4 * (e & 7)]) = p.v[j];
  }
}
// the same with the lazy BatchNorm apply: a thread's channel quad (e & 7 = threadIdx.x & 7) is fixed, so are its coefficients
__device__ __forceinline__ void patch_store_xf(const UpArgs& a, const TileXY& t, const Patch& p, float* sA, f32x4 sc, f32x4 sh,
                                               float nslope) {
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int e = threadIdx.x + 256 * j;
    const int pp = e >> 3;
    const int py = (pp * 1986) >> 16, px = pp - py * PW;
    const float inside = (t.y0 + py < a.H && t.x0 + px < a.W) ? 1.f : 0.f;
    f32x4 v = p.v[j];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float u = v[q] * sc[q] + sh[q] * inside;
      v[q] = fmaxf(u, u * nslope);
    }
    if (e < NP * 8) *reinterpret_cast<f32x4*>(&sA[pp * LDA + 4 * (e & 7)]) = v;
  }
}

CTVAE_PHASE_DECL(up)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void up_fwd_kernel(const UpArgs a) {
  CTVAE_PH(up, 0, 0);
  kernarg_warm<sizeof(UpArgs)>();
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;               // [297][36]
  float* sW = smem + NP * LDA;    // [9][32 ci][32 co]
  __shared__ float sS[4 * C * 3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rX = rsrc(a.X, (long)a.B * a.H * a.W * C * 4);
  const int OH = 2 * a.H, OW = 2 * a.W;
  const __amdgpu_buffer_rsrc_t rO = rsrc(a.out, (long)a.B * OH * OW * C * 4);

  // Every global load of the prologue is issued before the first one is used: the nine 16-byte weight loads per thread, the first
  // patch, bias and BatchNorm coefficients.  As `for (e = tid; e < N; e += 256) sW[e] = W[e]` the weights were nine load -> wait ->
  // LDS-store round trips in a row (the trip count depends on tid, so the loop is not unrolled), then three more for the scalars --
  // a third of this kernel's time at one tile per workgroup.
  constexpr int NW4 = NT * C * C / 4 / 256;
  static_assert(NW4 * 256 * 4 == NT * C * C, "weights: a whole number of 16-byte loads per thread");
  f32x4 wv[NW4];
#pragma unroll
  for (int j = 0; j < NW4; ++j) wv[j] = reinterpret_cast<const f32x4*>(a.Wt)[tid + 256 * j];
  Patch pt;
  int tile = blockIdx.x;
  TileXY cur = tile_xy(a, tile < a.ntiles ? tile : 0);
  if (tile < a.ntiles) patch_load(a, rX, cur, pt);
  const float bv = a.bias != nullptr ? a.bias[li] : 0.f;
  const float out_slope = a.act == ACT_LRELU ? kLeaky : (a.act == ACT_RELU ? 0.f : 1.f);
  float sn = 0.f, smean = 0.f, sm2 = 0.f;

  const bool xf = a.xf_scale != nullptr;
  const f32x4 xsc = xf ? *reinterpret_cast<const f32x4*>(a.xf_scale + 4 * (tid & 7)) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 xsh = xf ? *reinterpret_cast<const f32x4*>(a.xf_shift + 4 * (tid & 7)) : f32x4{0.f, 0.f, 0.f, 0.f};
  const float xns = a.xf_act == ACT_LRELU ? kLeaky : (a.xf_act == ACT_RELU ? 0.f : 1.f);
#pragma unroll
  for (int j = 0; j < NW4; ++j) reinterpret_cast<f32x4*>(sW)[tid + 256 * j] = wv[j];
  for (; tile < a.ntiles; tile += gridDim.x) {
    CTVAE_PH(up, 0, 1);
    if (xf) patch_store_xf(a, cur, pt, sA, xsc, xsh, xns);
    else patch_store(pt, sA);
    __syncthreads();
    CTVAE_PH(up, 0, 2);
    const int next = tile + gridDim.x;
    const TileXY nxt = tile_xy(a, next < a.ntiles ? next : 0);
    if (next < a.ntiles) patch_load(a, rX, nxt, pt);
    // Nine tap-GEMMs per 32-pixel row in the fixed class order {1,2,2,4} taps (host-checked).  The LDS fragments of tap
    // k+1 are read while the 16 MFMAs of tap k run (two register sets, sched_barrier keeps the order).
    struct Frag {
      f32x4 af[4];
      float bf[4][4];
    };
    auto read_tap = [&](int ly, int c, int t, Frag& f) {
      const float* ap = &sA[((ly + a.tdy[c][t]) * PW + li + a.tdx[c][t]) * LDA + 4 * lh];
      const float* wp = &sW[(a.twt[c][t] * C + 4 * lh) * C + li];
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        f.af[kg] = *reinterpret_cast<const f32x4*>(ap + kg * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) f.bf[kg][q] = wp[(kg * 8 + q) * C];
      }
    };
    auto mfma_tap = [&](const Frag& f, f32x16& acc) {
#pragma unroll
      for (int kg = 0; kg < 4; ++kg)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.af[kg][q], f.bf[kg][q], acc, 0, 0, 0);
    };
    auto epilogue = [&](int ly, int c, f32x16& acc) {
      // bias, statistics, 128 B per pixel out
      const unsigned rowoff = (unsigned)((((cur.b * OH + 2 * (cur.y0 + ly) + a.cpy[c]) * OW + 2 * cur.x0 + a.cpx[c]) * C + li)) * 4u;
      float m1 = 0.f;
      // identity / LeakyReLU / ReLU as max(t, t * slope), decided once per launch: `act_fwd(v, a.act)` per element was a branch tree
      // (with the tanh code behind it) sixteen times per epilogue, 128 per tile -- a fifth of this kernel's cycles
      if (a.act == ACT_TANH) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[r] += bv;
          m1 += acc[r];
          st1(rO, rowoff + (unsigned)((8 * (r >> 2) + 4 * lh + (r & 3)) * 2 * C) * 4u, act_fwd(acc[r], ACT_TANH));
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[r] += bv;
          m1 += acc[r];
          st1(rO, rowoff + (unsigned)((8 * (r >> 2) + 4 * lh + (r & 3)) * 2 * C) * 4u, fmaxf(acc[r], acc[r] * out_slope));
        }
      }
      if (a.bn_part != nullptr) {
        m1 *= (1.f / 16.f);
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) q += (acc[r] - m1) * (acc[r] - m1);
        const float ntot = sn + 16.f, d = m1 - smean;
        smean += d * (16.f / ntot);
        sm2 += q + d * d * (sn * 16.f / ntot);
        sn = ntot;
      }
    };
    auto zero = [](f32x16& acc) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    };
    // one scheduling region per tap: the next tap's 20 LDS reads are spread behind the 16 MFMAs of this one (pinned order: an MFMA,
    // then the reads that issue while it runs) instead of all in front of the first MFMA
#define UP_STEP(LY, C_, T_, CUR, NXT, NLY, NC, NT_, HAS_NEXT)          \
    if (HAS_NEXT) read_tap(NLY, NC, NT_, NXT);                          \
    mfma_tap(CUR, acc);                                                 \
    _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) {                 \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                \
    }                                                                   \
    __builtin_amdgcn_sched_barrier(0);
    Frag fa, fb;
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
      const int ly = wave + 4 * i;
      f32x16 acc;
      read_tap(ly, 0, 0, fa);
      zero(acc);
      UP_STEP(ly, 0, 0, fa, fb, ly, 1, 0, true)
      epilogue(ly, 0, acc);
      zero(acc);
      UP_STEP(ly, 1, 0, fb, fa, ly, 1, 1, true)
      UP_STEP(ly, 1, 1, fa, fb, ly, 2, 0, true)
      epilogue(ly, 1, acc);
      zero(acc);
      UP_STEP(ly, 2, 0, fb, fa, ly, 2, 1, true)
      UP_STEP(ly, 2, 1, fa, fb, ly, 3, 0, true)
      epilogue(ly, 2, acc);
      zero(acc);
      UP_STEP(ly, 3, 0, fb, fa, ly, 3, 1, true)
      UP_STEP(ly, 3, 1, fa, fb, ly, 3, 2, true)
      UP_STEP(ly, 3, 2, fb, fa, ly, 3, 3, true)
      UP_STEP(ly, 3, 3, fa, fb, ly, 0, 0, false)
      epilogue(ly, 3, acc);
    }
#undef UP_STEP
    CTVAE_PH(up, 0, 3);
    __syncthreads();
    CTVAE_PH(up, 0, 4);
    cur = nxt;
  }
  if (a.bn_part != nullptr) {
    {
      const float on = __shfl_xor(sn, 32, 64), om = __shfl_xor(smean, 32, 64), oq = __shfl_xor(sm2, 32, 64);
      const float ntot = sn + on;
      if (ntot > 0.f) {
        const float d = om - smean;
        sm2 = sm2 + oq + d * d * (sn * on / ntot);
        smean = smean + d * (on / ntot);
      }
      sn = ntot;
    }
    if (lh == 0) {
      float* st = &sS[(wave * C + li) * 3];
      st[0] = sn; st[1] = smean; st[2] = sm2;
    }
    __syncthreads();
    if (tid < C) {
      float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float nb = sS[(w * C + tid) * 3], mb = sS[(w * C + tid) * 3 + 1], qb = sS[(w * C + tid) * 3 + 2];
        if (nb > 0.f) {
          const float ntot = n + nb, d = mb - mean;
          mean += d * (nb / ntot);
          m2 += qb + d * d * (n * nb / ntot);
          n = ntot;
        }
      }
      float* p = a.bn_part + ((long)blockIdx.x * C + tid) * 3;
      p[0] = n; p[1] = mean; p[2] = m2;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// dW[t][ci][co] = sum over the class's output pixels of x[q + d_t][ci] * dy[2q + parity][co].
// Tap counts per class are the k3 s2 p1 pattern {1,2,2,4} (checked on the host): nine accumulators, static indices.
template <int CLS>
struct ClsInfo {
  static constexpr int ntaps = CLS == 0 ? 1 : (CLS == 3 ? 4 : 2);
  static constexpr int first = CLS == 0 ? 0 : (CLS == 1 ? 1 : (CLS == 2 ? 3 : 5));
};

// Register budget: ONE wave per SIMD (512 registers: the nine accumulators live in AGPRs, nothing spills).  Cut for two waves
// (256 registers) the FUSED variant spilled 168 bytes, and every scratch reload in the step loop is a vector-memory operation
// whose `s_waitcnt vmcnt(0)` also drains the dy / y loads just issued for the next step -- the read-ahead overlapped nothing:
// 45 -> 27 us at bs = 64 (one workgroup per CU anyway), and still ahead at bs = 256 where a second workgroup per CU is given up
// (step 1.582 -> 1.567 ms).
template <bool FUSED>
__global__ __launch_bounds__(256, 1) void up_wgrad_kernel(const UpArgs a) {
  kernarg_warm<sizeof(UpArgs)>();
  __shared__ __attribute__((aligned(16))) float sA[NP * LDA];   // 42.8 KB
  __shared__ __attribute__((aligned(16))) float sR[4 * C * C];  // 16 KB: cross-wave merge, one tap at a time
  __shared__ float sB[4 * C];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rX = rsrc(a.X, (long)a.B * a.H * a.W * C * 4);
  const int OH = 2 * a.H, OW = 2 * a.W;
  const long obytes = (long)a.B * OH * OW * C * 4;
  const __amdgpu_buffer_rsrc_t rG = rsrc(a.dY, obytes);
  const __amdgpu_buffer_rsrc_t rYb = rsrc(a.dy_y, FUSED ? obytes : 0);
  const __amdgpu_buffer_rsrc_t rGY = rsrc(a.gy_out, FUSED ? obytes : 0);
  const float k1 = FUSED ? a.dy_coef[li] : 0.f, k2 = FUSED ? a.dy_coef[C + li] : 0.f, k3 = FUSED ? a.dy_coef[2 * C + li] : 0.f;
  const float ksc = FUSED ? a.dy_coef[3 * C + li] : 0.f, ksh = FUSED ? a.dy_coef[4 * C + li] : 0.f;
  const float nslope = a.dy_act == ACT_LRELU ? kLeaky : (a.dy_act == ACT_RELU ? 0.f : 1.f);   // host admits only these
  const bool xf = a.xf_scale != nullptr;
  const f32x4 xsc = xf ? *reinterpret_cast<const f32x4*>(a.xf_scale + 4 * (tid & 7)) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 xsh = xf ? *reinterpret_cast<const f32x4*>(a.xf_shift + 4 * (tid & 7)) : f32x4{0.f, 0.f, 0.f, 0.f};
  const float xns = a.xf_act == ACT_LRELU ? kLeaky : (a.xf_act == ACT_RELU ? 0.f : 1.f);

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  CTVAE_PH(up, 1, 0);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const TileXY cur = tile_xy(a, tile);
    CTVAE_PH(up, 1, 1);
    // no cross-tile prefetch here: the nine accumulators need the registers (the CU's second workgroup covers the
    // wait); the patch goes through in two rounds of 5 float4 per thread for the same reason
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 v[NLD / 2];
#pragma unroll
      for (int j = 0; j < NLD / 2; ++j) {
        const int e = tid + 256 * (j + h * (NLD / 2));
        const int pp = e >> 3, c4 = e & 7;
        const int py = (pp * 1986) >> 16, px = pp - py * PW;
        const int iy = cur.y0 + py, ix = cur.x0 + px;
        const bool ok = e < NP * 8 && iy < a.H && ix < a.W;
        v[j] = ld4(rX, ok ? (unsigned)(((cur.b * a.H + iy) * a.W + ix) * C + 4 * c4) * 4u : kOOBu);
      }
      if (xf) {   // lazy BatchNorm apply of the previous block (channel quad e & 7 = tid & 7: per-thread coefficients)
#pragma unroll
        for (int j = 0; j < NLD / 2; ++j) {
          const int e = tid + 256 * (j + h * (NLD / 2));
          const int pp = e >> 3;
          const int py = (pp * 1986) >> 16, px = pp - py * PW;
          const float inside = (cur.y0 + py < a.H && cur.x0 + px < a.W) ? 1.f : 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float u = v[j][q] * xsc[q] + xsh[q] * inside;
            v[j][q] = fmaxf(u, u * xns);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NLD / 2; ++j) {
        const int e = tid + 256 * (j + h * (NLD / 2));
        if (e < NP * 8) *reinterpret_cast<f32x4*>(&sA[(e >> 3) * LDA + 4 * (e & 7)]) = v[j];
      }
    }
    __syncthreads();

    CTVAE_PH(up, 1, 2);
    // one step = (row, class, half row): 8 pixel pairs.  Its dy (and y) loads are issued TWO steps ahead (three register slots):
    // with one wave per SIMD a step of the one-tap class is 8 MFMAs = 0.2 us, far less than a memory round trip
    float dv[3][8], yv[FUSED ? 3 : 1][FUSED ? 8 : 1];
    auto step_off = [&](int ly, int c, int h) {
      return (unsigned)((((cur.b * OH + 2 * (cur.y0 + ly) + a.cpy[c]) * OW + 2 * cur.x0 + a.cpx[c] + 2 * (16 * h + lh)) * C + li)) * 4u;
    };
    auto dy_load = [&](unsigned off, auto slot_c) {
      constexpr int slot = decltype(slot_c)::value;
#pragma unroll
      for (int j = 0; j < 8; ++j) dv[slot][j] = ld1(rG, off + (unsigned)(4 * j * C) * 4u);
      if constexpr (FUSED) {
#pragma unroll
        for (int j = 0; j < 8; ++j) yv[slot][j] = ld1(rYb, off + (unsigned)(4 * j * C) * 4u);
      }
    };
    auto run_step = [&](int ly, auto cls_c, int h, unsigned off, auto slot_c) {
      constexpr int CLS = decltype(cls_c)::value;
      constexpr int slot = decltype(slot_c)::value;
      using CI = ClsInfo<CLS>;
      int xo[CI::ntaps];
#pragma unroll
      for (int t = 0; t < CI::ntaps; ++t) xo[t] = ((ly + a.tdy[CLS][t]) * PW + 16 * h + lh + a.tdx[CLS][t]) * LDA + li;
      // x operands are fetched one pixel pair ahead; sched_barrier keeps the compiler from hoisting all 32 LDS
      // reads of the step (live ranges would push the nine accumulators out of the register file)
      float xa[CI::ntaps], xb[CI::ntaps];
#pragma unroll
      for (int t = 0; t < CI::ntaps; ++t) xa[t] = sA[xo[t]];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j + 1 < 8) {
#pragma unroll
          for (int t = 0; t < CI::ntaps; ++t) xb[t] = sA[xo[t] + 2 * (j + 1) * LDA];
        }
        float d = dv[slot][j];
        if constexpr (FUSED) {   // BatchNorm-backward apply on load; g_y goes out for the data-gradient kernel
          const float yy = yv[slot][j];
          const float g1 = (yy * ksc + ksh) > 0.f ? d : d * nslope;   // LeakyReLU / ReLU / identity derivative
          d = k1 * g1 + k2 * yy + k3;
          st1(rGY, off + (unsigned)(4 * j * C) * 4u, d);
        }
        bsum += d;
#pragma unroll
        for (int t = 0; t < CI::ntaps; ++t)
          acc[CI::first + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[t], d, acc[CI::first + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < CI::ntaps; ++t) xa[t] = xb[t];
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // 16 steps of the tile: step S = (row S / 8, class (S % 8) / 2, half S & 1), register slot S % 3
    unsigned soff[3];
    auto issue = [&](auto s_c) {
      constexpr int S = decltype(s_c)::value;
      soff[S % 3] = step_off(wave + 4 * (S / 8), (S % 8) / 2, S & 1);
      dy_load(soff[S % 3], std::integral_constant<int, S % 3>{});
    };
    auto do_step = [&](auto s_c) {
      constexpr int S = decltype(s_c)::value;
      if constexpr (S + 2 < 16) issue(std::integral_constant<int, S + 2>{});
      run_step(wave + 4 * (S / 8), std::integral_constant<int, (S % 8) / 2>{}, S & 1, soff[S % 3], std::integral_constant<int, S % 3>{});
    };
    issue(std::integral_constant<int, 0>{});
    issue(std::integral_constant<int, 1>{});
#define UPW_STEP(S_) do_step(std::integral_constant<int, S_>{});
    UPW_STEP(0) UPW_STEP(1) UPW_STEP(2) UPW_STEP(3) UPW_STEP(4) UPW_STEP(5) UPW_STEP(6) UPW_STEP(7)
    CTVAE_PH(up, 1, 3);
    UPW_STEP(8) UPW_STEP(9) UPW_STEP(10) UPW_STEP(11) UPW_STEP(12) UPW_STEP(13) UPW_STEP(14) UPW_STEP(15)
#undef UPW_STEP
    CTVAE_PH(up, 1, 4);
    __syncthreads();
  }
  // ---- merge the 4 waves tap by tap (fixed order) into this workgroup's slab [9][32 ci][32 co] ----
  bsum += __shfl_xor(bsum, 32, 64);
  if (lh == 0) sB[wave * C + li] = bsum;
  auto merge_tap = [&](int wtap, const f32x16& v) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) sR[(wave * C + 8 * (r >> 2) + 4 * lh + (r & 3)) * C + li] = v[r];
    __syncthreads();
    for (int e = tid; e < C * C; e += 256) {
      const float s4 = ((sR[e] + sR[C * C + e]) + sR[2 * C * C + e]) + sR[3 * C * C + e];
      a.out[((long)blockIdx.x * NT + wtap) * C * C + e] = s4;
    }
  };
  merge_tap(a.twt[0][0], acc[0]);
  merge_tap(a.twt[1][0], acc[1]);
  merge_tap(a.twt[1][1], acc[2]);
  merge_tap(a.twt[2][0], acc[3]);
  merge_tap(a.twt[2][1], acc[4]);
  merge_tap(a.twt[3][0], acc[5]);
  merge_tap(a.twt[3][1], acc[6]);
  merge_tap(a.twt[3][2], acc[7]);
  merge_tap(a.twt[3][3], acc[8]);
  if (a.pbias != nullptr && tid < C) a.pbias[(long)blockIdx.x * C + tid] = ((sB[tid] + sB[C + tid]) + sB[2 * C + tid]) + sB[3 * C + tid];
  CTVAE_PH(up, 1, 5);
}

void fill(UpArgs& a, const ConvGeom& g) {
  a.B = g.B; a.H = g.gH; a.W = g.gW;
  a.tiles_y = g.gH / TH; a.tiles_x = g.gW / TW;
  a.ntiles = g.B * a.tiles_y * a.tiles_x;
  for (int c = 0; c < NCLS; ++c) {
    a.ntaps[c] = g.ntaps[c]; a.cpy[c] = g.py[c]; a.cpx[c] = g.px[c];
    for (int t = 0; t < MAXT; ++t) {
      const bool ok = t < g.ntaps[c];
      a.tdy[c][t] = ok ? g.taps[c][t].dy : 0;
      a.tdx[c][t] = ok ? g.taps[c][t].dx : 0;
      a.twt[c][t] = ok ? g.taps[c][t].wtap : 0;
    }
  }
}

constexpr size_t kUpFwdSmem = (size_t)(NP * LDA + NT * C * C) * 4;   // 79 632 B: two workgroups per CU

}  // namespace

// forward geometry (build_geom kind 1) of a 32 -> 32 channel, k3 s2 p1 op1 transposed convolution
bool upconv_supported(const ConvGeom& g) {
  if (!packed_weights(g) || g.wT != 0 || g.gC != C || g.sC != C || g.wCi != C || g.wCo != C) return false;
  if (g.is != 1 || g.os != 2 || g.ncls != NCLS) return false;
  if (g.sH != 2 * g.gH || g.sW != 2 * g.gW || g.gH % TH != 0 || g.gW % TW != 0) return false;
  int total = 0;
  for (int c = 0; c < NCLS; ++c) {
    if (g.ntaps[c] < 1 || g.ntaps[c] > MAXT) return false;
    total += g.ntaps[c];
    for (int t = 0; t < g.ntaps[c]; ++t) {
      const Tap& tp = g.taps[c][t];
      if (tp.dy < 0 || tp.dy > 1 || tp.dx < 0 || tp.dx > 1 || tp.wtap < 0 || tp.wtap >= NT) return false;
    }
  }
  // both kernels walk the classes in the k3 s2 p1 pattern: {1,2,2,4} taps
  if (g.ntaps[0] != 1 || g.ntaps[1] != 2 || g.ntaps[2] != 2 || g.ntaps[3] != 4) return false;
  return total == NT && (long)g.B * g.sH * g.sW * C < (1L << 29);
}
bool upconv_wgrad_supported(const ConvGeom& g) { return upconv_supported(g); }

static const int kUpWgs = [] { const char* e = getenv("CTVAE_UP_WGS"); return e ? atoi(e) : 512; }();   // diagnostic override
int upconv_rows(const ConvGeom& g) {
  const int nt = g.B * (g.gH / TH) * (g.gW / TW);
  return nt < kUpWgs ? nt : kUpWgs;
}

int launch_upconv_forward(const ConvGeom& g, const float* X, const float* W, const float* bias, float* S, int act,
                          float* bn_part, hipStream_t st, const InXform* xf) {
  UpArgs a{};
  fill(a, g);
  a.X = X; a.Wt = W; a.bias = bias; a.out = S; a.act = act; a.bn_part = bn_part;
  if (xf != nullptr && xf->scale != nullptr) {
    if (xf->act != ACT_NONE && xf->act != ACT_RELU && xf->act != ACT_LRELU) return kErrBadArg;
    a.xf_scale = xf->scale; a.xf_shift = xf->shift; a.xf_act = xf->act;
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(up_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kUpFwdSmem);
    attr_set = true;
  }
  ProfScope ps("up_fwd_kernel", st, 2.0 * a.ntiles * TH * TW * NT * C * C, 4.0 * a.ntiles * TH * TW * C * 5.0);
  hipLaunchKernelGGL(up_fwd_kernel, dim3(upconv_rows(g)), dim3(256), kUpFwdSmem, st, a);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

// partial slabs [parts][9][32][32] (+ bias partials [parts][32]) into ws; the caller reduces them
int launch_upconv_wgrad(const ConvGeom& g, const float* X, const float* dY, float* ws, float** part_out, float** pbias_out,
                        int* nparts, bool want_bias, hipStream_t st, const DyXform* dyx, const InXform* xf) {
  UpArgs a{};
  fill(a, g);
  a.X = X; a.dY = dY;
  if (xf != nullptr && xf->scale != nullptr) {
    if (xf->act != ACT_NONE && xf->act != ACT_RELU && xf->act != ACT_LRELU) return kErrBadArg;
    a.xf_scale = xf->scale; a.xf_shift = xf->shift; a.xf_act = xf->act;
  }
  if (dyx != nullptr && dyx->y != nullptr && dyx->act != ACT_NONE && dyx->act != ACT_RELU && dyx->act != ACT_LRELU) return kErrBadArg;
  if (dyx != nullptr && dyx->y != nullptr) { a.dy_y = dyx->y; a.dy_coef = dyx->coef; a.gy_out = dyx->gy_out; a.dy_act = dyx->act; }
  const int nwg = upconv_rows(g);
  a.out = ws;
  a.pbias = want_bias ? ws + (size_t)nwg * NT * C * C : nullptr;
  ProfScope ps("up_wgrad_kernel", st, 2.0 * a.ntiles * TH * TW * NT * C * C, 4.0 * a.ntiles * TH * TW * C * 5.0);
  if (a.dy_y != nullptr) hipLaunchKernelGGL(up_wgrad_kernel<true>, dim3(nwg), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(up_wgrad_kernel<false>, dim3(nwg), dim3(256), 0, st, a);
  CTVAE_LAUNCH_CHECK();
  *part_out = a.out;
  *pbias_out = a.pbias;
  *nparts = nwg;
  return 0;
}

}  // namespace ctvae
