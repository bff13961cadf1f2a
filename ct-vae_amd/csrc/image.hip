// Image-side 3x3 convolution between a 32-channel feature map and the 3-channel picture (vanilla_vae.py:73-74,
// `nn.Conv2d(hidden_dims[-1], out_channels=3, kernel_size=3, padding=1)` + Tanh), forward, weight gradient and data
// gradient, on the f32 matrix cores.
//
// With N = 3 an im2col GEMM tile would be 90 % padding.  These kernels turn the problem round so that the taps become
// the GEMM's N (or K) dimension and the spatial gather happens inside LDS:
//
//   forward   Z[p'][(t,co)] = sum_ci a[p'][ci] * W[t][ci][co]        one 32-wide MFMA tile: M = pixels, K = 32, N = 27
//             r[p][co]      = tanh(bias + sum_t Z[p + off_t][(t,co)])  9-point gather of the LDS-resident Z
//   wgrad     dW[t][ci][co] = sum_p' a[p'][ci] * g[p' - off_t][co]    M = ci (32), N = (t,co) (27), K = pixels: one
//                                                                     32x32 accumulator per wave for the whole launch
//   dgrad     ga[p][ci]     = sum_(t,co) g[p + off_t][co] * W[t][ci][co]   M = pixels, K = 27, N = ci (32)
//
// One workgroup = one 8 x 32 pixel tile of one image at a time (persistent loop over tiles, next tile's global loads
// in flight while the current one computes).  The 32-channel operand is staged ONCE per tile (tile + 1-pixel halo) with
// 16-B coalesced loads; the optional per-channel affine + LeakyReLU of the BatchNorm in front of the layer is applied
// there (lazy BatchNorm apply: the normalised activation never exists in HBM).  The data gradient also emits the
// BatchNorm's backward sums per workgroup (see TapGemmArgs::bnb_*), reading y with the same 128-B-per-pixel pattern
// it writes ga with.  All reductions run in a fixed order: bit-reproducible.
#include <type_traits>

#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

#ifdef CTVAE_PHASE_TIMING
// diagnostic build only (tools/phase_probe.py): timestamps of the phases of the first tiles of each workgroup
__device__ unsigned long long g_img_phase[1024 * 32];
#define IPHASE(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024 && (i) < 32) g_img_phase[blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)
extern "C" int ctvae_debug_img_phase_read(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_img_phase), (size_t)n * 8);
}
#else
#define IPHASE(i) do {} while (0)
#endif

namespace {

constexpr int TH = 8, TW = 32, PH = 10, PW = 34, NP = PH * PW /*340*/, NPP = 352 /*11 x 32*/;
constexpr int LDA = 36;          // floats per staged pixel row (32 + 4: ds_read_b128 of 8 rows hits 32 distinct banks)
constexpr int LDZ = 33;          // floats per Z row (odd: pixel-per-lane reads are conflict-free)
constexpr int GH = 12, GW = 36;  // zero-ringed gradient grid of the wgrad kernel (tile + 2)
constexpr int C = 32, NO = 3, NT = 9, NJ = 27;
constexpr unsigned kOOBi = 0x80000000u;

struct ImgArgs {
  const float* X;      // [B,H,W,32]
  const float* Wt;     // [9][32][3]
  const float* bias;
  const float* dY;     // [B,H,W,3]
  const float* scale;  // lazy BatchNorm apply on X (may be null)
  const float* shift;
  float* out;
  float* pbias;
  const float* bn_y;   // dgrad: fused BatchNorm-backward sums (may be null)
  const float* bn_mean;
  const float* bn_invstd;
  const float* bn_gamma;
  const float* bn_beta;
  float* bn_part;
  int bn_act, in_act, act;
  int B, H, W, tiles_y, tiles_x, ntiles;
  int tdy[NT], tdx[NT], twt[NT];
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
__device__ __forceinline__ TileXY tile_xy(const ImgArgs& a, int tile) {
  const int per = a.tiles_y * a.tiles_x;
  const int b = tile / per, r = tile - b * per;
  const int ty = r / a.tiles_x;
  return TileXY{b, ty * TH, (r - ty * a.tiles_x) * TW};
}

__device__ __forceinline__ TileXY tile_xy_hw(int tiles_y, int tiles_x, int tile) {
  const int per = tiles_y * tiles_x;
  const int b = tile / per, r = tile - b * per;
  const int ty = r / tiles_x;
  return TileXY{b, ty * TH, (r - ty * tiles_x) * TW};
}
// table[t] for a lane-varying t without a private-memory array
__device__ __forceinline__ int tap_dy9(const int* tab, int t) {
  int w = 0;
#pragma unroll
  for (int k = 0; k < NT; ++k) w = (t == k) ? tab[k] : w;
  return w;
}

// ---- staging of the 32-channel patch (tile + halo): 340 pixels x 8 float4, 11 per thread -------------------------
constexpr int NLD = (NP * 8 + 255) / 256;   // 11

struct Patch {
  f32x4 v[NLD];
  unsigned ok;   // bit j: load j is inside the image
};

// slot arithmetic redone per tile: fewer live registers (img_fwd_kernel runs at 3 waves per SIMD, 170 VGPRs)
__device__ __forceinline__ void patch_load_calc(const ImgArgs& a, __amdgpu_buffer_rsrc_t rX, const TileXY& t, Patch& p) {
  unsigned okm = 0;
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int e = threadIdx.x + 256 * j;
    const int pp = e >> 3, c4 = e & 7;
    const int py = (pp * 241) >> 13, px = pp - py * PW;   // pp / 34 for pp < 352
    const int iy = t.y0 - 1 + py, ix = t.x0 - 1 + px;
    const bool ok = e < NP * 8 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    p.v[j] = ld4(rX, ok ? (unsigned)(((t.b * a.H + iy) * a.W + ix) * C + 4 * c4) * 4u : kOOBi);
    okm |= (ok ? 1u : 0u) << j;
  }
  p.ok = okm;
}

// per-thread constants of the 11 patch slots (tile independent): byte offset relative to the patch origin and (py, px)
struct PatchIdx {
  unsigned rel[NLD];
  int pyx[NLD];   // py << 16 | px, or -1 for the unused tail slots
};
__device__ __forceinline__ void patch_index(const ImgArgs& a, PatchIdx& ix) {
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int e = threadIdx.x + 256 * j;
    const int pp = e >> 3, c4 = e & 7;
    const int py = (pp * 241) >> 13, px = pp - py * PW;   // pp / 34 for pp < 352
    ix.rel[j] = (unsigned)((py * a.W + px) * C + 4 * c4) * 4u;
    ix.pyx[j] = e < NP * 8 ? ((py << 16) | px) : -1;
  }
}
__device__ __forceinline__ void patch_load(const ImgArgs& a, __amdgpu_buffer_rsrc_t rX, const TileXY& t, const PatchIdx& ix,
                                           Patch& p) {
  // origin = pixel (y0 - 1, x0 - 1); it may lie outside the image (negative offset), every in-range slot adds back >= that
  const unsigned origin = (unsigned)(((t.b * a.H + t.y0 - 1) * a.W + t.x0 - 1) * C) * 4u;
  unsigned okm = 0;
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int py = ix.pyx[j] >> 16, px = ix.pyx[j] & 0xffff;
    const bool ok = ix.pyx[j] >= 0 && (unsigned)(t.y0 - 1 + py) < (unsigned)a.H && (unsigned)(t.x0 - 1 + px) < (unsigned)a.W;
    p.v[j] = ld4(rX, ok ? origin + ix.rel[j] : kOOBi);
    okm |= (ok ? 1u : 0u) << j;
  }
  p.ok = okm;
}

__device__ __forceinline__ void patch_store(const ImgArgs& a, const Patch& p, float* sA) {
  const bool xform = a.scale != nullptr;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (xform) {   // the thread's channel quad is the same for all its loads (256 % 8 == 0)
    sc = *reinterpret_cast<const f32x4*>(a.scale + 4 * (threadIdx.x & 7));
    sh = *reinterpret_cast<const f32x4*>(a.shift + 4 * (threadIdx.x & 7));
  }
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int e = threadIdx.x + 256 * j;
    if (e < NP * 8) {
      f32x4 v = p.v[j];
      if (xform) {
        const bool ok = (p.ok >> j) & 1u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // zero padding stays zero; the activation is decided once per quad (common.hpp act_slope_fwd)
          const float t = v[k] * sc[k] + sh[k];
          v[k] = ok ? (a.in_act == ACT_TANH ? act_fwd(t, ACT_TANH) : act_slope_fwd(t, act_slope(a.in_act))) : 0.f;
        }
      }
      *reinterpret_cast<f32x4*>(&sA[(e >> 3) * LDA + 4 * (e & 7)]) = v;
    }
  }
}

// weight matrix column of lane j = 3*t + co for rows ci (forward/wgrad) -- tap table lookups without private arrays
__device__ __forceinline__ int tap_wt(const ImgArgs& a, int t) {
  int w = 0;
#pragma unroll
  for (int k = 0; k < NT; ++k) w = (t == k) ? a.twt[k] : w;
  return w;
}
__device__ __forceinline__ int tap_dy(const ImgArgs& a, int t) {
  int w = 0;
#pragma unroll
  for (int k = 0; k < NT; ++k) w = (t == k) ? a.tdy[k] : w;
  return w;
}
__device__ __forceinline__ int tap_dx(const ImgArgs& a, int t) {
  int w = 0;
#pragma unroll
  for (int k = 0; k < NT; ++k) w = (t == k) ? a.tdx[k] : w;
  return w;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void img_fwd_kernel(const ImgArgs a) {
  kernarg_warm<sizeof(ImgArgs)>();
  __shared__ __attribute__((aligned(16))) float sA[NPP * LDA];   // patch; reused for Z [340][33]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rX = rsrc(a.X, (long)a.B * a.H * a.W * C * 4);
  const __amdgpu_buffer_rsrc_t rO = rsrc(a.out, (long)a.B * a.H * a.W * NO * 4);

  // B operand, resident for the whole launch: breg[kg*4+q] = W'[ci = kg*8 + 4*lh + q][j = li]
  float breg[16];
  {
    const int t = li / 3, co = li - 3 * t;
    const int wt = tap_wt(a, t);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int ci = (kk >> 2) * 8 + 4 * lh + (kk & 3);
      breg[kk] = li < NJ ? a.Wt[(wt * C + ci) * NO + co] : 0.f;
    }
  }
  float bias_r[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) bias_r[n] = a.bias != nullptr ? a.bias[n] : 0.f;
  int zoff[NT];   // Z row offset of tap t relative to the output pixel's own patch row
#pragma unroll
  for (int t = 0; t < NT; ++t) zoff[t] = (a.tdy[t] * PW + a.tdx[t]) * LDZ + 3 * t;
  for (int e = tid; e < (NPP - NP) * LDA; e += 256) sA[NP * LDA + e] = 0.f;   // MFMA rows beyond the patch

  Patch pt;
  int tile = blockIdx.x;
  TileXY cur = tile_xy(a, tile < a.ntiles ? tile : 0);
  if (tile < a.ntiles) patch_load_calc(a, rX, cur, pt);
  int it = 0;
  IPHASE(0);
  for (; tile < a.ntiles; tile += gridDim.x, ++it) {
    patch_store(a, pt, sA);
    IPHASE(1 + 6 * it);
    __syncthreads();
    IPHASE(2 + 6 * it);
    const int next = tile + gridDim.x;
    const TileXY nxt = tile_xy(a, next < a.ntiles ? next : 0);
    if (next < a.ntiles) patch_load_calc(a, rX, nxt, pt);   // in flight during the MFMA / gather phases

    IPHASE(3 + 6 * it);
    // ---- Z = patch x W' : wave w takes the 32-row blocks w, w+4, w+8 ----
    f32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int mt = wave + 4 * i;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      if (mt < NPP / 32) {
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
          const f32x4 af = *reinterpret_cast<const f32x4*>(&sA[(mt * 32 + li) * LDA + kg * 8 + 4 * lh]);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], breg[kg * 4 + q], acc[i], 0, 0, 0);
        }
      }
    }
    IPHASE(4 + 6 * it);
    __syncthreads();   // every wave is done reading the patch: its memory becomes Z
    float* sZ = sA;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int mt = wave + 4 * i;
      if (mt < NPP / 32 && li < NJ) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mt * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
          if (row < NP) sZ[row * LDZ + li] = acc[i][r];
        }
      }
    }
    __syncthreads();
    IPHASE(5 + 6 * it);
    // ---- gather: one thread per output pixel ----
    {
      const int ly = tid >> 5, lx = tid & 31;
      const float* z = sZ + ((ly + 1) * PW + lx + 1) * LDZ;
      float o[NO];
#pragma unroll
      for (int n = 0; n < NO; ++n) o[n] = bias_r[n];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int n = 0; n < NO; ++n) o[n] += z[zoff[t] + n];
      const unsigned off = (unsigned)(((cur.b * a.H + cur.y0 + ly) * a.W + cur.x0 + lx) * NO) * 4u;
      if (a.act == ACT_TANH) {   // the reference's final layer (vanilla_vae.py:73-75); decided once, not per value
#pragma unroll
        for (int n = 0; n < NO; ++n) st1(rO, off + 4u * n, act_fwd(o[n], ACT_TANH));
      } else {
#pragma unroll
        for (int n = 0; n < NO; ++n) st1(rO, off + 4u * n, act_slope_fwd(o[n], act_slope(a.act)));
      }
    }
    IPHASE(6 + 6 * it);
    __syncthreads();   // Z consumed before the next patch lands
    cur = nxt;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same forward with the 32-channel operand going from global memory STRAIGHT into the MFMA A registers: lane (pixel li,
// half lh) of a 32-pixel block loads channels [16 lh, 16 lh + 16) of its patch pixel with four 16-byte loads (two lanes cover
// the pixel's 128 bytes), applies the lazy BatchNorm + activation in registers, and the K steps pair channel s with channel
// 16 + s (the B registers are permuted to match).  No staging of the patch in LDS, no prefetch buffer in registers: LDS holds
// only Z (340 x 29 floats = 39 KB), so FOUR workgroups fit a CU (img_fwd_kernel: 51 KB and 162 VGPRs -> three), and a wave's
// next block is in flight while it multiplies the current one.
constexpr int LDZ2 = 29;   // odd: the pixel-per-lane gather reads are conflict-free; 4 x 340 x 29 x 4 B = 157.8 KB per CU
__global__ __launch_bounds__(256, 4) void img_fwd2_kernel(const ImgArgs a) {
  kernarg_warm<sizeof(ImgArgs)>();
  __shared__ float sZ[NP * LDZ2];
  __shared__ __attribute__((aligned(16))) float sSS[2 * C];   // BatchNorm scale | shift of the lazy apply
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rX = rsrc(a.X, (long)a.B * a.H * a.W * C * 4);
  const __amdgpu_buffer_rsrc_t rO = rsrc(a.out, (long)a.B * a.H * a.W * NO * 4);

  // B operand, resident for the whole launch: step s multiplies channel 16 lh + s: breg[s] = W'[ci = 16 lh + s][j = li]
  float breg[16];
  {
    const int t = li / 3, co = li - 3 * t;
    const int wt = tap_wt(a, t);
#pragma unroll
    for (int s = 0; s < 16; ++s) breg[s] = li < NJ ? a.Wt[(wt * C + 16 * lh + s) * NO + co] : 0.f;
  }
  const bool xform = a.scale != nullptr;
  if (xform && tid < 2 * C) sSS[tid] = tid < C ? a.scale[tid] : a.shift[tid - C];   // read per block: 32 registers less
  float bias_r[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) bias_r[n] = a.bias != nullptr ? a.bias[n] : 0.f;
  int zoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) zoff[t] = (a.tdy[t] * PW + a.tdx[t]) * LDZ2 + 3 * t;

  // the wave's blocks of 32 patch pixels: mt = wave, wave + 4, wave + 8 (the last exists for waves 0..2: 11 blocks of 340 + 12)
  int ppy[3], ppx[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int pp = (wave + 4 * i) * 32 + li;
    ppy[i] = (pp * 241) >> 13;
    ppx[i] = pp - ppy[i] * PW;
    if (pp >= NP) ppy[i] = -1000;     // beyond the patch: never inside the image
  }
  auto blk_load = [&](const TileXY& t, int i, f32x4 (&v)[4]) -> bool {
    const int iy = t.y0 - 1 + ppy[i], ix = t.x0 - 1 + ppx[i];
    const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    const unsigned off = ok ? (unsigned)(((t.b * a.H + iy) * a.W + ix) * C + 16 * lh) * 4u : kOOBi;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = ld4(rX, ok ? off + 16u * q : kOOBi);
    return ok;
  };
  const bool in_tanh = a.in_act == ACT_TANH;
  const float in_slope = act_slope(a.in_act);
  auto blk_mma = [&](int i, const f32x4 (&v)[4], bool ok) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 x4 = v[q];
      if (xform) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(&sSS[16 * lh + 4 * q]);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(&sSS[C + 16 * lh + 4 * q]);
        if (in_tanh) {
#pragma unroll
          for (int k = 0; k < 4; ++k) x4[k] = ok ? act_fwd(x4[k] * sc[k] + sh[k], ACT_TANH) : 0.f;
        } else {      // identity / LeakyReLU / ReLU: one select per value between the MFMAs, not a branch tree per value
#pragma unroll
          for (int k = 0; k < 4; ++k) x4[k] = ok ? act_slope_fwd(x4[k] * sc[k] + sh[k], in_slope) : 0.f;     // zero padding stays zero
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4[k], breg[4 * q + k], acc, 0, 0, 0);
    }
    const int mt = wave + 4 * i;
    if (li < NJ) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
        if (row < NP) sZ[row * LDZ2 + li] = acc[r];
      }
    }
  };
  const int nblk = wave < 3 ? 3 : 2;
  __syncthreads();   // sSS
  // all of a tile's blocks are requested one tile ahead: the loads of tile t + 1 go out before tile t's gather phase
  f32x4 va[4], vb[4], vc[4];
  bool oka = false, okb = false, okc = false;
  int tile = blockIdx.x;
  TileXY cur = tile_xy(a, tile < a.ntiles ? tile : 0);
  if (tile < a.ntiles) {
    oka = blk_load(cur, 0, va);
    okb = blk_load(cur, 1, vb);
    if (nblk == 3) okc = blk_load(cur, 2, vc);
  }
  for (; tile < a.ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    const TileXY nxt = tile_xy(a, next < a.ntiles ? next : 0);
    blk_mma(0, va, oka);
    blk_mma(1, vb, okb);
    if (nblk == 3) blk_mma(2, vc, okc);
    __syncthreads();
    if (next < a.ntiles) {
      oka = blk_load(nxt, 0, va);
      okb = blk_load(nxt, 1, vb);
      if (nblk == 3) okc = blk_load(nxt, 2, vc);
    }
    {   // gather: one thread per output pixel
      const int ly = tid >> 5, lx = tid & 31;
      const float* z = sZ + ((ly + 1) * PW + lx + 1) * LDZ2;
      float o[NO];
#pragma unroll
      for (int n = 0; n < NO; ++n) o[n] = bias_r[n];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int n = 0; n < NO; ++n) o[n] += z[zoff[t] + n];
      const unsigned off = (unsigned)(((cur.b * a.H + cur.y0 + ly) * a.W + cur.x0 + lx) * NO) * 4u;
      if (a.act == ACT_TANH) {   // the reference's final layer (vanilla_vae.py:73-75); decided once, not per value
#pragma unroll
        for (int n = 0; n < NO; ++n) st1(rO, off + 4u * n, act_fwd(o[n], ACT_TANH));
      } else {
#pragma unroll
        for (int n = 0; n < NO; ++n) st1(rO, off + 4u * n, act_slope_fwd(o[n], act_slope(a.act)));
      }
    }
    __syncthreads();   // Z consumed before the next tile's blocks land
    cur = nxt;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// partial dW slab per workgroup: out[blockIdx.x][wtap*32 + ci][co]; bias partial pbias[blockIdx.x][co]
__global__ __launch_bounds__(256) void img_wgrad_kernel(const ImgArgs a) {
  __shared__ __attribute__((aligned(16))) float sA[NPP * LDA];
  __shared__ __attribute__((aligned(16))) float sG[GH * GW * 4];   // cell (gy,gx) <-> tile pixel (gy-2, gx-2); ring = 0
  __shared__ float sB[4 * NO];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rX = rsrc(a.X, (long)a.B * a.H * a.W * C * 4);
  const __amdgpu_buffer_rsrc_t rG = rsrc(a.dY, (long)a.B * a.H * a.W * NO * 4);

  for (int e = tid; e < GH * GW * 4; e += 256) sG[e] = 0.f;
  for (int e = tid; e < (NPP - NP) * LDA; e += 256) sA[NP * LDA + e] = 0.f;   // K-loop padding steps read these rows
  // lane j = li = 3t + co reads g[p' - off_t][co]: patch pixel (py,px) <-> grid cell (py + 1 - dy_t, px + 1 - dx_t)
  const int t_l = li / 3, co_l = li - 3 * t_l;
  const bool jvalid = li < NJ;
  const int gbase = ((1 - tap_dy(a, t_l)) * GW + (1 - tap_dx(a, t_l)) + lh) * 4 + co_l;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bs[NO] = {0.f, 0.f, 0.f};

  Patch pt;
  float gv[NO];
  const int ly = tid >> 5, lx = tid & 31;
  auto g_load = [&](const TileXY& t) {
    const unsigned off = (unsigned)(((t.b * a.H + t.y0 + ly) * a.W + t.x0 + lx) * NO) * 4u;
#pragma unroll
    for (int n = 0; n < NO; ++n) gv[n] = ld1(rG, off + 4u * n);
  };
  PatchIdx pix;
  patch_index(a, pix);
  int tile = blockIdx.x;
  if (tile < a.ntiles) {
    const TileXY c0 = tile_xy(a, tile);
    patch_load(a, rX, c0, pix, pt);
    g_load(c0);
  }
  __syncthreads();   // sG ring zeroed
  int it = 0;
  IPHASE(0);
  for (; tile < a.ntiles; tile += gridDim.x, ++it) {
    patch_store(a, pt, sA);
    IPHASE(1 + 5 * it);
    {
      float* cell = &sG[((ly + 2) * GW + lx + 2) * 4];
#pragma unroll
      for (int n = 0; n < NO; ++n) {
        cell[n] = gv[n];
        bs[n] += gv[n];
      }
    }
    __syncthreads();
    IPHASE(2 + 5 * it);
    const int next = tile + gridDim.x;
    if (next < a.ntiles) {
      const TileXY nxt = tile_xy(a, next);
      patch_load(a, rX, nxt, pix, pt);
      g_load(nxt);
    }
    IPHASE(3 + 5 * it);
    // K loop over the 340 patch pixels, two per MFMA (lh picks the odd one).  Wave w takes the patch rows w, w+4, w+8:
    // inside a row every operand address is base + compile-time offset, so a step is two ds_read_b32 with immediate
    // offsets and one MFMA -- no index arithmetic, no branches, nothing between a load and its MFMA (dealing the 170
    // steps round-robin instead cost ~25 scalar instructions per step and ran at 255 cycles per 64-cycle MFMA).
    // Rows are double-buffered in registers: row r+1 is read while the 17 MFMAs of row r run.
    {
      constexpr int RS = PW / 2;   // 17 steps per patch row
      const int a_lane = lh * LDA + li;
      const int g_lane = jvalid ? gbase : 0;   // lanes j >= 27 walk the (zero) top ring row of the gradient grid
      float av[2][RS], gv[2][RS];
      auto rd = [&](int py, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const float* ap = sA + py * PW * LDA + a_lane;
        const float* gp = sG + (jvalid ? py * GW * 4 : 0) + g_lane;
#pragma unroll
        for (int j = 0; j < RS; ++j) {
          av[slot][j] = ap[2 * j * LDA];
          gv[slot][j] = gp[2 * j * 4];
        }
      };
      auto mm = [&](auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int j = 0; j < RS; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[slot][j], gv[slot][j], acc, 0, 0, 0);
      };
      using S0 = std::integral_constant<int, 0>;
      using S1 = std::integral_constant<int, 1>;
      const int wv = __builtin_amdgcn_readfirstlane(wave);
      rd(wv, S0{});
      rd(wv + 4, S1{});
      __builtin_amdgcn_sched_barrier(0);
      mm(S0{});
      __builtin_amdgcn_sched_barrier(0);
      if (wv + 8 < PH) rd(wv + 8, S0{});
      __builtin_amdgcn_sched_barrier(0);
      mm(S1{});
      __builtin_amdgcn_sched_barrier(0);
      if (wv + 8 < PH) mm(S0{});
    }
    IPHASE(4 + 5 * it);
    __syncthreads();
    IPHASE(5 + 5 * it);
  }
  // ---- merge the 4 waves in a fixed order ----
  float* sR = sA;   // [4][32][32]
#pragma unroll
  for (int r = 0; r < 16; ++r) sR[(wave * 32 + 8 * (r >> 2) + 4 * lh + (r & 3)) * 32 + li] = acc[r];
  float bw[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) bw[n] = wave_sum(bs[n]);
  if (lane == 0)
#pragma unroll
    for (int n = 0; n < NO; ++n) sB[wave * NO + n] = bw[n];
  __syncthreads();
  for (int e = tid; e < C * NJ; e += 256) {
    const int ci = e / NJ, j = e - ci * NJ;
    const float v = ((sR[(0 * 32 + ci) * 32 + j] + sR[(1 * 32 + ci) * 32 + j]) + sR[(2 * 32 + ci) * 32 + j]) + sR[(3 * 32 + ci) * 32 + j];
    const int t = j / 3, co = j - 3 * t;
    a.out[((long)blockIdx.x * (NT * C) + tap_wt(a, t) * C + ci) * NO + co] = v;
  }
  if (a.pbias != nullptr && tid < NO) a.pbias[(long)blockIdx.x * NO + tid] = ((sB[tid] + sB[NO + tid]) + sB[2 * NO + tid]) + sB[3 * NO + tid];
}

// ---------------------------------------------------------------------------------------------------------------------
// ga [B,H,W,32] from g [B,H,W,3]; optional BatchNorm-backward sums bn_part[blockIdx.x][32][2]
__global__ __launch_bounds__(256, 4) void img_dgrad_kernel(const ImgArgs a) {
  kernarg_warm<sizeof(ImgArgs)>();
  __shared__ __attribute__((aligned(16))) float sG[NP * 4];
  __shared__ float sS[4 * C * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rG = rsrc(a.dY, (long)a.B * a.H * a.W * NO * 4);
  const long obytes = (long)a.B * a.H * a.W * C * 4;
  const __amdgpu_buffer_rsrc_t rO = rsrc(a.out, obytes);
  const bool bn = a.bn_part != nullptr;
  const __amdgpu_buffer_rsrc_t rY = rsrc(a.bn_y, bn ? obytes : 0);

  // k = 2s + lh = 3t + co: A element = g[p + off_t][co] (LDS offset koff), B element = W[t][ci = li][co]
  int koff[14];
  float bw[14];
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int k0 = 2 * s, k1 = 2 * s + 1;
    const int t0 = k0 / 3, c0 = k0 % 3, t1 = (k1 < NJ ? k1 : 0) / 3, c1 = (k1 < NJ ? k1 : 0) % 3;
    const int o0 = (a.tdy[t0] * PW + a.tdx[t0]) * 4 + c0, o1 = (a.tdy[t1] * PW + a.tdx[t1]) * 4 + c1;
    const float w0 = a.Wt[(a.twt[t0] * C + li) * NO + c0];
    const float w1 = k1 < NJ ? a.Wt[(a.twt[t1] * C + li) * NO + c1] : 0.f;
    koff[s] = lh ? (k1 < NJ ? o1 : 0) : o0;
    bw[s] = lh ? w1 : w0;
  }
  const float bmean = bn ? a.bn_mean[li] : 0.f, binv = bn ? a.bn_invstd[li] : 0.f;
  const float bgm = bn ? a.bn_gamma[li] : 0.f, bbt = bn ? a.bn_beta[li] : 0.f;
  float s1 = 0.f, s2 = 0.f;

  // gradient patch: 340 pixels x 3 floats, threads 0..339 (+ second pass for 84 more) stage one pixel each
  float gq[2][NO];
  auto g_load = [&](const TileXY& t) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pp = tid + 256 * j;
      const int py = (pp * 241) >> 13, px = pp - py * PW;
      const int iy = t.y0 - 1 + py, ix = t.x0 - 1 + px;
      const bool ok = pp < NP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const unsigned off = ok ? (unsigned)(((t.b * a.H + iy) * a.W + ix) * NO) * 4u : kOOBi;
#pragma unroll
      for (int n = 0; n < NO; ++n) gq[j][n] = ld1(rG, ok ? off + 4u * n : kOOBi);
    }
  };
  int tile = blockIdx.x;
  TileXY cur = tile_xy(a, tile < a.ntiles ? tile : 0);
  if (tile < a.ntiles) g_load(cur);
  for (; tile < a.ntiles; tile += gridDim.x) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pp = tid + 256 * j;
      if (pp < NP)
#pragma unroll
        for (int n = 0; n < NO; ++n) sG[pp * 4 + n] = gq[j][n];
    }
    __syncthreads();
    const int next = tile + gridDim.x;
    const TileXY nxt = tile_xy(a, next < a.ntiles ? next : 0);
    if (next < a.ntiles) g_load(nxt);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ly = wave + 4 * i;   // tile row = one 32-pixel MFMA block
      const unsigned rowoff = (unsigned)(((cur.b * a.H + cur.y0 + ly) * a.W + cur.x0) * C + li) * 4u;
      float yv[16];
      if (bn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) yv[r] = ld1(rY, rowoff + (unsigned)((8 * (r >> 2) + 4 * lh + (r & 3)) * C) * 4u);
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* gp = &sG[((ly + 1) * PW + li + 1) * 4];
#pragma unroll
      for (int s = 0; s < 14; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[koff[s]], bw[s], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) st1(rO, rowoff + (unsigned)((8 * (r >> 2) + 4 * lh + (r & 3)) * C) * 4u, acc[r]);
      if (bn) {
        if (a.bn_act == ACT_TANH) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float xh = (yv[r] - bmean) * binv;
            const float g1 = acc[r] * act_bwd_from_out(act_fwd(bgm * xh + bbt, ACT_TANH), ACT_TANH);
            s1 += g1;
            s2 += g1 * xh;
          }
        } else {
          const float bsl = act_slope(a.bn_act);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float xh = (yv[r] - bmean) * binv;
            const float g1 = acc[r] * act_slope_bwd(bgm * xh + bbt, bsl);
            s1 += g1;
            s2 += g1 * xh;
          }
        }
      }
    }
    __syncthreads();
    cur = nxt;
  }
  if (bn) {
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lh == 0) {
      sS[(wave * C + li) * 2] = s1;
      sS[(wave * C + li) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < C * 2) {
      const float v = ((sS[tid] + sS[C * 2 + tid]) + sS[2 * C * 2 + tid]) + sS[3 * C * 2 + tid];
      a.bn_part[(long)blockIdx.x * C * 2 + tid] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The whole backward of the picture-side conv behind a BatchNorm + activation (vanilla_vae.py:71-74, final_layer.1-3) in
// ONE pass over y: img_dgrad_kernel with its BatchNorm-backward sums, plus the weight gradient.  The data gradient already
// holds, per lane, y[pixel(r, lh)][ci = li] for the 16 pixels r of its 32-pixel row block -- which is the A operand
// (M = ci, K = pixel) of  dW[t][ci][co] = sum_p a[p][ci] * g[p + off_t][co]  as it stands, and the B operand g[p + off_t][co]
// (N = j = 3t + co) sits in the staged gradient patch.  16 more MFMAs per row block into one accumulator per wave replace
// img_wgrad_kernel's own pass over the 134 MB activation (VanillaVAE bs = 256: 67 us).  a = act(gamma * xhat + beta) is the value
// the BatchNorm-backward sums need anyway.  Slabs out[blockIdx.x][wtap*32 + ci][co] and bias partials pbias[blockIdx.x][co] like
// img_wgrad_kernel; all merges in a fixed order.
__global__ __launch_bounds__(256, 4) void img_bwd_fused_kernel(const ImgArgs a, float* __restrict__ wpart, float* __restrict__ wpbias) {
  kernarg_warm<sizeof(ImgArgs) + 16>();
  const float bn_slope = act_slope(a.bn_act);
  __shared__ __attribute__((aligned(16))) float sG[NP * 4];
  __shared__ __attribute__((aligned(16))) float sR[4 * 32 * 32];   // end of the launch: the 4 waves' dW accumulators
  __shared__ float sS[4 * C * 2];
  __shared__ float sB[4 * NO];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rG = rsrc(a.dY, (long)a.B * a.H * a.W * NO * 4);
  const long obytes = (long)a.B * a.H * a.W * C * 4;
  const __amdgpu_buffer_rsrc_t rO = rsrc(a.out, obytes);
  const __amdgpu_buffer_rsrc_t rY = rsrc(a.bn_y, obytes);

  int koff[14];
  float bw[14];
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int k0 = 2 * s, k1 = 2 * s + 1;
    const int t0 = k0 / 3, c0 = k0 % 3, t1 = (k1 < NJ ? k1 : 0) / 3, c1 = (k1 < NJ ? k1 : 0) % 3;
    const int o0 = (a.tdy[t0] * PW + a.tdx[t0]) * 4 + c0, o1 = (a.tdy[t1] * PW + a.tdx[t1]) * 4 + c1;
    const float w0 = a.Wt[(a.twt[t0] * C + li) * NO + c0];
    const float w1 = k1 < NJ ? a.Wt[(a.twt[t1] * C + li) * NO + c1] : 0.f;
    koff[s] = lh ? (k1 < NJ ? o1 : 0) : o0;
    bw[s] = lh ? w1 : w0;
  }
  // weight gradient: lane column j = li = 3t + co (columns 27..31 are computed from a valid address and dropped)
  const int jj = li < NJ ? li : NJ - 1;
  const int t_l = jj / 3, co_l = jj - 3 * t_l;
  const int wlane = ((1 + tap_dy(a, t_l)) * PW + (1 + tap_dx(a, t_l)) + 4 * lh) * 4 + co_l;
  const float bmean = a.bn_mean[li], binv = a.bn_invstd[li], bgm = a.bn_gamma[li], bbt = a.bn_beta[li];
  float s1 = 0.f, s2 = 0.f;
  float bs[NO] = {0.f, 0.f, 0.f};
  f32x16 wacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) wacc[r] = 0.f;

  float gq[2][NO];
  auto g_load = [&](const TileXY& t) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pp = tid + 256 * j;
      const int py = (pp * 241) >> 13, px = pp - py * PW;
      const int iy = t.y0 - 1 + py, ix = t.x0 - 1 + px;
      const bool ok = pp < NP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const unsigned off = ok ? (unsigned)(((t.b * a.H + iy) * a.W + ix) * NO) * 4u : kOOBi;
#pragma unroll
      for (int n = 0; n < NO; ++n) gq[j][n] = ld1(rG, ok ? off + 4u * n : kOOBi);
    }
  };
  int tile = blockIdx.x;
  TileXY cur = tile_xy(a, tile < a.ntiles ? tile : 0);
  if (tile < a.ntiles) g_load(cur);
  for (; tile < a.ntiles; tile += gridDim.x) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pp = tid + 256 * j;
      if (pp < NP) {
        const int py = (pp * 241) >> 13, px = pp - py * PW;
        const bool inner = py >= 1 && py <= TH && px >= 1 && px <= TW;     // the tile itself: counted once for d bias
#pragma unroll
        for (int n = 0; n < NO; ++n) {
          sG[pp * 4 + n] = gq[j][n];
          bs[n] += inner ? gq[j][n] : 0.f;
        }
      }
    }
    __syncthreads();
    const int next = tile + gridDim.x;
    const TileXY nxt = tile_xy(a, next < a.ntiles ? next : 0);
    if (next < a.ntiles) g_load(nxt);
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
      const int ly = wave + 4 * i;   // tile row = one 32-pixel MFMA block
      const unsigned rowoff = (unsigned)(((cur.b * a.H + cur.y0 + ly) * a.W + cur.x0) * C + li) * 4u;
      float yv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) yv[r] = ld1(rY, rowoff + (unsigned)((8 * (r >> 2) + 4 * lh + (r & 3)) * C) * 4u);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* gp = &sG[((ly + 1) * PW + li + 1) * 4];
#pragma unroll
      for (int s = 0; s < 14; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[koff[s]], bw[s], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) st1(rO, rowoff + (unsigned)((8 * (r >> 2) + 4 * lh + (r & 3)) * C) * 4u, acc[r]);
      const float* gw = &sG[ly * PW * 4 + wlane];
      // the BatchNorm's activation is identity / LeakyReLU / ReLU (host-checked): one select for the value, one for the derivative
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float xh = (yv[r] - bmean) * binv;
        const float t = bgm * xh + bbt;
        const float av = act_slope_fwd(t, bn_slope);
        const float g1 = acc[r] * act_slope_bwd(t, bn_slope);
        s1 += g1;
        s2 += g1 * xh;
        wacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gw[(8 * (r >> 2) + (r & 3)) * 4], wacc, 0, 0, 0);
      }
    }
    __syncthreads();
    cur = nxt;
  }
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  if (lh == 0) {
    sS[(wave * C + li) * 2] = s1;
    sS[(wave * C + li) * 2 + 1] = s2;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) sR[(wave * 32 + 8 * (r >> 2) + 4 * lh + (r & 3)) * 32 + li] = wacc[r];
  float bwv[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) bwv[n] = wave_sum(bs[n]);
  if (lane == 0)
#pragma unroll
    for (int n = 0; n < NO; ++n) sB[wave * NO + n] = bwv[n];
  __syncthreads();
  if (tid < C * 2) {
    const float v = ((sS[tid] + sS[C * 2 + tid]) + sS[2 * C * 2 + tid]) + sS[3 * C * 2 + tid];
    a.bn_part[(long)blockIdx.x * C * 2 + tid] = v;
  }
  for (int e = tid; e < C * NJ; e += 256) {
    const int ci = e / NJ, j = e - ci * NJ;
    const float v = ((sR[(0 * 32 + ci) * 32 + j] + sR[(1 * 32 + ci) * 32 + j]) + sR[(2 * 32 + ci) * 32 + j]) + sR[(3 * 32 + ci) * 32 + j];
    const int t = j / 3, co = j - 3 * t;
    wpart[((long)blockIdx.x * (NT * C) + tap_wt(a, t) * C + ci) * NO + co] = v;
  }
  if (wpbias != nullptr && tid < NO) wpbias[(long)blockIdx.x * NO + tid] = ((sB[tid] + sB[NO + tid]) + sB[2 * NO + tid]) + sB[3 * NO + tid];
}

// =====================================================================================================================
// Picture-side stride-2 convolution 3 -> 32 channels (vanilla_vae.py:28-29, encoder.0: Conv2d(3, 32, 3, stride 2, pad 1)).
// Same idea with the roles swapped: K = (tap, input channel) = 27, N = 32 output channels, the 3-channel patch
// (17 x 65 pixels for an 8 x 32 output tile) lives in LDS and the MFMA A operand is read from it with per-k offsets.
constexpr int EPH = 2 * TH + 1, EPW = 2 * TW + 1, ENP = EPH * EPW;   // 17 x 65 = 1105
constexpr int ENLD = (ENP + 255) / 256;                               // 5 pixels per thread

struct EncArgs {
  const float* X;     // [B,2H,2W,3]
  const float* Wt;    // [9][3][32]
  const float* bias;
  const float* dY;    // wgrad: [B,H,W,32]
  float* out;         // fwd: y [B,H,W,32]; wgrad: slabs [nwg][27][32]
  float* pbias;       // wgrad: [nwg][32]
  float* bn_part;     // fwd: [nwg][32][3] (count, mean, M2) of y, may be null
  // wgrad, optional: dY is g_a of the BatchNorm behind this layer; g_y = k1*g_a*act'(y*scale+shift) + k2*y + k3 is formed on
  // load from dy_y and dy_coef [7][32] (rows as ctvae_bn_backward coef_out, 5/6 = d gamma / d beta, committed by block 0)
  const float* dy_y;
  const float* dy_coef;
  float* dgamma;
  float* dbeta;
  int dy_act, bn_accumulate;
  int act;
  int B, H, W, tiles_y, tiles_x, ntiles;   // H, W: OUTPUT size
  int tdy[NT], tdx[NT], twt[NT];
};

struct EncPatch {
  float v[ENLD][NO];
};

__device__ __forceinline__ void enc_patch_load(const EncArgs& a, __amdgpu_buffer_rsrc_t rX, const TileXY& t, EncPatch& p) {
  const int IH = 2 * a.H, IW = 2 * a.W;
#pragma unroll
  for (int j = 0; j < ENLD; ++j) {
    const int pp = threadIdx.x + 256 * j;
    const int py = (pp * 1009) >> 16, px = pp - py * EPW;   // pp / 65 for pp < 1280
    const int iy = 2 * t.y0 - 1 + py, ix = 2 * t.x0 - 1 + px;
    const bool ok = pp < ENP && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
    const unsigned off = (unsigned)(((t.b * IH + iy) * IW + ix) * NO) * 4u;
#pragma unroll
    for (int n = 0; n < NO; ++n) p.v[j][n] = ld1(rX, ok ? off + 4u * n : kOOBi);
  }
}
__device__ __forceinline__ void enc_patch_store(const EncPatch& p, float* sX) {
#pragma unroll
  for (int j = 0; j < ENLD; ++j) {
    const int pp = threadIdx.x + 256 * j;
    if (pp < ENP)
#pragma unroll
      for (int n = 0; n < NO; ++n) sX[pp * 4 + n] = p.v[j][n];
  }
}

__global__ __launch_bounds__(256, 4) void img_enc_fwd_kernel(const EncArgs a) {
  kernarg_warm<sizeof(EncArgs)>();
  __shared__ __attribute__((aligned(16))) float sX[ENP * 4];
  __shared__ float sS[4 * C * 3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rX = rsrc(a.X, (long)a.B * a.H * a.W * 4 * NO * 4);
  const __amdgpu_buffer_rsrc_t rO = rsrc(a.out, (long)a.B * a.H * a.W * C * 4);
  // k = 2s + lh = 3t + ci: A element = x[2q + off_t][ci] (LDS offset koff), B element = W[t][ci][co = li]
  int koff[14];
  float bw[14];
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int k0 = 2 * s, k1 = 2 * s + 1;
    const int t0 = k0 / 3, c0 = k0 % 3, t1 = (k1 < NJ ? k1 : 0) / 3, c1 = (k1 < NJ ? k1 : 0) % 3;
    const int o0 = (a.tdy[t0] * EPW + a.tdx[t0]) * 4 + c0, o1 = (a.tdy[t1] * EPW + a.tdx[t1]) * 4 + c1;
    const float w0 = a.Wt[(a.twt[t0] * NO + c0) * C + li];
    const float w1 = k1 < NJ ? a.Wt[(a.twt[t1] * NO + c1) * C + li] : 0.f;
    koff[s] = lh ? (k1 < NJ ? o1 : 0) : o0;
    bw[s] = lh ? w1 : w0;
  }
  const float bv = a.bias != nullptr ? a.bias[li] : 0.f;
  const float out_slope = act_slope(a.act);    // identity / LeakyReLU / ReLU (host-checked)
  float sn = 0.f, smean = 0.f, sm2 = 0.f;   // running (count, mean, M2) of this lane's channel

  EncPatch pt;
  int tile = blockIdx.x;
  TileXY cur = tile_xy_hw(a.tiles_y, a.tiles_x, tile < a.ntiles ? tile : 0);
  if (tile < a.ntiles) enc_patch_load(a, rX, cur, pt);
  for (; tile < a.ntiles; tile += gridDim.x) {
    enc_patch_store(pt, sX);
    __syncthreads();
    const int next = tile + gridDim.x;
    const TileXY nxt = tile_xy_hw(a.tiles_y, a.tiles_x, next < a.ntiles ? next : 0);
    if (next < a.ntiles) enc_patch_load(a, rX, nxt, pt);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ly = wave + 4 * i;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* xp = &sX[((2 * ly + 1) * EPW + 2 * li + 1) * 4];
#pragma unroll
      for (int s = 0; s < 14; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xp[koff[s]], bw[s], acc, 0, 0, 0);
      const unsigned rowoff = (unsigned)(((cur.b * a.H + cur.y0 + ly) * a.W + cur.x0) * C + li) * 4u;
      float m1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r] += bv;
        m1 += acc[r];
        st1(rO, rowoff + (unsigned)((8 * (r >> 2) + 4 * lh + (r & 3)) * C) * 4u, act_slope_fwd(acc[r], out_slope));
      }
      if (a.bn_part != nullptr) {   // Chan merge of this block's 16 values into the lane's running statistics
        m1 *= (1.f / 16.f);
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) q += (acc[r] - m1) * (acc[r] - m1);
        const float nt = sn + 16.f, d = m1 - smean;
        smean += d * (16.f / nt);
        sm2 += q + d * d * (sn * 16.f / nt);
        sn = nt;
      }
    }
    __syncthreads();
    cur = nxt;
  }
  if (a.bn_part != nullptr) {
    {   // the two lane halves hold the same channel
      const float on = __shfl_xor(sn, 32, 64), om = __shfl_xor(smean, 32, 64), oq = __shfl_xor(sm2, 32, 64);
      const float nt = sn + on;
      if (nt > 0.f) {
        const float d = om - smean;
        sm2 = sm2 + oq + d * d * (sn * on / nt);
        smean = smean + d * (on / nt);
      }
      sn = nt;
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
          const float nt = n + nb, d = mb - mean;
          mean += d * (nb / nt);
          m2 += qb + d * d * (n * nb / nt);
          n = nt;
        }
      }
      float* p = a.bn_part + ((long)blockIdx.x * C + tid) * 3;
      p[0] = n; p[1] = mean; p[2] = m2;
    }
  }
}

// dW[t][ci][co] = sum_q x[2q + off_t][ci] * dy[q][co]: M = co, N = (t,ci), K = output pixels.  dy goes from global memory
// straight into the MFMA A operand (128 B per pixel), x comes from the LDS patch.
template <bool FUSED>
__global__ __launch_bounds__(256, 4) void img_enc_wgrad_kernel(const EncArgs a) {
  kernarg_warm<sizeof(EncArgs)>();
  __shared__ __attribute__((aligned(16))) float sX[ENP * 4];
  __shared__ __attribute__((aligned(16))) float sR[4 * 32 * 32];
  __shared__ float sB[4 * C];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rX = rsrc(a.X, (long)a.B * a.H * a.W * 4 * NO * 4);
  const __amdgpu_buffer_rsrc_t rG = rsrc(a.dY, (long)a.B * a.H * a.W * C * 4);
  // lane j = li = 3t + ci reads x[2q + off_t][ci]
  const int t_l = li / 3, ci_l = li - 3 * t_l;
  const bool jvalid = li < NJ;
  const int xbase = ((tap_dy9(a.tdy, t_l) + 1) * EPW + tap_dy9(a.tdx, t_l) + 1 + 2 * lh) * 4 + ci_l;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;

  constexpr bool fused = FUSED;
  const __amdgpu_buffer_rsrc_t rY = rsrc(a.dy_y, fused ? (long)a.B * a.H * a.W * C * 4 : 0);
  const float k1 = fused ? a.dy_coef[li] : 0.f, k2 = fused ? a.dy_coef[C + li] : 0.f, k3 = fused ? a.dy_coef[2 * C + li] : 0.f;
  const float ksc = fused ? a.dy_coef[3 * C + li] : 0.f, ksh = fused ? a.dy_coef[4 * C + li] : 0.f;
  const float nslope = a.dy_act == ACT_LRELU ? kLeaky : (a.dy_act == ACT_RELU ? 0.f : 1.f);   // host admits only these
  if (fused && blockIdx.x == 0 && tid < C && a.dgamma != nullptr) {
    a.dgamma[tid] = (a.bn_accumulate ? a.dgamma[tid] : 0.f) + a.dy_coef[5 * C + tid];
    a.dbeta[tid] = (a.bn_accumulate ? a.dbeta[tid] : 0.f) + a.dy_coef[6 * C + tid];
  }
  EncPatch pt;
  float dv[2][16];   // this wave's dy operand: rows ly = wave, wave + 4; 16 pixel pairs each
  auto dy_load = [&](const TileXY& t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned rowoff = (unsigned)(((t.b * a.H + t.y0 + wave + 4 * i) * a.W + t.x0) * C + li) * 4u;
#pragma unroll
      for (int j = 0; j < 16; ++j) dv[i][j] = ld1(rG, rowoff + (unsigned)((2 * j + lh) * C) * 4u);
      if constexpr (FUSED) {   // BatchNorm-backward apply on load: no data gradient follows this layer, g_y never reaches memory
        float yv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) yv[j] = ld1(rY, rowoff + (unsigned)((2 * j + lh) * C) * 4u);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float g1 = (yv[j] * ksc + ksh) > 0.f ? dv[i][j] : dv[i][j] * nslope;
          dv[i][j] = k1 * g1 + k2 * yv[j] + k3;
        }
      }
    }
  };
  int tile = blockIdx.x;
  if (tile < a.ntiles) {
    const TileXY c0 = tile_xy_hw(a.tiles_y, a.tiles_x, tile);
    enc_patch_load(a, rX, c0, pt);
  }
  for (; tile < a.ntiles; tile += gridDim.x) {
    const TileXY cur = tile_xy_hw(a.tiles_y, a.tiles_x, tile);
    enc_patch_store(pt, sX);
    dy_load(cur);
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < a.ntiles) enc_patch_load(a, rX, tile_xy_hw(a.tiles_y, a.tiles_x, next), pt);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ly = wave + 4 * i;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float xv = sX[xbase + ((2 * ly) * EPW + 4 * j) * 4];
        xv = jvalid ? xv : 0.f;
        bsum += dv[i][j];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[i][j], xv, acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) sR[(wave * 32 + 8 * (r >> 2) + 4 * lh + (r & 3)) * 32 + li] = acc[r];
  bsum += __shfl_xor(bsum, 32, 64);
  if (lh == 0) sB[wave * C + li] = bsum;
  __syncthreads();
  for (int e = tid; e < NJ * C; e += 256) {
    const int j = e / C, co = e - j * C;
    const float v = ((sR[(0 * 32 + co) * 32 + j] + sR[(1 * 32 + co) * 32 + j]) + sR[(2 * 32 + co) * 32 + j]) + sR[(3 * 32 + co) * 32 + j];
    const int t = j / 3, ci = j - 3 * t;
    a.out[((long)blockIdx.x * NJ + tap_dy9(a.twt, t) * NO + ci) * C + co] = v;
  }
  if (a.pbias != nullptr && tid < C) a.pbias[(long)blockIdx.x * C + tid] = ((sB[tid] + sB[C + tid]) + sB[2 * C + tid]) + sB[3 * C + tid];
}

bool taps_ok(const ConvGeom& g) {
  if (g.ncls != 1 || g.is != 1 || g.os != 1 || g.ntaps[0] != NT) return false;
  if (g.gH != g.sH || g.gW != g.sW || g.sH % TH != 0 || g.sW % TW != 0) return false;
  for (int t = 0; t < NT; ++t) {
    const Tap& tp = g.taps[0][t];
    if (tp.dy < -1 || tp.dy > 1 || tp.dx < -1 || tp.dx > 1 || tp.wtap < 0 || tp.wtap >= NT) return false;
  }
  return (long)g.B * g.sH * g.sW * C < (1L << 29);
}

void fill(ImgArgs& a, const ConvGeom& g) {
  a.B = g.B; a.H = g.sH; a.W = g.sW;
  a.tiles_y = g.sH / TH; a.tiles_x = g.sW / TW;
  a.ntiles = g.B * a.tiles_y * a.tiles_x;
  for (int t = 0; t < NT; ++t) {
    a.tdy[t] = g.taps[0][t].dy; a.tdx[t] = g.taps[0][t].dx; a.twt[t] = g.taps[0][t].wtap;
  }
}

}  // namespace

// forward / wgrad geometry (build_geom kind 0): 32 -> 3 channels
bool img_conv_supported(const ConvGeom& g) { return packed_weights(g) && g.wT == 0 && g.gC == C && g.sC == NO && g.wCi == C && g.wCo == NO && taps_ok(g); }
// dgrad geometry (kind 2): gathers the 3-channel gradient, scatters 32 channels
bool img_dgrad_supported(const ConvGeom& g) { return packed_weights(g) && g.wT == 1 && g.gC == NO && g.sC == C && g.wCi == C && g.wCo == NO && taps_ok(g); }

// encoder.0 geometry (kind 0, stride 2): gathers 3 channels with is = 2, scatters 32
bool img_enc_supported(const ConvGeom& g) {
  if (!packed_weights(g) || g.wT != 0 || g.gC != NO || g.sC != C || g.wCi != NO || g.wCo != C) return false;
  if (g.ncls != 1 || g.is != 2 || g.os != 1 || g.ntaps[0] != NT) return false;
  if (g.gH != 2 * g.sH || g.gW != 2 * g.sW || g.sH % TH != 0 || g.sW % TW != 0) return false;
  for (int t = 0; t < NT; ++t) {
    const Tap& tp = g.taps[0][t];
    if (tp.dy < -1 || tp.dy > 1 || tp.dx < -1 || tp.dx > 1 || tp.wtap < 0 || tp.wtap >= NT) return false;
  }
  return (long)g.B * g.sH * g.sW * C < (1L << 29);
}

constexpr int kImgEncWgs = 1024;
int img_enc_rows(const ConvGeom& g) {
  const int nt = g.B * (g.sH / TH) * (g.sW / TW);
  return nt < kImgEncWgs ? nt : kImgEncWgs;
}

static void fill_enc(EncArgs& a, const ConvGeom& g) {
  a.B = g.B; a.H = g.sH; a.W = g.sW;
  a.tiles_y = g.sH / TH; a.tiles_x = g.sW / TW;
  a.ntiles = g.B * a.tiles_y * a.tiles_x;
  for (int t = 0; t < NT; ++t) {
    a.tdy[t] = g.taps[0][t].dy; a.tdx[t] = g.taps[0][t].dx; a.twt[t] = g.taps[0][t].wtap;
  }
}

int launch_img_enc_forward(const ConvGeom& g, const float* X, const float* W, const float* bias, float* S, int act,
                           float* bn_part, hipStream_t st) {
  EncArgs a{};
  fill_enc(a, g);
  a.X = X; a.Wt = W; a.bias = bias; a.out = S; a.act = act; a.bn_part = bn_part;
  ProfScope ps("img_enc_fwd_kernel", st, 2.0 * a.ntiles * TH * TW * NT * C * NO, 4.0 * a.ntiles * TH * TW * (C + 4 * NO));
  hipLaunchKernelGGL(img_enc_fwd_kernel, dim3(img_enc_rows(g)), dim3(256), 0, st, a);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

// partial slabs [parts][27][32] (+ bias partials [parts][32]) into ws; the caller reduces them
int launch_img_enc_wgrad(const ConvGeom& g, const float* X, const float* dY, float* ws, float** part_out, float** pbias_out,
                         int* nparts, bool want_bias, hipStream_t st, const DyXform* dyx) {
  EncArgs a{};
  fill_enc(a, g);
  a.X = X; a.dY = dY;
  if (dyx != nullptr && dyx->y != nullptr) {
    if (dyx->gy_out != nullptr || (dyx->act != ACT_NONE && dyx->act != ACT_RELU && dyx->act != ACT_LRELU)) return kErrBadArg;
    if ((dyx->dgamma != nullptr) != (dyx->dbeta != nullptr)) return kErrBadArg;
    a.dy_y = dyx->y; a.dy_coef = dyx->coef; a.dy_act = dyx->act;
    a.dgamma = dyx->dgamma; a.dbeta = dyx->dbeta; a.bn_accumulate = dyx->bn_accumulate;
  }
  const int nwg = a.ntiles < 512 ? a.ntiles : 512;
  a.out = ws;
  a.pbias = want_bias ? ws + (size_t)nwg * NJ * C : nullptr;
  ProfScope ps("img_enc_wgrad_kernel", st, 2.0 * a.ntiles * TH * TW * NT * C * NO, 4.0 * a.ntiles * TH * TW * (C + 4 * NO));
  if (a.dy_y != nullptr) hipLaunchKernelGGL(img_enc_wgrad_kernel<true>, dim3(nwg), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(img_enc_wgrad_kernel<false>, dim3(nwg), dim3(256), 0, st, a);
  CTVAE_LAUNCH_CHECK();
  *part_out = a.out;
  *pbias_out = a.pbias;
  *nparts = nwg;
  return 0;
}

// persistent grids (diagnostic overrides: CTVAE_IMG_FWD_WGS / CTVAE_IMG_WGRAD_WGS / CTVAE_IMG_DGRAD_WGS)
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static const int kImgFwdWgs = env_int("CTVAE_IMG_FWD_WGS", 768), kImgWgradWgs = env_int("CTVAE_IMG_WGRAD_WGS", 512),
                 kImgDgradWgs = env_int("CTVAE_IMG_DGRAD_WGS", 1024);

int img_wgrad_parts(const ConvGeom& g) {
  const int nt = g.B * (g.sH / TH) * (g.sW / TW);
  return nt < kImgWgradWgs ? nt : kImgWgradWgs;
}
int img_dgrad_rows(const ConvGeom& g) {
  const int nt = g.B * (g.sH / TH) * (g.sW / TW);
  return nt < kImgDgradWgs ? nt : kImgDgradWgs;
}

int launch_img_forward(const ConvGeom& g, const float* X, const float* W, const float* bias, float* S, int act,
                       const InXform* xf, hipStream_t st) {
  ImgArgs a{};
  fill(a, g);
  a.X = X; a.Wt = W; a.bias = bias; a.out = S; a.act = act;
  if (xf != nullptr && xf->scale != nullptr) { a.scale = xf->scale; a.shift = xf->shift; a.in_act = xf->act; }
  static const int v2 = env_int("CTVAE_IMG_FWD2", 1), wgs2 = env_int("CTVAE_IMG_FWD2_WGS", 1024);   // diagnostic
  ProfScope ps("img_fwd_kernel", st, 2.0 * a.ntiles * TH * TW * NT * C * NO, 4.0 * a.ntiles * TH * TW * (C + NO));
  if (v2) {   // A operand straight from global memory, four workgroups per CU
    const int nwg = a.ntiles < wgs2 ? a.ntiles : wgs2;
    hipLaunchKernelGGL(img_fwd2_kernel, dim3(nwg), dim3(256), 0, st, a);
    CTVAE_LAUNCH_CHECK();
    return 0;
  }
  const int nwg = a.ntiles < kImgFwdWgs ? a.ntiles : kImgFwdWgs;
  hipLaunchKernelGGL(img_fwd_kernel, dim3(nwg), dim3(256), 0, st, a);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

// partial slabs [parts][9*32][3] (+ bias partials [parts][3]) into ws; the caller reduces them
int launch_img_wgrad(const ConvGeom& g, const float* X, const float* dY, float* ws, float** part_out, float** pbias_out,
                     int* nparts, bool want_bias, const InXform* xf, hipStream_t st) {
  ImgArgs a{};
  fill(a, g);
  a.X = X; a.dY = dY;
  if (xf != nullptr && xf->scale != nullptr) { a.scale = xf->scale; a.shift = xf->shift; a.in_act = xf->act; }
  const int nwg = img_wgrad_parts(g);
  a.out = ws;
  a.pbias = want_bias ? ws + (size_t)nwg * NT * C * NO : nullptr;
  ProfScope ps("img_wgrad_kernel", st, 2.0 * a.ntiles * TH * TW * NT * C * NO, 4.0 * a.ntiles * TH * TW * (C + NO));
  hipLaunchKernelGGL(img_wgrad_kernel, dim3(nwg), dim3(256), 0, st, a);
  CTVAE_LAUNCH_CHECK();
  *part_out = a.out;
  *pbias_out = a.pbias;
  *nparts = nwg;
  return 0;
}

// img_dgrad (with the BatchNorm-backward sums of bnb) + img_wgrad of the SAME layer in one launch: x of the weight gradient
// is a = act(BN(bnb->y)).  Slabs [parts][9*32][3] (+ bias partials [parts][3]) into ws, parts = img_dgrad_rows(g).
int launch_img_backward_fused(const ConvGeom& g, const float* dY, const float* W, float* dX, const BnBwdFuse* bnb, float* ws,
                              float** part_out, float** pbias_out, int* nparts, bool want_bias, hipStream_t st) {
  if (bnb->act == ACT_TANH) return kErrBadArg;   // the kernel applies the BatchNorm's activation as max-free selects (identity / LeakyReLU / ReLU)
  ImgArgs a{};
  fill(a, g);
  a.dY = dY; a.Wt = W; a.out = dX;
  a.bn_y = bnb->y; a.bn_mean = bnb->mean; a.bn_invstd = bnb->invstd; a.bn_gamma = bnb->gamma; a.bn_beta = bnb->beta;
  a.bn_act = bnb->act; a.bn_part = bnb->part;
  const int nwg = img_dgrad_rows(g);
  float* part = ws;
  float* pb = want_bias ? ws + (size_t)nwg * NT * C * NO : nullptr;
  ProfScope ps("img_bwd_fused_kernel", st, 4.0 * a.ntiles * TH * TW * NT * C * NO, 4.0 * a.ntiles * TH * TW * (C * 2 + NO));
  hipLaunchKernelGGL(img_bwd_fused_kernel, dim3(nwg), dim3(256), 0, st, a, part, pb);
  CTVAE_LAUNCH_CHECK();
  *part_out = part;
  *pbias_out = pb;
  *nparts = nwg;
  return 0;
}
size_t img_backward_fused_ws_floats(const ConvGeom& g) { return (size_t)img_dgrad_rows(g) * (NT * C * NO + NO); }

int launch_img_dgrad(const ConvGeom& g, const float* dY, const float* W, float* dX, const BnBwdFuse* bnb, hipStream_t st) {
  ImgArgs a{};
  fill(a, g);
  a.dY = dY; a.Wt = W; a.out = dX;
  if (bnb != nullptr && bnb->part != nullptr) {
    a.bn_y = bnb->y; a.bn_mean = bnb->mean; a.bn_invstd = bnb->invstd; a.bn_gamma = bnb->gamma; a.bn_beta = bnb->beta;
    a.bn_act = bnb->act; a.bn_part = bnb->part;
  }
  const int nwg = img_dgrad_rows(g);
  ProfScope ps("img_dgrad_kernel", st, 2.0 * a.ntiles * TH * TW * NT * C * NO,
               4.0 * a.ntiles * TH * TW * (C * (a.bn_part != nullptr ? 2 : 1) + NO));
  hipLaunchKernelGGL(img_dgrad_kernel, dim3(nwg), dim3(256), 0, st, a);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
