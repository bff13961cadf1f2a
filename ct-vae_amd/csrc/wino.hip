// Winograd F(2x2, 3x3) for the 3x3 / stride-1 / pad-1 convolutions of the residual stacks (mcq_vae.py:57-70 via
// vq_vae.ResidualLayer, 13 of them per MCQ-VAE / CT-MCQ-VAE step, forward + data gradient):
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 multiplies per 4 outputs and channel pair instead of 36: the sixteen "frequency" GEMMs
//     M[f][tile][n] = sum_c V[f][tile][c] * U[f][c][n]
// together are 2.25x less MFMA work than the direct 9-tap GEMM.  On gfx950 the f32 MFMA pipe is the bound of these
// layers, so the saving is real time.  fp32 Winograd F(2,3) adds a relative error of a few 1e-7 per
// transform stage -- far inside the 1e-4 parity bound (tests/test_ops_gpu.py::test_winograd_*).
//
// One kernel does everything between the NHWC input and the NHWC output:
//   workgroup = 64 tiles x 64 output channels, 4 waves as 2 (tiles) x 2 (channels); a wave keeps all 16 frequency
//   accumulators of its 32 x 32 block in registers (16 x f32x16 = 256 accumulator registers, one wave per SIMD);
//   per chunk of 8 input channels: raw pixels of the workgroup's tile block (+1 halo, every pixel read once) -> LDS,
//   B^T d B per (tile, channel pair) LDS -> registers -> LDS as V[f][tile][8], pre-transformed filters U[f][n][8]
//   (wino_weight_kernel, a few MB per layer and step) -> LDS, then 16 x 4 MFMA 32x32x2 per wave;
//   epilogue: A^T M A is LANE-LOCAL (a lane holds the same (tile, n) element of all 16 accumulators), then bias /
//   activation and 128-byte stores.
// The data gradient of the same layer is the same computation with the filter taps mirrored and Ci/Co swapped: only
// the weight transform differs (tap table + transposed read).
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

namespace {

constexpr int MT = 64;        // tiles per workgroup
constexpr int NT = 64;        // output channels per workgroup
constexpr int KC8 = 8;        // input channels per chunk
constexpr int LDV = 12;       // padded row of V / U in LDS (floats): conflict-free ds_read_b128 over 16 consecutive rows
constexpr int RS = 8;         // raw pixel row in LDS (floats)
constexpr int NPMAX = 640;    // raw pixels per workgroup block (incl. halo)
constexpr int RAW_ITEMS = (NPMAX * 2 + 255) / 256;   // float4 loads per thread and chunk
static_assert(RAW_ITEMS * 128 <= NPMAX, "raw staging writes every item unconditionally");

constexpr unsigned kOOBw = 0x80000000u;

__device__ __forceinline__ f32x4 wld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
// voff: per-lane byte offset (kOOBw = outside -> 0), soff: wave-uniform byte offset added by the instruction (SGPR operand:
// the per-chunk part of an address costs no VALU instruction)
__device__ __forceinline__ f32x4 wld4s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wrsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

#ifdef CTVAE_PHASE_TIMING
// diagnostic build only (tools/wino_phase_probe.py): shader-clock cycles per loop phase, summed over the chunks, per wave
__device__ long long g_wino_phase[8 * 4096];
#define WPH(i) do { const long long now_ = clock64(); if (lane == 0) tph[i] += now_ - tlast; tlast = clock64(); } while (0)
extern "C" int ctvae_debug_wino_phase_read(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino_phase), (size_t)n * 8);
}
#else
#define WPH(i) do {} while (0)
#endif

// a - b on both halves in one instruction (the compiler packs f32x2 additions but splits subtractions in two)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

struct WinoArgs {
  const float* X;      // [B][H][W][K]
  const float* Ut;     // [16][K/8][N][8]
  const float* bias;   // [N] or null
  const float* add;    // [B][H][W][N] or null: added before the activation (residual skip / its gradient)
  float* Y;            // [B][H][W][N]
  int B, H, W, K, N;
  int bh, bw, nb;      // workgroup block: nb images x (bh x bw) tiles, nb*bh*bw == 64
  int by_n, bx_n;      // blocks per image (TH/bh, TW/bw)
  int act;
};

// ---- filter transform ------------------------------------------------------------------------------------------
// W packed [9][wCi][wCo].  Effective filter g[ky][kx] (k = gathered channel, n = produced channel) =
//   wT == 0: W[tap[ky][kx]][k][n]     wT == 1: W[tap[ky][kx]][n][k]
// Ut[f = 4i+j][k/8][n][k%8] = (G g G^T)[i][j].  thread = (n fastest, k).
struct WTaps { int t[9]; };

__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ Wp, float* __restrict__ Ut, int K, int N,
                                                          int wCi, int wCo, int wT, WTaps taps) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)K * N) return;
  int k, n;
  if (wT == 0) { n = (int)(e % N); k = (int)(e / N); }     // reads contiguous along n
  else { k = (int)(e % K); n = (int)(e / K); }             // W[t][n][k]: reads contiguous along k
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 9; ++a) {
    const long idx = wT == 0 ? ((long)taps.t[a] * wCi + k) * wCo + n : ((long)taps.t[a] * wCi + n) * wCo + k;
    g[a / 3][a % 3] = Wp[idx];
  }
  float t[4][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    t[0][j] = g[0][j];
    t[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
    t[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
    t[3][j] = g[2][j];
  }
  float* dst = Ut + (((long)(k >> 3)) * N + n) * 8 + (k & 7);
  const long fs = (long)(K >> 3) * N * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float u0 = t[i][0], u1 = 0.5f * (t[i][0] + t[i][1] + t[i][2]), u2 = 0.5f * (t[i][0] - t[i][1] + t[i][2]), u3 = t[i][2];
    dst[(long)(4 * i + 0) * fs] = u0;
    dst[(long)(4 * i + 1) * fs] = u1;
    dst[(long)(4 * i + 2) * fs] = u2;
    dst[(long)(4 * i + 3) * fs] = u3;
  }
}

// Forward launch of a layer whose data gradient will also run Winograd: both filter sets in one launch.
//   Uf[f][ci/8][co][ci%8] = (G g G^T)[f]                   (operand of the forward pass,   K = Ci, N = Co)
//   Ub[f][co/8][ci][co%8] = (G g~ G^T)[f], g~ = g mirrored  (operand of the data gradient,  K = Co, N = Ci)
// Mirroring the 3x3 taps only permutes the Winograd rows/columns (0 <-> 3), so Ub[(i,j)] = Uf[(p(i),p(j))], p = (3,1,2,0).
// workgroup = a 32 ci x 32 co tile of W ([9][Ci][Co], tap = ky*3+kx): 128-byte coalesced tap loads, the 16 transformed
// values of every (ci,co) go through LDS, and both layouts leave as 1-KB contiguous runs of 16-byte stores.
constexpr int W2_LD = 33;
__device__ __forceinline__ void wino_weight2_body(const float* __restrict__ Wp, float* __restrict__ Uf, float* __restrict__ Ub,
                                                  int Ci, int Co, int bx, int by, float* sW2) {
  const int tid = threadIdx.x;
  const int ci0 = bx * 32, co0 = by * 32;
  {
    const int co_l = tid & 31;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci_l = (tid >> 5) + 8 * r;
      float g[3][3];
#pragma unroll
      for (int a = 0; a < 9; ++a) g[a / 3][a % 3] = Wp[((long)a * Ci + ci0 + ci_l) * Co + co0 + co_l];
      float t[4][3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        t[0][j] = g[0][j];
        t[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        t[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        t[3][j] = g[2][j];
      }
      float* dst = sW2 + ci_l * W2_LD + co_l;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dst[(4 * i + 0) * 32 * W2_LD] = t[i][0];
        dst[(4 * i + 1) * 32 * W2_LD] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
        dst[(4 * i + 2) * 32 * W2_LD] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
        dst[(4 * i + 3) * 32 * W2_LD] = t[i][2];
      }
    }
  }
  __syncthreads();
  const long fs = (long)Ci * Co;
  const int o = (tid >> 6) & 3, x_l = (tid >> 1) & 31, half = tid & 1;
#pragma unroll 4
  for (int f = 0; f < 16; ++f) {
    const float* src = sW2 + f * 32 * W2_LD;
    // forward set: octet o of ci, row co = x_l, elements ci%8 = 4*half .. 4*half+3
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = src[(8 * o + 4 * half + k) * W2_LD + x_l];
    *reinterpret_cast<f32x4*>(Uf + f * fs + ((long)(ci0 / 8 + o) * Co + co0 + x_l) * 8 + 4 * half) = v;
    // data-gradient set: octet o of co, row ci = x_l, elements co%8 = 4*half .. 4*half+3, mirrored frequency
    const int i = f >> 2, j = f & 3;
    const int fo = ((i == 0 ? 3 : (i == 3 ? 0 : i)) << 2) | (j == 0 ? 3 : (j == 3 ? 0 : j));
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = src[x_l * W2_LD + 8 * o + 4 * half + k];
    *reinterpret_cast<f32x4*>(Ub + fo * fs + ((long)(co0 / 8 + o) * Ci + ci0 + x_l) * 8 + 4 * half) = v;
  }
}

__global__ __launch_bounds__(256) void wino_weight2_kernel(const float* __restrict__ Wp, float* __restrict__ Uf,
                                                           float* __restrict__ Ub, int Ci, int Co) {
  extern __shared__ __attribute__((aligned(16))) float sW2[];     // [16][32 ci][W2_LD]
  wino_weight2_body(Wp, Uf, Ub, Ci, Co, blockIdx.x, blockIdx.y, sW2);
}

// Every Winograd layer's filter transform of a training step in ONE launch (blockIdx.z = layer): the weights only change
// at the optimizer step, and thirteen 7-us launches of 8 x 8 workgroups each are mostly launch floor.
constexpr int kWinoBatchMax = 32;
struct WinoBatch {
  const float* W[kWinoBatchMax];
  float* Uf[kWinoBatchMax];
  float* Ub[kWinoBatchMax];
  int Ci[kWinoBatchMax], Co[kWinoBatchMax];
};
__global__ __launch_bounds__(256) void wino_weight2_batch_kernel(const WinoBatch t) {
  extern __shared__ __attribute__((aligned(16))) float sW2[];
  const int l = blockIdx.z;
  if ((int)blockIdx.x * 32 >= t.Ci[l] || (int)blockIdx.y * 32 >= t.Co[l]) return;
  wino_weight2_body(t.W[l], t.Uf[l], t.Ub[l], t.Ci[l], t.Co[l], blockIdx.x, blockIdx.y, sW2);
}

// ---- the GEMM --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino_conv_kernel(WinoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sV = smem;                          // [2][16][MT][8]  (double-buffered, rows swizzled, see a_rd)
  float* sU = sV + 2 * 16 * MT * 8;          // [2][16][NT][8]
  float* sRaw = sU + 2 * 16 * NT * 8;        // [NPMAX][RS]
  int* sOut = reinterpret_cast<int*>(sRaw + NPMAX * RS);   // [MT] output pixel index of the tile's (0,0) output, -1 = none

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int ntiles = a.N / NT;
  // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2); the ntiles
  // workgroups that read the same input block are given ids with the same id % 8, so the block is fetched once per XCD
  int mt, nt;
  {
    const int mtiles = gridDim.x / ntiles;
    if ((mtiles & 7) == 0) {
      const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
      mt = (slot / ntiles) * 8 + xcd;
      nt = slot - (slot / ntiles) * ntiles;
    } else {
      mt = blockIdx.x / ntiles;
      nt = blockIdx.x - mt * ntiles;
    }
  }
  const int PH = 2 * a.bh + 2, PW = 2 * a.bw + 2, NP = a.nb * PH * PW;
  const int K = a.K, N = a.N;

  // block origin
  int b0, y0, x0;
  if (a.nb > 1) { b0 = mt * a.nb; y0 = 0; x0 = 0; }
  else {
    const int per = a.by_n * a.bx_n;
    b0 = mt / per;
    const int r = mt - b0 * per;
    y0 = (r / a.bx_n) * 2 * a.bh;
    x0 = (r - (r / a.bx_n) * a.bx_n) * 2 * a.bw;
  }

  const __amdgpu_buffer_rsrc_t rX = wrsrc(a.X, (long)a.B * a.H * a.W * K * 4);
  const __amdgpu_buffer_rsrc_t rU = wrsrc(a.Ut, (long)16 * K * N * 4);

  // raw-load items: e -> (pixel p = e>>1, half = e&1)
  unsigned roff[RAW_ITEMS];
#pragma unroll
  for (int i = 0; i < RAW_ITEMS; ++i) {
    const int e = tid + 256 * i, p = e >> 1, half = e & 1;
    unsigned off = kOOBw;
    if (p < NP) {
      const int img = p / (PH * PW), r = p - img * (PH * PW), py = r / PW, px = r - py * PW;
      const int b = b0 + img, y = y0 + py - 1, x = x0 + px - 1;
      if (b < a.B && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W)
        off = ((unsigned)((b * a.H + y) * a.W + x) * (unsigned)K + 4u * half) * 4u;
    }
    roff[i] = off;
  }
  // U-load items: e = tid + 256*i -> f = e>>7, r = e&127 (float4 r of the 512-float row block of frequency f)
  const unsigned u_fs = (unsigned)(K >> 3) * (unsigned)N * 8u * 4u;            // bytes between frequencies
  const unsigned u_base = ((unsigned)nt * NT * 8u + 4u * (tid & 127)) * 4u + (unsigned)(tid >> 7) * u_fs;

  // transform role: tile m_t = tid>>2, channel pair q = tid&3
  const int m_t = tid >> 2, q = tid & 3;
  int rp0;
  {
    const int per = a.bh * a.bw;
    const int img = m_t / per, r = m_t - img * per, ty = r / a.bw, tx = r - ty * a.bw;
    rp0 = (img * PH + 2 * ty) * PW + 2 * tx;
    if (q == 0) {
      const int b = b0 + img;
      sOut[m_t] = b < a.B ? (b * a.H + y0 + 2 * ty) * a.W + x0 + 2 * tx : -1;
    }
  }

  f32x16 acc[16];
#pragma unroll
  for (int f = 0; f < 16; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  f32x4 rr[RAW_ITEMS], ru[8];
  const int nchunks = K / KC8;
  // LDS element offsets (floats) inside one [16][64][8] buffer; rows are 8 floats, the two 4-float halves of a row are
  // swapped for rows 8..15 (mod 16) so that ds_read_b128 over 16 consecutive rows touches all 64 banks once
  const int swz_t = (m_t >> 3) & 1;
  const int v_st = m_t * 8 + ((((q >> 1) ^ swz_t)) << 2) + (q & 1) * 2;          // transform store of channels 2q, 2q+1
  const int u_row = (tid & 127) >> 1;
  const int u_st = u_row * 8 + ((((tid & 1) ^ ((u_row >> 3) & 1))) << 2);        // U store of float4 (tid&127) of a frequency
  const int ma = wm * 32 + li, nbr = wn * 32 + li;
  const int a_rd = ma * 8 + ((lh ^ ((ma >> 3) & 1)) << 2);
  const int b_rd = nbr * 8 + ((lh ^ ((nbr >> 3) & 1)) << 2);
  constexpr int FB = MT * 8;           // floats per frequency
  constexpr int BUF = 16 * FB;         // floats per buffer

  // One chunk = 16 steps of (4 MFMA of frequency f) + a slice of the data movement for the NEXT chunks.  Everything that
  // moves data (13 LDS stores of the prefetched raw/U registers, 13 global loads two chunks ahead, 16 patch reads, the
  // transform and its 16 stores) is issued between MFMA groups: measured per chunk, the same instructions issued as
  // separate phases cost 855 + 725 cycles (LDS store and texture-address bandwidth shared by the 4 waves) next to 4096
  // cycles of MFMA, with one wave per SIMD and nothing to overlap them with.
  auto st_raw = [&](int i) {      // items beyond NP hold zeros and land in the unused tail of sRaw (RAW_ITEMS*128 <= NPMAX)
    const int e = tid + 256 * i;
    *reinterpret_cast<f32x4*>(sRaw + (e >> 1) * RS + 4 * (e & 1)) = rr[i];
  };
  auto st_u = [&](float* su, int i) { *reinterpret_cast<f32x4*>(su + (2 * i + (tid >> 7)) * FB + u_st) = ru[i]; };
  unsigned uoff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) uoff[i] = u_base + (unsigned)(2 * i) * u_fs;
  auto ld_raw = [&](int c, int i) { rr[i] = wld4s(rX, roff[i], (unsigned)c * 32u); };
  auto ld_u = [&](int c, int i) { ru[i] = wld4s(rU, uoff[i], (unsigned)c * (unsigned)N * 32u); };
  f32x2 d[4][4], t[4][4];
  auto rd_patch = [&](int i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const f32x2*>(sRaw + (rp0 + i * PW + j) * RS + 2 * q);
  };
  auto tf_rows = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = pk_sub(d[0][j], d[2][j]);
      t[1][j] = d[1][j] + d[2][j];
      t[2][j] = pk_sub(d[2][j], d[1][j]);
      t[3][j] = pk_sub(d[1][j], d[3][j]);
    }
  };
  auto tf_store = [&](float* sv, int i) {
    float* dst = sv + (4 * i) * FB + v_st;
    *reinterpret_cast<f32x2*>(dst) = pk_sub(t[i][0], t[i][2]);
    *reinterpret_cast<f32x2*>(dst + FB) = t[i][1] + t[i][2];
    *reinterpret_cast<f32x2*>(dst + 2 * FB) = pk_sub(t[i][2], t[i][1]);
    *reinterpret_cast<f32x2*>(dst + 3 * FB) = pk_sub(t[i][1], t[i][3]);
  };
  f32x4 fa, fb, na, nb;
  auto rd_frag = [&](const float* sv, const float* su, int f) {
    na = *reinterpret_cast<const f32x4*>(sv + f * FB + a_rd);
    nb = *reinterpret_cast<const f32x4*>(su + f * FB + b_rd);
  };
#define WINO_MFMA4(f)                                                                                   \
  _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                      \
      acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s_], fb[s_], acc[f], 0, 0, 0)
#define WINO_FENCE() __builtin_amdgcn_sched_barrier(0)

  // prologue: chunk 0 -> buffer 0, chunk 1 -> registers
#pragma unroll
  for (int i = 0; i < RAW_ITEMS; ++i) ld_raw(0, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) ld_u(0, i);
#pragma unroll
  for (int i = 0; i < RAW_ITEMS; ++i) st_raw(i);
#pragma unroll
  for (int i = 0; i < 8; ++i) st_u(sU, i);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RAW_ITEMS; ++i) ld_raw(1, i);       // (chunk index beyond K only reads unused data: see below)
#pragma unroll
  for (int i = 0; i < 8; ++i) ld_u(1, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) rd_patch(i);
  tf_rows();
#pragma unroll
  for (int i = 0; i < 4; ++i) tf_store(sV, i);
  __syncthreads();

  // main loop: all chunks but the last.  Loads run two chunks ahead; past the last chunk they fetch neighbouring
  // (in-bounds or zero-filled) data that is never stored.
#ifdef CTVAE_PHASE_TIMING
  long long cseg_[5] = {0, 0, 0, 0, 0};
  long long ctl_ = clock64();
  const long long cts_ = ctl_;
#define CSEG(i) do { const long long now_ = clock64(); cseg_[i] += now_ - ctl_; ctl_ = now_; } while (0)
#else
#define CSEG(i) do {} while (0)
#endif
  for (int c = 0; c + 1 < nchunks; ++c) {
    const int cur = c & 1;
    const float* svc = sV + cur * BUF;
    const float* suc = sU + cur * BUF;
    float* svn = sV + (cur ^ 1) * BUF;
    float* sun = sU + (cur ^ 1) * BUF;
    rd_frag(svc, suc, 0);
    fa = na; fb = nb;
    WINO_FENCE();
    // steps 0..7: one data-movement instruction (or two) behind each single MFMA, so that an LDS / texture queue that
    // is momentarily full stalls the wave while an MFMA is executing, not between two of them
#define WINO_M1(f, s_) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s_], fb[s_], acc[f], 0, 0, 0); WINO_FENCE()
    rd_frag(svc, suc, 1);
    WINO_M1(0, 0); st_raw(0); WINO_M1(0, 1); st_raw(1); WINO_M1(0, 2); st_u(sun, 0); WINO_M1(0, 3);
    fa = na; fb = nb; rd_frag(svc, suc, 2);
    WINO_M1(1, 0); st_raw(2); WINO_M1(1, 1); st_raw(3); WINO_M1(1, 2); st_u(sun, 1); WINO_M1(1, 3);
    fa = na; fb = nb; rd_frag(svc, suc, 3);
    WINO_M1(2, 0); st_raw(4); st_u(sun, 2); WINO_M1(2, 1); st_u(sun, 3); WINO_M1(2, 2); ld_raw(c + 2, 0); WINO_M1(2, 3); ld_raw(c + 2, 1);
    fa = na; fb = nb; rd_frag(svc, suc, 4);
    WINO_M1(3, 0); st_u(sun, 4); WINO_M1(3, 1); st_u(sun, 5); ld_raw(c + 2, 2); WINO_M1(3, 2); ld_raw(c + 2, 3); WINO_M1(3, 3); ld_u(c + 2, 0);
    fa = na; fb = nb; rd_frag(svc, suc, 5);
    WINO_M1(4, 0); st_u(sun, 6); WINO_M1(4, 1); st_u(sun, 7); ld_raw(c + 2, 4); WINO_M1(4, 2); ld_u(c + 2, 1); WINO_M1(4, 3); ld_u(c + 2, 2);
    fa = na; fb = nb; rd_frag(svc, suc, 6);
    WINO_M1(5, 0); ld_u(c + 2, 3); WINO_M1(5, 1); ld_u(c + 2, 4); WINO_M1(5, 2); ld_u(c + 2, 5); WINO_M1(5, 3);
    fa = na; fb = nb; rd_frag(svc, suc, 7);
    WINO_M1(6, 0); ld_u(c + 2, 6); WINO_M1(6, 1); ld_u(c + 2, 7); WINO_M1(6, 2); WINO_M1(6, 3);
    fa = na; fb = nb; rd_frag(svc, suc, 8);
    WINO_MFMA4(7); fa = na; fb = nb; WINO_FENCE();
#undef WINO_M1
    CSEG(0);
    __syncthreads();                                 // raw pixels of chunk c+1 are visible
    CSEG(1);
    // steps 8..15: patch reads, transform and V stores, again behind single MFMAs
#define WINO_M1(f, s_) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s_], fb[s_], acc[f], 0, 0, 0); WINO_FENCE()
    rd_frag(svc, suc, 9);
    WINO_M1(8, 0); rd_patch(0); WINO_M1(8, 1); rd_patch(1); WINO_M1(8, 2); WINO_M1(8, 3);
    fa = na; fb = nb; rd_frag(svc, suc, 10);
    WINO_M1(9, 0); rd_patch(2); WINO_M1(9, 1); rd_patch(3); WINO_M1(9, 2); WINO_M1(9, 3);
    fa = na; fb = nb; rd_frag(svc, suc, 11);
    WINO_MFMA4(10); fa = na; fb = nb; WINO_FENCE();
    rd_frag(svc, suc, 12); tf_rows();
    WINO_MFMA4(11); fa = na; fb = nb; WINO_FENCE();
    rd_frag(svc, suc, 13);
    WINO_M1(12, 0); tf_store(svn, 0); WINO_M1(12, 1); WINO_M1(12, 2); WINO_M1(12, 3);
    fa = na; fb = nb; rd_frag(svc, suc, 14);
    WINO_M1(13, 0); tf_store(svn, 1); WINO_M1(13, 1); WINO_M1(13, 2); WINO_M1(13, 3);
    fa = na; fb = nb; rd_frag(svc, suc, 15);
    WINO_M1(14, 0); tf_store(svn, 2); WINO_M1(14, 1); WINO_M1(14, 2); WINO_M1(14, 3);
    fa = na; fb = nb;
    WINO_M1(15, 0); tf_store(svn, 3); WINO_M1(15, 1); WINO_M1(15, 2); WINO_M1(15, 3);
#undef WINO_M1
    CSEG(2);
    __syncthreads();                                 // V/U of chunk c+1 complete, chunk c's buffers free
    CSEG(3);
  }
#ifdef CTVAE_PHASE_TIMING
  if (lane == 0 && blockIdx.x < 256) {
    cseg_[4] = clock64() - cts_;
    for (int i = 0; i < 5; ++i) g_wino_phase[(blockIdx.x * 4 + wave) * 8 + i] = cseg_[i];
  }
#endif
  {                                                  // last chunk: MFMA only
    const int cur = (nchunks - 1) & 1;
    const float* svc = sV + cur * BUF;
    const float* suc = sU + cur * BUF;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      fa = *reinterpret_cast<const f32x4*>(svc + f * FB + a_rd);
      fb = *reinterpret_cast<const f32x4*>(suc + f * FB + b_rd);
      WINO_MFMA4(f);
    }
  }
#undef WINO_MFMA4
#undef WINO_FENCE

  // ---- epilogue: A^T M A, lane-local ----
  const int col = nt * NT + wn * 32 + li;
  const float bv = a.bias != nullptr ? a.bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wm * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
    const int op = sOut[row];
    float t0[4], t1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t0[j] = acc[j][r] + acc[4 + j][r] + acc[8 + j][r];
      t1[j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
    }
    const float y00 = t0[0] + t0[1] + t0[2], y01 = t0[1] - t0[2] - t0[3];
    const float y10 = t1[0] + t1[1] + t1[2], y11 = t1[1] - t1[2] - t1[3];
    if (op >= 0) {
      float* dst = a.Y + (long)op * N + col;
      float s00 = bv, s01 = bv, s10 = bv, s11 = bv;
      if (a.add != nullptr) {
        const float* ad = a.add + (long)op * N + col;
        s00 += ad[0]; s01 += ad[N]; s10 += ad[(long)a.W * N]; s11 += ad[(long)a.W * N + N];
      }
      if (a.act == ACT_TANH) {
        dst[0] = act_fwd(y00 + s00, ACT_TANH);
        dst[N] = act_fwd(y01 + s01, ACT_TANH);
        dst[(long)a.W * N] = act_fwd(y10 + s10, ACT_TANH);
        dst[(long)a.W * N + N] = act_fwd(y11 + s11, ACT_TANH);
      } else {     // identity / LeakyReLU / ReLU: one select per value, no branch tree per value (common.hpp act_slope_fwd)
        const float osl = act_slope(a.act);
        dst[0] = act_slope_fwd(y00 + s00, osl);
        dst[N] = act_slope_fwd(y01 + s01, osl);
        dst[(long)a.W * N] = act_slope_fwd(y10 + s10, osl);
        dst[(long)a.W * N + N] = act_slope_fwd(y11 + s11, osl);
      }
    }
  }
}

// ---- the same GEMM for launches with fewer than 256 of the 64 x 64 workgroups ------------------------------------------
// (CT-MCQ-VAE: 128 images per GPU and pass -> 128 workgroups for 256 CUs.)  Workgroup = 64 tiles x 32 output channels,
// twice as many of them; the waves of a workgroup split the FREQUENCIES instead of the channels: wave (wm, fh) holds the
// 8 accumulators f = 4i + 2fh + jj (all four rows i, columns j = 2fh + jj) of a 32 x 32 block.  The row pass of A^T M A
// stays lane-local; the column pass needs both waves' columns, so each wave forms its partial outputs, hands the half
// it does not store through LDS (8 KB per wave) and finishes the other half.  Staging of the input block and the
// transform are those of wino_conv_kernel; the filter chunk is half as large (16 KB).
__global__ __launch_bounds__(256) void wino_conv_fs_kernel(WinoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NTs = 32;
  constexpr int FBv = MT * 8, FBu = NTs * 8;          // floats per frequency
  constexpr int BUFv = 16 * FBv, BUFu = 16 * FBu;     // floats per buffer
  float* sV = smem;                          // [2][16][MT][8]
  float* sU = sV + 2 * BUFv;                 // [2][16][NTs][8]
  float* sRaw = sU + 2 * BUFu;               // [NPMAX][RS]
  int* sOut = reinterpret_cast<int*>(sRaw + NPMAX * RS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, fh = wave & 1, li = lane & 31, lh = lane >> 5;
  const int ntiles = a.N / NTs;
  int mt, nt;                                // XCD-aware order, see wino_conv_kernel
  {
    const int mtiles = gridDim.x / ntiles;
    if ((mtiles & 7) == 0) {
      const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
      mt = (slot / ntiles) * 8 + xcd;
      nt = slot - (slot / ntiles) * ntiles;
    } else {
      mt = blockIdx.x / ntiles;
      nt = blockIdx.x - mt * ntiles;
    }
  }
  const int PH = 2 * a.bh + 2, PW = 2 * a.bw + 2, NP = a.nb * PH * PW;
  const int K = a.K, N = a.N;

  int b0, y0, x0;
  if (a.nb > 1) { b0 = mt * a.nb; y0 = 0; x0 = 0; }
  else {
    const int per = a.by_n * a.bx_n;
    b0 = mt / per;
    const int r = mt - b0 * per;
    y0 = (r / a.bx_n) * 2 * a.bh;
    x0 = (r - (r / a.bx_n) * a.bx_n) * 2 * a.bw;
  }
  const __amdgpu_buffer_rsrc_t rX = wrsrc(a.X, (long)a.B * a.H * a.W * K * 4);
  const __amdgpu_buffer_rsrc_t rU = wrsrc(a.Ut, (long)16 * K * N * 4);

  unsigned roff[RAW_ITEMS];
#pragma unroll
  for (int i = 0; i < RAW_ITEMS; ++i) {
    const int e = tid + 256 * i, p = e >> 1, half = e & 1;
    unsigned off = kOOBw;
    if (p < NP) {
      const int img = p / (PH * PW), r = p - img * (PH * PW), py = r / PW, px = r - py * PW;
      const int b = b0 + img, y = y0 + py - 1, x = x0 + px - 1;
      if (b < a.B && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W)
        off = ((unsigned)((b * a.H + y) * a.W + x) * (unsigned)K + 4u * half) * 4u;
    }
    roff[i] = off;
  }
  // U-load items: e = tid + 256*i (i < 4) -> f = e>>6, float4 r = e&63 of the 256-float row block of frequency f
  const unsigned u_fs = (unsigned)(K >> 3) * (unsigned)N * 8u * 4u;
  const unsigned u_base = ((unsigned)nt * NTs * 8u + 4u * (tid & 63)) * 4u + (unsigned)(tid >> 6) * u_fs;

  const int m_t = tid >> 2, q = tid & 3;
  int rp0;
  {
    const int per = a.bh * a.bw;
    const int img = m_t / per, r = m_t - img * per, ty = r / a.bw, tx = r - ty * a.bw;
    rp0 = (img * PH + 2 * ty) * PW + 2 * tx;
    if (q == 0) {
      const int b = b0 + img;
      sOut[m_t] = b < a.B ? (b * a.H + y0 + 2 * ty) * a.W + x0 + 2 * tx : -1;
    }
  }

  f32x16 acc[8];                                       // a8 = 2i + jj  <->  f = 4i + 2fh + jj
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  f32x4 rr[RAW_ITEMS], ru[4];
  const int nchunks = K / KC8;
  const int swz_t = (m_t >> 3) & 1;
  const int v_st = m_t * 8 + ((((q >> 1) ^ swz_t)) << 2) + (q & 1) * 2;
  const int u_row = (tid & 63) >> 1;
  const int u_st = u_row * 8 + ((((tid & 1) ^ ((u_row >> 3) & 1))) << 2);
  const int ma = wm * 32 + li;
  const int a_rd = ma * 8 + ((lh ^ ((ma >> 3) & 1)) << 2) + 2 * fh * FBv;      // + this wave's column offset
  const int b_rd = li * 8 + ((lh ^ ((li >> 3) & 1)) << 2) + 2 * fh * FBu;

  auto st_raw = [&](int i) {
    const int e = tid + 256 * i;
    *reinterpret_cast<f32x4*>(sRaw + (e >> 1) * RS + 4 * (e & 1)) = rr[i];
  };
  auto st_u = [&](float* su, int i) { *reinterpret_cast<f32x4*>(su + (4 * i + (tid >> 6)) * FBu + u_st) = ru[i]; };
  unsigned uoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) uoff[i] = u_base + (unsigned)(4 * i) * u_fs;
  auto ld_raw = [&](int c, int i) { rr[i] = wld4s(rX, roff[i], (unsigned)c * 32u); };
  auto ld_u = [&](int c, int i) { ru[i] = wld4s(rU, uoff[i], (unsigned)c * (unsigned)N * 32u); };
  f32x2 d[4][4], t[4][4];
  auto rd_patch = [&](int i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const f32x2*>(sRaw + (rp0 + i * PW + j) * RS + 2 * q);
  };
  auto tf_rows = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = pk_sub(d[0][j], d[2][j]);
      t[1][j] = d[1][j] + d[2][j];
      t[2][j] = pk_sub(d[2][j], d[1][j]);
      t[3][j] = pk_sub(d[1][j], d[3][j]);
    }
  };
  auto tf_store = [&](float* sv, int i) {
    float* dst = sv + (4 * i) * FBv + v_st;
    *reinterpret_cast<f32x2*>(dst) = pk_sub(t[i][0], t[i][2]);
    *reinterpret_cast<f32x2*>(dst + FBv) = t[i][1] + t[i][2];
    *reinterpret_cast<f32x2*>(dst + 2 * FBv) = pk_sub(t[i][2], t[i][1]);
    *reinterpret_cast<f32x2*>(dst + 3 * FBv) = pk_sub(t[i][1], t[i][3]);
  };
  f32x4 fa, fb, na, nb;
  auto rd_frag = [&](const float* sv, const float* su, int a8) {     // frequency 4*(a8>>1) + 2fh + (a8&1)
    const int fl = 4 * (a8 >> 1) + (a8 & 1);
    na = *reinterpret_cast<const f32x4*>(sv + fl * FBv + a_rd);
    nb = *reinterpret_cast<const f32x4*>(su + fl * FBu + b_rd);
  };
#define WINO_MFMA4(f)                                                                                   \
  _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                      \
      acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s_], fb[s_], acc[f], 0, 0, 0)
#define WINO_FENCE() __builtin_amdgcn_sched_barrier(0)

#pragma unroll
  for (int i = 0; i < RAW_ITEMS; ++i) ld_raw(0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) ld_u(0, i);
#pragma unroll
  for (int i = 0; i < RAW_ITEMS; ++i) st_raw(i);
#pragma unroll
  for (int i = 0; i < 4; ++i) st_u(sU, i);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RAW_ITEMS; ++i) ld_raw(1, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) ld_u(1, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) rd_patch(i);
  tf_rows();
#pragma unroll
  for (int i = 0; i < 4; ++i) tf_store(sV, i);
  __syncthreads();

  for (int c = 0; c + 1 < nchunks; ++c) {
    const int cur = c & 1;
    const float* svc = sV + cur * BUFv;
    const float* suc = sU + cur * BUFu;
    float* svn = sV + (cur ^ 1) * BUFv;
    float* sun = sU + (cur ^ 1) * BUFu;
    rd_frag(svc, suc, 0);
    fa = na; fb = nb;
    WINO_FENCE();
#define WINO_M1(f, s_) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s_], fb[s_], acc[f], 0, 0, 0); WINO_FENCE()
    rd_frag(svc, suc, 1);
    WINO_M1(0, 0); st_raw(0); WINO_M1(0, 1); st_raw(1); WINO_M1(0, 2); st_raw(2); WINO_M1(0, 3); st_u(sun, 0);
    fa = na; fb = nb; rd_frag(svc, suc, 2);
    WINO_M1(1, 0); st_raw(3); st_raw(4); WINO_M1(1, 1); st_u(sun, 1); st_u(sun, 2); WINO_M1(1, 2); st_u(sun, 3); ld_raw(c + 2, 0);
    WINO_M1(1, 3); ld_raw(c + 2, 1); ld_raw(c + 2, 2);
    fa = na; fb = nb; rd_frag(svc, suc, 3);
    WINO_M1(2, 0); ld_raw(c + 2, 3); WINO_M1(2, 1); ld_raw(c + 2, 4); WINO_M1(2, 2); ld_u(c + 2, 0); WINO_M1(2, 3); ld_u(c + 2, 1);
    fa = na; fb = nb; rd_frag(svc, suc, 4);
    WINO_M1(3, 0); ld_u(c + 2, 2); WINO_M1(3, 1); ld_u(c + 2, 3); WINO_M1(3, 2); WINO_M1(3, 3);
    fa = na; fb = nb;
    __syncthreads();                                 // raw pixels of chunk c+1 are visible
    rd_frag(svc, suc, 5);
    WINO_M1(4, 0); rd_patch(0); WINO_M1(4, 1); rd_patch(1); WINO_M1(4, 2); rd_patch(2); WINO_M1(4, 3); rd_patch(3);
    fa = na; fb = nb; rd_frag(svc, suc, 6); tf_rows();
    WINO_M1(5, 0); tf_store(svn, 0); WINO_M1(5, 1); WINO_M1(5, 2); WINO_M1(5, 3);
    fa = na; fb = nb; rd_frag(svc, suc, 7);
    WINO_M1(6, 0); tf_store(svn, 1); WINO_M1(6, 1); WINO_M1(6, 2); tf_store(svn, 2); WINO_M1(6, 3);
    fa = na; fb = nb;
    WINO_M1(7, 0); tf_store(svn, 3); WINO_M1(7, 1); WINO_M1(7, 2); WINO_M1(7, 3);
#undef WINO_M1
    __syncthreads();                                 // V/U of chunk c+1 complete, chunk c's buffers free
  }
  {
    const int cur = (nchunks - 1) & 1;
    const float* svc = sV + cur * BUFv;
    const float* suc = sU + cur * BUFu;
#pragma unroll
    for (int a8 = 0; a8 < 8; ++a8) {
      rd_frag(svc, suc, a8);
      fa = na; fb = nb;
      WINO_MFMA4(a8);
    }
  }
#undef WINO_MFMA4
#undef WINO_FENCE
  __syncthreads();                                   // every wave is done with sV: it becomes the exchange area

  // ---- epilogue ----
  float* sX = sV;                                    // [wave][8 rows][4 partials][64 lanes]
  float keep[8][4];                                  // partial outputs of the rows this wave stores: r in [8fh, 8fh+8)
#pragma unroll
  for (int rl = 0; rl < 8; ++rl)
#pragma unroll
    for (int v = 0; v < 4; ++v) keep[rl][v] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float t0[2], t1[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      t0[jj] = acc[jj][r] + acc[2 + jj][r] + acc[4 + jj][r];
      t1[jj] = acc[2 + jj][r] - acc[4 + jj][r] - acc[6 + jj][r];
    }
    float pv[4];
    // columns 0,1 (fh = 0):  y[.][0] += t[0] + t[1],  y[.][1] += t[1];   columns 2,3:  y[.][0] += t[2],  y[.][1] += -t[2] - t[3]
    pv[0] = fh == 0 ? t0[0] + t0[1] : t0[0];
    pv[1] = fh == 0 ? t0[1] : -t0[0] - t0[1];
    pv[2] = fh == 0 ? t1[0] + t1[1] : t1[0];
    pv[3] = fh == 0 ? t1[1] : -t1[0] - t1[1];
    const bool mine = (r >> 3) == fh;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (!mine) sX[((wave * 8 + (r & 7)) * 4 + v) * 64 + lane] = pv[v];       // the other wave finishes this row
      keep[r & 7][v] = mine ? pv[v] : keep[r & 7][v];
    }
  }
  __syncthreads();
  const int col = nt * NTs + li;
  const float bv = a.bias != nullptr ? a.bias[col] : 0.f;
  const int other = wave ^ 1;
#pragma unroll
  for (int rl = 0; rl < 8; ++rl) {
    float y[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) y[v] = keep[rl][v] + sX[((other * 8 + rl) * 4 + v) * 64 + lane];
    const int r = 8 * fh + rl;
    const int row = wm * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
    const int op = sOut[row];
    if (op >= 0) {
      float* dst = a.Y + (long)op * N + col;
      float s4[4] = {bv, bv, bv, bv};
      if (a.add != nullptr) {
        const float* ad = a.add + (long)op * N + col;
        s4[0] += ad[0]; s4[1] += ad[N]; s4[2] += ad[(long)a.W * N]; s4[3] += ad[(long)a.W * N + N];
      }
      if (a.act == ACT_TANH) {
        dst[0] = act_fwd(y[0] + s4[0], ACT_TANH);
        dst[N] = act_fwd(y[1] + s4[1], ACT_TANH);
        dst[(long)a.W * N] = act_fwd(y[2] + s4[2], ACT_TANH);
        dst[(long)a.W * N + N] = act_fwd(y[3] + s4[3], ACT_TANH);
      } else {
        const float osl = act_slope(a.act);
        dst[0] = act_slope_fwd(y[0] + s4[0], osl);
        dst[N] = act_slope_fwd(y[1] + s4[1], osl);
        dst[(long)a.W * N] = act_slope_fwd(y[2] + s4[2], osl);
        dst[(long)a.W * N + N] = act_slope_fwd(y[3] + s4[3], osl);
      }
    }
  }
}

// ---- weight gradient: F(3x3, 2x2) ---------------------------------------------------------------------------------
//     dW = sum_tiles A'^T [ (B^T d B) (.) (G dy G^T) ] A'     d = 4x4 input patch, dy = 2x2 output-gradient tile
// with the SAME B^T as the forward pass, G = [[1,0],[.5,.5],[.5,-.5],[0,1]], A'^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]].
// The sixteen GEMMs now reduce over TILES: Mw[f][ci][co] = sum_t Vx[f][t][ci] * Vy[f][t][co].
// workgroup = 64 ci x 64 co x one slice of the tile range (split over workgroups, slabs reduced by
// reduce_partials_kernel like every other weight gradient).  Per chunk of 8 tiles (2 tile rows x 4 tile columns of one
// image): raw X pixels (6 x 10 incl. halo) and raw dY pixels (8 x 4) -> LDS; waves 0,1 transform X, waves 2,3 dY:
// thread = (channel, tile row), it reads its 4 x 10 strip once (the row pass of B^T d B is shared by the four tiles),
// and writes V[f][channel][4 tiles] as ONE 16-byte LDS store per frequency; G's factors of 1/2 are applied to the
// accumulators at the end instead of per tile.  16 x 4 MFMA per wave and chunk, all data movement slotted between MFMA
// groups (see wino_conv_kernel); epilogue A'^T Mw A' lane-local -> slab[9][Ci][Co].
constexpr int WG_XPIX = 60;    // raw X pixels per chunk: 6 rows x 10 columns
constexpr int WG_XROWS = 64;   // rows allocated: the 4 x 256 staging items cover 64 pixel rows, the last 4 hold zeros
constexpr int WG_YPIX = 32;    // raw dY pixels per chunk: 8 tiles x 4
constexpr int RSW = 64;        // channels per raw pixel row

struct WinoWgArgs {
  const float* X;      // [B][H][W][Ci]
  const float* dY;     // [B][H][W][Co]
  float* part;         // [S][9][Ci][Co]
  float* pbias;        // [S][Co] or null: sum of dY over this slice's pixels (written by the ci-block-0 workgroups)
  int B, H, W, Ci, Co;
  int cy_n, cx_n;      // chunks per image: (H/4, W/8)
  int nchunks, chunks_per_split;
};

__global__ __launch_bounds__(256) void wino_wgrad_kernel(WinoWgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int FB = 64 * 8;                 // floats per frequency
  constexpr int BUF = 16 * FB;               // floats per buffer
  float* sVx = smem;                         // [2][16][64 ci][8 tiles]   (rows swizzled like wino_conv_kernel's)
  float* sVy = sVx + 2 * BUF;                // [2][16][64 co][8 tiles]
  float* sRX = sVy + 2 * BUF;                // [WG_XROWS][RSW]
  float* sRY = sRX + WG_XROWS * RSW;         // [WG_YPIX][RSW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  // XCD-aware order: all (ci block, co block) workgroups of one slice of the tile range read the same X and dY pixels
  // (about 1 MB each); workgroup ids go round-robin over the 8 XCDs, so the slices are dealt out per XCD
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if ((gridDim.z & 7) == 0) {
    const int nt = gridDim.x * gridDim.y;
    const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int slot = L >> 3, t = slot % nt;
    bz = (slot / nt) * 8 + (L & 7);
    bx = t % gridDim.x;
    by = t / gridDim.x;
  }
  const int ci0 = bx * 64, co0 = by * 64, split = bz;
  const int Ci = a.Ci, Co = a.Co, H = a.H, W = a.W;
  const __amdgpu_buffer_rsrc_t rX = wrsrc(a.X, (long)a.B * H * W * Ci * 4);
  const __amdgpu_buffer_rsrc_t rY = wrsrc(a.dY, (long)a.B * H * W * Co * 4);

  // raw-load items.  X: e = tid + 256*i (i < 4) -> pixel p = e>>4 (py = p/10, px = p%10), float4 e&15; p < 60
  //                  dY: e = tid + 256*i (i < 2) -> pixel p = e>>4 = tile*4 + pos, float4 e&15
  // halo flags of an item: 1 row above the chunk, 2 row below, 4 column left, 8 column right, 16 not a pixel of the block.
  // A chunk at the image border turns the matching flags into "outside": no per-chunk coordinate arithmetic per item.
  int x_flag[4];
  unsigned x_rel[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + 256 * i, p = e >> 4, py = p / 10 - 1, px = p % 10 - 1;
    x_flag[i] = (py == -1 ? 1 : 0) | (py == 4 ? 2 : 0) | (px == -1 ? 4 : 0) | (px == 8 ? 8 : 0) | (p < WG_XPIX ? 0 : 16);
    x_rel[i] = (unsigned)((py * W + px) * Ci + ci0 + 4 * (e & 15)) * 4u;
  }
  unsigned y_rel[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + 256 * i, t = e >> 6, pos = (e >> 4) & 3;
    const int y = 2 * (t >> 2) + (pos >> 1), x = 2 * (t & 3) + (pos & 1);
    y_rel[i] = (unsigned)((y * W + x) * Co + co0 + 4 * (e & 15)) * 4u;
  }
  f32x4 rx[4], ry[2];
  // position of the next chunk to load (uniform): image lb, chunk row lcy, chunk column lcx; stepped, never divided
  int lb, lcy, lcx;
  {
    const int per = a.cy_n * a.cx_n;
    const int c = split * a.chunks_per_split;
    lb = c / per;
    const int r = c - lb * per;
    lcy = r / a.cx_n;
    lcx = r - lcy * a.cx_n;
  }
  int pix0 = 0, cflag = 16;                          // first output pixel of that chunk, its border flags (| 16)
  auto chunk_here = [&](bool live) {
    pix0 = (lb * H + 4 * lcy) * W + 8 * lcx;
    cflag = live ? ((lcy == 0 ? 1 : 0) | (lcy == a.cy_n - 1 ? 2 : 0) | (lcx == 0 ? 4 : 0) | (lcx == a.cx_n - 1 ? 8 : 0) | 16) : 31;
  };
  auto chunk_step = [&]() {                          // branch-free (selects): no control flow inside the MFMA loop
    const int nx = lcx + 1;
    const bool wx = nx == a.cx_n;
    lcx = wx ? 0 : nx;
    const int ny = lcy + (wx ? 1 : 0);
    const bool wy = ny == a.cy_n;
    lcy = wy ? 0 : ny;
    lb += wy ? 1 : 0;
  };
  // (x_rel is negative for the halo row / column above-left of the chunk: it must be added to the chunk base BEFORE the
  //  buffer's range check sees it, so this one stays a VALU add)
  auto ld_x = [&](int i) {
    rx[i] = wld4(rX, (x_flag[i] & cflag) ? kOOBw : (unsigned)pix0 * (unsigned)Ci * 4u + x_rel[i]);
  };
  auto ld_y = [&](bool live, int i) { ry[i] = wld4s(rY, live ? y_rel[i] : kOOBw, (unsigned)pix0 * (unsigned)Co * 4u); };
  auto st_x = [&](int i) {       // item 3 of the upper threads lies beyond pixel 59: zeros into the unused rows 60..63
    const int e = tid + 256 * i;
    *reinterpret_cast<f32x4*>(sRX + (e >> 4) * RSW + 4 * (e & 15)) = rx[i];
  };
  auto st_y = [&](int i) {
    const int e = tid + 256 * i;
    *reinterpret_cast<f32x4*>(sRY + (e >> 4) * RSW + 4 * (e & 15)) = ry[i];
  };

  // transform roles: EVERY thread = (channel chn, tile row tq, tile half th) and does the same work: the two tiles
  // 2th, 2th+1 of the X strip (columns 4th .. 4th+5) and the same two tiles of dY.  (Role branches -- some waves X,
  // others dY -- cost ~150 register moves per chunk at the joins, and left the X waves with twice the VALU work.)
  const int chn = tid & 63, tq = (tid >> 6) & 1, th = tid >> 7;
  const int v_st = chn * 8 + ((tq ^ ((chn >> 3) & 1)) << 2) + 2 * th;
  const int ma = wm * 32 + li, nbr = wn * 32 + li;
  const int a_rd = ma * 8 + ((lh ^ ((ma >> 3) & 1)) << 2);
  const int b_rd = nbr * 8 + ((lh ^ ((nbr >> 3) & 1)) << 2);

  float sx[4][6];       // strip rows 2tq..2tq+3, columns 4th..4th+5; afterwards the row-transformed strip
  float ey[2][4];       // dY [tile][pos]
  auto rd_x = [&](int r) {
#pragma unroll
    for (int x = 0; x < 6; ++x) sx[r][x] = sRX[((2 * tq + r) * 10 + 4 * th + x) * RSW + chn];
  };
  auto tf_x_rows = [&]() {
#pragma unroll
    for (int x = 0; x < 6; ++x) {
      const float d0 = sx[0][x], d1 = sx[1][x], d2 = sx[2][x], d3 = sx[3][x];
      sx[0][x] = d0 - d2; sx[1][x] = d1 + d2; sx[2][x] = d2 - d1; sx[3][x] = d1 - d3;
    }
  };
  auto st_vx = [&](float* sv, int i) {      // frequencies 4i .. 4i+3, two tiles each
    f32x2 v0, v1, v2, v3;
#pragma unroll
    for (int tx = 0; tx < 2; ++tx) {
      const float t0 = sx[i][2 * tx], t1 = sx[i][2 * tx + 1], t2 = sx[i][2 * tx + 2], t3 = sx[i][2 * tx + 3];
      v0[tx] = t0 - t2; v1[tx] = t1 + t2; v2[tx] = t2 - t1; v3[tx] = t1 - t3;
    }
    float* dst = sv + (4 * i) * FB + v_st;
    *reinterpret_cast<f32x2*>(dst) = v0;
    *reinterpret_cast<f32x2*>(dst + FB) = v1;
    *reinterpret_cast<f32x2*>(dst + 2 * FB) = v2;
    *reinterpret_cast<f32x2*>(dst + 3 * FB) = v3;
  };
  float bsum = 0.f;     // running sum of this thread's dY pixels = its share of the bias gradient
  auto rd_y = [&](int tx) {
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
      ey[tx][pos] = sRY[((tq * 4 + 2 * th + tx) * 4 + pos) * RSW + chn];
      bsum += ey[tx][pos];
    }
  };
  auto st_vy = [&](float* sv, int i) {      // 2*G e (2G)^T without the factors: rows {e0, e0+e1, e0-e1, e1}, same for columns
    f32x2 v0, v1, v2, v3;
#pragma unroll
    for (int tx = 0; tx < 2; ++tx) {
      const float e00 = ey[tx][0], e01 = ey[tx][1], e10 = ey[tx][2], e11 = ey[tx][3];
      float r0, r1;
      if (i == 0) { r0 = e00; r1 = e01; }
      else if (i == 1) { r0 = e00 + e10; r1 = e01 + e11; }
      else if (i == 2) { r0 = e00 - e10; r1 = e01 - e11; }
      else { r0 = e10; r1 = e11; }
      v0[tx] = r0; v1[tx] = r0 + r1; v2[tx] = r0 - r1; v3[tx] = r1;
    }
    float* dst = sv + (4 * i) * FB + v_st;
    *reinterpret_cast<f32x2*>(dst) = v0;
    *reinterpret_cast<f32x2*>(dst + FB) = v1;
    *reinterpret_cast<f32x2*>(dst + 2 * FB) = v2;
    *reinterpret_cast<f32x2*>(dst + 3 * FB) = v3;
  };

  f32x16 acc[16];
#pragma unroll
  for (int f = 0; f < 16; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  f32x4 fa, fb, na, nb;
  auto rd_frag = [&](const float* sx_, const float* sy_, int f) {
    na = *reinterpret_cast<const f32x4*>(sx_ + f * FB + a_rd);
    nb = *reinterpret_cast<const f32x4*>(sy_ + f * FB + b_rd);
  };
#define WINO_MFMA4(f)                                                                                   \
  _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                      \
      acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s_], fb[s_], acc[f], 0, 0, 0)
#define WINO_FENCE() __builtin_amdgcn_sched_barrier(0)

  const int c_begin = split * a.chunks_per_split;
  int c_end = c_begin + a.chunks_per_split;
  if (c_end > a.nchunks) c_end = a.nchunks;
  const int n = c_end - c_begin;             // >= 1 by construction of the grid
#ifdef CTVAE_PHASE_TIMING
  long long tq_[6];
  tq_[0] = clock64();
#endif

  // prologue: chunk 0 -> buffer 0, chunk 1 -> registers
  chunk_here(true);
#pragma unroll
  for (int i = 0; i < 4; ++i) ld_x(i);
#pragma unroll
  for (int i = 0; i < 2; ++i) ld_y(true, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) st_x(i);
#pragma unroll
  for (int i = 0; i < 2; ++i) st_y(i);
  __syncthreads();
  {
    const bool live = n > 1;
    chunk_step();
    chunk_here(live);
#pragma unroll
    for (int i = 0; i < 4; ++i) ld_x(i);
#pragma unroll
    for (int i = 0; i < 2; ++i) ld_y(live, i);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) rd_x(r);
  rd_y(0); rd_y(1);
  tf_x_rows();
#pragma unroll
  for (int i = 0; i < 4; ++i) { st_vx(sVx, i); st_vy(sVy, i); }
  __syncthreads();

#ifdef CTVAE_PHASE_TIMING
  tq_[1] = clock64();
  long long seg_[5] = {0, 0, 0, 0, 0};
  long long tl_ = clock64();
#define WSEG(i) do { const long long now_ = clock64(); seg_[i] += now_ - tl_; tl_ = now_; } while (0)
#else
#define WSEG(i) do {} while (0)
#endif
  for (int cc = 0; cc + 1 < n; ++cc) {
    const int cur = cc & 1;
    const float* sxc = sVx + cur * BUF;
    const float* syc = sVy + cur * BUF;
    float* sxn = sVx + (cur ^ 1) * BUF;
    float* syn = sVy + (cur ^ 1) * BUF;
    const bool live = cc + 2 < n;                                   // is there a chunk two ahead?
    rd_frag(sxc, syc, 0);
    chunk_step();
    chunk_here(live);
    fa = na; fb = nb;
    WINO_FENCE();
    // steps 0..5: registers (chunk cc+1) -> raw LDS, then refill them with chunk cc+2; one instruction behind each MFMA
#define WINO_M1(f, s_) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s_], fb[s_], acc[f], 0, 0, 0); WINO_FENCE()
    rd_frag(sxc, syc, 1);
    WINO_M1(0, 0); st_x(0); WINO_M1(0, 1); st_x(1); WINO_M1(0, 2); st_x(2); WINO_M1(0, 3); st_x(3);
    fa = na; fb = nb; rd_frag(sxc, syc, 2);
    WINO_M1(1, 0); st_y(0); WINO_M1(1, 1); st_y(1); WINO_M1(1, 2); ld_x(0); WINO_M1(1, 3); ld_x(1);
    fa = na; fb = nb; rd_frag(sxc, syc, 3);
    WINO_M1(2, 0); ld_x(2); WINO_M1(2, 1); ld_x(3); WINO_M1(2, 2); ld_y(live, 0); WINO_M1(2, 3); ld_y(live, 1);
    fa = na; fb = nb; rd_frag(sxc, syc, 4);
    WINO_MFMA4(3); fa = na; fb = nb; WINO_FENCE();
    rd_frag(sxc, syc, 5);
    WINO_MFMA4(4); fa = na; fb = nb; WINO_FENCE();
    rd_frag(sxc, syc, 6);
    WINO_MFMA4(5); fa = na; fb = nb; WINO_FENCE();
    WSEG(0);
    __syncthreads();                                 // raw pixels of chunk cc+1 are visible
    WSEG(1);
    rd_frag(sxc, syc, 7);
    WINO_M1(6, 0); rd_x(0); WINO_M1(6, 1); rd_x(1); WINO_M1(6, 2); rd_x(2); WINO_M1(6, 3); rd_x(3);
    fa = na; fb = nb; rd_frag(sxc, syc, 8);
    WINO_M1(7, 0); rd_y(0); WINO_M1(7, 1); rd_y(1); WINO_M1(7, 2); WINO_M1(7, 3);
    fa = na; fb = nb; rd_frag(sxc, syc, 9);
    WINO_MFMA4(8); fa = na; fb = nb; WINO_FENCE();
    rd_frag(sxc, syc, 10); tf_x_rows();
    WINO_MFMA4(9); fa = na; fb = nb; WINO_FENCE();
    rd_frag(sxc, syc, 11);
    WINO_M1(10, 0); st_vx(sxn, 0); WINO_M1(10, 1); st_vy(syn, 0); WINO_M1(10, 2); WINO_M1(10, 3);
    fa = na; fb = nb; rd_frag(sxc, syc, 12);
    WINO_M1(11, 0); st_vx(sxn, 1); WINO_M1(11, 1); st_vy(syn, 1); WINO_M1(11, 2); WINO_M1(11, 3);
    fa = na; fb = nb; rd_frag(sxc, syc, 13);
    WINO_M1(12, 0); st_vx(sxn, 2); WINO_M1(12, 1); st_vy(syn, 2); WINO_M1(12, 2); WINO_M1(12, 3);
    fa = na; fb = nb; rd_frag(sxc, syc, 14);
    WINO_M1(13, 0); st_vx(sxn, 3); WINO_M1(13, 1); st_vy(syn, 3); WINO_M1(13, 2); WINO_M1(13, 3);
    fa = na; fb = nb;
#undef WINO_M1
    WSEG(2);
    rd_frag(sxc, syc, 15);
    WINO_MFMA4(14); fa = na; fb = nb; WINO_FENCE();
    WINO_MFMA4(15); WINO_FENCE();
    WSEG(3);
    __syncthreads();                                 // V of chunk cc+1 complete, chunk cc's buffers free
    WSEG(4);
  }
#ifdef CTVAE_PHASE_TIMING
  tq_[2] = clock64();
#endif
  {                                                  // last chunk: MFMA only
    const int cur = (n - 1) & 1;
    const float* sxc = sVx + cur * BUF;
    const float* syc = sVy + cur * BUF;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      fa = *reinterpret_cast<const f32x4*>(sxc + f * FB + a_rd);
      fb = *reinterpret_cast<const f32x4*>(syc + f * FB + b_rd);
      WINO_MFMA4(f);
    }
  }
#undef WINO_MFMA4
#undef WINO_FENCE

#ifdef CTVAE_PHASE_TIMING
  tq_[3] = clock64();
#endif
  if (a.pbias != nullptr && bx == 0) {                 // bias gradient: the two tile rows of a channel meet in LDS
    __syncthreads();
    if (wave > 0) sRY[(wave - 1) * 64 + chn] = bsum;            // the four (tile row, tile half) shares of a channel
    __syncthreads();
    if (wave == 0) a.pbias[(long)split * Co + co0 + chn] = ((bsum + sRY[chn]) + sRY[64 + chn]) + sRY[128 + chn];
  }

  // ---- epilogue: G's halves (rows/columns 1,2 of the dY transform), then dW[ky][kx] = A'^T Mw A',
  //      A'^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]] ----
  float* slab = a.part + (long)split * 9 * Ci * Co;
  const int col = co0 + wn * 32 + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = ci0 + wm * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
    float m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sc = ((i == 1 || i == 2) ? 0.5f : 1.f) * ((j == 1 || j == 2) ? 0.5f : 1.f);
        m[i][j] = acc[4 * i + j][r] * sc;
      }
    float t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = m[0][j] + m[1][j] + m[2][j];
      t[1][j] = m[1][j] - m[2][j];
      t[2][j] = m[1][j] + m[2][j] - m[3][j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float* dst = slab + ((long)(3 * i) * Ci + row) * Co + col;
      dst[0] = t[i][0] + t[i][1] + t[i][2];
      dst[(long)Ci * Co] = t[i][1] - t[i][2];
      dst[2L * Ci * Co] = t[i][1] + t[i][2] - t[i][3];
    }
  }
#ifdef CTVAE_PHASE_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  tq_[4] = clock64();
  if (lane == 0) {
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (wg < 256) {
      for (int i = 0; i < 5; ++i) g_wino_phase[(wg * 4 + wave) * 8 + i] = tq_[i] - tq_[0];
      for (int i = 0; i < 5; ++i) g_wino_phase[8 * 1024 + (wg * 4 + wave) * 8 + i] = seg_[i];
    }
  }
#endif
}

// 3x3, stride 1, "same" padding, one class: effective tap table (ky' = dy + 1, kx' = dx + 1) or false
bool wino_taps(const ConvGeom& g, WTaps& wt) {
  if (!packed_weights(g) || g.is != 1 || g.os != 1 || g.ncls != 1 || g.ntaps[0] != 9 || g.gH != g.sH || g.gW != g.sW) return false;
  bool seen[9] = {};
  for (int t = 0; t < 9; ++t) {
    const Tap& tp = g.taps[0][t];
    if (tp.dy < -1 || tp.dy > 1 || tp.dx < -1 || tp.dx > 1) return false;
    const int a = (tp.dy + 1) * 3 + tp.dx + 1;
    if (seen[a]) return false;
    seen[a] = true;
    wt.t[a] = tp.wtap;
  }
  return true;
}

bool wino_block(const ConvGeom& g, int& bh, int& bw, int& nb) {
  const int H = g.gH, W = g.gW;
  if (H % 2 || W % 2) return false;
  const int TH = H / 2, TW = W / 2;
  if (TH * TW <= MT) {
    if (MT % (TH * TW)) return false;
    bh = TH; bw = TW; nb = MT / (TH * TW);
  } else {
    if (TH % 8 || TW % 8) return false;
    bh = bw = 8; nb = 1;
  }
  return nb * (2 * bh + 2) * (2 * bw + 2) <= NPMAX;
}

}  // namespace

// process-wide switch (ctvae_winograd_enable): on unless the environment says CTVAE_NO_WINOGRAD=1
static int g_wino_on = -1;
bool wino_enabled() {
  if (g_wino_on < 0) {
    const char* e = getenv("CTVAE_NO_WINOGRAD");
    g_wino_on = (e != nullptr && e[0] == '1') ? 0 : 1;
  }
  return g_wino_on != 0;
}
int wino_set_enabled(int on) {
  const int prev = wino_enabled() ? 1 : 0;
  g_wino_on = on ? 1 : 0;
  return prev;
}

size_t wino_ws_floats(const ConvGeom& g) { return (size_t)16 * g.gC * g.sC; }

// Is this launch a Winograd candidate?  (the caller still decides on epilogue features)
bool wino_supported(const ConvGeom& g, size_t ws_floats) {
  WTaps wt;
  int bh, bw, nb;
  if (!wino_taps(g, wt) || !wino_block(g, bh, bw, nb)) return false;
  if (g.gC % KC8 || g.gC < 64 || g.sC % NT) return false;
  if ((long)g.B * g.gH * g.gW * g.gC >= (1L << 29) || (long)g.B * g.sH * g.sW * g.sC >= (1L << 29)) return false;
  if ((long)16 * g.gC * g.sC >= (1L << 29)) return false;
  const int tiles = g.B * (g.gH / 2) * (g.gW / 2);
  const long wgs = (long)ceil_div(tiles, MT) * (g.sC / NT);
  if (wgs < 64) return false;                      // too few workgroups: the split-K direct kernel fills the chip better
  return wino_ws_floats(g) <= ws_floats;
}

// filters_ready: this launch's transformed filters (what an earlier forward launch left in its bwd_out) -> no transform.
// bwd_out (forward launches only): also write the data gradient's filter set there, in the same transform launch.
int launch_wino_filters_batch(int n, const float* const* W, float* const* Uf, float* const* Ub, const int* Ci, const int* Co,
                              hipStream_t st) {
  if (n <= 0 || !W || !Uf || !Ub || !Ci || !Co) return kErrBadArg;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_weight2_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  for (int l0 = 0; l0 < n; l0 += kWinoBatchMax) {
    WinoBatch t{};
    const int m = n - l0 < kWinoBatchMax ? n - l0 : kWinoBatchMax;
    int gx = 0, gy = 0;
    double bytes = 0;
    for (int l = 0; l < m; ++l) {
      if (!W[l0 + l] || !Uf[l0 + l] || !Ub[l0 + l] || Ci[l0 + l] % 32 || Co[l0 + l] % 32 || Ci[l0 + l] <= 0 || Co[l0 + l] <= 0)
        return kErrBadArg;
      t.W[l] = W[l0 + l]; t.Uf[l] = Uf[l0 + l]; t.Ub[l] = Ub[l0 + l]; t.Ci[l] = Ci[l0 + l]; t.Co[l] = Co[l0 + l];
      gx = t.Ci[l] / 32 > gx ? t.Ci[l] / 32 : gx;
      gy = t.Co[l] / 32 > gy ? t.Co[l] / 32 : gy;
      bytes += 4.0 * (9.0 + 32.0) * t.Ci[l] * t.Co[l];
    }
    ProfScope ps("wino_weight2_batch_kernel", st, 0.0, bytes);
    hipLaunchKernelGGL(wino_weight2_batch_kernel, dim3(gx, gy, m), dim3(256), (size_t)16 * 32 * W2_LD * 4, st, t);
    CTVAE_LAUNCH_CHECK();
  }
  return 0;
}

int launch_wino_conv(const ConvGeom& g, const float* X, const float* Wp, const float* bias, float* Y, int act, float* ws,
                     size_t ws_floats, hipStream_t st, const float* filters_ready, float* bwd_out, const float* add) {
  WTaps wt;
  int bh, bw, nb;
  if (!wino_supported(g, ws_floats) || !wino_taps(g, wt) || !wino_block(g, bh, bw, nb)) return kErrBadArg;
  const int K = g.gC, N = g.sC;
  const float* Ut = ws;
  bool plain_fwd = g.wT == 0;
  for (int a9 = 0; a9 < 9; ++a9) plain_fwd = plain_fwd && wt.t[a9] == a9;
  if (filters_ready != nullptr) {
    Ut = filters_ready;
  } else if (bwd_out != nullptr && plain_fwd && K % 32 == 0 && N % 32 == 0) {
    static bool attr2_set = false;
    if (!attr2_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_weight2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
      attr2_set = true;
    }
    ProfScope ps("wino_weight2_kernel", st, 0.0, 4.0 * (9.0 + 32.0) * K * N);
    hipLaunchKernelGGL(wino_weight2_kernel, dim3(K / 32, N / 32), dim3(256), (size_t)16 * 32 * W2_LD * 4, st, Wp, ws, bwd_out, K,
                       N);
    CTVAE_LAUNCH_CHECK();
  } else {
    ProfScope ps("wino_weight_kernel", st, 0.0, 4.0 * (9.0 + 16.0) * K * N);
    hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)(((long)K * N + 255) / 256)), dim3(256), 0, st, Wp, ws, K, N, g.wCi,
                       g.wCo, g.wT, wt);
    CTVAE_LAUNCH_CHECK();
  }
  WinoArgs a{};
  a.X = X; a.Ut = Ut; a.bias = bias; a.add = add; a.Y = Y;
  a.B = g.B; a.H = g.gH; a.W = g.gW; a.K = K; a.N = N;
  a.bh = bh; a.bw = bw; a.nb = nb;
  a.by_n = (g.gH / 2) / bh; a.bx_n = (g.gW / 2) / bw;
  a.act = act;
  const int mtiles = nb > 1 ? ceil_div(g.B, nb) : g.B * a.by_n * a.bx_n;
  // fewer than 200 of the 64 x 64 workgroups: 64 x 32 workgroups whose waves split the frequencies (twice as many)
  const bool fsplit = (long)mtiles * (N / NT) < 200;
  const size_t smem = (size_t)(2 * 16 * MT * 8 + 2 * 16 * (fsplit ? 32 : NT) * 8 + NPMAX * RS) * 4 + MT * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_conv_fs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  char name[128];
  snprintf(name, sizeof name, fsplit ? "wino_conv_fs_kernel" : "wino_conv_kernel");
  if (prof_detailed())
    snprintf(name, sizeof name, "%s B=%d %dx%d K=%d N=%d wT=%d", fsplit ? "wino_conv_fs_kernel" : "wino_conv_kernel", g.B, g.gH,
             g.gW, K, N, g.wT);
  // flops are counted as the direct convolution's (what the layer computes), bytes as input + output + filters
  ProfScope ps(name, st, 2.0 * 9.0 * (double)g.B * g.gH * g.gW * K * N,
               4.0 * ((double)g.B * g.gH * g.gW * (K + N) + 16.0 * K * N));
  if (fsplit) hipLaunchKernelGGL(wino_conv_fs_kernel, dim3((unsigned)(mtiles * (N / 32))), dim3(256), smem, st, a);
  else hipLaunchKernelGGL(wino_conv_kernel, dim3((unsigned)(mtiles * (N / NT))), dim3(256), smem, st, a);
  CTVAE_LAUNCH_CHECK();
  return 0;
}


static bool wino_wgrad_chunk(const ConvGeom& g) {      // chunks of 2 x 4 tiles = 4 x 8 output pixels
  return g.gH % 4 == 0 && g.gW % 8 == 0;
}

// weight gradient of a 3x3 / stride 1 / same-padding conv, bias-free caller (geometry of kind 0)
bool wino_wgrad_supported(const ConvGeom& g, size_t ws_floats, int* splits) {
  WTaps wt;
  if (g.wT != 0 || !wino_taps(g, wt) || !wino_wgrad_chunk(g)) return false;
  for (int a = 0; a < 9; ++a)
    if (wt.t[a] != a) return false;                 // dW is written as [ky*3+kx][Ci][Co]
  if (g.gC % 64 || g.sC % 64) return false;
  if ((long)g.B * g.gH * g.gW * g.gC >= (1L << 29) || (long)g.B * g.sH * g.sW * g.sC >= (1L << 29)) return false;
  const int tiles = g.B * (g.gH / 2) * (g.gW / 2);
  const int nchunks = tiles / 8;
  const int out_tiles = (g.gC / 64) * (g.sC / 64);
  int S = ceil_div(256, out_tiles);                 // one workgroup per CU
  if (S > nchunks / 8) S = nchunks / 8;             // at least 8 chunks per workgroup
  if (S < 1) return false;
  if ((long)S * out_tiles < 128) return false;
  while (S > 1 && (size_t)S * (9 * g.gC + 1) * g.sC > ws_floats) --S;
  if ((size_t)S * (9 * g.gC + 1) * g.sC > ws_floats) return false;
  if (splits != nullptr) *splits = S;
  return true;
}

// slabs at ws [nparts][9][Ci][Co]; with want_bias the dY sums follow at *pbias_out [nparts][Co]
int launch_wino_wgrad(const ConvGeom& g, const float* X, const float* dY, float* ws, size_t ws_floats, int* nparts,
                      bool want_bias, float** pbias_out, hipStream_t st) {
  int S = 0;
  if (!wino_wgrad_supported(g, ws_floats, &S)) return kErrBadArg;
  WinoWgArgs a{};
  a.X = X; a.dY = dY; a.part = ws;
  a.B = g.B; a.H = g.gH; a.W = g.gW; a.Ci = g.gC; a.Co = g.sC;
  a.cy_n = g.gH / 4; a.cx_n = g.gW / 8;
  a.nchunks = g.B * a.cy_n * a.cx_n;
  a.chunks_per_split = ceil_div(a.nchunks, S);
  S = ceil_div(a.nchunks, a.chunks_per_split);
  *nparts = S;
  a.pbias = want_bias ? ws + (size_t)S * 9 * a.Ci * a.Co : nullptr;
  if (pbias_out != nullptr) *pbias_out = a.pbias;
  char name[128];
  snprintf(name, sizeof name, "wino_wgrad_kernel");
  if (prof_detailed()) snprintf(name, sizeof name, "wino_wgrad_kernel B=%d %dx%d Ci=%d Co=%d S=%d", g.B, g.gH, g.gW, a.Ci, a.Co, S);
  ProfScope ps(name, st, 2.0 * 9.0 * (double)g.B * g.gH * g.gW * a.Ci * a.Co,
               4.0 * ((double)g.B * g.gH * g.gW * (a.Ci + a.Co) + 9.0 * S * a.Ci * a.Co));
  const size_t smem = (size_t)(4 * 16 * 64 * 8 + (WG_XROWS + WG_YPIX) * RSW) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(wino_wgrad_kernel, dim3(a.Ci / 64, a.Co / 64, S), dim3(256), smem, st, a);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
