// Picture-side transposed convolution of the MCQ-VAE / CT-MCQ-VAE decoders: ConvTranspose2d(64 -> 3, k4 s2 p1) (+bias,
// +Tanh), mcq_vae.py:229-237 -- 64 x 64 x 3 outputs from a 32 x 32 x 64 input, 67 MB read and 12.6 MB written per 256
// images.  As a thin VALU kernel it ran at 164 us (14 % of the VALU peak); the MFMA form below is HBM-bound.
//
// Z formulation (as image.hip does for the 3x3 picture-side convs): every INPUT pixel q is multiplied once with all 16
// taps of the filter,
//        Z[q][t*3 + c] = sum_ci in[q][ci] * W[t][ci][c]            (a [pixels x 64] x [64 x 48] GEMM on MFMA),
// and an OUTPUT pixel (2y+py, 2x+px) is the sum of the 4 Z entries of its parity class from a 2 x 2 neighbourhood of q:
//        out[2y+py][2x+px][c] = bias[c] + sum_{t in class(py,px)} Z[(y+dy_t, x+dx_t)][t*3 + c].
// workgroup = 8 x 32 input pixels (+1 halo: 340 patch pixels = 11 MFMA row tiles): the A operand goes from global memory
// straight into the MFMA registers (lane = pixel, its 32 channels of the lane half: 128 contiguous bytes), the 64 x 48
// filter matrix sits in 64 registers per lane for the whole workgroup, Z lives in LDS (340 x 49 floats), the gather
// writes 16 x 64 x 3 outputs as contiguous 12-byte pixels.  No MAC is spent on inserted zeros, none on the 61 unused
// output channels a 64-wide GEMM tile would pad.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

namespace {

constexpr int UTH = 8, UTW = 32;                 // input pixels per workgroup tile
constexpr int UPH = UTH + 2, UPW = UTW + 2;      // patch with halo
constexpr int UNP = UPH * UPW;                   // 340
constexpr int UMT = (UNP + 31) / 32;             // 11 row tiles
constexpr int ULDZ = 49;                         // Z row stride (floats)
constexpr int ULDW = 68;                         // staged filter row stride (floats): [n][64 k] + pad
constexpr unsigned kOOBu = 0x80000000u;

struct UpImgArgs {
  const float* X;        // [B][H][W][64]
  const float* Wp;       // [16][64][3]
  const float* bias;     // [3] or null
  float* Y;              // [B][2H][2W][3]
  int B, H, W;
  int tiles_y, tiles_x;
  int act;
  int wtap[16];          // [class*4 + ti] -> tap index into Wp
  int dy[16], dx[16];    // [class*4 + ti] -> input offset
};

__device__ __forceinline__ f32x4 uld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}

__global__ __launch_bounds__(256, 2) void upimg_fwd_kernel(UpImgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sZ = smem;                 // [UNP][ULDZ]; its first 48*ULDW floats hold the staged filter matrix at the start
  float* sWt = smem;                // [48][ULDW]: sWt[n][k], n = (class*4 + ti)*3 + c
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int per = a.tiles_y * a.tiles_x;
  const int b = blockIdx.x / per, r0 = blockIdx.x - b * per;
  const int ty = r0 / a.tiles_x, tx = r0 - ty * a.tiles_x;
  const int qy0 = ty * UTH, qx0 = tx * UTW;

  // ---- filter matrix -> LDS (transposed) -> registers ----
  {                                                        // all 12 loads of a thread before the first LDS store (they were 12 round trips in a row)
    float wv[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int e = tid + 256 * j;
      const int k = e / 48, n = e - k * 48;                // reads: 3 contiguous floats per (tap, k)
      const int t16 = n / 3, c = n - t16 * 3;
      wv[j] = a.Wp[((long)a.wtap[t16] * 64 + k) * 3 + c];
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int e = tid + 256 * j;
      const int k = e / 48, n = e - k * 48;
      sWt[n * ULDW + k] = wv[j];
    }
  }
  __syncthreads();
  f32x4 bw[2][8];                                          // B operand: column n = li (+32), k = 32*lh + 4j .. +3
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = nt * 32 + li;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      bw[nt][j] = n < 48 ? *reinterpret_cast<const f32x4*>(&sWt[n * ULDW + 32 * lh + 4 * j]) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();                                         // the staging area becomes part of Z

  // ---- Z = patch x filter matrix ----
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0,
                                                                      (int)((long)a.B * a.H * a.W * 64 * 4), 0x00020000);
  for (int mt = wave; mt < UMT; mt += 4) {
    const int p = mt * 32 + li;                            // patch pixel of this lane
    const int py = p / UPW, px = p - py * UPW;
    const int y = qy0 + py - 1, x = qx0 + px - 1;
    const bool ok = p < UNP && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
    const unsigned off = ok ? ((unsigned)((b * a.H + y) * a.W + x) * 64u + 32u * lh) * 4u : kOOBu;
    f32x4 av[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) av[j] = uld4(rX, ok ? off + 16u * j : kOOBu);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][u], bw[0][j][u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][u], bw[1][j][u], acc1, 0, 0, 0);
      }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
      if (row < UNP) {
        sZ[row * ULDZ + li] = acc0[r];
        if (li < 16) sZ[row * ULDZ + 32 + li] = acc1[r];
      }
    }
  }
  __syncthreads();

  // ---- gather: 16 x 64 output pixels x 3 channels ----
  const int OW = 2 * a.W, OH = 2 * a.H;
  const float b0 = a.bias ? a.bias[0] : 0.f, b1 = a.bias ? a.bias[1] : 0.f, b2 = a.bias ? a.bias[2] : 0.f;
#pragma unroll
  for (int j = 0; j < (2 * UTH * 2 * UTW) / 256; ++j) {
    const int idx = tid + 256 * j, oy = idx >> 6, ox = idx & 63;
    const int qy = oy >> 1, qx = ox >> 1, cls = (oy & 1) * 2 + (ox & 1);
    float v0 = b0, v1 = b1, v2 = b2;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const int t16 = cls * 4 + ti;
      const float* z = sZ + ((qy + a.dy[t16] + 1) * UPW + qx + a.dx[t16] + 1) * ULDZ + t16 * 3;
      v0 += z[0]; v1 += z[1]; v2 += z[2];
    }
    const int Y = 2 * qy0 + oy, X = 2 * qx0 + ox;
    if (Y < OH && X < OW) {
      float* dst = a.Y + ((long)(b * OH + Y) * OW + X) * 3;
      if (a.act == ACT_TANH) {
        dst[0] = act_fwd(v0, ACT_TANH);
        dst[1] = act_fwd(v1, ACT_TANH);
        dst[2] = act_fwd(v2, ACT_TANH);
      } else {
        const float osl = act_slope(a.act);
        dst[0] = act_slope_fwd(v0, osl);
        dst[1] = act_slope_fwd(v1, osl);
        dst[2] = act_slope_fwd(v2, osl);
      }
    }
  }
}

}  // namespace

// ConvTranspose2d(64 -> 3, k4 s2 p1) forward geometry (kind 1): 4 parity classes x 4 taps, 2 x 2 input neighbourhood
bool upimg_supported(const ConvGeom& g) {
  if (!packed_weights(g) || g.wT != 0 || g.is != 1 || g.os != 2 || g.ncls != 4 || g.gC != 64 || g.sC != 3) return false;
  if (g.gH % UTH || g.gW % UTW || g.sH != 2 * g.gH || g.sW != 2 * g.gW) return false;
  for (int c = 0; c < 4; ++c) {
    if (g.ntaps[c] != 4 || g.py[c] != c / 2 || g.px[c] != c % 2) return false;
    for (int t = 0; t < 4; ++t) {
      const Tap& tp = g.taps[c][t];
      if (tp.dy < -1 || tp.dy > 1 || tp.dx < -1 || tp.dx > 1 || tp.wtap < 0 || tp.wtap > 15) return false;
    }
  }
  return (long)g.B * g.gH * g.gW * 64 < (1L << 29);
}

int launch_upimg_forward(const ConvGeom& g, const float* X, const float* Wp, const float* bias, float* Y, int act,
                         hipStream_t st) {
  if (!upimg_supported(g)) return kErrBadArg;
  UpImgArgs a{};
  a.X = X; a.Wp = Wp; a.bias = bias; a.Y = Y;
  a.B = g.B; a.H = g.gH; a.W = g.gW;
  a.tiles_y = g.gH / UTH; a.tiles_x = g.gW / UTW;
  a.act = act;
  for (int c = 0; c < 4; ++c)
    for (int t = 0; t < 4; ++t) {
      a.wtap[c * 4 + t] = g.taps[c][t].wtap;
      a.dy[c * 4 + t] = g.taps[c][t].dy;
      a.dx[c * 4 + t] = g.taps[c][t].dx;
    }
  const size_t smem = (size_t)UNP * ULDZ * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(upimg_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  const double px = (double)g.B * g.gH * g.gW;
  ProfScope ps("upimg_fwd_kernel", st, 2.0 * px * 64 * 48, 4.0 * px * (64 + 12));
  hipLaunchKernelGGL(upimg_fwd_kernel, dim3((unsigned)(g.B * a.tiles_y * a.tiles_x)), dim3(256), smem, st, a);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
