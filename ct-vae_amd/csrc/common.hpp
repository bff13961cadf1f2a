// Common device/host helpers for the ctvae HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "geom.hpp"

namespace ctvae {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float kLeaky = 0.01f;  // nn.LeakyReLU() default (vanilla_vae.py:31)

enum Act : int { ACT_NONE = 0, ACT_LRELU = 1, ACT_RELU = 2, ACT_TANH = 3 };

__device__ __forceinline__ float act_fwd(float v, int act) {
  switch (act) {
    case ACT_LRELU: return v > 0.f ? v : v * kLeaky;
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_TANH: return 1.f - 2.f * __frcp_rn(__expf(2.f * v) + 1.f);   // |err| < 1e-6: hardware exp2 / rcp
    default: return v;
  }
}

// derivative expressed through the saved post-activation output o
__device__ __forceinline__ float act_bwd_from_out(float o, int act) {
  switch (act) {
    case ACT_LRELU: return o > 0.f ? 1.f : kLeaky;
    case ACT_RELU: return o > 0.f ? 1.f : 0.f;
    case ACT_TANH: return 1.f - o * o;
    default: return 1.f;
  }
}

// Launch-uniform activations in unrolled element loops: `act_fwd(v, act)` per element compiles to a branch tree per element (the
// tanh code behind it); hot loops decide ONCE -- `if (act == ACT_TANH) { loop with tanh } else { loop with act_slope_fwd }` --
// and apply identity / LeakyReLU / ReLU as one select: t > 0 ? t : t * slope (slope 1 / 0.01 / 0; ReLU yields -0 for t < 0).
__device__ __forceinline__ float act_slope(int act) { return act == ACT_LRELU ? kLeaky : (act == ACT_RELU ? 0.f : 1.f); }
__device__ __forceinline__ float act_slope_fwd(float v, float slope) { return v > 0.f ? v : v * slope; }
// d act / d t at pre-activation t (equals act_bwd_from_out(act_fwd(t))): t > 0 ? 1 : slope
__device__ __forceinline__ float act_slope_bwd(float t, float slope) { return t > 0.f ? 1.f : slope; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// The same sum for a FULL wave in convergent code, on the DPP data path: four adds inside the 16-lane rows, two row broadcasts,
// one v_readlane of lane 63 -- seven instructions instead of six dependent ds_bpermute round trips through the LDS crossbar
// (~130 cycles each: the two block sums of bn_fused_fwd_kernel were 1.2 us apiece, the four-value one of bn_fused_bwd_kernel
// 3.3 us, in-kernel clocks, round 3).  Fixed summation order; every lane gets the same bits.
__device__ __forceinline__ float wave_sum_full(float v) {
  auto bi = [](float x) { return __builtin_bit_cast(int, x); };
  auto bf = [](int x) { return __builtin_bit_cast(float, x); };
  v += bf(__builtin_amdgcn_update_dpp(0, bi(v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
  v += bf(__builtin_amdgcn_update_dpp(0, bi(v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
  v += bf(__builtin_amdgcn_update_dpp(0, bi(v), 0x141, 0xF, 0xF, true));    // row_half_mirror
  v += bf(__builtin_amdgcn_update_dpp(0, bi(v), 0x140, 0xF, 0xF, true));    // row_mirror: every lane holds its row's sum
  v += bf(__builtin_amdgcn_update_dpp(0, bi(v), 0x142, 0xA, 0xF, false));   // row_bcast:15 into rows 1 and 3
  v += bf(__builtin_amdgcn_update_dpp(0, bi(v), 0x143, 0xC, 0xF, false));   // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return bf(__builtin_amdgcn_readlane(bi(v), 63));
}

// double-precision counterpart of wave_sum_full (two 32-bit DPP moves per step)
__device__ __forceinline__ double wave_sum_d_full(double v) {
#define CTVAE_DPP_D(CTRL, RM, BC)                                                                          \
  {                                                                                                        \
    const long long b = __builtin_bit_cast(long long, v);                                                  \
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, RM, 0xF, BC);                               \
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, RM, 0xF, BC);                       \
    v += __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);                                 \
  }
  CTVAE_DPP_D(0xB1, 0xF, true)
  CTVAE_DPP_D(0x4E, 0xF, true)
  CTVAE_DPP_D(0x141, 0xF, true)
  CTVAE_DPP_D(0x140, 0xF, true)
  CTVAE_DPP_D(0x142, 0xA, false)
  CTVAE_DPP_D(0x143, 0xC, false)
#undef CTVAE_DPP_D
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* sm /* >=4 floats */) {
  v = wave_sum_full(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Touch every 64-byte line of the kernel-argument segment with ONE batch of scalar loads.  Kernels with large argument structs
// (geometry, tap tables, a dozen pointers) read them in the order of use: class index -> per-class counts -> tile constants ->
// tap table ..., each a dependent round trip to memory behind the launch's cache invalidate -- eight in front of the tile
// kernel's first operand load, 2.9 us per workgroup of a 9 us launch (tools/phase_probe.py, round 3).  After this batch they
// are scalar-cache hits (and L2 hits for the lane-indexed table load).  BYTES: explicit arguments (every load starts inside
// them: the implicit arguments behind them exist only in kernels that use them).
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
#if defined(__HIP_DEVICE_COMPILE__)
  const void* ka = __builtin_amdgcn_kernarg_segment_ptr();
  unsigned junk;
  asm volatile(
      ".set .Lkw_off, 0\n"
      ".rept %2\n"
      "s_load_dword %0, %1, .Lkw_off\n"
      ".set .Lkw_off, .Lkw_off + 64\n"
      ".endr\n"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(junk)
      : "s"(ka), "n"((BYTES + 63) / 64)
      : "memory");
#endif
}
// The same batch, which also returns the three argument dwords at byte offsets O0, O1, O2 -- the ones the kernel branches on
// first.  Read as ordinary arguments they are loads the compiler hoists ABOVE the batch (kernel arguments are invariant memory
// to it): one more round trip in front of it.
template <int BYTES, int O0, int O1, int O2>
__device__ __forceinline__ void kernarg_warm_get(int& v0, int& v1, int& v2) {
#if defined(__HIP_DEVICE_COMPILE__)
  const void* ka = __builtin_amdgcn_kernarg_segment_ptr();
  unsigned junk;
  asm volatile(
      "s_load_dword %1, %4, %6\n"
      "s_load_dword %2, %4, %7\n"
      "s_load_dword %3, %4, %8\n"
      ".set .Lkw_off, 0\n"
      ".rept %5\n"
      "s_load_dword %0, %4, .Lkw_off\n"
      ".set .Lkw_off, .Lkw_off + 64\n"
      ".endr\n"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(junk), "=&s"(v0), "=&s"(v1), "=&s"(v2)
      : "s"(ka), "n"((BYTES + 63) / 64), "n"(O0), "n"(O1), "n"(O2)
      : "memory");
#endif
}

// Lazy BatchNorm apply: the operand a kernel gathers is x' = act(x*scale[c] + shift[c]) of the tensor it is given
// (padding stays 0).  Only the thin (3-output-channel) kernels implement it: they stage every input element once.
struct InXform {
  const float* scale;
  const float* shift;
  int act;
};

// Lazy BatchNorm-backward apply: a weight-gradient kernel is handed g_a (gradient w.r.t. the BatchNorm+activation
// output) instead of g_y and forms g_y = k1*g_a*act'(y*scale+shift) + k2*y + k3 itself while loading, also writing it
// to gy_out for the data-gradient kernel that follows.  coef = [5][C]: k1, k2, k3, scale, shift (bn_bwd_finalize_kernel).
struct DyXform {
  const float* y;
  const float* coef;
  float* gy_out;     // may be null where no data gradient follows (image.hip img_enc_wgrad_kernel)
  int act;
  float* dgamma = nullptr;   // with gy_out null: the kernel also commits the BatchNorm's parameter gradients (coef rows 5, 6)
  float* dbeta = nullptr;
  int bn_accumulate = 0;
};

// Winograd filter hand-over between a layer's forward launch and its data-gradient launch (wino.hip)
struct WinoFilters {
  const float* ready;   // dgrad: the filter set the forward launch left behind (skip the transform)
  float* bwd_out;       // forward: where to leave the data gradient's filter set
};

// BatchNorm(+activation) layer whose output gradient a dgrad launch produces (fused backward sums)
struct BnBwdFuse {
  const float* y;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  int act;
  float* part;   // [rows][C][2], rows = tapgemm_bnb_rows()
};

// Split-K launch whose consumer sums the slices itself (bn.hip bn_fused_fwd/bwd_kernel): the tile kernel leaves its raw
// accumulators CHANNEL-MAJOR in part[S][N][R] (R = B*sH*sW scattered pixels; a lane's four consecutive rows are one 16-byte
// store, and the channel owners of the consumer read their columns as contiguous runs) and no finishing launch is issued.
// splitk is filled by the launcher: <= 1 means the plan did not split and the launch ran as usual (part untouched).
// Rows are in class-major order r' = cls * (B*Qh*Qw) + m of the launch's geometry (RowMap: r' -> scattered pixel).
struct SplitKRaw {
  float* part;
  int splitk;
  int pixel_major = 0;   // 1: the ordinary layout part[S][R][N] (pixel rows as in the output tensor) -- for element-wise consumers
                         // that sum the slices while they read their operand (gauss_latent_*, splitk_permute_kernel)
};

// r' = cls * Mc + m (m = (b, qy, qx) of a ConvGeom's class grid) -> pixel index in the scattered tensor [B][sH][sW]
struct RowMap {
  int ncls, Mc, Qh, Qw, sH, sW, os;
  unsigned pyp, pxp;     // the classes' parity offsets, 8 bits each: a table indexed by a per-thread class is a LOAD from the kernel
                         // arguments in front of every row's own load (two dependent round trips per row)
  int lgMc, lgQhw, lgQw; // log2 of Mc, Qh*Qw, Qw when ALL THREE are powers of two (the models' shapes), else -1: three integer
                         // divisions per row were ~100 vector instructions in front of every row's access (bn_fused_*: 16 waves per CU)
};

inline RowMap row_map_of(const ConvGeom& g) {
  RowMap m{};
  m.ncls = g.ncls; m.Mc = g.B * g.Qh * g.Qw; m.Qh = g.Qh; m.Qw = g.Qw; m.sH = g.sH; m.sW = g.sW; m.os = g.os;
  static_assert(kMaxCls <= 4, "RowMap packs four classes");
  for (int c = 0; c < kMaxCls; ++c) { m.pyp |= (unsigned)(g.py[c] & 0xff) << (8 * c); m.pxp |= (unsigned)(g.px[c] & 0xff) << (8 * c); }
  auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (v > 0 && (1 << l) == v) ? l : -1; };
  m.lgMc = lg2(m.Mc); m.lgQhw = lg2(g.Qh * g.Qw); m.lgQw = lg2(g.Qw);
  if (m.lgMc < 0 || m.lgQhw < 0 || m.lgQw < 0) m.lgMc = m.lgQhw = m.lgQw = -1;
  return m;
}

__device__ __forceinline__ int row_map_pixel(const RowMap& m, int r) {
  if (m.ncls == 1) return r;
  int cls, b, qy, qx;
  if (m.lgMc >= 0) {
    cls = r >> m.lgMc;
    const int mm = r & (m.Mc - 1);
    b = mm >> m.lgQhw;
    const int rr = mm & ((1 << m.lgQhw) - 1);
    qy = rr >> m.lgQw;
    qx = rr & (m.Qw - 1);
  } else {
    cls = r / m.Mc;
    const int mm = r - cls * m.Mc;
    const int qhw = m.Qh * m.Qw;
    b = mm / qhw;
    const int rr = mm - b * qhw;
    qy = rr / m.Qw;
    qx = rr - qy * m.Qw;
  }
  const int py = (int)((m.pyp >> (8 * cls)) & 0xffu), px = (int)((m.pxp >> (8 * cls)) & 0xffu);
  return (b * m.sH + qy * m.os + py) * m.sW + qx * m.os + px;
}

struct TapGemmPlan {
  int BM, BN, mtiles, ntiles, splitk;
  int thin;      // 1: VALU thin-layer kernels (thin.hip) instead of the MFMA tile kernel
  int bn_parts;  // rows of (count, mean, M2) partials a BN-statistics epilogue would write
};

#define CTVAE_LAUNCH_CHECK()                   \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)

// argument error (distinct from hipError_t values, which are small positive numbers)
constexpr int kErrBadArg = -22;
constexpr int kErrWorkspace = -12;

}  // namespace ctvae
