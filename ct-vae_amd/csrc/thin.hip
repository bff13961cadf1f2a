// Thin (3-channel output) layers: the image-side convolutions whose GEMM has N <= 4
// (vanilla_vae.py:73 final_layer.3; mcq_vae.py:233 decoder's last ConvTranspose2d, forward and wgrad;
// the 3-channel-INPUT side keeps the masked MFMA path: broadcasting 3 values per tap through vector loads is TA-bound).  An MFMA tile would be >=90 % padding there, so these run on the VALU with the
// wide channel dimension on the lanes (coalesced 128/256-B rows), the weights of a lane held in registers for
// the whole launch, and the narrow operand broadcast.  They are bandwidth/issue bound: roofline = HBM bytes.
//
//   narrow_out : S[m][n<NO] = sum_t sum_c      G[gpix][c] * W[t][c][n]          lanes = c/4, shuffle-reduced (NO <= 4)
//   wgrad_narrow_n : dW[t][c][n<NO] = sum_m X[gpix][c] * dY[spix][n]           lanes = c/4
// Geometry (taps, parity classes, weight orientation) is the same ConvGeom as the MFMA path (geom.hpp).
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

struct ThinArgs {
  ConvGeom g;
  const float* G;      // gathered operand (X for wgrad)
  const float* W;
  const float* bias;
  const float* add;
  const float* mask;
  const float* dY;     // wgrad only
  float* S;            // output (partials for wgrad)
  float* pbias;        // wgrad: bias partials [nwg][N]
  float* bn_part;      // narrow_in: per-workgroup (count, mean, M2) [nwg][N][3]
  int act, mask_act;
  int Mc, N;
  int groups_per_cls;  // pixel groups per class
};

__device__ __forceinline__ float wmat(const ConvGeom& g, const float* W, int wtap, int c, int n) {
  return g.wT ? W[((long)wtap * g.wCi + n) * g.wCo + c] : W[((long)wtap * g.wCi + c) * g.wCo + n];
}

// ---------------------------------------------------------------------------------------------------------
// narrow_out: LP = gC/4 lanes per pixel (each lane 4 channels), PPW = 64/LP pixels per wave step.
template <int LP, int NO, int TMAX>
__global__ __launch_bounds__(256) void thin_narrow_out_kernel(const ThinArgs a) {
  constexpr int PPW = 64 / LP;
  const ConvGeom& g = a.g;
  const int cls = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cq = lane % LP, slot = lane / LP;
  const int ntaps = g.ntaps[cls];

  float w[TMAX][4][NO];
  int tdy[TMAX], tdx[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    const bool tk = t < ntaps;
    const Tap tp = g.taps[cls][tk ? t : 0];
    tdy[t] = tp.dy;
    tdx[t] = tp.dx;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < NO; ++n) w[t][j][n] = tk ? wmat(g, a.W, tp.wtap, 4 * cq + j, n) : 0.f;
  }

  const int wave_global = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
  const int ngroups = (a.Mc + PPW - 1) / PPW;
  for (int grp = wave_global; grp < ngroups; grp += nwaves) {
    const int m = grp * PPW + slot;
    float acc[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) acc[n] = 0.f;
    int b = 0, qy = 0, qx = 0;
    const bool mok = m < a.Mc;
    if (mok) {
      decode_m(g, m, b, qy, qx);
      const int iy0 = qy * g.is, ix0 = qx * g.is;
      const long base = ((long)b * g.gH) * g.gW;
      // all tap loads are issued unconditionally (clamped address, zeroed afterwards) so they overlap
      f32x4 xv[TMAX];
#pragma unroll
      for (int t = 0; t < TMAX; ++t) {
        const int iy = iy0 + tdy[t], ix = ix0 + tdx[t];
        const bool ok = (t < ntaps) && (unsigned)iy < (unsigned)g.gH && (unsigned)ix < (unsigned)g.gW;
        const long off = ok ? (base + (long)iy * g.gW + ix) * g.gC : 0;
        f32x4 x = *reinterpret_cast<const f32x4*>(a.G + off + 4 * cq);
        if (!ok) x = f32x4{0.f, 0.f, 0.f, 0.f};
        xv[t] = x;
      }
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < NO; ++n) acc[n] += xv[t][j] * w[t][j][n];
    }
#pragma unroll
    for (int o = LP / 2; o > 0; o >>= 1)
#pragma unroll
      for (int n = 0; n < NO; ++n) acc[n] += __shfl_xor(acc[n], o, 64);
    if (mok && cq == 0) {
      const long idx0 = (long)scatter_pix(g, cls, b, qy, qx) * NO;
#pragma unroll
      for (int n = 0; n < NO; ++n) {
        float v = acc[n] + (a.bias != nullptr ? a.bias[n] : 0.f);
        if (a.add != nullptr) v += a.add[idx0 + n];
        v = act_fwd(v, a.act);
        if (a.mask != nullptr) v *= act_bwd_from_out(a.mask[idx0 + n], a.mask_act);
        a.S[idx0 + n] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// wgrad, narrow dY (NO <= 4 channels), wide X: lanes = channel quad of X.
template <int LP, int NO, int TMAX>
__global__ __launch_bounds__(256) void thin_wgrad_narrow_n_kernel(const ThinArgs a, int rows_total) {
  constexpr int PPW = 64 / LP;
  const ConvGeom& g = a.g;
  const int cls = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cq = lane % LP, slot = lane / LP;
  const int ntaps = g.ntaps[cls];
  int tdy[TMAX], tdx[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    const Tap tp = g.taps[cls][t < ntaps ? t : 0];
    tdy[t] = tp.dy;
    tdx[t] = tp.dx;
  }
  float acc[TMAX][4][NO];
#pragma unroll
  for (int t = 0; t < TMAX; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < NO; ++n) acc[t][j][n] = 0.f;
  float bsum[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) bsum[n] = 0.f;

  const int wave_global = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
  const int ngroups = (a.Mc + PPW - 1) / PPW;
  for (int grp = wave_global; grp < ngroups; grp += nwaves) {
    const int m = grp * PPW + slot;
    if (m < a.Mc) {
      int b, qy, qx;
      decode_m(g, m, b, qy, qx);
      const float* dyp = a.dY + (long)scatter_pix(g, cls, b, qy, qx) * NO;
      float dy[NO];
#pragma unroll
      for (int n = 0; n < NO; ++n) {
        dy[n] = dyp[n];
        bsum[n] += dy[n];
      }
      const int iy0 = qy * g.is, ix0 = qx * g.is;
      const long base = ((long)b * g.gH) * g.gW;
      f32x4 xv[TMAX];
#pragma unroll
      for (int t = 0; t < TMAX; ++t) {
        const int iy = iy0 + tdy[t], ix = ix0 + tdx[t];
        const bool ok = (t < ntaps) && (unsigned)iy < (unsigned)g.gH && (unsigned)ix < (unsigned)g.gW;
        const long off = ok ? (base + (long)iy * g.gW + ix) * g.gC : 0;
        f32x4 x = *reinterpret_cast<const f32x4*>(a.G + off + 4 * cq);
        if (!ok) x = f32x4{0.f, 0.f, 0.f, 0.f};
        xv[t] = x;
      }
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < NO; ++n) acc[t][j][n] += xv[t][j] * dy[n];
    }
  }
  // merge the PPW pixel slots (lanes cq + LP*s), then the waves
  __shared__ float sm[4][TMAX * 4 * NO][LP];
#pragma unroll
  for (int t = 0; t < TMAX; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < NO; ++n) {
        float v = acc[t][j][n];
#pragma unroll
        for (int o = LP; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        if (slot == 0) sm[wave][(t * 4 + j) * NO + n][cq] = v;
      }
  __syncthreads();
  for (int e = tid; e < TMAX * 4 * NO * LP; e += 256) {
    const int r = e / LP, q = e % LP;
    const int t = r / (4 * NO), j = (r / NO) % 4, n = r % NO;
    if (t < ntaps) {
      const float v = ((sm[0][r][q] + sm[1][r][q]) + sm[2][r][q]) + sm[3][r][q];
      const int wrow = g.taps[cls][t].wtap * g.gC + 4 * q + j;
      a.S[((long)blockIdx.x * rows_total + wrow) * NO + n] = v;
    }
  }
  if (a.pbias != nullptr) {
    __shared__ float sb[4][NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      float v = (cq == 0) ? bsum[n] : 0.f;   // every lane of a pixel slot saw the same dY: count it once
      v = wave_sum(v);
      if (lane == 0) sb[wave][n] = v;
    }
    __syncthreads();
    if (tid < NO) a.pbias[((long)blockIdx.x * g.ncls + cls) * NO + tid] = ((sb[0][tid] + sb[1][tid]) + sb[2][tid]) + sb[3][tid];
  }
}

// ---- host side ------------------------------------------------------------------------------------------
static int max_taps(const ConvGeom& g) {
  int m = 0;
  for (int c = 0; c < g.ncls; ++c) m = g.ntaps[c] > m ? g.ntaps[c] : m;
  return m;
}

bool thin_forward_supported(const ConvGeom& g) {
  const int N = g.sC, T = max_taps(g);
  if (N == 3 && (g.gC == 32 || g.gC == 64) && T <= 9 && g.wT == 0) return true;        // narrow_out (forward only)
  return false;
}

int thin_bn_parts(const ConvGeom& g) { return 0; }

int launch_thin_forward(const ConvGeom& g, const float* G, const float* W, const float* bias, const float* add,
                        const float* mask, int mask_act, float* S, int act, float* bn_part, hipStream_t st) {
  ThinArgs a{};
  a.g = g; a.G = G; a.W = W; a.bias = bias; a.add = add; a.mask = mask; a.S = S; a.bn_part = bn_part;
  a.act = act; a.mask_act = mask_act;
  a.Mc = g.B * g.Qh * g.Qw;
  a.N = g.sC;
  const int T = max_taps(g);
  dim3 grid(1024, g.ncls), block(256);
  const double macs = (double)a.Mc * a.N * g.gC * [&] { int s = 0; for (int c = 0; c < g.ncls; ++c) s += g.ntaps[c]; return s; }();
  const double bytes = 4.0 * ((double)g.B * g.gH * g.gW * g.gC + (double)g.B * g.sH * g.sW * g.sC);
  {
    if (bn_part != nullptr) return kErrBadArg;
    ProfScope ps("thin_narrow_out_kernel", st, 2.0 * macs, bytes);
    if (g.gC == 32) {
      if (T <= 4) hipLaunchKernelGGL((thin_narrow_out_kernel<8, 3, 4>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((thin_narrow_out_kernel<8, 3, 9>), grid, block, 0, st, a);
    } else {
      if (T <= 4) hipLaunchKernelGGL((thin_narrow_out_kernel<16, 3, 4>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((thin_narrow_out_kernel<16, 3, 9>), grid, block, 0, st, a);
    }
  }
  CTVAE_LAUNCH_CHECK();
  return 0;
}

bool thin_wgrad_supported(const ConvGeom& g) {
  const int N = g.sC, T = max_taps(g);
  if (g.wT != 0) return false;
  if (N == 3 && (g.gC == 32 || g.gC == 64) && T <= 9) return true;
  return false;
}

constexpr int kThinWgradWgs = 256;

size_t thin_wgrad_workspace_floats(const ConvGeom& g) {
  int taps = 0;
  for (int c = 0; c < g.ncls; ++c) taps += g.ntaps[c];
  return (size_t)kThinWgradWgs * ((size_t)taps * g.gC * g.sC + (size_t)g.ncls * g.sC);
}

// writes partials [kThinWgradWgs][rows_total*N] (+ bias partials) into ws; the caller reduces them
int launch_thin_wgrad(const ConvGeom& g, const float* X, const float* dY, float* ws, float** part_out, float** pbias_out,
                      int* nparts_w, int* nparts_b, bool want_bias, hipStream_t st) {
  ThinArgs a{};
  a.g = g; a.G = X; a.dY = dY;
  a.Mc = g.B * g.Qh * g.Qw;
  a.N = g.sC;
  int taps = 0;
  for (int c = 0; c < g.ncls; ++c) taps += g.ntaps[c];
  const int rows_total = taps * g.gC;
  a.S = ws;
  a.pbias = want_bias ? ws + (size_t)kThinWgradWgs * rows_total * a.N : nullptr;
  // every (workgroup, class) writes only its own class's rows: zero-fill is not needed because the reduce
  // below sums rows over workgroups and each row belongs to exactly one class -> all rows are written by all wgs
  const int T = max_taps(g);
  dim3 grid(kThinWgradWgs, g.ncls), block(256);
  const double macs = (double)a.Mc * a.N * rows_total;
  const double bytes = 4.0 * ((double)g.B * g.gH * g.gW * g.gC + (double)g.B * g.sH * g.sW * g.sC);
  {
    ProfScope ps("thin_wgrad_narrow_n_kernel", st, 2.0 * macs, bytes);
    if (g.gC == 32) {
      if (T <= 4) hipLaunchKernelGGL((thin_wgrad_narrow_n_kernel<8, 3, 4>), grid, block, 0, st, a, rows_total);
      else hipLaunchKernelGGL((thin_wgrad_narrow_n_kernel<8, 3, 9>), grid, block, 0, st, a, rows_total);
    } else {
      if (T <= 4) hipLaunchKernelGGL((thin_wgrad_narrow_n_kernel<16, 3, 4>), grid, block, 0, st, a, rows_total);
      else hipLaunchKernelGGL((thin_wgrad_narrow_n_kernel<16, 3, 9>), grid, block, 0, st, a, rows_total);
    }
  }
  CTVAE_LAUNCH_CHECK();
  *part_out = a.S;
  *pbias_out = a.pbias;
  *nparts_w = kThinWgradWgs;
  *nparts_b = kThinWgradWgs * g.ncls;
  return 0;
}

}  // namespace ctvae
