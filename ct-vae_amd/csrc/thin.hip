// Thin (3-channel output) layers: the image-side convolutions whose GEMM has N <= 4
// (vanilla_vae.py:73 final_layer.3; mcq_vae.py:233 the decoder's last ConvTranspose2d), forward and wgrad.
// An MFMA tile would be >=90 % padding there, so they run on the VALU -- and since they move the largest
// activations of the nets (32x64x64 per image) they are built around the memory system:
//   * one workgroup = one TH x TW tile of output pixels of one image (and one output-parity class); the
//     input patch (tile + halo) x all channels is staged ONCE into LDS with coalesced 16-B loads, so every
//     input element leaves L2 once instead of once per tap (9x for the 3x3 conv);
//   * lanes = (pixel slot, 4-channel group): each tap is one ds_read_b128 per lane, the weights of a lane's
//     4 channels x taps x 3 outputs stay in registers for the whole launch;
//   * forward: shuffle-reduce over the channel groups, bias + tanh, 12 B per pixel out;
//   * wgrad: per-lane accumulators over all the workgroup's tiles (persistent loop), then a fixed-order merge of
//     pixel slots / waves -> one partial slab per workgroup, reduced by the caller (deterministic);
//   * optional on-the-fly input transform a = act(x*scale[c] + shift[c]) while staging (fused BatchNorm apply).
// The 3-channel-INPUT side keeps the masked MFMA path: broadcasting 3 values per tap through vector loads is TA-bound.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

struct ThinArgs {
  ConvGeom g;
  const float* G;       // gathered operand (layer input X)
  const float* W;
  const float* bias;
  const float* dY;      // wgrad only
  const float* in_scale;  // optional per-channel affine + activation applied to G while staging
  const float* in_shift;
  float* S;             // forward output / wgrad partial slabs
  float* pbias;         // wgrad: bias partials [nwg][NO]
  int in_act;
  int act;
  int tiles_y, tiles_x;   // tiles per image and class
  int ntiles;             // B * tiles_y * tiles_x
  int dmin_y, dmin_x, PH, PW;
  unsigned inv_PW;        // ceil(65536 / PW): pp / PW == (pp * inv_PW) >> 16 for pp < 900
  int rows_total;
};

constexpr unsigned kOOBt = 0x80000000u;

// TH x TW tile, LP = gC/4 lanes per pixel.  LDS patch: [PH*PW][gC + 4] floats.
template <int LP, int NO, int TMAX, int TH, int TW, bool WGRAD>
__global__ __launch_bounds__(256) void thin_tile_kernel(const ThinArgs a) {
  kernarg_warm<sizeof(ThinArgs)>();
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int GC = LP * 4, LDP = GC + 4, PPW = 64 / LP, NPIX = TH * TW;
  constexpr int ITERS = NPIX / (4 * PPW);
  static_assert(NPIX % (4 * PPW) == 0, "tile must be a multiple of the pixels one pass covers");
  float* sP = smem;                                  // patch
  float* sDY = smem + a.PH * a.PW * LDP;             // wgrad: dY of the tile [NPIX][4]
  const ConvGeom& g = a.g;
  const int cls = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cq = lane % LP, slot = lane / LP;
  const int ntaps = g.ntaps[cls];

  float w[WGRAD ? 1 : TMAX][4][NO];
  float acc_w[WGRAD ? TMAX : 1][4][NO];
  int tofs[TMAX];                                    // LDS float offset of tap t relative to the pixel's patch origin
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    const bool tk = t < ntaps;
    const Tap tp = g.taps[cls][tk ? t : 0];
    tofs[t] = ((tp.dy - a.dmin_y) * a.PW + (tp.dx - a.dmin_x)) * LDP;
    if constexpr (!WGRAD) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < NO; ++n) w[t][j][n] = tk ? a.W[((long)tp.wtap * g.wCi + 4 * cq + j) * g.wCo + n] : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < NO; ++n) acc_w[t][j][n] = 0.f;
    }
  }
  float bsum[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) bsum[n] = 0.f;
  float bias_r[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) bias_r[n] = (!WGRAD && a.bias != nullptr) ? a.bias[n] : 0.f;

  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.G), 0, (int)((long)g.B * g.gH * g.gW * GC * 4), 0x00020000);
  const bool xform = a.in_scale != nullptr;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / (a.tiles_y * a.tiles_x);
    const int tr = tile - b * (a.tiles_y * a.tiles_x);
    const int qy0 = (tr / a.tiles_x) * TH, qx0 = (tr % a.tiles_x) * TW;
    const int iy_lo = qy0 * g.is + a.dmin_y, ix_lo = qx0 * g.is + a.dmin_x;
    __syncthreads();                                 // previous tile fully consumed
    // ---- stage the input patch (and dY) ---------------------------------------------------------------
    const int nvec = a.PH * a.PW * LP;
    // four 16-byte loads per thread are issued before the first is used (one load -> wait -> LDS store per iteration, as this loop
    // was, is a memory round trip per 4 KB of the patch); a slot past the patch loads the out-of-range offset and is not stored
    for (int e0 = 0; e0 < nvec; e0 += 4 * 256) {
      f32x4 v[4];
      int pp[4], c4[4];
      bool ok[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = e0 + tid + 256 * j;
        pp[j] = e / LP;
        c4[j] = e - pp[j] * LP;
        const int py = (int)(((unsigned)pp[j] * a.inv_PW) >> 16), px = pp[j] - py * a.PW;   // no integer division in the hot loop
        const int iy = iy_lo + py, ix = ix_lo + px;
        ok[j] = e < nvec && (unsigned)iy < (unsigned)g.gH && (unsigned)ix < (unsigned)g.gW;
        const unsigned off = ok[j] ? (unsigned)(((b * g.gH + iy) * g.gW + ix) * GC + 4 * c4[j]) * 4u : kOOBt;
        v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, (int)off, 0, 0));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (e0 + tid + 256 * j < nvec) {
          if (xform) {
            const f32x4 tsc = *reinterpret_cast<const f32x4*>(a.in_scale + 4 * c4[j]);
            const f32x4 tsh = *reinterpret_cast<const f32x4*>(a.in_shift + 4 * c4[j]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // padding stays 0
              const float t = v[j][k] * tsc[k] + tsh[k];
              v[j][k] = ok[j] ? (a.in_act == ACT_TANH ? act_fwd(t, ACT_TANH) : act_slope_fwd(t, act_slope(a.in_act))) : 0.f;
            }
          }
          *reinterpret_cast<f32x4*>(&sP[pp[j] * LDP + 4 * c4[j]]) = v[j];
        }
      }
    }
    if constexpr (WGRAD) {
      for (int e = tid; e < NPIX; e += 256) {
        const int ly = e / TW, lx = e - ly * TW;
        const int qy = qy0 + ly, qx = qx0 + lx;
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        if (qy < g.Qh && qx < g.Qw) {
          const float* p = a.dY + (long)scatter_pix(g, cls, b, qy, qx) * NO;
#pragma unroll
          for (int n = 0; n < NO; ++n) d[n] = p[n];
        }
        *reinterpret_cast<f32x4*>(&sDY[e * 4]) = d;
      }
    }
    __syncthreads();
    // ---- compute ---------------------------------------------------------------------------------------
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
      const int p = it * (4 * PPW) + wave * PPW + slot;
      const int ly = p / TW, lx = p - ly * TW;
      const float* base = sP + ((ly * g.is) * a.PW + lx * g.is) * LDP + 4 * cq;
      f32x4 xv[TMAX];
#pragma unroll
      for (int t = 0; t < TMAX; ++t) xv[t] = (t < ntaps) ? *reinterpret_cast<const f32x4*>(base + tofs[t]) : f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (!WGRAD) {
        float acc[NO];
#pragma unroll
        for (int n = 0; n < NO; ++n) acc[n] = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NO; ++n) acc[n] += xv[t][j] * w[t][j][n];
#pragma unroll
        for (int o = LP / 2; o > 0; o >>= 1)
#pragma unroll
          for (int n = 0; n < NO; ++n) acc[n] += __shfl_xor(acc[n], o, 64);
        // after the butterfly every lane of the slot holds the sums: lane cq < NO finishes output channel cq
        const int qy = qy0 + ly, qx = qx0 + lx;
        float mine = acc[0] + bias_r[0];
#pragma unroll
        for (int n = 1; n < NO; ++n) mine = (cq == n) ? acc[n] + bias_r[n] : mine;
        mine = act_fwd(mine, a.act);
        if (cq < NO && qy < g.Qh && qx < g.Qw) a.S[(long)scatter_pix(g, cls, b, qy, qx) * NO + cq] = mine;
      } else {
        const f32x4 d = *reinterpret_cast<const f32x4*>(&sDY[p * 4]);
#pragma unroll
        for (int n = 0; n < NO; ++n) bsum[n] += d[n];
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NO; ++n) acc_w[t][j][n] += xv[t][j] * d[n];
      }
    }
  }

  if constexpr (WGRAD) {
    // merge the PPW pixel slots (lanes cq + LP*s) by shuffles, then the 4 waves through LDS, in a fixed order
    __syncthreads();
    float* sm = smem;                                  // [4][TMAX*4*NO][LP]
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < NO; ++n) {
          float v = acc_w[t][j][n];
#pragma unroll
          for (int o = LP; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
          if (slot == 0) sm[(wave * (TMAX * 4 * NO) + (t * 4 + j) * NO + n) * LP + cq] = v;
        }
    float bs[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) bs[n] = wave_sum(cq == 0 ? bsum[n] : 0.f);   // every lane of a slot saw the same dY
    __syncthreads();
    for (int e = tid; e < TMAX * 4 * NO * LP; e += 256) {
      const int r = e / LP, q = e - r * LP;
      const int t = r / (4 * NO), j = (r / NO) % 4, n = r % NO;
      if (t < ntaps) {
        constexpr int R = TMAX * 4 * NO;
        const float v = ((sm[(0 * R + r) * LP + q] + sm[(1 * R + r) * LP + q]) + sm[(2 * R + r) * LP + q]) + sm[(3 * R + r) * LP + q];
        const int wrow = g.taps[cls][t].wtap * GC + 4 * q + j;
        a.S[((long)blockIdx.x * a.rows_total + wrow) * NO + n] = v;
      }
    }
    if (a.pbias != nullptr) {
      __syncthreads();
      if (lane == 0)
#pragma unroll
        for (int n = 0; n < NO; ++n) sm[wave * NO + n] = bs[n];
      __syncthreads();
      if (tid < NO) a.pbias[((long)blockIdx.x * g.ncls + cls) * NO + tid] = ((sm[tid] + sm[NO + tid]) + sm[2 * NO + tid]) + sm[3 * NO + tid];
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------
static int max_taps(const ConvGeom& g) {
  int m = 0;
  for (int c = 0; c < g.ncls; ++c) m = g.ntaps[c] > m ? g.ntaps[c] : m;
  return m;
}

static bool thin_shape_ok(const ConvGeom& g) {
  if (!packed_weights(g) || g.wT != 0 || g.sC != 3) return false;
  if (!(g.gC == 32 || g.gC == 64)) return false;
  const int TH = 8, TW = g.gC == 32 ? 32 : 16;
  if (g.Qh % TH != 0 || g.Qw % TW != 0) return false;
  return max_taps(g) <= 9;
}

bool thin_forward_supported(const ConvGeom& g) { return thin_shape_ok(g); }
bool thin_wgrad_supported(const ConvGeom& g) { return thin_shape_ok(g); }
int thin_bn_parts(const ConvGeom&) { return 0; }

constexpr int kThinWgs = 768;   // persistent workgroups per class (wgrad partial slabs): 3 per CU hide the LDS/VALU latency

size_t thin_wgrad_workspace_floats(const ConvGeom& g) {
  int taps = 0;
  for (int c = 0; c < g.ncls; ++c) taps += g.ntaps[c];
  return (size_t)kThinWgs * ((size_t)taps * g.gC * g.sC + (size_t)g.ncls * g.sC);
}

template <bool WGRAD>
static int launch_thin(ThinArgs& a, int nwg, hipStream_t st, const char* pname) {
  const ConvGeom& g = a.g;
  const int TH = 8, TW = g.gC == 32 ? 32 : 16;
  int dmin_y = 1 << 20, dmax_y = -(1 << 20), dmin_x = 1 << 20, dmax_x = -(1 << 20);
  for (int c = 0; c < g.ncls; ++c)
    for (int t = 0; t < g.ntaps[c]; ++t) {
      const Tap& tp = g.taps[c][t];
      dmin_y = tp.dy < dmin_y ? tp.dy : dmin_y; dmax_y = tp.dy > dmax_y ? tp.dy : dmax_y;
      dmin_x = tp.dx < dmin_x ? tp.dx : dmin_x; dmax_x = tp.dx > dmax_x ? tp.dx : dmax_x;
    }
  a.dmin_y = dmin_y; a.dmin_x = dmin_x;
  a.PH = (TH - 1) * g.is + (dmax_y - dmin_y) + 1;
  a.PW = (TW - 1) * g.is + (dmax_x - dmin_x) + 1;
  if (a.PH * a.PW >= 900) return kErrBadArg;
  a.inv_PW = (65536u + a.PW - 1) / a.PW;
  a.tiles_y = g.Qh / TH; a.tiles_x = g.Qw / TW;
  a.ntiles = g.B * a.tiles_y * a.tiles_x;
  int taps = 0;
  for (int c = 0; c < g.ncls; ++c) taps += g.ntaps[c];
  a.rows_total = taps * g.gC;
  const int T = max_taps(g);
  const int LDP = g.gC + 4;
  size_t smem = (size_t)a.PH * a.PW * LDP * 4 + (WGRAD ? (size_t)TH * TW * 16 : 0);
  const size_t merge = WGRAD ? (size_t)4 * (T <= 4 ? 4 : 9) * 4 * 3 * (g.gC / 4) * 4 : 0;
  if (merge > smem) smem = merge;
  if (smem > 160 * 1024) return kErrBadArg;
  dim3 grid(nwg < a.ntiles ? nwg : a.ntiles, g.ncls), block(256);
  const double macs = (double)g.B * g.Qh * g.Qw * g.sC * g.gC * taps;
  const double bytes = 4.0 * ((double)g.B * g.gH * g.gW * g.gC + (double)g.B * g.sH * g.sW * g.sC);
  ProfScope ps(pname, st, 2.0 * macs, bytes);
#define CTVAE_THIN(LP_, TM_, TW_)                                                                                     \
  do {                                                                                                                \
    auto kern = thin_tile_kernel<LP_, 3, TM_, 8, TW_, WGRAD>;                                                         \
    if (smem > 64 * 1024)                                                                                             \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL(kern, grid, block, smem, st, a);                                                               \
  } while (0)
  if (g.gC == 32) {
    if (T <= 4) CTVAE_THIN(8, 4, 32);
    else CTVAE_THIN(8, 9, 32);
  } else {
    if (T <= 4) CTVAE_THIN(16, 4, 16);
    else CTVAE_THIN(16, 9, 16);
  }
#undef CTVAE_THIN
  CTVAE_LAUNCH_CHECK();
  return 0;
}

bool img_conv_supported(const ConvGeom& g);
int launch_img_forward(const ConvGeom& g, const float* X, const float* W, const float* bias, float* S, int act,
                       const InXform* xf, hipStream_t st);
int launch_img_wgrad(const ConvGeom& g, const float* X, const float* dY, float* ws, float** part_out, float** pbias_out,
                     int* nparts, bool want_bias, const InXform* xf, hipStream_t st);

bool upimg_supported(const ConvGeom& g);
int launch_upimg_forward(const ConvGeom& g, const float* X, const float* Wp, const float* bias, float* Y, int act,
                         hipStream_t st);

int launch_thin_forward(const ConvGeom& g, const float* G, const float* W, const float* bias, const float* add,
                        const float* mask, int mask_act, float* S, int act, float* bn_part, hipStream_t st,
                        const InXform* xf) {
  if (add != nullptr || mask != nullptr || bn_part != nullptr) return kErrBadArg;
  if (img_conv_supported(g)) return launch_img_forward(g, G, W, bias, S, act, xf, st);   // 3x3 s1, 32 -> 3: MFMA form (image.hip)
  if (upimg_supported(g) && (xf == nullptr || xf->scale == nullptr))                     // ConvT k4 s2, 64 -> 3: MFMA form (upimg.hip)
    return launch_upimg_forward(g, G, W, bias, S, act, st);
  ThinArgs a{};
  a.g = g; a.G = G; a.W = W; a.bias = bias; a.S = S; a.act = act;
  if (xf != nullptr && xf->scale != nullptr) { a.in_scale = xf->scale; a.in_shift = xf->shift; a.in_act = xf->act; }
  return launch_thin<false>(a, 2048, st, "thin_tile_kernel<fwd>");
}

// writes partials [nwg][rows_total*N] (+ bias partials) into ws; the caller reduces them
int launch_thin_wgrad(const ConvGeom& g, const float* X, const float* dY, float* ws, float** part_out, float** pbias_out,
                      int* nparts_w, int* nparts_b, bool want_bias, hipStream_t st, const InXform* xf) {
  if (img_conv_supported(g)) {
    int np = 0;
    const int rc = launch_img_wgrad(g, X, dY, ws, part_out, pbias_out, &np, want_bias, xf, st);
    *nparts_w = np;
    *nparts_b = np;
    return rc;
  }
  ThinArgs a{};
  a.g = g; a.G = X; a.dY = dY;
  if (xf != nullptr && xf->scale != nullptr) { a.in_scale = xf->scale; a.in_shift = xf->shift; a.in_act = xf->act; }
  int taps = 0;
  for (int c = 0; c < g.ncls; ++c) taps += g.ntaps[c];
  const int rows_total = taps * g.gC;
  const int ntiles = g.B * (g.Qh / 8) * (g.Qw / (g.gC == 32 ? 32 : 16));
  const int nwg = ntiles < kThinWgs ? ntiles : kThinWgs;
  a.S = ws;
  a.pbias = want_bias ? ws + (size_t)nwg * rows_total * g.sC : nullptr;
  int rc = launch_thin<true>(a, nwg, st, "thin_tile_kernel<wgrad>");
  if (rc) return rc;
  *part_out = a.S;
  *pbias_out = a.pbias;
  *nparts_w = nwg;
  *nparts_b = nwg * g.ncls;
  return 0;
}

}  // namespace ctvae
