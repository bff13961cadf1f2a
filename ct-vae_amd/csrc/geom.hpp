// Tap-GEMM geometry shared by host launch code and device kernels (and by the host-side
// geometry emulation used in the CPU test-suite).
//
// Every convolution-like op on the hot path is the same gather-GEMM
//
//     S[spix(m)][n] = sum_t sum_c  G[gpix(m,t)][c] * Wmat[t][c][n]          m = (b, qy, qx)
//
// over NHWC tensors, where G is the *gathered* operand, S the *scattered* one:
//   gpix(m,t) = (b, qy*is + dy_t, qx*is + dx_t)   (zero outside the image)
//   spix(m)   = (b, qy*os + py,   qx*os + px)     per output-parity class (py,px)
//
//   Conv2d forward          (vanilla_vae.py:28, mcq_vae.py:170)  is=s os=1, 1 class
//   ConvTranspose2d forward (vanilla_vae.py:50, mcq_vae.py:223)  is=1 os=s, s*s classes
//   Conv2d dgrad            = ConvTranspose geometry, weights transposed per tap
//   ConvTranspose2d dgrad   = Conv geometry,          weights transposed per tap
//   Linear                  = 1x1 conv on a 1x1 image
// Weights always live in HBM as [tap][Ci_layer][Co_layer] (the packed layout behind the
// PyTorch-shaped parameter views).
#pragma once
#include <cstdlib>

#if defined(__HIPCC__)
#define CTVAE_HD __host__ __device__ __forceinline__
#else
#define CTVAE_HD inline
#endif

namespace ctvae {

constexpr int kMaxTaps = 16;
constexpr int kMaxCls = 4;

struct Tap {
  int dy, dx, wtap;
};

struct ConvGeom {
  int B;
  int gH, gW, gC;   // gathered tensor  [B,gH,gW,gC]
  int sH, sW, sC;   // scattered tensor [B,sH,sW,sC]
  int Qh, Qw;       // per-class grid of m
  int is, os;
  int ncls;
  int wCi, wCo;     // weight storage [ntaps_total][wCi][wCo]
  int wT;           // 0: Wmat[t][c][n] = W[t][c][n] (c<wCi,n<wCo); 1: Wmat[t][c][n] = W[t][n][c] (c<wCo,n<wCi)
  int wts, wrs;     // element W[t][r][j] sits at t*wts + r*wrs + j floats: (wCi*wCo, wCo) for the packed [tap][Ci][Co] layout,
                    // (wCo, ntaps*wCo) for [Ci][tap][Co] -- a Linear layer over torch.flatten(NCHW) read as a k x k convolution
  int ntaps[kMaxCls];
  int py[kMaxCls], px[kMaxCls];
  Tap taps[kMaxCls][kMaxTaps];
};

// m -> (b,qy,qx)
CTVAE_HD void decode_m(const ConvGeom& g, int m, int& b, int& qy, int& qx) {
  int qhw = g.Qh * g.Qw;
  b = m / qhw;
  int r = m - b * qhw;
  qy = r / g.Qw;
  qx = r - qy * g.Qw;
}

// pixel index (not multiplied by channels) in the gathered tensor, or -1 when outside
CTVAE_HD int gather_pix(const ConvGeom& g, int b, int qy, int qx, const Tap& t) {
  int iy = qy * g.is + t.dy, ix = qx * g.is + t.dx;
  if ((unsigned)iy >= (unsigned)g.gH || (unsigned)ix >= (unsigned)g.gW) return -1;
  return (b * g.gH + iy) * g.gW + ix;
}

CTVAE_HD bool packed_weights(const ConvGeom& g) { return g.wrs == g.wCo && g.wts == g.wCi * g.wCo; }

CTVAE_HD int scatter_pix(const ConvGeom& g, int cls, int b, int qy, int qx) {
  return (b * g.sH + qy * g.os + g.py[cls]) * g.sW + qx * g.os + g.px[cls];
}

// diagnostic: CTVAE_TAP_ORDER=0 keeps the plain (ky, kx) tap order of strided gathers
inline bool tap_order_grouped() {
#if defined(__HIP_DEVICE_COMPILE__)
  return true;
#else
  static const int v = [] { const char* e = getenv("CTVAE_TAP_ORDER"); return e ? atoi(e) : 1; }();
  return v != 0;
#endif
}

// ---- host-side builders -------------------------------------------------------------------
// kind: 0 conv fwd, 1 convT fwd, 2 conv dgrad, 3 convT dgrad.
// (B,H,W,Ci) is the LAYER's input tensor, Co the layer's output channels; k,s,p,op the layer's
// hyper-parameters.  Returns 0 on success, -1 on unsupported shapes.
inline int build_geom(ConvGeom& g, int kind, int B, int H, int W, int Ci, int Co, int k, int s, int p, int op) {
  g = ConvGeom{};
  const bool transposed_layer = (kind == 1 || kind == 3);
  int Ho, Wo;
  if (!transposed_layer) {
    Ho = (H + 2 * p - k) / s + 1;
    Wo = (W + 2 * p - k) / s + 1;
  } else {
    Ho = (H - 1) * s - 2 * p + k + op;
    Wo = (W - 1) * s - 2 * p + k + op;
  }
  if (Ho <= 0 || Wo <= 0 || k * k > 64 || s < 1 || s > 2) return -1;
  g.B = B;
  g.wCi = Ci;
  g.wCo = Co;
  g.wts = Ci * Co;
  g.wrs = Co;
  const bool fwd = (kind == 0 || kind == 1);
  g.wT = fwd ? 0 : 1;
  // "conv-like" geometry: gather with stride, scatter dense.  "convT-like": gather dense, scatter strided.
  const bool convlike = (kind == 0 || kind == 3);
  if (fwd) {
    g.gH = H; g.gW = W; g.gC = Ci; g.sH = Ho; g.sW = Wo; g.sC = Co;
  } else {
    g.gH = Ho; g.gW = Wo; g.gC = Co; g.sH = H; g.sW = W; g.sC = Ci;
  }
  if (convlike) {
    // scattered index q; gathered = q*s + (ky - p)
    g.is = s; g.os = 1; g.ncls = 1;
    g.Qh = g.sH; g.Qw = g.sW;
    g.py[0] = g.px[0] = 0;
    if (k * k > kMaxTaps) return -1;
    int n = 0;
    if (s == 2 && tap_order_grouped()) {
      // Stride 2: taps whose offsets have the same parity read the SAME rows / columns of the gathered tensor, one output
      // pixel apart (ky = 0 and ky = 2 of a 3x3 both read the odd rows).  In plain (ky, kx) order a workgroup comes back to a
      // row several K chunks later -- with ~128 workgroups per L2 that is past the L2's capacity on the large tensors and the
      // row is fetched again (PMC: final_layer.0's data gradient fetched 2.1 x its 134 MB operand).  Grouped by parity the
      // re-read follows at once and hits the L2.  The weights follow their tap (wtap), results differ only in summation order.
      for (int qy = 0; qy < 2; ++qy)
        for (int qx = 0; qx < 2; ++qx)
          for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx)
              if ((((ky - p) & 1) == (qy ^ 1)) && (((kx - p) & 1) == (qx ^ 1))) g.taps[0][n++] = Tap{ky - p, kx - p, ky * k + kx};
    } else {
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) g.taps[0][n++] = Tap{ky - p, kx - p, ky * k + kx};
    }
    g.ntaps[0] = n;
  } else {
    // scattered index o = q*s + par; contributions from gathered i with o = i*s - p + ky  =>  i = q + (par + p - ky)/s
    g.is = 1; g.os = s; g.ncls = s * s;
    if (g.sH % s || g.sW % s) return -1;
    g.Qh = g.sH / s; g.Qw = g.sW / s;
    for (int c = 0; c < g.ncls; ++c) {
      int py = c / s, px = c % s, n = 0;
      g.py[c] = py; g.px[c] = px;
      for (int ky = 0; ky < k; ++ky) {
        if ((py + p - ky) % s != 0) continue;
        for (int kx = 0; kx < k; ++kx) {
          if ((px + p - kx) % s != 0) continue;
          if (n >= kMaxTaps) return -1;
          // C++ division truncates toward zero; (par+p-ky) is an exact multiple of s here
          g.taps[c][n++] = Tap{(py + p - ky) / s, (px + p - kx) / s, ky * k + kx};
        }
      }
      g.ntaps[c] = n;
    }
  }
  return 0;
}

}  // namespace ctvae
