// TEST INFRASTRUCTURE: host emulation of the tap-GEMM *index arithmetic* (ct-vae_amd/csrc/geom.hpp) with
// naive loops, so that the geometry tables (tap lists, parity classes, weight orientation) can be
// checked against torch's conv ops in the CPU-only test-suite.  Not part of the product library.
#include <cstring>
#include <vector>

#include "../../ct-vae_amd/csrc/geom.hpp"

using namespace ctvae;

extern "C" {

// kind | 0x100: the weight block is [Ci][taps][Co] (CTVAE_W_CI_TAP of include/ctvae_hip.h, api.hip conv_geom)
static int geom_of(ConvGeom& g, int kind, int B, int H, int Wd, int Ci, int Co, int k, int s, int p, int op) {
  if (build_geom(g, kind & 0xff, B, H, Wd, Ci, Co, k, s, p, op)) return -1;
  if (kind & 0x100) {
    g.wts = Co;
    g.wrs = k * k * Co;
  }
  return 0;
}

// S = tapgemm(G, W) for geometry kind 0..3 (see build_geom); tensors NHWC, W packed [taps][Ci][Co]
int emul_tapgemm(int kind, const float* G, const float* W, float* S, int B, int H, int Wd, int Ci, int Co, int k, int s,
                 int p, int op) {
  ConvGeom g;
  if (geom_of(g, kind, B, H, Wd, Ci, Co, k, s, p, op)) return -1;
  const int Mc = g.B * g.Qh * g.Qw, N = g.sC;
  std::memset(S, 0, sizeof(float) * (size_t)g.B * g.sH * g.sW * g.sC);
  for (int cls = 0; cls < g.ncls; ++cls)
    for (int m = 0; m < Mc; ++m) {
      int b, qy, qx;
      decode_m(g, m, b, qy, qx);
      const int sp = scatter_pix(g, cls, b, qy, qx);
      for (int n = 0; n < N; ++n) {
        double acc = 0.0;
        for (int t = 0; t < g.ntaps[cls]; ++t) {
          const Tap& tp = g.taps[cls][t];
          const int gp = gather_pix(g, b, qy, qx, tp);
          if (gp < 0) continue;
          for (int c = 0; c < g.gC; ++c) {
            const float w = g.wT ? W[(size_t)tp.wtap * g.wts + (size_t)n * g.wrs + c] : W[(size_t)tp.wtap * g.wts + (size_t)c * g.wrs + n];
            acc += (double)G[(size_t)gp * g.gC + c] * w;
          }
        }
        S[(size_t)sp * N + n] = (float)acc;
      }
    }
  return 0;
}

// dW[wtap][c][n] = sum_m X[gpix][c]*dY[spix][n] in forward geometry kind 0/1
int emul_wgrad(int kind, const float* X, const float* dY, float* dW, int B, int H, int Wd, int Ci, int Co, int k, int s,
               int p, int op) {
  ConvGeom g;
  if ((kind & 0xff) > 1 || geom_of(g, kind, B, H, Wd, Ci, Co, k, s, p, op)) return -1;
  const int Mc = g.B * g.Qh * g.Qw, N = g.sC;
  std::vector<double> acc((size_t)k * k * Ci * Co, 0.0);
  for (int cls = 0; cls < g.ncls; ++cls)
    for (int m = 0; m < Mc; ++m) {
      int b, qy, qx;
      decode_m(g, m, b, qy, qx);
      const int sp = scatter_pix(g, cls, b, qy, qx);
      for (int t = 0; t < g.ntaps[cls]; ++t) {
        const Tap& tp = g.taps[cls][t];
        const int gp = gather_pix(g, b, qy, qx, tp);
        if (gp < 0) continue;
        for (int c = 0; c < g.gC; ++c)
          for (int n = 0; n < N; ++n)
            acc[(size_t)tp.wtap * g.wts + (size_t)c * g.wrs + n] += (double)X[(size_t)gp * g.gC + c] * dY[(size_t)sp * N + n];
      }
    }
  for (size_t i = 0; i < acc.size(); ++i) dW[i] = (float)acc[i];
  return 0;
}
}
