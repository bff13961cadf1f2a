// Shared declarations of the tap-GEMM kernels (tapgemm.hip: masked small-channel variant + dispatch;
// tapgemm_fast.hip: the vector/MFMA main kernel).
#pragma once
#include "common.hpp"

namespace ctvae {

struct TapGemmArgs {
  ConvGeom g;
  const float* G;
  const float* W;
  const float* bias;
  const float* add;
  const float* mask;
  float* S;
  float* bn_part;  // optional [ncls*mtiles][N][3] per-tile (count, mean, M2) of the pre-activation output
  // optional: S is the gradient w.r.t. the OUTPUT a = act(BN(y)) of a BatchNorm layer -> the epilogue also emits that
  // layer's backward sums per tile, bnb_part[ncls*mtiles][N][2] = (sum g', sum g'*xhat), g' = S*act'(gamma*xhat+beta),
  // xhat = (y-mean)*invstd: the separate pass over (g_a, y) of bn.hip:bn_bwd_partial_kernel is not needed then
  const float* bnb_y;
  const float* bnb_mean;
  const float* bnb_invstd;
  const float* bnb_gamma;
  const float* bnb_beta;
  float* bnb_part;
  int bnb_act;
  float* part;     // split-K partial sums [splitk][B*sH*sW][N] (raw accumulators), used when splitk > 1
  int splitk;
  // lazy BatchNorm apply (InXform, fast kernel, forward orientation only): the gathered operand is act(G*scale[c] + shift[c])
  // of the tensor bound as G, formed between the global load and the LDS store (padding and rows beyond Mc stay 0)
  const float* xf_scale;
  const float* xf_shift;
  int xf_act;
  int out_pix;     // P > 1 (dense scatter, no split-K): output column n = c*P + p is stored at column p*(N/P) + c -- an nn.Linear whose
                   // output is .view(B, C, h, w) (NCHW, vanilla_vae.py:102) leaves the NHWC tensor [B, h, w, C] itself (no layout launch)
  int part_T;      // split-K partials channel-major [splitk][N][B*sH*sW] (SplitKRaw) instead of pixel-major [splitk][B*sH*sW][N]
  int lgQw, lgQhw; // log2(Qw), log2(Qh*Qw) when both are powers of two, else -1 (generic division)
  int act;
  int mask_act;
  int Mc;      // B*Qh*Qw
  int N;       // sC
  int mtiles;  // per class
  int ntiles;
  // fast kernel only: parity class of workgroup row blockIdx.y (+ blockIdx.z when cls_rot) -- see launch_fast_cfg
  int cls_order[kMaxCls];
  int cls_rot;
  // the tap table and tap counts once more, in LAUNCH order (row i = class cls_order[i]): a workgroup indexes them with what it
  // knows from blockIdx alone, so the lane-indexed table load goes out together with the first kernel-argument loads instead of
  // behind the cls_order lookup -- one dependent round trip less in front of the first operand tile (round 3)
  int ntaps_l[kMaxCls];
  Tap taps_l[kMaxCls][kMaxTaps];
};

constexpr int KC = 32;
constexpr int LDK = KC + 4;  // padded K-contiguous LDS row (floats)

// m -> (b, qy, qx), by shifts when the class grid is a power of two
__device__ __forceinline__ void decode_m_fast(const TapGemmArgs& a, int m, int& b, int& qy, int& qx) {
  if (a.lgQw >= 0) {
    b = m >> a.lgQhw;
    const int rr = m & ((1 << a.lgQhw) - 1);
    qy = rr >> a.lgQw;
    qx = rr & ((1 << a.lgQw) - 1);
  } else {
    decode_m(a.g, m, b, qy, qx);
  }
}

int launch_tapgemm_fast(const TapGemmArgs& a, const TapGemmPlan& plan, int pf, hipStream_t st);

}  // namespace ctvae
