// Argument block of the grouped-Linear kernels (glinear.hip); see include/ctvae_hip.h ctvae_glinear_*.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace ctvae {

constexpr int kGLinMaxSeg = 4;

struct GLinArgs {
  const float* x;            // [B*64][ldx]
  int ldx, K;
  int nseg, N;               // nseg segments of N output columns
  const float* W[kGLinMaxSeg];       // element (g, n, k) at W[s] + g*wgs[s] + n*ldw[s] + k
  int ldw[kGLinMaxSeg];
  long wgs[kGLinMaxSeg];
  const float* bias[kGLinMaxSeg];    // element (g, n) at bias[s] + g*bgs[s] + n   (may be null)
  int bgs[kGLinMaxSeg];
  const int* group[kGLinMaxSeg];     // [B] or null
  float* y;                  // [B*64][ldy], segment s at columns s*N
  int ldy;
  int B;
};

int launch_glinear_forward(const GLinArgs& a, hipStream_t st);
int launch_glinear_dgrad(const GLinArgs& a, float* dx, hipStream_t st);
size_t glinear_wgrad_ws_floats(int G, int N, int K, int S);
int launch_glinear_wgrad(const float* x, int ldx, int K, const float* dy, int ldy, int col0, int N, const int* group, int G,
                         int B, float* dW, int ldo, float* dbias, int accumulate, float* ws, size_t ws_floats, hipStream_t st);

}  // namespace ctvae
