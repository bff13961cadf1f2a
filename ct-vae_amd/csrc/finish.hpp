// Finishing passes behind the GEMM launches: the split-K sum + epilogue of the tile kernels (tapgemm.hip) and the slab
// reduction of the weight-gradient kernels (wgrad.hip).  Shared here so that ONE launch can serve both when a layer's
// backward pass was issued as a pair (ctvae_conv_backward): reduce_partials_kernel takes the split-K job as extra blocks.
#pragma once
#include "common.hpp"

namespace ctvae {

struct ReduceJob {
  const float* part;
  float* dst;
  long n;
  int S;
  long stride;
  int small;   // block shape
  int vec;     // float4 lanes (n % 4 == 0 and stride % 4 == 0)
};

// S[i] = epi(sum_z part[z][i]) for the split-K path (N % 4 == 0); nblk blocks of 256 threads walk the n4 float4 elements
struct SplitKJob {
  const float* part;
  int splitk;
  long stride;
  const float* bias;
  const float* add;
  const float* mask;
  int mask_act, act;
  float* S;
  long n4;
  int N;
  int nblk;
};

__device__ __forceinline__ void splitk_finish_body(const SplitKJob& k, int blk) {
  const long gs = (long)k.nblk * 256;
  for (long i = (long)blk * 256 + threadIdx.x; i < k.n4; i += gs) {
    f32x4 v = reinterpret_cast<const f32x4*>(k.part)[i];
    for (int z = 1; z < k.splitk; ++z) {
      f32x4 t = reinterpret_cast<const f32x4*>(k.part + z * k.stride)[i];
      v += t;
    }
    const int col = (int)((i * 4) % k.N);
    if (k.bias != nullptr) v += *reinterpret_cast<const f32x4*>(k.bias + col);
    if (k.add != nullptr) v += reinterpret_cast<const f32x4*>(k.add)[i];
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = act_fwd(v[q], k.act);
    if (k.mask != nullptr) {
      f32x4 m = reinterpret_cast<const f32x4*>(k.mask)[i];
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] *= act_bwd_from_out(m[q], k.mask_act);
    }
    reinterpret_cast<f32x4*>(k.S)[i] = o;
  }
}

}  // namespace ctvae
