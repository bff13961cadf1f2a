// Pairwise edge scorer of the causal-transition layer (ct_mcq_vae.py:86-95,147-151: `graph_discovers[k]` =
// Linear(2D, H) -> LeakyReLU -> Linear(H, 1) -> Sigmoid evaluated on EVERY ordered pair of the N latent nodes).
//
// The first Linear is separable, W1 [x_i ; x_j] = W1a x_i + W1b x_j, so the host computes u = x W1a^T and
// v = x W1b^T + b1 ([B,N,H] each, two small GEMMs) and this file does the part that would otherwise materialise a
// [B,N,N,H] tensor (13 MB per sample at N = 64, H = 800) several times per step:
//
//     out[b,i,j] = sigmoid(b2 + sum_h w2[h] * lrelu(u[b,i,h] + v[b,j,h]))
//
// forward : thread = one (i,j) pair, the h-sum runs inside the thread; v chunks are staged transposed in LDS.
// backward: thread = one h, a workgroup owns (b, 256 h's) and walks all (i,j) pairs, so that
//           dU[b,i,h] and dV[b,j,h] are each produced by exactly ONE thread (no atomics, bit-reproducible);
//           dw2 / db2 leave as per-sample partials reduced by the caller.
// Both are VALU kernels (≈7 instructions per (pair, h)); nothing of size N*N*H ever touches memory.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

namespace {

constexpr int HC = 32;    // h chunk of the forward kernel
constexpr int JT = 64;    // j lanes per workgroup
constexpr int IT = 4;     // i rows per workgroup
constexpr int NMAX = 64;  // backward keeps u[i], dU[i] of one h in registers

__global__ __launch_bounds__(256) void pair_mlp_fwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                          const float* __restrict__ w2, const float* __restrict__ b2,
                                                          float* __restrict__ out, int N, int H, int ld, float slope, int w2_bs,
                                                          int b2_bs, const int* __restrict__ row_of) {
  __shared__ float sV[HC][JT + 1];
  __shared__ __attribute__((aligned(8))) float sU[IT][HC];
  __shared__ __attribute__((aligned(8))) float sW[HC];
  const int tid = threadIdx.x, j_l = tid & (JT - 1), i_l = tid / JT;
  const int b = blockIdx.y, i = blockIdx.x * IT + i_l;
  const float* ub = u + (long)b * N * ld;      // rows of stride ld (u / v may be column blocks of one wider GEMM output)
  const float* vb = v + (long)b * N * ld;
  const int wr = row_of ? row_of[b] : b;   // which row of the scorer bank this sample uses (row_of: the discoverer of its action)
  const float bias = b2 != nullptr ? b2[(long)wr * b2_bs] : 0.f;
  w2 += (long)wr * w2_bs;              // per-sample scorer (w2_bs = H) or one shared scorer (0)
  for (int j0 = 0; j0 < N; j0 += JT) {
    f32x2 acc2 = {0.f, 0.f};
    for (int h0 = 0; h0 < H; h0 += HC) {
      __syncthreads();
      // stage v[b, j0:j0+64, h0:h0+32] transposed (reads coalesced along h), u rows and w2
      for (int e = tid; e < JT * HC; e += 256) {
        const int jj = e / HC, hh = e - jj * HC;
        const int j = j0 + jj, h = h0 + hh;
        sV[hh][jj] = (j < N && h < H) ? vb[(long)j * ld + h] : 0.f;
      }
      if (tid < IT * HC) {
        const int ii = tid / HC, hh = tid - ii * HC;
        const int gi = blockIdx.x * IT + ii, h = h0 + hh;
        sU[ii][hh] = (gi < N && h < H) ? ub[(long)gi * ld + h] : 0.f;
      }
      if (tid < HC) sW[tid] = (h0 + tid < H) ? w2[h0 + tid] : 0.f;   // zero weight masks the h tail
      __syncthreads();
#pragma unroll
      for (int hh = 0; hh < HC; hh += 2) {       // two h per step in packed f32x2 registers
        const f32x2 uu = *reinterpret_cast<const f32x2*>(&sU[i_l][hh]);
        const f32x2 ww = *reinterpret_cast<const f32x2*>(&sW[hh]);
        const f32x2 vv = {sV[hh][j_l], sV[hh + 1][j_l]};
        const f32x2 t = uu + vv;
        const f32x2 ts = t * f32x2{slope, slope};
        acc2 += ww * f32x2{fmaxf(t[0], ts[0]), fmaxf(t[1], ts[1])};
      }
    }
    const int j = j0 + j_l;
    const float acc = acc2[0] + acc2[1];
    if (i < N && j < N) out[((long)b * N + i) * N + j] = 1.f / (1.f + __expf(-(acc + bias)));
  }
}

// Forward, N == 64 (the 64 latent nodes): grid (4, B) -- a workgroup scores 16 rows i x 64 columns j of one sample, a
// thread owns (i, 4 consecutive j).  lrelu(t) = slope*t + (1-slope)*relu(t): the linear part sums per node (a dot product
// of the node's row with w2, formed while the row is staged), the pair loop is add / max / fma only -- 2 instructions per
// (pair, h) with packed f32x2 arithmetic.  u / v chunks of 32 h are staged transposed in LDS ([h][node]) through registers,
// the next chunk's 16-byte global loads are in flight while the current one is evaluated.
constexpr int PH = 32;     // h per chunk
constexpr int PLS = 68;    // LDS row stride ([h][node]); 16-byte aligned rows

__global__ __launch_bounds__(256) void pair_mlp_fwd64_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                            const float* __restrict__ w2, const float* __restrict__ b2,
                                                            float* __restrict__ out, int H, int ld, float slope, int w2_bs,
                                                            int b2_bs, const int* __restrict__ row_of) {
  __shared__ __attribute__((aligned(16))) float sV[2][PH * PLS];
  __shared__ __attribute__((aligned(16))) float sU[2][PH * 16];
  __shared__ __attribute__((aligned(16))) float sW[2][PH];
  __shared__ float sAL[16], sAR[64];
  const int tid = threadIdx.x, b = blockIdx.y, i0 = blockIdx.x * 16;
  const int ti = tid >> 4, tj = (tid & 15) * 4;
  const int wr = row_of ? row_of[b] : b;
  const float bias = b2 != nullptr ? b2[(long)wr * b2_bs] : 0.f;
  w2 += (long)wr * w2_bs;
  const float* ub = u + ((long)b * 64 + i0) * ld;
  const float* vb = v + (long)b * 64 * ld;
  // staging roles: v: thread -> (node tid >> 2, 8 h at (tid & 3) * 8); u: threads < 128 -> (row tid >> 3, 4 h at (tid & 7) * 4)
  const int vn = tid >> 2, vh = (tid & 3) * 8, un = tid >> 3, uh = (tid & 7) * 4;
  f32x4 rv0, rv1, ru;
  float wv[8], wu[4];
  auto fetch = [&](int h0) {
    const bool ok0 = h0 + vh < H, ok1 = h0 + vh + 4 < H;      // H % 4 == 0
    rv0 = ok0 ? *reinterpret_cast<const f32x4*>(vb + (long)vn * ld + h0 + vh) : f32x4{0.f, 0.f, 0.f, 0.f};
    rv1 = ok1 ? *reinterpret_cast<const f32x4*>(vb + (long)vn * ld + h0 + vh + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = (h0 + vh + j < H) ? w2[h0 + vh + j] : 0.f;
    if (tid < 128) {
      const bool oku = h0 + uh < H;
      ru = oku ? *reinterpret_cast<const f32x4*>(ub + (long)un * ld + h0 + uh) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) wu[j] = (h0 + uh + j < H) ? w2[h0 + uh + j] : 0.f;
    }
  };
  float dotv = 0.f, dotu = 0.f;        // sum_h w2[h] * v[node][h] (this thread's h) / same for u
  f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
  const int nch = (H + PH - 1) / PH;
  fetch(0);
  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sV[buf][(vh + j) * PLS + vn] = rv0[j];
      sV[buf][(vh + 4 + j) * PLS + vn] = rv1[j];
      dotv += wv[j] * rv0[j] + wv[4 + j] * rv1[j];
    }
    if (tid < 128) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sU[buf][(uh + j) * 16 + un] = ru[j];
        dotu += wu[j] * ru[j];
      }
    }
    if (vn == 0) {                      // node 0's four staging threads hold w2[h0 + 0..31] between them
#pragma unroll
      for (int j = 0; j < 8; ++j) sW[buf][vh + j] = wv[j] * (1.f - slope);
    }
    __syncthreads();                    // one barrier per chunk: the buffers alternate
    if (c + 1 < nch) fetch((c + 1) * PH);
    const float* V = sV[buf];
    const float* U = sU[buf];
#pragma unroll 8
    for (int h = 0; h < PH; ++h) {
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(V + h * PLS + tj);
      const float ui = U[h * 16 + ti], wh = sW[buf][h];
      const f32x2 u2 = {ui, ui}, w2v = {wh, wh};
      const f32x2 t0 = u2 + f32x2{v4[0], v4[1]}, t1 = u2 + f32x2{v4[2], v4[3]};
      acc0 += w2v * f32x2{fmaxf(t0[0], 0.f), fmaxf(t0[1], 0.f)};
      acc1 += w2v * f32x2{fmaxf(t1[0], 0.f), fmaxf(t1[1], 0.f)};
    }
  }
  // the linear part: per-node dot products, summed over the threads that staged the node's row
  dotv += __shfl_xor(dotv, 1, 64);
  dotv += __shfl_xor(dotv, 2, 64);
  dotu += __shfl_xor(dotu, 1, 64);
  dotu += __shfl_xor(dotu, 2, 64);
  dotu += __shfl_xor(dotu, 4, 64);
  __syncthreads();
  if ((tid & 3) == 0) sAR[vn] = dotv;
  if (tid < 128 && (tid & 7) == 0) sAL[un] = dotu;
  __syncthreads();
  const float al = sAL[ti];
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = (j < 2 ? acc0[j & 1] : acc1[j & 1]) + slope * (al + sAR[tj + j]) + bias;
    o[j] = 1.f / (1.f + __expf(-a));
  }
  *reinterpret_cast<f32x4*>(out + ((long)b * 64 + i0 + ti) * 64 + tj) = o;
}

// grid (ceil(H/256), B).  dw2_part [B][H], dgs_part [B][gridDim.x] (sum of g*s*(1-s), identical for every h block)
__global__ __launch_bounds__(256) void pair_mlp_bwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                          const float* __restrict__ w2, const float* __restrict__ out,
                                                          const float* __restrict__ g_out, float* __restrict__ dU,
                                                          float* __restrict__ dV, float* __restrict__ dw2_part,
                                                          float* __restrict__ db2_part, int N, int H, int ld, int ldd, float slope,
                                                          int w2_bs, const int* __restrict__ row_of) {
  __shared__ __attribute__((aligned(16))) float sG[NMAX][NMAX];   // [j][i] = g * s * (1 - s)
  __shared__ float sRed[4];
  const int tid = threadIdx.x, b = blockIdx.y, h = blockIdx.x * 256 + tid;
  const bool hok = h < H;
  const long bo = (long)b * N * ld, bd = (long)b * N * ldd;
  float gsum = 0.f;
  for (int e = tid; e < NMAX * NMAX; e += 256) {
    const int i = e / NMAX, j = e - i * NMAX;   // coalesced along j
    float gs = 0.f;
    if (i < N && j < N) {
      const float s = out[((long)b * N + i) * N + j];
      gs = g_out[((long)b * N + i) * N + j] * s * (1.f - s);
    }
    sG[j][i] = gs;
    gsum += gs;
  }
  gsum = block_sum_256(gsum, sRed);   // contains the barrier that publishes sG
  if (tid == 0 && blockIdx.x == 0) db2_part[b] = gsum;

  // rows i are processed two at a time in packed f32x2 registers; v[b,j,h] is fetched one j ahead
  f32x2 ur[NMAX / 2], du[NMAX / 2];
#pragma unroll
  for (int i = 0; i < NMAX; ++i) {
    ur[i >> 1][i & 1] = (hok && i < N) ? u[bo + (long)i * ld + h] : 0.f;
    du[i >> 1][i & 1] = 0.f;
  }
  const float wh = hok ? w2[(long)(row_of ? row_of[b] : b) * w2_bs + h] : 0.f;
  f32x2 dw2 = {0.f, 0.f};
  float v_next = hok ? v[bo + h] : 0.f;
  for (int j = 0; j < N; ++j) {
    const float vj = v_next;
    if (j + 1 < N) v_next = hok ? v[bo + (long)(j + 1) * ld + h] : 0.f;
    const f32x2 vj2 = {vj, vj};
    f32x2 dvj2 = {0.f, 0.f};
#pragma unroll
    for (int i4 = 0; i4 < NMAX; i4 += 4) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(&sG[j][i4]);   // same address in every lane: broadcast
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x2 g2 = {g4[2 * q], g4[2 * q + 1]};
        const f32x2 t = ur[(i4 >> 1) + q] + vj2;
        const f32x2 sl = {t[0] > 0.f ? 1.f : slope, t[1] > 0.f ? 1.f : slope};
        const f32x2 gd = g2 * sl;
        du[(i4 >> 1) + q] += gd;
        dvj2 += gd;
        dw2 += gd * t;                // g * lrelu(t) = g * sl * t
      }
    }
    if (hok) dV[bd + (long)j * ldd + h] = wh * (dvj2[0] + dvj2[1]);
  }
  const float dw = dw2[0] + dw2[1];
  if (hok) {
#pragma unroll
    for (int i = 0; i < NMAX; ++i)
      if (i < N) dU[bd + (long)i * ldd + h] = wh * du[i >> 1][i & 1];
    dw2_part[(long)b * H + h] = dw;
  }
}

}  // namespace

int launch_pair_mlp_forward(const float* u, const float* v, int ld, const float* w2, const float* b2, float* out, int B, int N,
                            int H, float slope, int per_sample, const int* row_of, hipStream_t st) {
  if (B <= 0 || N <= 0 || H <= 0 || ld < H) return kErrBadArg;
  ProfScope ps((N == 64 && H % 4 == 0 && ld % 4 == 0) ? "pair_mlp_fwd64_kernel" : "pair_mlp_fwd_kernel", st, 3.0 * B * (double)N * N * H, 4.0 * B * (2.0 * N * H + (double)N * N));
  if (N == 64 && H % 4 == 0 && ld % 4 == 0) {
    hipLaunchKernelGGL(pair_mlp_fwd64_kernel, dim3(4, B), dim3(256), 0, st, u, v, w2, b2, out, H, ld, slope, per_sample ? H : 0,
                       per_sample ? 1 : 0, per_sample ? row_of : nullptr);
    CTVAE_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(pair_mlp_fwd_kernel, dim3((N + IT - 1) / IT, B), dim3(256), 0, st, u, v, w2, b2, out, N, H, ld, slope,
                     per_sample ? H : 0, per_sample ? 1 : 0, per_sample ? row_of : nullptr);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_pair_mlp_backward(const float* u, const float* v, int ld, const float* w2, const float* out, const float* g_out,
                             float* dU, float* dV, int ldd, float* dw2_part, float* db2_part, int B, int N, int H, float slope,
                             int per_sample, const int* row_of, hipStream_t st) {
  if (B <= 0 || N <= 0 || H <= 0 || N > NMAX || ld < H || ldd < H) return kErrBadArg;
  ProfScope ps("pair_mlp_bwd_kernel", st, 7.0 * B * (double)N * N * H, 4.0 * B * (4.0 * N * H + 2.0 * N * N));
  hipLaunchKernelGGL(pair_mlp_bwd_kernel, dim3((H + 255) / 256, B), dim3(256), 0, st, u, v, w2, out, g_out, dU, dV, dw2_part,
                     db2_part, N, H, ld, ldd, slope, per_sample ? H : 0, per_sample ? row_of : nullptr);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
