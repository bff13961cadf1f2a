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
#include "phase.hpp"
#include "prof.hpp"
#include "satmath.hpp"

CTVAE_PHASE_DECL(pair)

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

// Forward, N == 64, blocked: grid (4, B) as above, but a thread owns a 4 x 4 block of pairs (rows 4 tr.., columns 4 tc..) and each of
// the four waves takes a quarter of every chunk's hidden units; the partial sums of the waves meet in LDS at the end.  relu is the
// clamp of the add on operands staged pre-scaled by 2^-64 (satmath.hpp): per hidden unit and thread two 16-byte LDS reads feed
// 8 packed adds + 8 packed fmas for 16 pairs -- one instruction per (pair, h), LDS at a fifth of its rate.
constexpr int QH = 64;     // h per chunk

__global__ __launch_bounds__(256) void pair_mlp_fwd64b_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                                             float* __restrict__ out, int H, int ld, float slope, int w2_bs,
                                                             int b2_bs, const int* __restrict__ row_of) {
  __shared__ __attribute__((aligned(16))) float sV[2][QH * PLS];   // [h][node] * 2^-64; after the loop: the waves' partial sums
  __shared__ __attribute__((aligned(16))) float sU[2][QH * 16];
  __shared__ __attribute__((aligned(16))) float sW[2][QH];         // w2 * (1 - slope)
  __shared__ float sAL[16], sAR[64];
  const int tid = threadIdx.x, b = blockIdx.y, i0 = blockIdx.x * 16, wave = tid >> 6, lane = tid & 63;
  const int tr = lane >> 4, tc = lane & 15;
  const int wr = row_of ? row_of[b] : b;
  const float bias = b2 != nullptr ? b2[(long)wr * b2_bs] : 0.f;
  w2 += (long)wr * w2_bs;
  const float* ub = u + ((long)b * 64 + i0) * ld;
  const float* vb = v + (long)b * 64 * ld;
  // staging roles: v: thread -> (node tid >> 2, 16 h at (tid & 3) * 16); u: thread -> (row tid >> 4, 4 h at (tid & 15) * 4)
  const int vn = tid >> 2, vh = (tid & 3) * 16, un = tid >> 4, uh = (tid & 15) * 4;
  f32x4 rv[4], ru, wv[4], wu;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int h0) {                       // H % 4 == 0: a 4-group is inside or outside
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bool ok = h0 + vh + 4 * g < H;
      rv[g] = ok ? *reinterpret_cast<const f32x4*>(vb + (long)vn * ld + h0 + vh + 4 * g) : z4;
      wv[g] = ok ? *reinterpret_cast<const f32x4*>(w2 + h0 + vh + 4 * g) : z4;
    }
    const bool oku = h0 + uh < H;
    ru = oku ? *reinterpret_cast<const f32x4*>(ub + (long)un * ld + h0 + uh) : z4;
    wu = oku ? *reinterpret_cast<const f32x4*>(w2 + h0 + uh) : z4;
  };
  float dotv = 0.f, dotu = 0.f;
  f32x2 acc[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r][0] = acc[r][1] = f32x2{0.f, 0.f};
  const int nch = (H + QH - 1) / QH;
  const float oms = 1.f - slope;
  fetch(0);
  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sV[buf][(vh + 4 * g + j) * PLS + vn] = rv[g][j] * kSatDown;
        dotv += wv[g][j] * rv[g][j];
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sU[buf][(uh + j) * 16 + un] = ru[j] * kSatDown;
      dotu += wu[j] * ru[j];
    }
    if (un == 0) *reinterpret_cast<f32x4*>(&sW[buf][uh]) = wu * f32x4{oms, oms, oms, oms};   // row 0's 16 staging threads hold w2[h0 + 0..63]
    __syncthreads();                    // one barrier per chunk: the buffers alternate
    if (c + 1 < nch) fetch((c + 1) * QH);
    const float* V = sV[buf] + (16 * wave) * PLS + 4 * tc;
    const float* U = sU[buf] + (16 * wave) * 16 + 4 * tr;
    const float* W = sW[buf] + 16 * wave;
#pragma unroll 4
    for (int h = 0; h < 16; h += 2) {
      const f32x4 ua = *reinterpret_cast<const f32x4*>(U + h * 16), va = *reinterpret_cast<const f32x4*>(V + h * PLS);
      const f32x4 ub4 = *reinterpret_cast<const f32x4*>(U + (h + 1) * 16), vb4 = *reinterpret_cast<const f32x4*>(V + (h + 1) * PLS);
      const f32x2 wp = *reinterpret_cast<const f32x2*>(W + h);
      relu_fma_4x4<false>(acc, f32x2{ua[0], ua[1]}, f32x2{ua[2], ua[3]}, f32x2{va[0], va[1]}, f32x2{va[2], va[3]}, wp);
      relu_fma_4x4<true>(acc, f32x2{ub4[0], ub4[1]}, f32x2{ub4[2], ub4[3]}, f32x2{vb4[0], vb4[1]}, f32x2{vb4[2], vb4[3]}, wp);
    }
  }
  // the linear part: per-node dot products, summed over the threads that staged the node's row
  dotv = quad_sum(dotv);
  dotu += __shfl_xor(dotu, 1, 64);
  dotu += __shfl_xor(dotu, 2, 64);
  dotu += __shfl_xor(dotu, 4, 64);
  dotu += __shfl_xor(dotu, 8, 64);
  __syncthreads();                      // every wave is done with the staging buffers
  if ((tid & 3) == 0) sAR[vn] = dotv;
  if ((tid & 15) == 0) sAL[un] = dotu;
  float* P = sV[0];                     // [wave][16 rows][PLS]
#pragma unroll
  for (int r = 0; r < 4; ++r)
    *reinterpret_cast<f32x4*>(P + (wave * 16 + 4 * tr + r) * PLS + 4 * tc) = f32x4{acc[r][0][0], acc[r][0][1], acc[r][1][0], acc[r][1][1]};
  __syncthreads();
  const int ti = tid >> 4, tj = (tid & 15) * 4;
  f32x4 sum = *reinterpret_cast<const f32x4*>(P + ti * PLS + tj);
#pragma unroll
  for (int w = 1; w < 4; ++w) sum += *reinterpret_cast<const f32x4*>(P + (w * 16 + ti) * PLS + tj);
  const float al = sAL[ti];
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = sum[j] * kSatUp + slope * (al + sAR[tj + j]) + bias;
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

// Backward, N == 64 (the 64 latent nodes).  With lrelu'(t) = slope + (1 - slope) [t > 0] and G = g s (1 - s):
//     A[i,h]  = sum_j G[i,j] [u[i,h] + v[j,h] > 0]            Bv[j,h] = sum_i G[i,j] [u[i,h] + v[j,h] > 0]
//     dU[i,h] = w2[h] (slope RS[i] + (1 - slope) A[i,h])      dV[j,h] = w2[h] (slope CS[j] + (1 - slope) Bv[j,h])
//     dw2[h]  = slope (u . RS + v . CS) + (1 - slope) (sum_i u[i,h] A[i,h] + sum_j v[j,h] Bv[j,h])        (relu(t) = t [t > 0])
// RS / CS = row / column sums of G: everything linear in G leaves the pair loop, which is three packed instructions per TWO
// (pair, h): the step function as the clamp of the add (satmath.hpp), one fma into A, one into Bv.  The packed halves are two
// hidden units of the thread, so every operand of the loop is a natural register pair and G[i,j] a broadcast.
// grid (ceil(nw / WPB), B), nw = ceil(H / 32): a wave owns 32 hidden units; lane = (quarter q of the rows i, hl): h0 + hl and
// h0 + 16 + hl; u and A of the 16 rows stay in registers, G sits transposed in LDS (one 16-byte broadcast read per 4 rows),
// Bv meets in the quad through DPP adds.  One producer per output element, fixed summation order: bit-reproducible.
template <int WPB>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(4, 4))) void pair_mlp_bwd64_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                                 const float* __restrict__ w2, const float* __restrict__ out,
                                                                 const float* __restrict__ g_out, float* __restrict__ dU,
                                                                 float* __restrict__ dV, float* __restrict__ dw2_part,
                                                                 float* __restrict__ db2_part, int H, int ld, int ldd, float slope,
                                                                 int w2_bs, const int* __restrict__ row_of) {
  constexpr int GS = 68;
  __shared__ __attribute__((aligned(16))) float sG[64 * GS];      // [j][i]
  __shared__ float sRS[64], sCS[64];
  const int tid = threadIdx.x, b = blockIdx.y, wave = tid >> 6, lane = tid & 63;
  CTVAE_PH(pair, 1, 0);
  {                                                               // coalesced along j; every load is issued before the first use
    constexpr int NE = (4096 + 64 * WPB - 1) / (64 * WPB);
    float so[NE], go[NE];
#pragma unroll
    for (int t = 0; t < NE; ++t) {
      const int e = tid + t * 64 * WPB;
      so[t] = e < 4096 ? out[(long)b * 4096 + e] : 0.f;
      go[t] = e < 4096 ? g_out[(long)b * 4096 + e] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < NE; ++t) {
      const int e = tid + t * 64 * WPB;
      if (e < 4096) sG[(e & 63) * GS + (e >> 6)] = go[t] * so[t] * (1.f - so[t]);
    }
  }
  __syncthreads();
  CTVAE_PH(pair, 1, 1);
  if (tid < 64) {
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) s += sG[j * GS + tid];
    sRS[tid] = s;
    if (blockIdx.x == 0) {
      s = wave_sum(s);
      if (tid == 0) db2_part[b] = s;
    }
  } else if (tid < 128) {
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) s += sG[(tid - 64) * GS + i];
    sCS[tid - 64] = s;
  }
  __syncthreads();
  CTVAE_PH(pair, 1, 2);
  const int h0 = 32 * (blockIdx.x * WPB + wave);
  if (h0 >= H) return;
  const int q = lane & 3, ha = h0 + (lane >> 2), hb = ha + 16, i0 = 16 * q;
  const bool oka = ha < H, okb = hb < H;
  const int hca = oka ? ha : H - 1, hcb = okb ? hb : H - 1;       // out-of-range lanes read a valid column and store nothing
  const float oms = 1.f - slope;
  const float* up = u + (long)b * 64 * ld + (long)i0 * ld;
  f32x2 u2[16], A2[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    u2[s] = f32x2{up[hca], up[hcb]} * f32x2{kStepUp, kStepUp};
    up += ld;
    A2[s] = f32x2{0.f, 0.f};
  }
  const float* wrow = w2 + (long)(row_of ? row_of[b] : b) * w2_bs;
  const f32x2 wv = {wrow[hca], wrow[hcb]};
  const float* vp = v + (long)b * 64 * ld;
  float* dvp = dV + (long)b * 64 * ldd;
  f32x2 vn = {vp[hca], vp[hcb]};
  f32x2 dwj = {0.f, 0.f};
  const int st_h = (q & 1) ? hcb : hca;
  const float st_w = (q & 1) ? wv[1] : wv[0];
  CTVAE_PH(pair, 1, 3);
  auto column = [&](int j) {
    const f32x2 v2 = vn * f32x2{kStepUp, kStepUp};
    if (j + 1 < 64) vp += ld;
    vn = f32x2{vp[hca], vp[hcb]};                        // one column ahead; unconditional (the last row twice): no branch
    f32x2 bv0 = {0.f, 0.f}, bv1 = {0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(&sG[j * GS + i0 + 4 * s4]);
      step_fma2(A2[4 * s4], A2[4 * s4 + 1], bv0, bv1, u2[4 * s4], u2[4 * s4 + 1], v2, f32x2{g4[0], g4[1]});
      step_fma2(A2[4 * s4 + 2], A2[4 * s4 + 3], bv0, bv1, u2[4 * s4 + 2], u2[4 * s4 + 3], v2, f32x2{g4[2], g4[3]});
    }
    f32x2 bv = bv0 + bv1;
    bv[0] = quad_sum(bv[0]);
    bv[1] = quad_sum(bv[1]);
    const float cs = slope * sCS[j];
    const f32x2 dv = f32x2{cs, cs} + f32x2{oms, oms} * bv;
    dwj += v2 * dv;
    // every lane of the quad holds the sums: even lanes store ha, odd lanes hb -- unconditionally (a predicated store is a branch,
    // behind which the compiler can only wait vmcnt(0) for the next column's v, i.e. for this store's round trip).  Lanes past H
    // work on the clamped unit H - 1 and store ITS (correct) value once more.
    dvp[st_h] = st_w * ((q & 1) ? dv[1] : dv[0]);
    dvp += ldd;
  };
  column(0);                                  // outside the loop: its header then waits vmcnt(1) -- the load, not the previous column's store
  for (int j = 1; j < 64; ++j) column(j);
  CTVAE_PH(pair, 1, 4);
  f32x2 dwi = {0.f, 0.f};
  float* dup = dU + (long)b * 64 * ldd + (long)i0 * ldd;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float rs = slope * sRS[i0 + s];
    const f32x2 du = f32x2{rs, rs} + f32x2{oms, oms} * A2[s];
    dwi += u2[s] * du;
    if (oka) dup[ha] = wv[0] * du[0];
    if (okb) dup[hb] = wv[1] * du[1];
    dup += ldd;
  }
  dwi[0] = quad_sum(dwi[0]);
  dwi[1] = quad_sum(dwi[1]);
  if (q == 0) {
    if (oka) dw2_part[(long)b * H + ha] = (dwi[0] + dwj[0]) * kStepDown;
    if (okb) dw2_part[(long)b * H + hb] = (dwi[1] + dwj[1]) * kStepDown;
  }
  CTVAE_PH(pair, 1, 5);
}

}  // namespace

int launch_pair_mlp_forward(const float* u, const float* v, int ld, const float* w2, const float* b2, float* out, int B, int N,
                            int H, float slope, int per_sample, const int* row_of, hipStream_t st) {
  if (B <= 0 || N <= 0 || H <= 0 || ld < H) return kErrBadArg;
  ProfScope ps((N == 64 && H % 4 == 0 && ld % 4 == 0) ? "pair_mlp_fwd64_kernel" : "pair_mlp_fwd_kernel", st, 3.0 * B * (double)N * N * H, 4.0 * B * (2.0 * N * H + (double)N * N));
  static const bool old_fwd = getenv("CTVAE_PAIR_FWD_OLD") != nullptr;      // diagnostic: the (i, 4 j) kernel
  if (N == 64 && H % 4 == 0 && ld % 4 == 0 && !old_fwd && (reinterpret_cast<uintptr_t>(w2) & 15) == 0) {   // w2 rows are read as float4
    hipLaunchKernelGGL(pair_mlp_fwd64b_kernel, dim3(4, B), dim3(256), 0, st, u, v, w2, b2, out, H, ld, slope, per_sample ? H : 0,
                       per_sample ? 1 : 0, per_sample ? row_of : nullptr);
    CTVAE_LAUNCH_CHECK();
    return 0;
  }
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
  static const bool old_bwd = getenv("CTVAE_PAIR_BWD_OLD") != nullptr;      // diagnostic: the any-N kernel at N == 64 too
  if (N == 64 && !old_bwd) {
    ProfScope ps("pair_mlp_bwd64_kernel", st, 5.0 * B * (double)N * N * H,      // clamped add + two fmas per (pair, h)
                 4.0 * B * (4.0 * N * H + 2.0 * N * N));
    // 4-wave workgroups: ceil(nw / 4) per sample.  (5-wave workgroups tile H = 800 exactly but only two of them fit a CU's wave
    // slots, 4 + 4 + ... per SIMD: 640 workgroups then take two rounds -- 69 us against 56 us.)
    const int nw = (H + 31) / 32;
    hipLaunchKernelGGL(pair_mlp_bwd64_kernel<4>, dim3((nw + 3) / 4, B), dim3(256), 0, st, u, v, w2, out, g_out, dU, dV, dw2_part,
                       db2_part, H, ld, ldd, slope, per_sample ? H : 0, per_sample ? row_of : nullptr);
    CTVAE_LAUNCH_CHECK();
    return 0;
  }
  ProfScope ps("pair_mlp_bwd_kernel", st, 7.0 * B * (double)N * N * H, 4.0 * B * (4.0 * N * H + 2.0 * N * N));
  hipLaunchKernelGGL(pair_mlp_bwd_kernel, dim3((H + 255) / 256, B), dim3(256), 0, st, u, v, w2, out, g_out, dU, dV, dw2_part,
                     db2_part, N, H, ld, ldd, slope, per_sample ? H : 0, per_sample ? row_of : nullptr);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
