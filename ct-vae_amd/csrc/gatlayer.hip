// One GATv2Conv layer of CausalTransition.graph_transitioner (ct_mcq_vae.py:103-114: gnn.GATv2Conv(in, out, edge_dim=1,
// heads)) on a batch of dense weighted graphs over the 64 latent nodes, fused:
//
//   forward   S[r,c]   = sum_k att[k] * lrelu(xl[r,k] + xr[c,k] + a'[r,c] * we[k], 0.2)        (r source, c target)
//             alpha    = softmax over the sources r of the kept pairs (edges a[r,c] != 0, r != c, plus the self loop)
//             out[c,:] = act(sum_r alpha[r,c] * xl[r,:] + bias)
//   a' = a off the diagonal; a'[c,c] = mean of the incoming edge attributes of c (add_self_loops, fill_value 'mean').
//
// One workgroup per (sample, head slot): both projections of the head sit transposed in LDS, a thread owns a 4 x 4 block
// of (source, target) pairs (two 16-byte LDS reads feed 16 pair evaluations per channel), the score tile never leaves
// the CU: masked column softmax and the alpha-weighted aggregation run on the LDS-resident tile.  As torch ops the same
// layer was a score kernel + masked_fill + softmax + einsum over [B,H,65,65] tensors (and their autograd mirrors).
//
// What is NOT here on purpose: the action node the reference appends (ct_mcq_vae.py:203-209).  Its row of the padded
// adjacency is zero (no outgoing edge) and its own output row is discarded (:221), so it never reaches a latent node;
// the 64 latent nodes are the whole graph as far as the layer's result and every gradient are concerned.
//
// Head slots: a sample evaluates Hs <= H heads; head_map[b*Hs + hs] names the head whose att / we / bias the slot uses
// (null: slot == head).  CausalTransition._compute_y reads only head 0 and head 1+action of the LAST layer (:224-226), so
// that layer runs two slots per sample instead of all 13 / 21 heads.
//
//   gat_layer_fwd_kernel   above
//   gat_layer_bwd_kernel   d alpha = xl . g^T, softmax backward -> dS; the aggregation's share of d xl; d a' per head slot
//                          (dS * sum_k att we lrelu'(.)); bias-gradient partials
//   gat_proj_bwd_kernel    thread = (channel, quarter of the sources): d xl, d xr, d att, d we from dS -- every element has
//                          one producer, partial sums meet in shuffles (no atomics: bit-reproducible)
//   gat_adj_reduce_kernel  d a = edge * (sum_slots d a' + (sum_slots d a'[c,c]) / deg[c])
#include "common.hpp"
#include "gatlayer.hpp"
#include "phase.hpp"
#include "prof.hpp"
#include "satmath.hpp"

CTVAE_PHASE_DECL(gat)

namespace ctvae {

namespace {

constexpr int GN = 64;   // nodes per graph
constexpr int LS = 68;   // LDS row stride of the [channel][node] operand tiles (16-byte aligned rows)
constexpr int SS = 65;   // LDS row stride of the [source][target] tile

__device__ __forceinline__ float relu(float v) { return fmaxf(v, 0.f); }

// The thread's 4 x 4 block of the adjacency (rows 4tr.., columns 4tc..) with the self-loop attribute on the diagonal.
// keep: bit (4i+j) set for pairs inside the softmax (edge or self loop).  scratch: >= 2*16*64 floats of LDS.
__device__ __forceinline__ void load_adj_block(const float* __restrict__ adj, int b, int tr, int tc, float (&a)[4][4],
                                               unsigned& keep, float* scratch, float* sLoop, float* sDeg) {
  const int tid = threadIdx.x;
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cd[4] = {0.f, 0.f, 0.f, 0.f};
  keep = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(adj + ((long)b * GN + 4 * tr + i) * GN + 4 * tc);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool diag = (tr == tc) && (i == j);
      const bool e = v[j] != 0.f && !diag;          // existing self loops are removed first
      a[i][j] = e ? v[j] : 0.f;
      if (e) { keep |= 1u << (4 * i + j); cs[j] += v[j]; cd[j] += 1.f; }
      if (diag) keep |= 1u << (4 * i + j);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    scratch[tr * GN + 4 * tc + j] = cs[j];
    scratch[16 * GN + tr * GN + 4 * tc + j] = cd[j];
  }
  __syncthreads();
  if (tid < GN) {
    float s = 0.f, d = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) { s += scratch[t * GN + tid]; d += scratch[16 * GN + t * GN + tid]; }
    sDeg[tid] = d;
    sLoop[tid] = s / fmaxf(d, 1.f);
  }
  __syncthreads();
  if (tr == tc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i][i] = sLoop[4 * tc + i];
  }
}

// xl / xr of one head slot, transposed into LDS: T[k][n].  16-byte global loads, all of a thread's loads issued before the
// first LDS store (<= 8 per thread for C <= 128), so their latencies overlap; C % 4 == 0.
__device__ __forceinline__ void stage_T(const float* __restrict__ src, long row0, int ld, int col0, int C, float* T, float scale = 1.f) {
  const int c4 = C >> 2, n4 = GN * c4;
  f32x4 v[8];
  int nn[8], kk[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int e = threadIdx.x + 256 * it;
    const int n = e / c4;
    nn[it] = n;
    kk[it] = (e - n * c4) * 4;
    if (e < n4) v[it] = *reinterpret_cast<const f32x4*>(src + (row0 + n) * ld + col0 + kk[it]);
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    if (threadIdx.x + 256 * it < n4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) T[(kk[it] + j) * LS + nn[it]] = v[it][j] * scale;
    }
  }
}

// The pair loop of the forward score and of the backward's edge-attribute gradient: thread = 4 x 4 block of (source, target) pairs,
// acc[r][c] += wa.y * sat(l[k][r] + r[k][c] + a[r][c] * wa.x) over the C channels (score_2x4, satmath.hpp).  Operands of four
// channels are read from LDS as one group while the previous group is evaluated (a 16-byte read takes longer than the 24
// instructions of one channel, and only two waves share a SIMD); the last group's read-ahead lands in the arrays that follow
// (in bounds, unused).  C % 4 == 0.
__device__ __forceinline__ void pair_loop(f32x2 (&acc)[4][2], const f32x2 (&a2)[4][2], const float* pl, const float* pr, const float* pw, int C) {
  struct Group { f32x4 l[4], r[4]; f32x2 w[4]; };
  auto load = [&](Group& g, int k) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g.l[j] = *reinterpret_cast<const f32x4*>(pl + (k + j) * LS);
      g.r[j] = *reinterpret_cast<const f32x4*>(pr + (k + j) * LS);
      g.w[j] = *reinterpret_cast<const f32x2*>(pw + 2 * (k + j));
    }
  };
  auto eval = [&](const Group& g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 r01 = {g.r[j][0], g.r[j][1]}, r23 = {g.r[j][2], g.r[j][3]};
      score_2x4(acc[0][0], acc[0][1], acc[1][0], acc[1][1], f32x2{g.l[j][0], g.l[j][1]}, r01, r23, g.w[j], a2[0][0], a2[0][1], a2[1][0], a2[1][1]);
      score_2x4(acc[2][0], acc[2][1], acc[3][0], acc[3][1], f32x2{g.l[j][2], g.l[j][3]}, r01, r23, g.w[j], a2[2][0], a2[2][1], a2[3][0], a2[3][1]);
    }
  };
  // two register sets take turns (a rotating copy costs ten v_mov per channel on top of the 24 instructions of the work)
  Group ga, gb;
  load(ga, 0);
  for (int k = 0; k < C; k += 8) {
    load(gb, k + 4);                 // k + 4 == C: the arrays that follow, unused
    eval(ga);
    if (k + 4 < C) {
      load(ga, k + 8);
      eval(gb);
    }
  }
}

// MFMA helper of the fused layer kernels: D[m][n] = sum_{r < 64} A(r, m) * B[n][r] for one 32 x 32 tile; A(r, m) sits at
// A[r * ars + m * ams], B rows have stride LS and hold 64 consecutive r.  v_mfma_f32_32x32x2_f32 takes two values of the reduction
// index per step (lane >> 5 picks which); the r order is permuted so that a lane's B operands of four steps are ONE 16-byte LDS
// read (r = 8 q + 4 (lane >> 5) + j, conflict-free for row stride 68).  Result: lane holds n = lane & 31, register i holds
// m = 8 (i >> 2) + 4 (lane >> 5) + (i & 3).
__device__ __forceinline__ f32x16 mfma_tile_64(const float* A, int ars, int ams, const float* B, int lane) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int lh = lane >> 5, ln = lane & 31;
  const float* ap = A + (4 * lh) * ars + ln * ams;
  const float* bp = B + ln * LS + 4 * lh;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + 8 * q);
    const float a0 = ap[(8 * q) * ars], a1 = ap[(8 * q + 1) * ars], a2 = ap[(8 * q + 2) * ars], a3 = ap[(8 * q + 3) * ars];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b4[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b4[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b4[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b4[3], acc, 0, 0, 0);
  }
  return acc;
}

// Forward.  The per-pair part of the score is relu: lrelu(m) = slope m + (1 - slope) relu(m), the linear part sums per node.  xl, xr
// and we are staged scaled by 2^-64 so that relu is the clamp of the packed fma that forms m (satmath.hpp): 1.5 instructions per
// (pair, channel).  The alpha-weighted aggregation out = alpha^T xl is four to eight 32 x 32 MFMA tiles straight from the LDS
// operands, written to global memory from the accumulators.
__global__ __launch_bounds__(256, 2) void gat_layer_fwd_kernel(GatLayerArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = a.C;
  float* XL = smem;                 // [C][LS]  xl * 2^-64
  float* XR = XL + C * LS;          // [C][LS]  xr * 2^-64
  float* Ss = XR + C * LS;          // [GN][SS]
  float* sWA = Ss + GN * SS;        // [C + 4][2]  {we * 2^-64, att * (1 - slope)}   (8-byte aligned: every array before it has an even length)
  float* sAtt = sWA + 2 * (C + 4);  // [C]
  float* sP = sAtt + C;             // [2][GN][4] partial sums of att . xl / att . xr (quarter of the channels each)
  sP += (4 - ((sP - smem) & 3)) & 3;
  float* sCol = sP + 2 * GN * 4;    // [4][GN]
  float* sLoop = sCol + 4 * GN;     // [GN]
  float* sDeg = sLoop + GN;         // [GN]
  const int tid = threadIdx.x, hs = blockIdx.x, b = blockIdx.y, lane = tid & 63, wave = tid >> 6;
  const int head = a.head_map ? a.head_map[b * a.Hs + hs] : hs;
  const int tr = tid >> 4, tc = tid & 15;
  CTVAE_PH(gat, 0, 0);
  stage_T(a.xl, (long)b * GN, a.ld, hs * C, C, XL, kSatDown);
  stage_T(a.xr, (long)b * GN, a.ld, hs * C, C, XR, kSatDown);
  for (int k = tid; k < C; k += 256) {
    const float w = a.we[head * C + k], t = a.att[head * C + k];
    sWA[2 * k] = w * kSatDown;
    sWA[2 * k + 1] = t * (1.f - a.slope);
    sAtt[k] = t;
  }
  CTVAE_PH(gat, 0, 1);
  float av[4][4];
  unsigned keep;
  load_adj_block(a.adj, b, tr, tc, av, keep, Ss, sLoop, sDeg);      // two barriers inside: the staging above is visible
  CTVAE_PH(gat, 0, 2);
  // lrelu(m) = slope*m + (1-slope)*relu(m): the first term is linear in xl, xr, a' and is summed per node, not per pair
  float aw;                                                            // sum_k att we * 2^-64
  {
    const int n = tid & (GN - 1), q = tid >> 6, cq = C >> 2;
    float pl = 0.f, pr = 0.f;
#pragma unroll 5
    for (int k = q * cq; k < (q + 1) * cq; ++k) {
      const float t = sAtt[k];
      pl += t * XL[k * LS + n];
      pr += t * XR[k * LS + n];
    }
    sP[n * 4 + q] = pl;
    sP[(GN + n) * 4 + q] = pr;
    float s = 0.f;
    for (int k = lane; k < C; k += 64) s += sAtt[k] * sWA[2 * k];
    aw = wave_sum(s);
  }
  CTVAE_PH(gat, 0, 3);
  f32x2 acc[4][2], a2[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      acc[i][jp] = f32x2{0.f, 0.f};
      a2[i][jp] = f32x2{av[i][2 * jp], av[i][2 * jp + 1]};
    }
  pair_loop(acc, a2, XL + 4 * tr, XR + 4 * tc, sWA, C);
  CTVAE_PH(gat, 0, 4);
  __syncthreads();
  CTVAE_PH(gat, 0, 5);
  {
    float al[4], ar[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(sP + (4 * tr + i) * 4), v = *reinterpret_cast<const f32x4*>(sP + (GN + 4 * tc + i) * 4);
      al[i] = (u[0] + u[1]) + (u[2] + u[3]);
      ar[i] = (v[0] + v[1]) + (v[2] + v[3]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float s = (a.slope * (al[i] + ar[j] + av[i][j] * aw) + acc[i][j >> 1][j & 1]) * kSatUp;
        Ss[(4 * tr + i) * SS + 4 * tc + j] = ((keep >> (4 * i + j)) & 1u) ? s : -INFINITY;
      }
  }
  __syncthreads();
  CTVAE_PH(gat, 0, 6);
  // softmax over the sources r of every target column c: thread = (c, quarter of the rows)
  {
    const int c = tid & (GN - 1), q = tid >> 6;
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, Ss[(16 * q + i) * SS + c]);
    sCol[q * GN + c] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sCol[c], sCol[GN + c]), fmaxf(sCol[2 * GN + c], sCol[3 * GN + c]));   // finite: the self loop is kept
    __syncthreads();
    float e[16], sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      e[i] = __expf(Ss[(16 * q + i) * SS + c] - mx);
      sum += e[i];
    }
    sCol[q * GN + c] = sum;
    __syncthreads();
    const float inv = 1.f / (sCol[c] + sCol[GN + c] + sCol[2 * GN + c] + sCol[3 * GN + c]);
    float* al = a.alpha + (((long)b * a.Hs + hs) * GN + 16 * q) * GN + c;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = e[i] * inv;
      Ss[(16 * q + i) * SS + c] = p;
      al[i * GN] = p;
    }
  }
  __syncthreads();
  CTVAE_PH(gat, 0, 7);
  // out[c][k] = 2^64 sum_r alpha[r][c] * xl'[r][k]: tile = (32 targets, 32 channels); tiles past C read the next array (discarded)
  {
    const int nt2 = 2 * ((C + 31) >> 5), lh = lane >> 5, ln = lane & 31;
    for (int t = wave; t < nt2; t += 4) {
      const int mt = t & 1, k = 32 * (t >> 1) + ln;
      const f32x16 o = mfma_tile_64(Ss + 32 * mt, SS, 1, XL + 32 * (t >> 1) * LS, lane);
      if (k < C) {
        const float bias = a.bias[head * C + k];
        float* op = a.out + ((long)b * GN + 32 * mt + 4 * lh) * a.ldo + hs * C + k;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float v = o[i] * kSatUp + bias;
          if (a.act == ACT_LRELU) v = v > 0.f ? v : v * kLeaky;
          op[(long)(8 * (i >> 2) + (i & 3)) * a.ldo] = v;
        }
      }
    }
  }
  CTVAE_PH(gat, 0, 8);
}

// Backward of the layer up to dS (gat_proj_bwd_kernel takes it from there): d alpha = xl G^T and the aggregation's share of d xl,
// alpha G, as MFMA tiles from the LDS operands; softmax backward on the accumulators; the edge-attribute gradient
// d a'[r][c] = dS (slope sum_k att we + (1 - slope) sum_k att we [m > 0]) with the step function as the clamp of the packed fma that
// forms m (operands scaled by 2^60, satmath.hpp): 1.5 instructions per (pair, channel).
__global__ __launch_bounds__(256, 2) void gat_layer_bwd_kernel(GatBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const GatLayerArgs& a = p.f;
  const int C = a.C;
  float* XL = smem;                 // [C][LS]  xl * 2^60
  float* R2 = XL + C * LS;          // [C][LS]: G[k][c], then xr[k][c] * 2^60
  float* Ss = R2 + C * LS;          // [GN][SS] scratch of load_adj_block, then alpha, then dS
  float* sWA = Ss + GN * SS;        // [C + 4][2]  {we * 2^60, att we (1 - slope)}
  float* sT = sWA + 2 * (C + 4);    // [2][GN]
  float* sLoop = sT + 2 * GN;
  float* sDeg = sLoop + GN;
  const int tid = threadIdx.x, hs = blockIdx.x, b = blockIdx.y, lane = tid & 63, wave = tid >> 6, lh = lane >> 5, ln = lane & 31;
  const int head = a.head_map ? a.head_map[b * a.Hs + hs] : hs;
  const int tr = tid >> 4, tc = tid & 15;
  CTVAE_PH(gat, 1, 0);
  float alv[16];                                         // alpha of the head: in flight while the operands are staged
  {
    const float* al = a.alpha + ((long)b * a.Hs + hs) * GN * GN;
#pragma unroll
    for (int t = 0; t < 16; ++t) alv[t] = al[tid + 256 * t];
  }
  stage_T(a.xl, (long)b * GN, a.ld, hs * C, C, XL, kStepUp);
  {                                                     // G[k][c] = g_out[c][k] * act'(out[c][k])
    const int c4 = C >> 2, n4 = GN * c4;
    f32x4 gv[8], ov[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int e = tid + 256 * it;
      const int n = e / c4;
      const long o = ((long)b * GN + n) * a.ldo + hs * C + (e - n * c4) * 4;
      if (e < n4) {
        gv[it] = *reinterpret_cast<const f32x4*>(p.g_out + o);
        if (a.act == ACT_LRELU) ov[it] = *reinterpret_cast<const f32x4*>(a.out + o);
      }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int e = tid + 256 * it;
      if (e < n4) {
        const int n = e / c4, k = (e - n * c4) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float g = gv[it][j];
          if (a.act == ACT_LRELU) g *= ov[it][j] > 0.f ? 1.f : kLeaky;
          R2[(k + j) * LS + n] = g;
        }
      }
    }
  }
  for (int k = tid; k < C; k += 256) {
    const float w = a.we[head * C + k];
    sWA[2 * k] = w * kStepUp;
    sWA[2 * k + 1] = a.att[head * C + k] * w * (1.f - a.slope);
  }
  CTVAE_PH(gat, 1, 1);
  float av[4][4];
  unsigned keep;
  load_adj_block(a.adj, b, tr, tc, av, keep, Ss, sLoop, sDeg);     // two barriers inside: the staging above is visible
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int e = tid + 256 * t;
    Ss[(e >> 6) * SS + (e & 63)] = alv[t];
  }
  CTVAE_PH(gat, 1, 2);
  if (tid < C) {                                        // bias gradient: sum over the targets
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < GN; ++c) s += R2[tid * LS + c];
    p.dbias_part[((long)b * a.Hs + hs) * C + tid] = s;
  }
  float aw;                                             // slope * sum_k att we
  {
    float s = 0.f;
    for (int k = lane; k < C; k += 64) s += a.att[head * C + k] * a.we[head * C + k];
    aw = wave_sum(s) * a.slope;
  }
  CTVAE_PH(gat, 1, 3);
  // d alpha[r][c] = sum_k xl[r][k] * G[k][c]: one 32 x 32 tile per wave, two channels per MFMA step
  const int mt = wave >> 1, nt = wave & 1;
  f32x16 da;
  {
#pragma unroll
    for (int i = 0; i < 16; ++i) da[i] = 0.f;
    const float* ap = XL + lh * LS + 32 * mt + ln;
    const float* bp = R2 + lh * LS + 32 * nt + ln;
    // four MFMAs per trip with their eight LDS operands read together (`#pragma unroll 4` on the runtime-bounded loop was refused by
    // the compiler: one ds_read pair -> wait -> MFMA per trip, the matrix pipe idle for an LDS latency fifty times per tile)
    int k = 0;
    for (; k + 8 <= C; k += 8) {
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { av[u] = ap[(k + 2 * u) * LS]; bv[u] = bp[(k + 2 * u) * LS]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) da = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], da, 0, 0, 0);
    }
    for (; k < C; k += 2) da = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k * LS], bp[k * LS], da, 0, 0, 0);
  }
  __syncthreads();                                      // alpha is in Ss
  CTVAE_PH(gat, 1, 4);
  // softmax backward per target column: dS = alpha * (d alpha - sum_r alpha * d alpha).  The lane holds column c = 32 nt + ln and
  // the rows r = 32 mt + 8 (i >> 2) + 4 lh + (i & 3).
  float alr[16];
  {
    const float* sp = Ss + (32 * mt + 4 * lh) * SS + 32 * nt + ln;
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      da[i] *= kStepDown;
      alr[i] = sp[(8 * (i >> 2) + (i & 3)) * SS];
      part += alr[i] * da[i];
    }
    part += __shfl_xor(part, 32, 64);
    if (lh == 0) sT[mt * GN + 32 * nt + ln] = part;
  }
  // the aggregation's share of d xl[r][k] = sum_c alpha[r][c] * G[k][c]: tile = (32 sources, 32 channels), straight to global memory
  {
    const int nt2 = 2 * ((C + 31) >> 5);
    for (int t = wave; t < nt2; t += 4) {
      const int m2 = t & 1, k = 32 * (t >> 1) + ln;
      const f32x16 o = mfma_tile_64(Ss + 32 * m2 * SS, 1, SS, R2 + 32 * (t >> 1) * LS, lane);
      if (k < C) {
        float* op = p.dxl + ((long)b * GN + 32 * m2 + 4 * lh) * p.ldd + hs * C + k;
#pragma unroll
        for (int i = 0; i < 16; ++i) op[(long)(8 * (i >> 2) + (i & 3)) * p.ldd] = o[i];
      }
    }
  }
  __syncthreads();                                      // alpha and G are dead, sT is complete
  CTVAE_PH(gat, 1, 5);
  {
    const float tcol = sT[32 * nt + ln] + sT[GN + 32 * nt + ln];
    float* sp = Ss + (32 * mt + 4 * lh) * SS + 32 * nt + ln;
    float* gp = p.dS + (((long)b * a.Hs + hs) * GN + 32 * mt + 4 * lh) * GN + 32 * nt + ln;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float d = alr[i] * (da[i] - tcol);
      sp[(8 * (i >> 2) + (i & 3)) * SS] = d;
      gp[(8 * (i >> 2) + (i & 3)) * GN] = d;
    }
  }
  stage_T(a.xr, (long)b * GN, a.ld, hs * C, C, R2, kStepUp);
  __syncthreads();
  CTVAE_PH(gat, 1, 6);
  // d a'[r][c] = dS * (slope * sum_k att we + (1 - slope) * sum_k att we [m > 0]): thread = 4 x 4 block of pairs
  f32x2 t[4][2], a2[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      t[i][jp] = f32x2{0.f, 0.f};
      a2[i][jp] = f32x2{av[i][2 * jp], av[i][2 * jp + 1]};
    }
  pair_loop(t, a2, XL + 4 * tr, R2 + 4 * tc, sWA, C);
  CTVAE_PH(gat, 1, 7);
  {
    float* dA = p.dattr + (((long)b * a.Hs + hs) * GN + 4 * tr) * GN + 4 * tc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = Ss[(4 * tr + i) * SS + 4 * tc + j] * (aw + t[i][j >> 1][j & 1]);
      *reinterpret_cast<f32x4*>(dA + i * GN) = v;
    }
  }
  CTVAE_PH(gat, 1, 8);
}

// d xl (added to what gat_layer_bwd_kernel left), d xr, d att, d we from dS.  With lrelu'(m) = slope + (1 - slope) [m > 0] and
// P[r,c,k] = dS[r,c] [m[r,c,k] > 0] everything is three sums of P plus terms linear in dS that never enter the pair loop:
//     Ql[r,k] = sum_c P        Qr[c,k] = sum_r P        Qa[k] = sum_rc P a'[r,c]          RS / CS = row / column sums of dS, DA = sum dS a'
//     d xl[r,k] += att_k (slope RS[r] + (1-slope) Ql)      d xr[c,k] = att_k (slope CS[c] + (1-slope) Qr)      d we[k] = att_k (slope DA + (1-slope) Qa)
//     d att[k]   = sum_r xl[r,k] (slope RS[r] + (1-slope) Ql[r,k]) + sum_c xr[c,k] (slope CS[c] + (1-slope) Qr[c,k]) + we_k (slope DA + (1-slope) Qa[k])
// (dS relu(m) = P m and m = xl + xr + a' we).  grid (Hs, B), 256 threads: thread = (4 channels kq -- two packed pairs --, 8 sources
// q8); xl and Ql of its (source, channel)s stay in registers, dS and a' of the head sit transposed in LDS (16-byte reads shared by
// the lanes of a channel group), the step function is the clamp of a packed add on operands scaled by 2^60 (satmath.hpp): five
// packed instructions per two (pair, channel)s.  Qr meets in the 8 lanes through DPP adds: one producer per output element,
// fixed order, no atomics.
__global__ __launch_bounds__(256, 4) void gat_proj_bwd_kernel(GatBwdArgs p) {
  constexpr int TS = GN + 4;
  __shared__ __attribute__((aligned(16))) float sG[GN * TS];   // [c][r] dS
  __shared__ __attribute__((aligned(16))) float sA[GN * TS];   // [c][r] a' (self-loop mean on the diagonal)
  __shared__ float sRS[GN], sCS[GN], sDAc[GN], sLoop[GN];
  const GatLayerArgs& a = p.f;
  const int C = a.C;
  const int tid = threadIdx.x, hs = blockIdx.x, b = blockIdx.y;
  const int head = a.head_map ? a.head_map[b * a.Hs + hs] : hs;
  CTVAE_PH(gat, 2, 0);
  {
    const float* dS = p.dS + ((long)b * a.Hs + hs) * GN * GN;
    const float* adj = a.adj + (long)b * GN * GN;
    float gv[16], av[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      gv[t] = dS[tid + 256 * t];
      av[t] = adj[tid + 256 * t];
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int e = tid + 256 * t, r = e >> 6, c = e & 63;
      sG[c * TS + r] = gv[t];
      sA[c * TS + r] = (r == c) ? 0.f : av[t];
    }
  }
  __syncthreads();
  if (tid < GN) {                    // self-loop attribute: mean of the incoming edges of target c = tid; CS[c]
    float s = 0.f, d = 0.f, cs = 0.f;
#pragma unroll 8
    for (int r = 0; r < GN; ++r) {
      const float v = sA[tid * TS + r];
      s += v;
      d += v != 0.f ? 1.f : 0.f;
      cs += sG[tid * TS + r];
    }
    sLoop[tid] = s / fmaxf(d, 1.f);
    sCS[tid] = cs;
  } else if (tid < 2 * GN) {         // RS[r]
    const int r = tid - GN;
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < GN; ++c) s += sG[c * TS + r];
    sRS[r] = s;
  }
  __syncthreads();
  if (tid < GN) sA[tid * TS + tid] = sLoop[tid];
  __syncthreads();
  if (tid < GN) {                    // DA = sum dS a'
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < GN; ++r) s += sG[tid * TS + r] * sA[tid * TS + r];
    s = wave_sum(s);
    if (tid == 0) sDAc[0] = s;
  }
  __syncthreads();
  CTVAE_PH(gat, 2, 1);
  const int kq = tid >> 3, q8 = tid & 7, k0 = 4 * kq, r0 = 8 * q8;
  if (k0 >= C) return;               // C % 4 == 0: a channel group is inside or outside
  const float slope = a.slope, oms = 1.f - a.slope;
  const f32x2 up = {kStepUp, kStepUp};
  f32x2 xl2[2][8], ql[2][8];
  {
    const float* xp = a.xl + ((long)b * GN + r0) * a.ld + hs * C + k0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(xp + (long)r * a.ld);
      xl2[0][r] = f32x2{x[0], x[1]} * up;
      xl2[1][r] = f32x2{x[2], x[3]} * up;
      ql[0][r] = ql[1][r] = f32x2{0.f, 0.f};
    }
  }
  const float* wep = a.we + head * C + k0;
  const float* atp = a.att + head * C + k0;
  const f32x2 at2[2] = {{atp[0], atp[1]}, {atp[2], atp[3]}};
  const f32x2 ws2[2] = {f32x2{wep[0], wep[1]} * up, f32x2{wep[2], wep[3]} * up};
  const float* xrp = a.xr + (long)b * GN * a.ld + hs * C + k0;
  float* dxrp = p.dxr + (long)b * GN * p.ldd + hs * C + k0;
  f32x4 xr_next = *reinterpret_cast<const f32x4*>(xrp);
  f32x2 qa[2] = {{0.f, 0.f}, {0.f, 0.f}}, dc[2] = {{0.f, 0.f}, {0.f, 0.f}};
  auto target = [&](int c) {
    const f32x4 xr4 = xr_next;
    if (c + 1 < GN) xrp += a.ld;
    xr_next = *reinterpret_cast<const f32x4*>(xrp);                     // one target ahead; unconditional (the last row twice): no branch
    const f32x2 xr2[2] = {f32x2{xr4[0], xr4[1]} * up, f32x2{xr4[2], xr4[3]} * up};
    const f32x4 g4a = *reinterpret_cast<const f32x4*>(&sG[c * TS + r0]), g4b = *reinterpret_cast<const f32x4*>(&sG[c * TS + r0 + 4]);
    const f32x4 a4a = *reinterpret_cast<const f32x4*>(&sA[c * TS + r0]), a4b = *reinterpret_cast<const f32x4*>(&sA[c * TS + r0 + 4]);
    f32x2 qr[2] = {{0.f, 0.f}, {0.f, 0.f}};
#define CTVAE_PROJ(R, A2, G2)                                                                                              \
    proj_step2x2(ql[0][R], ql[0][R + 1], ql[1][R], ql[1][R + 1], qr[0], qr[1], qa[0], qa[1], A2, G2, ws2[0], ws2[1], xl2[0][R], \
                 xl2[0][R + 1], xl2[1][R], xl2[1][R + 1], xr2[0], xr2[1])
    CTVAE_PROJ(0, (f32x2{a4a[0], a4a[1]}), (f32x2{g4a[0], g4a[1]}));
    CTVAE_PROJ(2, (f32x2{a4a[2], a4a[3]}), (f32x2{g4a[2], g4a[3]}));
    CTVAE_PROJ(4, (f32x2{a4b[0], a4b[1]}), (f32x2{g4b[0], g4b[1]}));
    CTVAE_PROJ(6, (f32x2{a4b[2], a4b[3]}), (f32x2{g4b[2], g4b[3]}));
#undef CTVAE_PROJ
    const float cs = slope * sCS[c];
    f32x4 o;
    float q0 = qr[0][0], q1 = qr[0][1], q2 = qr[1][0], q3 = qr[1][1];
    oct_sum4(q0, q1, q2, q3);
    const f32x2 qs[2] = {{q0, q1}, {q2, q3}};
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const f32x2 q = qs[pp];
      const f32x2 d = f32x2{cs, cs} + f32x2{oms, oms} * q;
      dc[pp] += xr2[pp] * d;
      o[2 * pp] = at2[pp][0] * d[0];
      o[2 * pp + 1] = at2[pp][1] * d[1];
    }
    // every lane of the 8-lane group holds the same sums and stores them (same 16 bytes: one request): a store under `if (q8 == 0)`
    // is a branch, behind which the compiler can only wait vmcnt(0) for the next target's xr -- i.e. for this store's round trip
    *reinterpret_cast<f32x4*>(dxrp) = o;
    dxrp += p.ldd;
  };
  // the first target outside the loop: the loop header then sees [load, store] outstanding on both of its edges and waits vmcnt(1)
  // (with the prologue's loads on the entry edge it waits vmcnt(0): the previous target's store, a memory round trip per target)
  target(0);
  for (int c = 1; c < GN; ++c) target(c);
  CTVAE_PH(gat, 2, 2);
  f32x2 dr[2] = {{0.f, 0.f}, {0.f, 0.f}};
  {
    float* dp = p.dxl + ((long)b * GN + r0) * p.ldd + hs * C + k0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float rs = slope * sRS[r0 + r];
      const f32x2 d0 = f32x2{rs, rs} + f32x2{oms, oms} * ql[0][r], d1 = f32x2{rs, rs} + f32x2{oms, oms} * ql[1][r];
      dr[0] += xl2[0][r] * d0;
      dr[1] += xl2[1][r] * d1;
      f32x4 v = *reinterpret_cast<const f32x4*>(dp);
      v += f32x4{at2[0][0] * d0[0], at2[0][1] * d0[1], at2[1][0] * d1[0], at2[1][1] * d1[1]};
      *reinterpret_cast<f32x4*>(dp) = v;
      dp += p.ldd;
    }
  }
  const float da = slope * sDAc[0];
  f32x4 datt, dwe;
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float lin = da + oms * oct_sum(qa[pp][h]);                   // slope DA + (1 - slope) Qa[k]
      datt[2 * pp + h] = (oct_sum(dr[pp][h]) + dc[pp][h]) * kStepDown + wep[2 * pp + h] * lin;
      dwe[2 * pp + h] = at2[pp][h] * lin;
    }
  if (q8 == 0) {
    const long o = ((long)b * a.Hs + hs) * C + k0;
    *reinterpret_cast<f32x4*>(p.datt_part + o) = datt;
    *reinterpret_cast<f32x4*>(p.dwe_part + o) = dwe;
  }
  CTVAE_PH(gat, 2, 3);
}

// d adj[b][r][c] = edge[r][c] * (sum_slots d a'[r][c] + (sum_slots d a'[c][c]) / deg[c]); grid (4, B), 256 threads: a workgroup owns
// 16 rows of the sample (thread = (row, 4 columns)); the in-degrees come from the whole adjacency (16 KB, every workgroup counts
// them itself), the diagonal of the summed slots from Hs scattered loads per column.  All loads of a thread are issued before the
// first use: with one workgroup per sample and a loop over the slots this was a 29 us launch for 27 MB.
__global__ __launch_bounds__(256) void gat_adj_reduce_kernel(const float* __restrict__ dattr, const float* __restrict__ adj,
                                                            float* __restrict__ dadj, int Hs, int accumulate) {
  __shared__ float sD[16][GN];
  __shared__ float sDiag[GN], sDeg[GN];
  const int tid = threadIdx.x, b = blockIdx.y, tr = tid >> 4, tc = tid & 15, r = 16 * blockIdx.x + tr;
  const float* ab = adj + (long)b * GN * GN;
  f32x4 cnt4[4];                                       // rows tr, tr + 16, tr + 32, tr + 48 of the whole adjacency: column counts
#pragma unroll
  for (int i = 0; i < 4; ++i) cnt4[i] = *reinterpret_cast<const f32x4*>(ab + (tr + 16 * i) * GN + 4 * tc);
  const f32x4 av = *reinterpret_cast<const f32x4*>(ab + r * GN + 4 * tc);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  float dg = 0.f;
  for (int h0 = 0; h0 < Hs; h0 += 4) {                 // four slots' loads in flight (a plain loop over the slots waits for each), added in slot order
    f32x4 t[4];
    float d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int h = h0 + u < Hs ? h0 + u : Hs - 1;
      t[u] = *reinterpret_cast<const f32x4*>(dattr + (((long)b * Hs + h) * GN + r) * GN + 4 * tc);
      d[u] = dattr[(((long)b * Hs + h) * GN + (tid & (GN - 1))) * GN + (tid & (GN - 1))];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (h0 + u < Hs) { s += t[u]; dg += d[u]; }
  }
  const f32x4 prev = accumulate ? *reinterpret_cast<const f32x4*>(dadj + ((long)b * GN + r) * GN + 4 * tc) : f32x4{0.f, 0.f, 0.f, 0.f};
  float cd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (cnt4[i][j] != 0.f && tr + 16 * i != 4 * tc + j) cd[j] += 1.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) sD[tr][4 * tc + j] = cd[j];
  if (tid < GN) sDiag[tid] = dg;
  __syncthreads();
  if (tid < GN) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sD[q][tid];
    sDeg[tid] = fmaxf(t, 1.f);
  }
  __syncthreads();
  f32x4 o = prev;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * tc + j;
    const bool e = av[j] != 0.f && r != c;
    o[j] += e ? s[j] + sDiag[c] / sDeg[c] : 0.f;
  }
  *reinterpret_cast<f32x4*>(dadj + ((long)b * GN + r) * GN + 4 * tc) = o;
}

size_t fwd_smem(int C) { return ((size_t)2 * C * LS + GN * SS + 2 * (C + 4) + C + 4 + 2 * GN * 4 + 4 * GN + 2 * GN) * sizeof(float); }
size_t bwd_smem(int C) { return ((size_t)2 * C * LS + GN * SS + 2 * (C + 4) + 2 * GN + 2 * GN) * sizeof(float); }

bool args_ok(const GatLayerArgs& a) {
  return a.xl && a.xr && a.adj && a.we && a.att && a.bias && a.out && a.alpha && a.B > 0 && a.Hs > 0 && a.C >= 16 && a.C <= 128 &&
         a.C % 4 == 0 && a.ld % 4 == 0 && a.ldo % 4 == 0 && a.ld >= a.Hs * a.C && a.ldo >= a.Hs * a.C && a.slope > 0.f && a.slope < 1.f && (a.act == ACT_NONE || a.act == ACT_LRELU);
}

}  // namespace

int launch_gat_layer_forward(const GatLayerArgs& a, hipStream_t st) {
  if (!args_ok(a)) return kErrBadArg;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gat_layer_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const double pairs = (double)a.B * a.Hs * GN * GN;
  // per (pair, channel): add + clamped fma + fma in the pair loop (5 FLOP), the aggregation's multiply-add on MFMA (2)
  ProfScope ps(a.C <= 64 ? "gat_layer_fwd_kernel[C<=64]" : "gat_layer_fwd_kernel[C>64]", st, 7.0 * pairs * a.C, 4.0 * a.B * a.Hs * (3.0 * GN * a.C + 2.0 * GN * GN));
  const dim3 grid(a.Hs, a.B);
  hipLaunchKernelGGL(gat_layer_fwd_kernel, grid, dim3(256), fwd_smem(a.C), st, a);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_gat_layer_backward(const GatBwdArgs& p, float* dadj, int accumulate_dadj, hipStream_t st) {
  const GatLayerArgs& a = p.f;
  if (!args_ok(a) || !p.g_out || !p.dS || !p.dattr || !p.dxl || !p.dxr || !p.dbias_part || !p.datt_part || !p.dwe_part ||
      p.ldd < a.Hs * a.C || p.ldd % 4)
    return kErrBadArg;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gat_layer_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const double pairs = (double)a.B * a.Hs * GN * GN;
  const dim3 grid(a.Hs, a.B);
  {
    // pair loop 5 FLOP per (pair, channel), d alpha and the aggregation's d xl on MFMA 2 each
    ProfScope ps(a.C <= 64 ? "gat_layer_bwd_kernel[C<=64]" : "gat_layer_bwd_kernel[C>64]", st, 9.0 * pairs * a.C, 4.0 * a.B * a.Hs * (4.0 * GN * a.C + 4.0 * GN * GN));
    hipLaunchKernelGGL(gat_layer_bwd_kernel, grid, dim3(256), bwd_smem(a.C), st, p);
    CTVAE_LAUNCH_CHECK();
  }
  {
    ProfScope ps("gat_proj_bwd_kernel", st, 9.0 * pairs * a.C, 4.0 * a.B * a.Hs * (4.0 * GN * a.C + 2.0 * GN * GN));
    hipLaunchKernelGGL(gat_proj_bwd_kernel, grid, dim3(256), 0, st, p);
    CTVAE_LAUNCH_CHECK();
  }
  if (dadj != nullptr) {
    ProfScope ps("gat_adj_reduce_kernel", st, 0.0, 4.0 * a.B * (a.Hs + 2.0) * GN * GN);
    hipLaunchKernelGGL(gat_adj_reduce_kernel, dim3(4, a.B), dim3(256), 0, st, p.dattr, a.adj, dadj, a.Hs, accumulate_dadj);
    CTVAE_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace ctvae
