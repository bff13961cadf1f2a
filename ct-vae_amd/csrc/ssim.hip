// The multi-scale SSIM reconstruction loss of MSSIMVAE (mssim_vae.py:182-279) and its gradient w.r.t. the first picture.
//
//   level l = 0..4 on pictures of side S = 64 >> l (2x2 average pooling between levels, mssim_vae.py:265-266):
//     mu1 = G*a, mu2 = G*b, s11 = G*(a a) - mu1^2, s22 = G*(b b) - mu2^2, s12 = G*(a b) - mu1 mu2      (per channel, zero padding 5)
//     cs = (2 s12 + C2) / (s11 + s22 + C2),   ssim = (2 mu1 mu2 + C1) / (mu1^2 + mu2^2 + C1) * cs,    means over B, C, H, W
//   loss = 1 - prod_{i<4} (mcs_i^w_i * mssim_4^w_4)          (mssim_vae.py:274-279: the last level's power multiplies EACH factor)
//   G = g (x) g with the reference's 11-tap window g[x] ~ exp(+(x-5)^2 / (2 1.5^2)) (positive exponent, mssim_vae.py:203-206),
//   handed in by the host.
//
// One workgroup per (image, channel): both pictures of that channel live in LDS with a zeroed 5-pixel border, every thread
// computes strips of four pixels from an 11 x 14 window (the five moments share the loaded values).  Forward: per-level sums of the
// two maps per workgroup, merged in a fixed order by ssim_finish_kernel, which also leaves d loss / d (map element) per level for
// the backward pass.  Backward: the whole pyramid of both pictures in LDS, levels from coarse to fine: per pixel the partial
// derivatives P, Q, R of (alpha ssim + beta cs) w.r.t. mu1, G*(a a), G*(a b); grad a = G*P + 2 a G*Q + b G*R (G is symmetric, zero
// padding on both sides) plus a quarter of the coarser level's gradient of the pixel's 2x2 block.  Bit-reproducible.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

namespace {

constexpr int S0 = 64, NLEV = 5, KW = 11, BRD = 5;
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

struct SsimArgs {
  const float* a;     // [B,64,64,C] the picture that receives the gradient (reconstruction)
  const float* b;     // [B,64,64,C]
  float* part;        // [NLEV][B*C][2] sums of the ssim / cs maps per workgroup
  const float* coef;  // backward: [NLEV][2] = d loss / d (ssim map element), d loss / d (cs map element)
  const float* g_up;  // backward: upstream gradient (scalar on the device)
  float* g_a;         // backward: [B,64,64,C]
  int B, C;
  float g[KW];
};

__host__ __device__ constexpr int lev_side(int l) { return S0 >> l; }
__host__ __device__ constexpr int lev_pitch(int l) { return (S0 >> l) + 2 * BRD; }
__host__ __device__ constexpr int lev_floats(int l) { return lev_pitch(l) * lev_pitch(l); }
constexpr int lev_off_c(int l) { return l == 0 ? 0 : lev_off_c(l - 1) + lev_floats(l - 1); }
constexpr int kPyrFloats = lev_off_c(NLEV);   // 8436
__device__ __forceinline__ int lev_off(int l) {   // (no recursion on the device)
  return l == 0 ? 0 : (l == 1 ? lev_off_c(1) : (l == 2 ? lev_off_c(2) : (l == 3 ? lev_off_c(3) : lev_off_c(4))));
}

__device__ __forceinline__ float block_sum(float v, float* red /* [4] */) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// the five windowed moments of a strip of four pixels (y, x0 .. x0+3); pa / pb point at the strip's window origin
__device__ __forceinline__ void moments5(const float* pa, const float* pb, int pitch, const float (&g)[KW], float (&m1)[4], float (&m2)[4],
                                         float (&e11)[4], float (&e22)[4], float (&e12)[4]) {
#pragma unroll
  for (int o = 0; o < 4; ++o) m1[o] = m2[o] = e11[o] = e22[o] = e12[o] = 0.f;
  for (int dy = 0; dy < KW; ++dy) {
    float va[14], vb[14], aa[14], bb[14], ab[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      va[k] = pa[dy * pitch + k];
      vb[k] = pb[dy * pitch + k];
      aa[k] = va[k] * va[k];
      bb[k] = vb[k] * vb[k];
      ab[k] = va[k] * vb[k];
    }
    const float gy = g[dy];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float h1 = 0.f, h2 = 0.f, h11 = 0.f, h22 = 0.f, h12 = 0.f;
#pragma unroll
      for (int dx = 0; dx < KW; ++dx) {
        const float w = g[dx];
        h1 += w * va[o + dx];
        h2 += w * vb[o + dx];
        h11 += w * aa[o + dx];
        h22 += w * bb[o + dx];
        h12 += w * ab[o + dx];
      }
      m1[o] += gy * h1; m2[o] += gy * h2; e11[o] += gy * h11; e22[o] += gy * h22; e12[o] += gy * h12;
    }
  }
}

// a picture level in LDS: interior (S x S) at offset BRD, BRD in a pitch x pitch square whose border is zero
__device__ __forceinline__ void load_level0(const float* src, int C, int img, int ch, float* dst) {
  constexpr int P = lev_pitch(0);
  for (int e = threadIdx.x; e < P * P; e += 256) {
    const int y = e / P - BRD, x = e % P - BRD;
    float v = 0.f;
    if ((unsigned)y < (unsigned)S0 && (unsigned)x < (unsigned)S0) v = src[(((long)img * S0 + y) * S0 + x) * C + ch];
    dst[e] = v;
  }
}
// 2x2 average pooling of level l (at src) into level l+1 (at dst, its own pitch; border zeroed)
__device__ __forceinline__ void pool_level(const float* src, int l, float* dst) {
  const int Ps = lev_pitch(l), Sd = lev_side(l + 1), Pd = lev_pitch(l + 1);
  for (int e = threadIdx.x; e < Pd * Pd; e += 256) {
    const int y = e / Pd - BRD, x = e % Pd - BRD;
    float v = 0.f;
    if ((unsigned)y < (unsigned)Sd && (unsigned)x < (unsigned)Sd) {
      const float* p = src + (BRD + 2 * y) * Ps + BRD + 2 * x;
      v = ((p[0] + p[1]) + (p[Ps] + p[Ps + 1])) * 0.25f;
    }
    dst[e] = v;
  }
}

// ---- forward: per-level sums of the ssim and cs maps of one (image, channel) ------------------------------------------------
__global__ __launch_bounds__(256) void ssim_fwd_kernel(const SsimArgs a) {
  extern __shared__ float smem[];
  float* pa = smem;                 // the whole pyramid of a, then of b
  float* pb = smem + kPyrFloats;
  __shared__ float red[4];
  const int wg = blockIdx.x, img = wg / a.C, ch = wg - img * a.C;
  float g[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) g[k] = a.g[k];
  load_level0(a.a, a.C, img, ch, pa);
  load_level0(a.b, a.C, img, ch, pb);
  __syncthreads();
  for (int l = 0; l + 1 < NLEV; ++l) {
    pool_level(pa + lev_off(l), l, pa + lev_off(l + 1));
    pool_level(pb + lev_off(l), l, pb + lev_off(l + 1));
    __syncthreads();
  }
  for (int l = 0; l < NLEV; ++l) {
    const int S = lev_side(l), P = lev_pitch(l);
    const float* la = pa + lev_off(l);
    const float* lb = pb + lev_off(l);
    float ssum = 0.f, csum = 0.f;
    for (int st = threadIdx.x; st < S * S / 4; st += 256) {
      const int y = st / (S / 4), x0 = (st - y * (S / 4)) * 4;
      float m1[4], m2[4], e11[4], e22[4], e12[4];
      moments5(la + y * P + x0, lb + y * P + x0, P, g, m1, m2, e11, e22, e12);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const float mu11 = m1[o] * m1[o], mu22 = m2[o] * m2[o], mu12 = m1[o] * m2[o];
        const float v1 = 2.f * (e12[o] - mu12) + kC2, v2 = (e11[o] - mu11) + (e22[o] - mu22) + kC2;
        csum += v1 / v2;
        ssum += ((2.f * mu12 + kC1) * v1) / ((mu11 + mu22 + kC1) * v2);
      }
    }
    const float st = block_sum(ssum, red), ct = block_sum(csum, red);
    if (threadIdx.x == 0) {
      a.part[((long)l * gridDim.x + wg) * 2] = st;
      a.part[((long)l * gridDim.x + wg) * 2 + 1] = ct;
    }
  }
}

// ---- finish: level means, the loss, and d loss / d (map element) per level --------------------------------------------------
__global__ __launch_bounds__(256) void ssim_finish_kernel(const float* __restrict__ part, int nwg, float* __restrict__ loss,
                                                          float* __restrict__ coef, float w0, float w1, float w2, float w3, float w4) {
  __shared__ float red[4];
  __shared__ float ms[NLEV], mc[NLEV];
  const float w[NLEV] = {w0, w1, w2, w3, w4};
  for (int l = 0; l < NLEV; ++l) {
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < nwg; i += 256) {
      s += part[((long)l * nwg + i) * 2];
      c += part[((long)l * nwg + i) * 2 + 1];
    }
    const float st = block_sum(s, red), ct = block_sum(c, red);
    if (threadIdx.x == 0) {
      const float n = (float)nwg * (float)(lev_side(l) * lev_side(l));
      ms[l] = st / n;
      mc[l] = ct / n;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float p4 = powf(ms[NLEV - 1], w[NLEV - 1]);
    float out = 1.f;
    for (int i = 0; i + 1 < NLEV; ++i) out *= powf(mc[i], w[i]) * p4;
    loss[0] = 1.f - out;
    for (int l = 0; l < NLEV; ++l) {
      const float n = (float)nwg * (float)(lev_side(l) * lev_side(l));
      // d out / d mc_l = out w_l / mc_l (l < 4);  d out / d ms_4 = out * 4 w_4 / ms_4;  loss = 1 - out
      coef[2 * l] = l == NLEV - 1 ? -out * (float)(NLEV - 1) * w[l] / ms[l] / n : 0.f;
      coef[2 * l + 1] = l < NLEV - 1 ? -out * w[l] / mc[l] / n : 0.f;
    }
  }
}

// ---- backward: gradient w.r.t. picture a ----------------------------------------------------------------------------------
// filter three maps with the window for a strip of four pixels
__device__ __forceinline__ void filter3(const float* pp, const float* pq, const float* pr, int pitch, const float (&g)[KW], float (&fp)[4],
                                        float (&fq)[4], float (&fr)[4]) {
#pragma unroll
  for (int o = 0; o < 4; ++o) fp[o] = fq[o] = fr[o] = 0.f;
  for (int dy = 0; dy < KW; ++dy) {
    float vp[14], vq[14], vr[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      vp[k] = pp[dy * pitch + k];
      vq[k] = pq[dy * pitch + k];
      vr[k] = pr[dy * pitch + k];
    }
    const float gy = g[dy];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float hp = 0.f, hq = 0.f, hr = 0.f;
#pragma unroll
      for (int dx = 0; dx < KW; ++dx) {
        const float w = g[dx];
        hp += w * vp[o + dx];
        hq += w * vq[o + dx];
        hr += w * vr[o + dx];
      }
      fp[o] += gy * hp; fq[o] += gy * hq; fr[o] += gy * hr;
    }
  }
}

constexpr int kPqrFloats = lev_floats(0);                        // one map with border at the finest level
constexpr int kCarryA = S0 * S0, kCarryB = (S0 / 2) * (S0 / 2);  // gradient of the level being formed / of the coarser one
constexpr size_t kSsimFwdSmem = (size_t)2 * kPyrFloats * 4;
constexpr size_t kSsimBwdSmem = (size_t)(2 * kPyrFloats + 3 * kPqrFloats + kCarryA + kCarryB) * 4;   // 153.7 KB

__global__ __launch_bounds__(256) void ssim_bwd_kernel(const SsimArgs a) {
  extern __shared__ float smem[];
  float* pa = smem;
  float* pb = pa + kPyrFloats;
  float* mP = pb + kPyrFloats;
  float* mQ = mP + kPqrFloats;
  float* mR = mQ + kPqrFloats;
  float* cA = mR + kPqrFloats;   // levels 0, 2, 4
  float* cB = cA + kCarryA;      // levels 1, 3
  const int wg = blockIdx.x, img = wg / a.C, ch = wg - img * a.C;
  float g[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) g[k] = a.g[k];
  load_level0(a.a, a.C, img, ch, pa);
  load_level0(a.b, a.C, img, ch, pb);
  __syncthreads();
  for (int l = 0; l + 1 < NLEV; ++l) {
    pool_level(pa + lev_off(l), l, pa + lev_off(l + 1));
    pool_level(pb + lev_off(l), l, pb + lev_off(l + 1));
    __syncthreads();
  }
  for (int l = NLEV - 1; l >= 0; --l) {
    const int S = lev_side(l), P = lev_pitch(l);
    const float* la = pa + lev_off(l);
    const float* lb = pb + lev_off(l);
    const float alpha = a.coef[2 * l], beta = a.coef[2 * l + 1];
    for (int e = threadIdx.x; e < P * P; e += 256) mP[e] = mQ[e] = mR[e] = 0.f;   // borders (and the previous level's maps)
    __syncthreads();
    for (int st = threadIdx.x; st < S * S / 4; st += 256) {
      const int y = st / (S / 4), x0 = (st - y * (S / 4)) * 4;
      float m1[4], m2[4], e11[4], e22[4], e12[4];
      moments5(la + y * P + x0, lb + y * P + x0, P, g, m1, m2, e11, e22, e12);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const float mu11 = m1[o] * m1[o], mu22 = m2[o] * m2[o], mu12 = m1[o] * m2[o];
        const float v1 = 2.f * (e12[o] - mu12) + kC2, v2 = (e11[o] - mu11) + (e22[o] - mu22) + kC2;
        const float A1 = 2.f * mu12 + kC1, A2 = mu11 + mu22 + kC1;
        const float cs = v1 / v2, lum = A1 / A2;
        const float dl = 2.f * (m2[o] - lum * m1[o]) / A2;            // d lum / d mu1
        const float dc = 2.f * (cs * m1[o] - m2[o]) / v2;             // d cs / d mu1 (through s12 and s11)
        const float k = alpha * lum + beta;                           // weight of cs in alpha ssim + beta cs
        const int idx = (BRD + y) * P + BRD + x0 + o;
        mP[idx] = alpha * cs * dl + k * dc;
        mQ[idx] = -k * cs / v2;                                       // d / d G*(a a)
        mR[idx] = 2.f * k / v2;                                       // d / d G*(a b)
      }
    }
    __syncthreads();
    float* cur = (l & 1) ? cB : cA;
    const float* prev = (l & 1) ? cA : cB;     // gradient of level l + 1, side S / 2
    for (int st = threadIdx.x; st < S * S / 4; st += 256) {
      const int y = st / (S / 4), x0 = (st - y * (S / 4)) * 4;
      float fp[4], fq[4], fr[4];
      filter3(mP + y * P + x0, mQ + y * P + x0, mR + y * P + x0, P, g, fp, fq, fr);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const float av = la[(BRD + y) * P + BRD + x0 + o], bv = lb[(BRD + y) * P + BRD + x0 + o];
        float gr = fp[o] + 2.f * av * fq[o] + bv * fr[o];
        if (l + 1 < NLEV) gr += 0.25f * prev[(y >> 1) * (S >> 1) + ((x0 + o) >> 1)];
        cur[y * S + x0 + o] = gr;
      }
    }
    __syncthreads();
  }
  const float up = a.g_up != nullptr ? a.g_up[0] : 1.f;
  for (int e = threadIdx.x; e < S0 * S0; e += 256) a.g_a[((long)img * S0 * S0 + e) * a.C + ch] = up * cA[e];
}

}  // namespace

int launch_ssim_forward(const float* a, const float* b, const float* window, float* part, float* loss, float* coef, int B, int C, int H,
                        int W, const float* weights, hipStream_t st) {
  if (!a || !b || !window || !part || !loss || !coef || !weights || B <= 0 || C <= 0 || H != S0 || W != S0) return kErrBadArg;
  SsimArgs s{};
  s.a = a; s.b = b; s.part = part; s.B = B; s.C = C;
  for (int k = 0; k < KW; ++k) s.g[k] = window[k];
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ssim_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSsimFwdSmem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ssim_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSsimBwdSmem);
    attr = true;
  }
  {
    ProfScope ps("ssim_fwd_kernel", st, 0.0, 8.0 * (double)B * C * S0 * S0);
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3(B * C), dim3(256), kSsimFwdSmem, st, s);
    CTVAE_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(ssim_finish_kernel, dim3(1), dim3(256), 0, st, part, B * C, loss, coef, weights[0], weights[1], weights[2],
                     weights[3], weights[4]);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_ssim_backward(const float* a, const float* b, const float* window, const float* coef, const float* g_loss, float* g_a, int B,
                         int C, int H, int W, hipStream_t st) {
  if (!a || !b || !window || !coef || !g_a || B <= 0 || C <= 0 || H != S0 || W != S0) return kErrBadArg;
  SsimArgs s{};
  s.a = a; s.b = b; s.coef = coef; s.g_up = g_loss; s.g_a = g_a; s.B = B; s.C = C;
  for (int k = 0; k < KW; ++k) s.g[k] = window[k];
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ssim_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSsimBwdSmem);
    attr = true;
  }
  ProfScope ps("ssim_bwd_kernel", st, 0.0, 12.0 * (double)B * C * S0 * S0);
  hipLaunchKernelGGL(ssim_bwd_kernel, dim3(B * C), dim3(256), kSsimBwdSmem, st, s);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
