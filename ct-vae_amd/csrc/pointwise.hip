// HBM-bound pointwise kernels of the training step: layout changes at the NCHW API boundary,
// activation backward, Gaussian reparameterisation (vanilla_vae.py:107-117) and the flat fused Adam
// update (experiment.py:158-160).  Roofline for all of them: bytes moved / 8 TB/s.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

// out[b][p][c] = in[b][c][p]   (NCHW -> NHWC with p = h*W+w), or the inverse with to_nhwc = 0
__global__ __launch_bounds__(256) void permute_cp_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                         int C, int P, int to_nhwc) {
  const long n = (long)B * C * P;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    if (to_nhwc) {  // i indexes out [b][p][c]
      int c = (int)(i % C);
      long t = i / C;
      int p = (int)(t % P);
      long b = t / P;
      out[i] = in[(b * C + c) * P + p];
    } else {  // i indexes out [b][c][p]
      int p = (int)(i % P);
      long t = i / P;
      int c = (int)(t % C);
      long b = t / C;
      out[i] = in[(b * P + p) * C + c];
    }
  }
}

// The picture's case, C = 3 and P % 4 == 0: one thread moves 4 pixels with three 16-byte loads and three 16-byte stores
// (the element-wise kernel above spends two 64-bit divisions and a 4-byte access pair per element: 11 us for the 12.6 MB batch
// of VanillaVAE bs = 256 at the head of every step).
__global__ __launch_bounds__(256) void permute3_kernel(const float* __restrict__ in, float* __restrict__ out, long quads, int P4,
                                                       int to_nhwc) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // quad index: (b, p/4)
  if (i >= quads) return;
  const long b = i / P4;
  const int q = (int)(i - b * P4);
  const f32x4* planes = reinterpret_cast<const f32x4*>(in) + b * 3 * P4 + q;        // NCHW side: plane c at + c*P4
  f32x4* oplanes = reinterpret_cast<f32x4*>(out) + b * 3 * P4 + q;
  const f32x4* pix = reinterpret_cast<const f32x4*>(in) + (b * P4 + q) * 3;         // NHWC side: 12 consecutive floats
  f32x4* opix = reinterpret_cast<f32x4*>(out) + (b * P4 + q) * 3;
  if (to_nhwc) {
    const f32x4 r = planes[0], g = planes[P4], bl = planes[2 * P4];
    opix[0] = f32x4{r[0], g[0], bl[0], r[1]};
    opix[1] = f32x4{g[1], bl[1], r[2], g[2]};
    opix[2] = f32x4{bl[2], r[3], g[3], bl[3]};
  } else {
    const f32x4 v0 = pix[0], v1 = pix[1], v2 = pix[2];
    oplanes[0] = f32x4{v0[0], v0[3], v1[2], v2[1]};
    oplanes[P4] = f32x4{v0[1], v1[0], v1[3], v2[2]};
    oplanes[2 * P4] = f32x4{v0[2], v1[1], v2[0], v2[3]};
  }
}

// gin = gout * act'(out)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ out,
                                                      float* __restrict__ gin, long n, int act) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) gin[i] = gout[i] * act_bwd_from_out(out[i], act);
}

// out = act(in)   (standalone nn.LeakyReLU sites, mcq_vae.py:185,216, when not fused into a producer)
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, long n, int act) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = act_fwd(in[i], act);
}

// z = eps * exp(0.5*logvar) + mu          (row strides allow mu/logvar to be column slices of one head GEMM)
__global__ __launch_bounds__(256) void reparam_fwd_kernel(const float* __restrict__ mu, long mu_rs, const float* __restrict__ lv,
                                                          long lv_rs, const float* __restrict__ eps, float* __restrict__ z,
                                                          int B, int L) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * L) return;
  const int b = i / L, d = i - b * L;
  z[i] = eps[i] * expf(0.5f * lv[b * lv_rs + d]) + mu[b * mu_rs + d];
}

// g_mu = g_z ; g_lv = g_z * eps * 0.5 * exp(0.5*logvar)
__global__ __launch_bounds__(256) void reparam_bwd_kernel(const float* __restrict__ gz, const float* __restrict__ lv, long lv_rs,
                                                          const float* __restrict__ eps, float* __restrict__ gmu,
                                                          float* __restrict__ glv, int B, int L) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * L) return;
  const int b = i / L, d = i - b * L;
  const float g = gz[i];
  gmu[i] = g;
  glv[i] = g * eps[i] * 0.5f * expf(0.5f * lv[b * lv_rs + d]);
}

// ---- Gaussian latent as one node (vanilla_vae.py:85-92,107-122): heads [B][2L] = fc_mu | fc_var output ----------------------
// forward : z = eps * exp(0.5*logvar) + mu with mu = heads[:, :L], logvar = heads[:, L:]; eps given, or drawn here:
//           Philox4x32-10 keyed by rng[0] with counter (rng[1], element quad), Box-Muller -> N(0,1); eps is kept for backward.
// backward: g_heads[:, :L] = g_mu + g_z,  g_heads[:, L:] = g_logvar + g_z * eps * 0.5 * exp(0.5*logvar) in one launch (autograd
//           otherwise adds the two contributions with a launch each and concatenates with a third); when the noise was drawn
//           here, thread 0 advances rng[1] -- the forward kernel of the next step runs strictly later.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// slices != nullptr: the heads are still S raw split-K slices [S][B][2L] of the fc_mu | fc_var GEMM (SplitKRaw, pixel-major):
// this kernel sums them (+ bias), writes `heads` for the loss / the backward pass and goes on -- the split-K finish launch between
// the GEMM and the latent disappears
__global__ __launch_bounds__(256) void gauss_latent_fwd_kernel(float* __restrict__ heads, const float* __restrict__ eps_in,
                                                              const unsigned long long* __restrict__ rng, float* __restrict__ eps_out,
                                                              float* __restrict__ z, int B, int L, const float* __restrict__ slices,
                                                              int S, const float* __restrict__ bias) {
  const int q = blockIdx.x * 256 + threadIdx.x;          // one thread = four consecutive elements (L % 4 == 0)
  const int i = 4 * q;
  if (i >= B * L) return;
  const int b = i / L, d = i - b * L;
  f32x4 e;
  if (eps_in != nullptr) {
    e = *reinterpret_cast<const f32x4*>(eps_in + i);
  } else {
    const unsigned long long key = rng[0], ctr = rng[1];
    unsigned o[4];
    philox4x32_10((unsigned)q, (unsigned)ctr, (unsigned)(ctr >> 32), 0x5eedu, (unsigned)key, (unsigned)(key >> 32), o);
    // Box-Muller on two pairs; u in (0, 1]
    const float u0 = ((o[0] >> 8) + 1) * (1.f / 16777216.f), u1 = (o[1] >> 8) * (1.f / 16777216.f);
    const float u2 = ((o[2] >> 8) + 1) * (1.f / 16777216.f), u3 = (o[3] >> 8) * (1.f / 16777216.f);
    const float r0 = sqrtf(-2.f * __logf(u0)), r1 = sqrtf(-2.f * __logf(u2));
    float s0, c0, s1, c1;
    __sincosf(6.28318530718f * u1, &s0, &c0);
    __sincosf(6.28318530718f * u3, &s1, &c1);
    e = f32x4{r0 * c0, r0 * s0, r1 * c1, r1 * s1};
  }
  f32x4 mu, lv;
  if (slices != nullptr) {
    const long o = (long)b * 2 * L + d, st = (long)B * 2 * L;
    mu = f32x4{0.f, 0.f, 0.f, 0.f};
    lv = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < S; s0 += 8) {   // eight slices of loads in flight (a slice index beyond S re-reads the last one, weight 0)
      f32x4 tm[8], tl[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long so = (long)(s0 + u < S ? s0 + u : S - 1) * st + o;
        tm[u] = *reinterpret_cast<const f32x4*>(slices + so);
        tl[u] = *reinterpret_cast<const f32x4*>(slices + so + L);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (s0 + u < S) { mu += tm[u]; lv += tl[u]; }
      }
    }
    if (bias != nullptr) {
      mu += *reinterpret_cast<const f32x4*>(bias + d);
      lv += *reinterpret_cast<const f32x4*>(bias + L + d);
    }
    *reinterpret_cast<f32x4*>(heads + o) = mu;
    *reinterpret_cast<f32x4*>(heads + o + L) = lv;
  } else {
    mu = *reinterpret_cast<const f32x4*>(heads + (long)b * 2 * L + d);
    lv = *reinterpret_cast<const f32x4*>(heads + (long)b * 2 * L + L + d);
  }
  f32x4 zz;
#pragma unroll
  for (int k = 0; k < 4; ++k) zz[k] = e[k] * expf(0.5f * lv[k]) + mu[k];
  *reinterpret_cast<f32x4*>(z + i) = zz;
  *reinterpret_cast<f32x4*>(eps_out + i) = e;
}

__global__ __launch_bounds__(256) void gauss_latent_bwd_kernel(const float* __restrict__ g_mu, const float* __restrict__ g_lv,
                                                              const float* __restrict__ g_z, const float* __restrict__ heads,
                                                              const float* __restrict__ eps, float* __restrict__ g_heads,
                                                              unsigned long long* __restrict__ rng_bump, int B, int L, int gz_slices) {
  // gz_slices > 0: g_z is gz_slices raw split-K slices [S][B][L] of decoder_input's data gradient, summed here
  const int i = 4 * (blockIdx.x * 256 + threadIdx.x);
  if (i == 0 && rng_bump != nullptr) rng_bump[1] += 1ull;
  if (i >= B * L) return;
  const int b = i / L, d = i - b * L;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 gz = zero;
  if (g_z) {
    const int S = gz_slices > 0 ? gz_slices : 1;
    for (int s0 = 0; s0 < S; s0 += 8) {   // eight slices of loads in flight
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(g_z + (long)(s0 + u < S ? s0 + u : S - 1) * B * L + i);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (s0 + u < S) gz += t[u];
      }
    }
  }
  const f32x4 gm = g_mu ? *reinterpret_cast<const f32x4*>(g_mu + i) : zero;
  const f32x4 gl = g_lv ? *reinterpret_cast<const f32x4*>(g_lv + i) : zero;
  const f32x4 lv = *reinterpret_cast<const f32x4*>(heads + (long)b * 2 * L + L + d);
  const f32x4 e = *reinterpret_cast<const f32x4*>(eps + i);
  f32x4 o1;
#pragma unroll
  for (int k = 0; k < 4; ++k) o1[k] = gl[k] + gz[k] * e[k] * 0.5f * expf(0.5f * lv[k]);
  *reinterpret_cast<f32x4*>(g_heads + (long)b * 2 * L + d) = gm + gz;
  *reinterpret_cast<f32x4*>(g_heads + (long)b * 2 * L + L + d) = o1;
}

// ---- Adam ---------------------------------------------------------------------------------------------
// state[0]=step (as float), [1]=lr, [2]=beta1, [3]=beta2, [4]=eps, [5]=weight_decay, [6]=beta1^t, [7]=beta2^t,
// [16 + 16k], k = 0..64: ticket counters (bits), zero between launches -- kAdamStateFloats floats in all
constexpr int kAdamTicketOffset = 16, kAdamMaxWgs = 2048, kAdamStateFloats = kAdamTicketOffset + 16 * (1 + kAdamMaxWgs / 32);
// torch.optim.Adam (no amsgrad): g += wd*p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// The step counter's advance rides in this launch: every thread forms beta^t of THIS step from the stored beta^(t-1), and the
// workgroup that finishes last -- every other one has read the state by then -- writes the advanced state back.  With ONE
// ticket counter the 2048 same-address atomics serialised behind the kernel (19.5 -> 37.7 us, tools/negative), and fewer,
// fatter workgroups stream slower (512: 27.5 us, 256: 43 us): hence the two-level ticket below.
template <bool VEC>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, float* state, long n, float grad_scale) {
  const float lr = state[1], b1 = state[2], b2 = state[3], eps = state[4], wd = state[5];
  const float b1t = state[6] * b1, b2t = state[7] * b2;
  const float bc1 = 1.f - b1t, bc2 = 1.f - b2t;
  const float step_size = lr / bc1, bc2s = sqrtf(bc2);
  const long stride = (long)gridDim.x * 256;
  auto upd = [&](float& pi, float gi, float& mi, float& vi) {
    gi = gi * grad_scale + wd * pi;
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    pi = pi - step_size * (mi / (sqrtf(vi) / bc2s + eps));
  };
  const long n4 = VEC ? n / 4 : 0;
  if constexpr (VEC) {   // all four buffers 16-byte aligned: 16-byte accesses, two quads per thread in flight
    f32x4* p4 = reinterpret_cast<f32x4*>(p);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    f32x4* m4 = reinterpret_cast<f32x4*>(m);
    f32x4* v4 = reinterpret_cast<f32x4*>(v);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 2 * stride) {
      const long i2 = i + stride;
      const bool two = i2 < n4;
      f32x4 pa = p4[i], ga = g4[i], ma = m4[i], va = v4[i];
      f32x4 pb = two ? p4[i2] : pa, gb = two ? g4[i2] : ga, mb = two ? m4[i2] : ma, vb = two ? v4[i2] : va;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float pq = pa[q], mq = ma[q], vq = va[q];
        upd(pq, ga[q], mq, vq);
        pa[q] = pq; ma[q] = mq; va[q] = vq;
      }
      m4[i] = ma; v4[i] = va; p4[i] = pa;
      if (two) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float pq = pb[q], mq = mb[q], vq = vb[q];
          upd(pq, gb[q], mq, vq);
          pb[q] = pq; mb[q] = mq; vb[q] = vq;
        }
        m4[i2] = mb; v4[i2] = vb; p4[i2] = pb;
      }
    }
  }
  for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float pi = p[i], mi = m[i], vi = v[i];
    upd(pi, g[i], mi, vi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
  }
  __syncthreads();   // every thread of this workgroup holds its copy of the state
  if (threadIdx.x == 0) {
    // two-level ticket, every counter on its own 64-byte line: 32 workgroups share a first-level counter, the last of each
    // group takes a ticket of the top-level one (<= 64 groups)
    unsigned* tk = reinterpret_cast<unsigned*>(state + kAdamTicketOffset);
    const unsigned grp = blockIdx.x >> 5, ngrp = (gridDim.x + 31) >> 5;
    const unsigned gsize = gridDim.x - grp * 32 < 32 ? gridDim.x - grp * 32 : 32;
    if (atomicAdd(tk + 16 * (1 + grp), 1u) == gsize - 1) {
      atomicExch(tk + 16 * (1 + grp), 0u);
      if (atomicAdd(tk, 1u) == ngrp - 1) {
        atomicExch(tk, 0u);
        state[0] += 1.f;
        state[6] = b1t;
        state[7] = b2t;
      }
    }
  }
}

// Straight-through Bernoulli sample through a 2-class Gumbel-softmax (tau = 1, hard) of
// log(clamp([1-p, p], 1e-4)) (ct_mcq_vae.py:126,177-183).  noise = 2 standard Gumbel draws/element.
__global__ __launch_bounds__(256) void gumbel_st_fwd_kernel(const float* __restrict__ p, const float* __restrict__ noise,
                                                            float* __restrict__ out, float* __restrict__ soft, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float pi = p[i];
    const float a0 = logf(fmaxf(1.f - pi, 1e-4f)) + noise[2 * i];
    const float a1 = logf(fmaxf(pi, 1e-4f)) + noise[2 * i + 1];
    const float y1 = 1.f / (1.f + expf(a0 - a1));
    const float hard = a1 > a0 ? 1.f : 0.f;      // argmax keeps the first maximum on ties
    out[i] = (hard - y1) + y1;                   // y_hard - y_soft.detach() + y_soft
    soft[i] = y1;
  }
}

__global__ __launch_bounds__(256) void gumbel_st_bwd_kernel(const float* __restrict__ go, const float* __restrict__ p,
                                                            const float* __restrict__ soft, float* __restrict__ gp, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float pi = p[i], y1 = soft[i];
    const float d = (pi > 1e-4f ? 1.f / pi : 0.f) + ((1.f - pi) > 1e-4f ? 1.f / (1.f - pi) : 0.f);
    gp[i] = go[i] * y1 * (1.f - y1) * d;
  }
}

static inline unsigned grid_for(long n, int cap = 4096) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

int launch_permute(const float* in, float* out, int B, int C, int P, int to_nhwc, hipStream_t st) {
  if (C == 3 && P % 4 == 0 && (reinterpret_cast<uintptr_t>(in) % 16) == 0 && (reinterpret_cast<uintptr_t>(out) % 16) == 0) {
    ProfScope ps("permute3_kernel", st, 0.0, 8.0 * (double)B * C * P);
    const long quads = (long)B * (P / 4);
    hipLaunchKernelGGL(permute3_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, in, out, quads, P / 4, to_nhwc);
    CTVAE_LAUNCH_CHECK();
    return 0;
  }
  ProfScope ps("permute_cp_kernel", st, 0.0, 8.0 * (double)B * C * P);
  hipLaunchKernelGGL(permute_cp_kernel, dim3(grid_for((long)B * C * P)), dim3(256), 0, st, in, out, B, C, P, to_nhwc);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_act_bwd(const float* gout, const float* out, float* gin, long n, int act, hipStream_t st) {
  ProfScope ps("act_bwd_kernel", st, 0.0, 12.0 * (double)n);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, gout, out, gin, n, act);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_act_fwd(const float* in, float* out, long n, int act, hipStream_t st) {
  hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, out, n, act);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_reparam_fwd(const float* mu, long mu_rs, const float* lv, long lv_rs, const float* eps, float* z, int B, int L,
                       hipStream_t st) {
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(ceil_div(B * L, 256)), dim3(256), 0, st, mu, mu_rs, lv, lv_rs, eps, z, B, L);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_reparam_bwd(const float* gz, const float* lv, long lv_rs, const float* eps, float* gmu, float* glv, int B, int L,
                       hipStream_t st) {
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(ceil_div(B * L, 256)), dim3(256), 0, st, gz, lv, lv_rs, eps, gmu, glv, B, L);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_gumbel_fwd(const float* p, const float* noise, float* out, float* soft, long n, hipStream_t st) {
  hipLaunchKernelGGL(gumbel_st_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, noise, out, soft, n);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_gumbel_bwd(const float* go, const float* p, const float* soft, float* gp, long n, hipStream_t st) {
  hipLaunchKernelGGL(gumbel_st_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, go, p, soft, gp, n);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

size_t adam_state_floats() { return kAdamStateFloats; }

int launch_adam(float* p, const float* g, float* m, float* v, float* state, long n, float grad_scale, hipStream_t st) {
  ProfScope ps("adam_kernel", st, 0.0, 28.0 * (double)n);
  static const int wgs = [] { const char* e = getenv("CTVAE_ADAM_WGS"); const int w = e ? atoi(e) : 1024; return w > kAdamMaxWgs ? kAdamMaxWgs : w; }();   // diagnostic
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) % 16) == 0;
  if (vec) hipLaunchKernelGGL(adam_kernel<true>, dim3(grid_for(n / 4, wgs)), dim3(256), 0, st, p, g, m, v, state, n, grad_scale);
  else hipLaunchKernelGGL(adam_kernel<false>, dim3(grid_for(n, wgs)), dim3(256), 0, st, p, g, m, v, state, n, grad_scale);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_gauss_latent_fwd(float* heads, const float* eps_in, const unsigned long long* rng, float* eps_out, float* z, int B,
                            int L, hipStream_t st, const float* slices, int S, const float* bias) {
  if (!heads || !eps_out || !z || (!eps_in && !rng) || B <= 0 || L <= 0 || L % 4 || (slices != nullptr && S < 1)) return kErrBadArg;
  ProfScope ps("gauss_latent_fwd_kernel", st, 0.0, 4.0 * (5.0 + (slices ? 2.0 * S : 0.0)) * B * L);
  hipLaunchKernelGGL(gauss_latent_fwd_kernel, dim3((B * L / 4 + 255) / 256), dim3(256), 0, st, heads, eps_in, rng, eps_out, z, B, L,
                     slices, S, bias);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

int launch_gauss_latent_bwd(const float* g_mu, const float* g_lv, const float* g_z, const float* heads, const float* eps,
                            float* g_heads, unsigned long long* rng_bump, int B, int L, hipStream_t st, int gz_slices) {
  if (!heads || !eps || !g_heads || B <= 0 || L <= 0 || L % 4 || (gz_slices > 0 && !g_z)) return kErrBadArg;
  ProfScope ps("gauss_latent_bwd_kernel", st, 0.0, 4.0 * (7.0 + (gz_slices > 1 ? gz_slices - 1.0 : 0.0)) * B * L);
  hipLaunchKernelGGL(gauss_latent_bwd_kernel, dim3((B * L / 4 + 255) / 256), dim3(256), 0, st, g_mu, g_lv, g_z, heads, eps, g_heads,
                     rng_bump, B, L, gz_slices);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

// out[b][c][p] (NCHW) = sum_s slices[s][(b*P + p)*C + c]: the split-K finish of a data gradient written NHWC and the
// NHWC -> NCHW layout change behind it (decoder.0's data gradient in front of decoder_input, vanilla_vae.py:102) as one launch.
// Tiny tensors ([B,512,2,2]): thread = one output element, reads along c are strided by design.
__global__ __launch_bounds__(256) void splitk_permute_kernel(const float* __restrict__ slices, int S, float* __restrict__ out, int B,
                                                            int C, int P) {
  const long n = (long)B * C * P, i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int p = (int)(i % P), c = (int)((i / P) % C), b = (int)(i / ((long)P * C));
  const long src = ((long)b * P + p) * C + c;
  float v = slices[src];
  for (int s = 1; s < S; ++s) v += slices[(long)s * n + src];
  out[i] = v;
}

int launch_splitk_permute(const float* slices, int S, float* out, int B, int C, int P, hipStream_t st) {
  if (!slices || !out || S < 1 || B <= 0 || C <= 0 || P <= 0) return kErrBadArg;
  const long n = (long)B * C * P;
  ProfScope ps("splitk_permute_kernel", st, 0.0, 4.0 * (S + 1.0) * n);
  hipLaunchKernelGGL(splitk_permute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slices, S, out, B, C, P);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
