// Tap-GEMM main kernel (vector path: gathered channels % 32 == 0, N % 4 == 0).  See tapgemm.hip for the op
// list and geom.hpp for the geometry.
//
// Design rule measured on gfx950 (tools/mfma_probe.hip): v_mfma_f32_32x32x2_f32 runs on the SIMD's vector
// datapath, so VALU instructions of ANY wave on that SIMD do not overlap with it -- SIMD time = MFMA time +
// VALU time.  Hence everything around the 64-cycle MFMAs is written to issue as few VALU instructions as
// possible:
//   * global loads go through buffer resources: 32-bit byte offsets, the wave-uniform part of an address rides
//     in the scalar offset, and an out-of-range offset returns 0 -> zero padding / tile edges cost one select;
//   * per-row offsets and the separable y/x validity masks of the im2col gather are computed once per tile;
//   * the epilogue uses 32-bit indices, drops stores by sending them out of range (no branches), and takes
//     per-row scatter indices from arithmetic (dense scatter) or one 16-byte LDS read per 4 rows (parity classes);
//   * fragments of the next 8-deep k-group are read from LDS before the MFMAs of the current one;
//   * per-tap constants sit in lanes and are fetched with v_readlane (no scalar loads, no division in the loop).
// Workgroup = 4 waves, tile BM x BN = (WM*TM*32) x (WN*TN*32), K chunks of 32; grids with <= 4 workgroups per CU run
// the double-buffered, explicitly software-pipelined loop (PF = 3), larger grids the single-buffer loop (PF = 0).
#include <type_traits>

#include "prof.hpp"
#include "pair.hpp"
#include "tapgemm.hpp"
#include "wgrad_fast.hpp"

namespace ctvae {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0x80000000u;  // byte offset beyond any buffer we bind (tensors are < 2 GiB)

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned voff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

#ifdef CTVAE_PHASE_TIMING
// diagnostic build only (tools/phase_probe.py): per-workgroup timestamps of the kernel phases, 100 MHz clock
__device__ unsigned long long g_phase[8 * 8192];
#define PHASE(i) do { if (threadIdx.x == 0 && blockIdx.z == 0) g_phase[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 8 + (i)] = wall_clock64(); } while (0)
// where the pipelined loop's time goes, summed over the chunks of a workgroup (wave 0, shader clock): [0] waiting for the global
// loads + LDS stores, [1] issuing the next loads / LDS reads + the first eight MFMAs, [2] the barrier, [3] reads + the last eight MFMAs
__device__ long long g_loop[4 * 8192];
#define LOOPT(var) const long long var = clock64()
#define LOOPACC(i, a, b) loopt_[i] += (b) - (a)
#define LOOPDECL long long loopt_[4] = {0, 0, 0, 0}
#define LOOPOUT do { if (threadIdx.x == 0 && blockIdx.z == 0) for (int q_ = 0; q_ < 4; ++q_) g_loop[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 4 + q_] = loopt_[q_]; } while (0)
extern "C" int ctvae_debug_loop_read(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_loop), (size_t)n * 8); }
extern "C" int ctvae_debug_phase_read(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), (size_t)n * 8);
}
extern "C" int ctvae_debug_wphase_read(unsigned long long* out, int n) {   // the weight-gradient role of conv_bwd_pair_kernel
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wphase), (size_t)n * 8);
}
#else
#define PHASE(i) do {} while (0)
#define LOOPT(var) do {} while (0)
#define LOOPACC(i, a, b) do {} while (0)
#define LOOPDECL do {} while (0)
#define LOOPOUT do {} while (0)
#endif

template <int WM, int WN, int TM, int TN, bool WT, int PF, bool XF = false, int PD = 1>
__global__ __launch_bounds__(256) void tapgemm_fast_kernel(const TapGemmArgs a) {
  __shared__ __attribute__((aligned(16))) float sAbuf[(PF == 3 ? 2 : 1) * (WM * TM * 32) * LDK];
  __shared__ __attribute__((aligned(16))) float sBbuf[(PF == 3 ? 2 : 1) * (WT ? (WN * TN * 32) * LDK : KC * (WN * TN * 32))];
  __shared__ __attribute__((aligned(16))) int sOut[WM * TM * 32];
  PHASE(0);
  int k_cls_rot, k_mtiles, k_ntiles;
  kernarg_warm_get<sizeof(TapGemmArgs), offsetof(TapGemmArgs, cls_rot), offsetof(TapGemmArgs, mtiles), offsetof(TapGemmArgs, ntiles)>(
      k_cls_rot, k_mtiles, k_ntiles);
  const int vbx = blockIdx.x, vby = blockIdx.y, vbz = blockIdx.z, vgx = k_mtiles * k_ntiles;   // = gridDim.x (launch_fast_cfg)
#include "tapgemm_fast_body.inc"
}

// One launch for the two independent GEMMs of a layer's backward pass: workgroups [0, nA) run the data-gradient tile kernel
// (64 x 64 tiles, pipelined loop), the others the weight-gradient kernel (64 x 64 tiles of dW per pixel slice).  Both
// launches on their own are one lock-step round of workgroups that pays the kernel boundary (previous kernel's L2
// write-back, launch gap, prologue, store burst: ~6.5 us) for ~20 us of MFMA work; sharing a launch
// pays it once and lets the weight-gradient workgroups start while the data-gradient's stores drain.
// XFB: the weight-gradient role applies the previous block's BatchNorm + activation to its X operand on load
// PDA: prefetch depth (chunks) of the data-gradient role's global loads (4 for launches of at most two workgroups per CU)
template <bool XFB, int PDA = 1>
__global__ __launch_bounds__(256) void conv_bwd_pair_kernel(const TapGemmArgs a, const WgradArgs w, int lgQw, int lgQhw, int lgC,
                                                            int nA, int gxA, int gyA, int gxB, int gyB, int role_xcd_on) {
  // the data-gradient kernel's arrays; the weight-gradient workgroups use the first 8 KB of each for their X / dY chunks
  __shared__ __attribute__((aligned(16))) float sAbuf[2 * 64 * LDK];
  __shared__ __attribute__((aligned(16))) float sBbuf[2 * 64 * LDK];
  __shared__ __attribute__((aligned(16))) int sOut[64];
  __shared__ unsigned sPix[2][MC];
  __shared__ unsigned sMsk[2][MC];
  __shared__ unsigned sOutB[2][MC];
  static_assert(2 * 64 * LDK >= MC * 64, "chunk buffers of the weight-gradient kernel fit");
  // both roles read their arguments behind ONE batch of scalar loads (kernarg_warm); the three the data-gradient role
  // branches on first come back with it
  constexpr int kOffW = (sizeof(TapGemmArgs) + 7) / 8 * 8, kOffI = kOffW + sizeof(WgradArgs);
  static_assert(alignof(TapGemmArgs) == 8 && alignof(WgradArgs) == 8 && sizeof(WgradArgs) % 8 == 0, "kernel-argument layout");
  int k_nA, k_gxA, k_gyA;
  kernarg_warm_get<kOffI + 9 * 4, kOffI + 12, kOffI + 16, kOffI + 20>(k_nA, k_gxA, k_gyA);
  const int L = blockIdx.x;
  if (L < k_nA) {
    constexpr int WM = 2, WN = 2, TM = 1, TN = 1, PF = 3;
    constexpr bool WT = true, XF = false;
    constexpr int PD = PDA;
    const int vbx = L % k_gxA, vr = L / k_gxA, vby = vr % k_gyA, vbz = vr / k_gyA, vgx = k_gxA;
    const int k_cls_rot = a.cls_rot, k_ntiles = a.ntiles;
#include "tapgemm_fast_body.inc"
  } else {
    const int Lb = L - k_nA;
    // Role-aware XCD placement (round 3).  Workgroups go to the 8 XCDs round-robin by their launch index L.  The data-gradient
    // role gives XCD X a contiguous eighth of its tiles, i.e. of the dY pixel range it gathers; the weight-gradient role used to
    // deal its pixel slices to the XCDs by (slice & 7), so every L2 fetched dY once for each role.  With S % 8 == 0 slices the
    // weight-gradient workgroups of XCD X now take the slices [X*S/8, (X+1)*S/8) -- the same eighth of the pixels -- and all
    // output tiles of a slice stay on one XCD as before.  Measured (VanillaVAE bs = 256, two --pmc passes each): fetch traffic of
    // the BatchNorm layers' paired launches 51.2 -> 39.7 MB per launch, 3.35 -> 3.28 GB per step -- and 1.599 -> 1.612 ms per
    // step (three alternating runs), the slice count rounded to a multiple of 8 costing more than the L2 hits return to an
    // MFMA-bound launch.  Hence OFF by default (CTVAE_PAIR_ROLE_XCD=1 turns it on, e.g. where the fabric is shared with RCCL).
    if (gyB % 8 == 0 && role_xcd_on) {
      const int X = L & 7, j = Lb >> 3, T = gxB;
      wgrad_fast_body<2, 2, 1, 1, XFB, true>(w, lgQw, lgQhw, lgC, sAbuf, sBbuf, sPix, sMsk, sOutB, j % T, X * (gyB >> 3) + j / T, gxB, gyB);
    } else {
      wgrad_fast_body<2, 2, 1, 1, XFB>(w, lgQw, lgQhw, lgC, sAbuf, sBbuf, sPix, sMsk, sOutB, Lb % gxB, Lb / gxB, gxB, gyB);
    }
  }
}

template <int WM, int WN, int TM, int TN>
static int launch_fast_cfg(const TapGemmArgs& a, int pf, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  TapGemmArgs args = a;
  args.mtiles = ceil_div(a.Mc, BM);
  args.ntiles = ceil_div(a.N, BN);
  const bool wt = a.g.wT != 0;
  dim3 grid(args.mtiles * args.ntiles, a.g.ncls, a.splitk > 1 ? a.splitk : 1), block(256);
  // Parity classes of a stride-2 3x3 layer reduce over 1, 2, 2 and 4 taps: workgroups of one class are 4x longer than those
  // of another.  Workgroups are dealt to the CUs in linear order (x, then y = class row, then z = K slice), so with the
  // plain order the CUs that got the 4-tap rows finish last while the others idle (resident grids), or the longest
  // workgroups start last (larger grids).  Resident grids (<= 1024 workgroups): class rows in the order
  // [fewest, middle, most, middle] taps, so that the rows 256 workgroups apart -- which share CUs when a class row is half
  // a round -- pair up as (1 + 4) and (2 + 2), and rotated by the K slice when the launch is split (a CU then sees three
  // different classes instead of three times the same).  VanillaVAE bs=256: 1.882 -> 1.849 ms per step (M=4x1024 sk=3 launches
  // 43 -> 33 us, M=4x4096 42 -> 38 us).  Larger grids keep the plain order: longest classes first 1.853 ms, classes
  // interleaved in groups of 8 workgroups 1.880 ms against 1.849 / 1.857 ms (same box).
  for (int c = 0; c < kMaxCls; ++c) args.cls_order[c] = c;
  args.cls_rot = 0;
  static const int no_cls_order = [] { const char* e = getenv("CTVAE_NO_CLS_ORDER"); return e ? atoi(e) : 0; }();   // diagnostic
  if (a.g.ncls == 4 && !no_cls_order) {
    int idx[4] = {0, 1, 2, 3};
    for (int i = 1; i < 4; ++i)
      for (int j = i; j > 0 && a.g.ntaps[idx[j]] < a.g.ntaps[idx[j - 1]]; --j) { const int t = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = t; }
    const long total = (long)grid.x * grid.y * grid.z;
    if (total <= 1024) {
      args.cls_order[0] = idx[0]; args.cls_order[1] = idx[1]; args.cls_order[2] = idx[3]; args.cls_order[3] = idx[2];
      args.cls_rot = grid.z > 1 ? 1 : 0;
    } else {
      static const int big = [] { const char* e = getenv("CTVAE_CLS_BIG"); return e ? atoi(e) : 0; }();   // diagnostic: 1 longest first, 2 interleaved -- both measured slower (below)
      if (big == 1) for (int c = 0; c < 4; ++c) args.cls_order[c] = idx[3 - c];
      if (big == 2 && grid.x % 8 == 0 && grid.z == 1) args.cls_rot = 2;
    }
  }
  for (int i = 0; i < kMaxCls; ++i) {   // launch-order copies of the tap table (TapGemmArgs::taps_l)
    const int c = args.cls_order[i];
    args.ntaps_l[i] = a.g.ntaps[c];
    for (int t = 0; t < kMaxTaps; ++t) args.taps_l[i][t] = a.g.taps[c][t];
  }
  // (deep prefetch -- four register sets of loads in flight, template parameter PD = 4 of the tile kernel -- was measured for
  // launches of at most two workgroups per CU and LOST: bs = 64 step 0.825 -> 0.857 ms; tools/negative/README.md.  Not instantiated.)
  constexpr bool deep = false;
  char name[160];
  snprintf(name, sizeof name, "tapgemm_fast_kernel<%d,%d,%d,%d,%s,%d%s>", WM, WN, TM, TN, wt ? "true" : "false", pf,
           (!wt && a.xf_scale != nullptr) ? ",true" : (deep ? ",false,4" : ""));
  double macs = 0;
  for (int c = 0; c < a.g.ncls; ++c) macs += (double)a.Mc * a.N * a.g.ntaps[c] * a.g.gC;
  if constexpr (WM == 2 && WN == 2 && TM == 1 && TN == 1) {
    PairCtx* pc = pair_ctx();
    if (pc != nullptr && wt && pf == 3 && !pc->haveA) {   // ctvae_conv_backward: issued by pair_flush()
      pc->haveA = true;
      pc->A = args;
      pc->gxA = grid.x; pc->gyA = grid.y; pc->gzA = grid.z;
      pc->flopsA = 2.0 * macs;
      pc->bytesA = 4.0 * ((double)a.g.B * a.g.gH * a.g.gW * a.g.gC + (double)a.g.B * a.g.sH * a.g.sW * a.g.sC);
      return 0;
    }
  }
  if (prof_detailed()) {
    size_t l = strlen(name);
    snprintf(name + l, sizeof name - l, " M=%dx%d N=%d C=%d taps=%d sk=%d", a.g.ncls, a.Mc, a.N, a.g.gC, a.g.ntaps[a.g.ncls - 1],
             a.splitk);
  }
  const double bytes = 4.0 * ((double)a.g.B * a.g.gH * a.g.gW * a.g.gC + (double)a.g.B * a.g.sH * a.g.sW * a.g.sC);
  ProfScope ps(name, st, 2.0 * macs, bytes);
  if (!wt && a.xf_scale != nullptr) {   // lazy BatchNorm apply on the gathered operand (forward launches only)
    if (pf == 3) hipLaunchKernelGGL((tapgemm_fast_kernel<WM, WN, TM, TN, false, 3, true>), grid, block, 0, st, args);
    else hipLaunchKernelGGL((tapgemm_fast_kernel<WM, WN, TM, TN, false, 0, true>), grid, block, 0, st, args);
  } else if (!wt) {
    // PF 1 / 2 (plain double buffer, two-chunk register prefetch) were measured and lost to 3: not instantiated
    if (pf == 3) hipLaunchKernelGGL((tapgemm_fast_kernel<WM, WN, TM, TN, false, 3>), grid, block, 0, st, args);
    else hipLaunchKernelGGL((tapgemm_fast_kernel<WM, WN, TM, TN, false, 0>), grid, block, 0, st, args);
  } else {
    if (pf == 3) hipLaunchKernelGGL((tapgemm_fast_kernel<WM, WN, TM, TN, true, 3>), grid, block, 0, st, args);
    else hipLaunchKernelGGL((tapgemm_fast_kernel<WM, WN, TM, TN, true, 0>), grid, block, 0, st, args);
  }
  CTVAE_LAUNCH_CHECK();
  return 0;
}

PairCtx*& pair_ctx() {
  static thread_local PairCtx* ctx = nullptr;
  return ctx;
}

int pair_flush(PairCtx& c, hipStream_t st) {
  if (c.haveA && c.haveB) {
    const unsigned nA = c.gxA * c.gyA * c.gzA, nB = c.gxB * c.gyB;
    ProfScope ps("conv_bwd_pair_kernel", st, c.flopsA + c.flopsB, c.bytesA + c.bytesB);
    static const int role_xcd = [] { const char* e = getenv("CTVAE_PAIR_ROLE_XCD"); return e ? atoi(e) : 0; }();   // 1 = role-aware placement (see the kernel)
#define CTVAE_PAIR(XFB_, PD_)                                                                                                  \
  hipLaunchKernelGGL((conv_bwd_pair_kernel<XFB_, PD_>), dim3(nA + nB), dim3(256), 0, st, c.A, c.B, c.lgQw, c.lgQhw, c.lgC, (int)nA, \
                     (int)c.gxA, (int)c.gyA, (int)c.gxB, (int)c.gyB, role_xcd)
    if (c.B.xf_scale != nullptr) CTVAE_PAIR(true, 1);
    else CTVAE_PAIR(false, 1);
#undef CTVAE_PAIR
    CTVAE_LAUNCH_CHECK();
  } else if (c.haveA) {
    ProfScope ps("tapgemm_fast_kernel<2,2,1,1,true,3>", st, c.flopsA, c.bytesA);
    hipLaunchKernelGGL((tapgemm_fast_kernel<2, 2, 1, 1, true, 3>), dim3(c.gxA, c.gyA, c.gzA), dim3(256), 0, st, c.A);
    CTVAE_LAUNCH_CHECK();
  } else if (c.haveB) {
    const int rc = launch_wgrad_fast_recorded(c, st);
    if (rc) return rc;
  }
  if (c.haveRed || (c.haveSK && c.haveBF)) {   // slab reduction, with the data gradient's split-K finish (and the BatchNorm-backward finalize of
    const int rc = launch_finish_recorded(c, st);   // the layer below) as extra blocks of the same launch
    if (rc) return rc;
  } else {
    if (c.haveSK) {
      const int rc = launch_splitk_recorded(c, st);
      if (rc) return rc;
    }
    if (c.haveBF) {
      const int rc = launch_bn_bwd_finalize_job(c.bf, st);
      if (rc) return rc;
    }
  }
  for (auto& f : c.later) {
    const int rc = f();
    if (rc) return rc;
  }
  return 0;
}

int launch_tapgemm_fast(const TapGemmArgs& a, const TapGemmPlan& plan, int pf, hipStream_t st) {
  if (plan.BN == 32) return launch_fast_cfg<4, 1, 1, 1>(a, pf, st);
  if (plan.BM == 128) return launch_fast_cfg<2, 2, 2, 1>(a, pf, st);
  return launch_fast_cfg<2, 2, 1, 1>(a, pf, st);
}

}  // namespace ctvae
