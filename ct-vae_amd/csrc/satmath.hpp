// relu / step functions of the pair kernels (pairmlp.hip, gatlayer.hip) as the VOP3P `clamp` output modifier of a packed-f32
// instruction: the result of v_pk_add_f32 / v_pk_fma_f32 ... clamp is min(max(x, 0), 1) per half (NaN -> 0), at no cost beyond
// the add / fma itself -- gfx950 has no packed max, a relu or a compare + select costs one plain VALU instruction per ELEMENT.
// tools/clamp_probe.hip checks on the device that the hardware honours the modifier (the compiler never emits it for f32).
//
//   relu(t)  = 2^64  * sat(t * 2^-64)     for t < 2^64      operands are staged pre-scaled by kSatDown; power-of-two scaling is
//   [t > 0]  =         sat(t * 2^60)      for t >= 2^-60    exact in binary floating point, so sums of pre-scaled operands are
//                                                            the scaled sums bit for bit (far from overflow / denormals here:
//                                                            |t| of O(1e-6 .. 1e3))
// The helpers take NATURAL register pairs (two floats that already sit in an aligned VGPR pair); `_lo` / `_hi` broadcast one
// half of the first operand through op_sel, so that no v_mov is needed to build {x, x}.
#pragma once
#include "common.hpp"

namespace ctvae {

constexpr float kSatDown = 5.421010862427522e-20f;   // 2^-64
constexpr float kSatUp = 18446744073709551616.f;     // 2^64
constexpr float kStepUp = 1152921504606846976.f;     // 2^60
constexpr float kStepDown = 8.673617379884035e-19f;  // 2^-60

// {sat(a.x + b.x), sat(a.y + b.y)}
__device__ __forceinline__ f32x2 pk_add_sat(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// {sat(a.x + b.x), sat(a.x + b.y)}
__device__ __forceinline__ f32x2 pk_add_sat_lo(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// {sat(a.y + b.x), sat(a.y + b.y)}
__device__ __forceinline__ f32x2 pk_add_sat_hi(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// {sat(a.x * b.x + c.x), sat(a.y * b.y + c.y)}
__device__ __forceinline__ f32x2 pk_fma_sat(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Two rows of a step-function accumulation in one scheduling unit (the compiler otherwise splits the shared product into
// mul + add, builds {g, g} with moves and defers half of the fmas past the loop body -- 64+ live temporaries):
//   ind_r = sat(ua|ub + v)        acc_r += g_r * ind_r        col += g_r * ind_r        g = {g_a, g_b} broadcast through op_sel
__device__ __forceinline__ void step_fma2(f32x2& acc_a, f32x2& acc_b, f32x2& col_a, f32x2& col_b, f32x2 ua, f32x2 ub, f32x2 v, f32x2 g) {
  f32x2 ia, ib;
  asm("v_pk_add_f32 %[ia], %[ua], %[v] clamp\n\t"
      "v_pk_add_f32 %[ib], %[ub], %[v] clamp\n\t"
      "v_pk_fma_f32 %[aa], %[g], %[ia], %[aa] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[ca], %[g], %[ia], %[ca] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[ab], %[g], %[ib], %[ab] op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %[cb], %[g], %[ib], %[cb] op_sel:[1,0,0]"
      : [aa] "+v"(acc_a), [ab] "+v"(acc_b), [ca] "+v"(col_a), [cb] "+v"(col_b), [ia] "=&v"(ia), [ib] "=&v"(ib)
      : [ua] "v"(ua), [ub] "v"(ub), [v] "v"(v), [g] "v"(g));
}

// One hidden unit of the forward pair scorer on a 4 x 4 block of pairs: acc[r][c] += w * sat(u_r + v_c), rows u = {u01, u23},
// columns v = {v01, v23}, w = one half (HI) of the pair wp.  16 instructions, all operands natural register pairs.
template <bool HI>
__device__ __forceinline__ void relu_fma_4x4(f32x2 (&acc)[4][2], f32x2 u01, f32x2 u23, f32x2 v01, f32x2 v23, f32x2 wp) {
  f32x2 t0, t1, t2, t3;
#define CTVAE_ROWS2(UP, A0, A1, A2_, A3)                                                                           \
  if (HI)                                                                                                          \
    asm("v_pk_add_f32 %[t0], %[u], %[va] op_sel_hi:[0,1] clamp\n\t"                                                \
        "v_pk_add_f32 %[t1], %[u], %[vb] op_sel_hi:[0,1] clamp\n\t"                                                \
        "v_pk_add_f32 %[t2], %[u], %[va] op_sel:[1,0] clamp\n\t"                                                   \
        "v_pk_add_f32 %[t3], %[u], %[vb] op_sel:[1,0] clamp\n\t"                                                   \
        "v_pk_fma_f32 %[a0], %[w], %[t0], %[a0] op_sel:[1,0,0]\n\t"                                                \
        "v_pk_fma_f32 %[a1], %[w], %[t1], %[a1] op_sel:[1,0,0]\n\t"                                                \
        "v_pk_fma_f32 %[a2], %[w], %[t2], %[a2] op_sel:[1,0,0]\n\t"                                                \
        "v_pk_fma_f32 %[a3], %[w], %[t3], %[a3] op_sel:[1,0,0]"                                                     \
        : [a0] "+v"(A0), [a1] "+v"(A1), [a2] "+v"(A2_), [a3] "+v"(A3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), \
          [t3] "=&v"(t3)                                                                                           \
        : [u] "v"(UP), [va] "v"(v01), [vb] "v"(v23), [w] "v"(wp));                                                 \
  else                                                                                                             \
    asm("v_pk_add_f32 %[t0], %[u], %[va] op_sel_hi:[0,1] clamp\n\t"                                                \
        "v_pk_add_f32 %[t1], %[u], %[vb] op_sel_hi:[0,1] clamp\n\t"                                                \
        "v_pk_add_f32 %[t2], %[u], %[va] op_sel:[1,0] clamp\n\t"                                                   \
        "v_pk_add_f32 %[t3], %[u], %[vb] op_sel:[1,0] clamp\n\t"                                                   \
        "v_pk_fma_f32 %[a0], %[w], %[t0], %[a0] op_sel_hi:[0,1,1]\n\t"                                             \
        "v_pk_fma_f32 %[a1], %[w], %[t1], %[a1] op_sel_hi:[0,1,1]\n\t"                                             \
        "v_pk_fma_f32 %[a2], %[w], %[t2], %[a2] op_sel_hi:[0,1,1]\n\t"                                             \
        "v_pk_fma_f32 %[a3], %[w], %[t3], %[a3] op_sel_hi:[0,1,1]"                                                  \
        : [a0] "+v"(A0), [a1] "+v"(A1), [a2] "+v"(A2_), [a3] "+v"(A3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), \
          [t3] "=&v"(t3)                                                                                           \
        : [u] "v"(UP), [va] "v"(v01), [vb] "v"(v23), [w] "v"(wp))
  CTVAE_ROWS2(u01, acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
  CTVAE_ROWS2(u23, acc[2][0], acc[2][1], acc[3][0], acc[3][1]);
#undef CTVAE_ROWS2
}

// Two sources (r, r + 1) x four channels (two packed pairs A, B) of the GATv2 projection gradients (gatlayer.hip gat_proj_bwd_kernel):
//   ind_r = sat(a_r * w + xl_r + xr)      ql_r += g_r * ind_r      qr += g_r * ind_r      qa += (g_r a_r) * ind_r
// a = {a_r, a_r+1} and g likewise (natural pairs out of 16-byte LDS reads, broadcast per half through op_sel); w, xl_r, xr = two
// channels each.  21 instructions for 8 (pair, channel)s; the temporaries never leave the block.
__device__ __forceinline__ void proj_step2x2(f32x2& qlA0, f32x2& qlA1, f32x2& qlB0, f32x2& qlB1, f32x2& qrA, f32x2& qrB, f32x2& qaA,
                                             f32x2& qaB, f32x2 a, f32x2 g, f32x2 wA, f32x2 wB, f32x2 xlA0, f32x2 xlA1, f32x2 xlB0,
                                             f32x2 xlB1, f32x2 xrA, f32x2 xrB) {
  f32x2 t0, t1, t2, t3, ga;
  asm("v_pk_mul_f32 %[ga], %[g], %[a]\n\t"
      "v_pk_fma_f32 %[t0], %[a], %[wA], %[xA0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[t1], %[a], %[wA], %[xA1] op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %[t2], %[a], %[wB], %[xB0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[t3], %[a], %[wB], %[xB1] op_sel:[1,0,0]\n\t"
      "v_pk_add_f32 %[t0], %[t0], %[xrA] clamp\n\t"
      "v_pk_add_f32 %[t1], %[t1], %[xrA] clamp\n\t"
      "v_pk_add_f32 %[t2], %[t2], %[xrB] clamp\n\t"
      "v_pk_add_f32 %[t3], %[t3], %[xrB] clamp\n\t"
      "v_pk_fma_f32 %[qA0], %[g], %[t0], %[qA0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[qB0], %[g], %[t2], %[qB0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[qrA], %[g], %[t0], %[qrA] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[qrB], %[g], %[t2], %[qrB] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[qaA], %[ga], %[t0], %[qaA] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[qaB], %[ga], %[t2], %[qaB] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[qA1], %[g], %[t1], %[qA1] op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %[qB1], %[g], %[t3], %[qB1] op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %[qrA], %[g], %[t1], %[qrA] op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %[qrB], %[g], %[t3], %[qrB] op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %[qaA], %[ga], %[t1], %[qaA] op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %[qaB], %[ga], %[t3], %[qaB] op_sel:[1,0,0]"
      : [qA0] "+v"(qlA0), [qA1] "+v"(qlA1), [qB0] "+v"(qlB0), [qB1] "+v"(qlB1), [qrA] "+v"(qrA), [qrB] "+v"(qrB), [qaA] "+v"(qaA),
        [qaB] "+v"(qaB), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [ga] "=&v"(ga)
      : [a] "v"(a), [g] "v"(g), [wA] "v"(wA), [wB] "v"(wB), [xA0] "v"(xlA0), [xA1] "v"(xlA1), [xB0] "v"(xlB0), [xB1] "v"(xlB1),
        [xrA] "v"(xrA), [xrB] "v"(xrB));
}

// One channel of the GATv2 score on two source rows x four target columns (gatlayer.hip):
//   acc[r][c] += at * sat(l_r + r_c + a[r][c] * w)           l = {l_r, l_r+1}, wa = {w, at}, columns in two packed pairs
// 12 instructions for 8 (pair, channel)s.
__device__ __forceinline__ void score_2x4(f32x2& acc00, f32x2& acc01, f32x2& acc10, f32x2& acc11, f32x2 l, f32x2 r01, f32x2 r23, f32x2 wa,
                                          f32x2 a00, f32x2 a01, f32x2 a10, f32x2 a11) {
  f32x2 t0, t1, t2, t3;
  asm("v_pk_add_f32 %[t0], %[l], %[r01] op_sel_hi:[0,1]\n\t"
      "v_pk_add_f32 %[t1], %[l], %[r23] op_sel_hi:[0,1]\n\t"
      "v_pk_add_f32 %[t2], %[l], %[r01] op_sel:[1,0]\n\t"
      "v_pk_add_f32 %[t3], %[l], %[r23] op_sel:[1,0]\n\t"
      "v_pk_fma_f32 %[t0], %[a00], %[wa], %[t0] op_sel_hi:[1,0,1] clamp\n\t"
      "v_pk_fma_f32 %[t1], %[a01], %[wa], %[t1] op_sel_hi:[1,0,1] clamp\n\t"
      "v_pk_fma_f32 %[t2], %[a10], %[wa], %[t2] op_sel_hi:[1,0,1] clamp\n\t"
      "v_pk_fma_f32 %[t3], %[a11], %[wa], %[t3] op_sel_hi:[1,0,1] clamp\n\t"
      "v_pk_fma_f32 %[c00], %[wa], %[t0], %[c00] op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %[c01], %[wa], %[t1], %[c01] op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %[c10], %[wa], %[t2], %[c10] op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %[c11], %[wa], %[t3], %[c11] op_sel:[1,0,0]"
      : [c00] "+v"(acc00), [c01] "+v"(acc01), [c10] "+v"(acc10), [c11] "+v"(acc11), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
        [t3] "=&v"(t3)
      : [l] "v"(l), [r01] "v"(r01), [r23] "v"(r23), [wa] "v"(wa), [a00] "v"(a00), [a01] "v"(a01), [a10] "v"(a10), [a11] "v"(a11));
}

// Same block for the backward's step function: acc[r][c] += c_k * [l_r + r_c + a[r][c] * w > 0], operands pre-scaled by 2^60,
// wa = {w * 2^60, att_k we_k (1 - slope)}.  (The instruction sequence is the forward's: sat() of the scaled sum IS the step.)
#define step_2x4 score_2x4

// sum over the 4 / 8 lanes of an aligned lane group, result in every lane of the group (DPP adds, no LDS)
__device__ __forceinline__ float quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ float oct_sum(float v) {
  v = quad_sum(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  return v;
}

// oct_sum of four values at once: the adds carry the DPP modifier themselves (the compiler emits v_mov_b32_dpp + add and pads every
// step with s_nop for the VALU-write -> DPP-read hazard; interleaving four independent chains covers the two wait states).
__device__ __forceinline__ void oct_sum4(float& a, float& b, float& c, float& d) {
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

}  // namespace ctvae
