// extern "C" entry points of libctvae_hip.so (see include/ctvae_hip.h).  Thin argument checks + dispatch
// into the kernel launchers; nothing here allocates or synchronises.
#include "../../include/ctvae_hip.h"

#include "common.hpp"
#include "gatlayer.hpp"
#include "glinear.hpp"
#include "pair.hpp"

namespace ctvae {
int launch_tapgemm(const ConvGeom& g, const float* G, const float* W, const float* bias, const float* add,
                   const float* mask, int mask_act, float* S, int act, float* bn_part, float* ws, size_t ws_floats,
                   hipStream_t st, const BnBwdFuse* bnb = nullptr, const InXform* xf = nullptr,
                   const WinoFilters* wf = nullptr, SplitKRaw* raw = nullptr, int out_pix = 0);
bool out_pix_supported(const ConvGeom& g, size_t ws_floats, int out_pix);
bool wino_supported(const ConvGeom& g, size_t ws_floats);
bool wino_enabled();
int wino_set_enabled(int on);
int launch_crop_resize_u8(const unsigned char* img, const long long* rows, float* out, int B, int N, int H, int W, int crop,
                          int S, hipStream_t st);
bool thin_forward_supported(const ConvGeom& g);
bool thin_wgrad_supported(const ConvGeom& g);
int tapgemm_bnb_rows(const ConvGeom& g, size_t ws_floats);
bool img_dgrad_supported(const ConvGeom& g);
void tapgemm_plan(const ConvGeom& g, size_t ws_floats, TapGemmPlan& p);
int launch_wgrad(const ConvGeom& g, const float* X, const float* dY, float* dW, float* dbias, float* ws, size_t ws_bytes,
                 int accumulate, hipStream_t st, const InXform* xf = nullptr, const DyXform* dyx = nullptr);
int launch_bn_forward(const float* y, int R, int C, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float momentum, float eps, int training, int act, float* out, float* save_mean,
                      float* save_invstd, float* ws, size_t ws_bytes, long long* nbt, hipStream_t st,
                      float* coef_out = nullptr);
int launch_bn_finish_forward(const float* y, int R, int C, int nparts, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps, int training, int act,
                             float* out, float* save_mean, float* save_invstd, float* ws, long long* nbt, hipStream_t st,
                             float* coef_out = nullptr);
size_t bn_workspace_floats(int C, int nparts);
int launch_bn_backward(const float* ga, const float* beta, const float* y, int R, int C, const float* gamma,
                       const float* save_mean, const float* save_invstd, int act, float* gy, float* dgamma, float* dbeta,
                       int accumulate, float* ws, size_t ws_bytes, const float* part_in, int part_rows, hipStream_t st,
                       float* coef_out = nullptr, const float* coef_in = nullptr);
int launch_wino_filters_batch(int n, const float* const* W, float* const* Uf, float* const* Ub, const int* Ci, const int* Co,
                              hipStream_t st);
bool upconv_wgrad_supported(const ConvGeom& g);
bool img_enc_supported(const ConvGeom& g);
bool img_conv_supported(const ConvGeom& g);
int launch_img_backward_fused(const ConvGeom& g, const float* dY, const float* W, float* dX, const BnBwdFuse* bnb, float* ws,
                              float** part_out, float** pbias_out, int* nparts, bool want_bias, hipStream_t st);
size_t img_backward_fused_ws_floats(const ConvGeom& g);
int wgrad_finish_slabs(const float* part, float* dW, long n, int nparts, const float* pbias, float* dbias, long nb, int accumulate,
                       hipStream_t st);
int launch_permute(const float* in, float* out, int B, int C, int P, int to_nhwc, hipStream_t st);
int launch_gat_score(int mode, const float* xl, const float* xr, const float* attr, const float* we, const float* att, float* out,
                     int B, int N, int H, int C, float slope, hipStream_t st);
int launch_gat_score_backward(const float* xl, const float* xr, const float* attr, const float* we, const float* att,
                              const float* g, float* dxl, float* dxr, float* datt_part, float* dwe_part, int B, int N, int H,
                              int C, float slope, hipStream_t st);
int launch_gauss_latent_fwd(float* heads, const float* eps_in, const unsigned long long* rng, float* eps_out, float* z, int B,
                            int L, hipStream_t st, const float* slices, int S, const float* bias);
int launch_splitk_permute(const float* slices, int S, float* out, int B, int C, int P, hipStream_t st);
int launch_gauss_latent_bwd(const float* g_mu, const float* g_lv, const float* g_z, const float* heads, const float* eps,
                            float* g_heads, unsigned long long* rng_bump, int B, int L, hipStream_t st, int gz_slices);
int launch_ct_reg_forward(const float* adj, const float* graph, const float* uni, float* part, float ckl, float cgs, float cpt,
                          int B, int N, hipStream_t st);
int launch_ct_reg_backward(const float* adj, const float* graph, const float* uni, const float* part, const float* g_loss,
                           float ckl, float cgs, float cpt, float* d_adj, float* d_graph, int B, int N, hipStream_t st);
int launch_ct_blend_softmax_forward(const float* y, const float* mask, float* probs, long R, int Hs, int D, hipStream_t st);
int launch_ct_blend_softmax_backward(const float* g, const float* probs, const float* y, const float* mask, float* dy, float* dmask,
                                     long R, int Hs, int D, hipStream_t st);
int launch_ct_latent_ce_forward(const float* probs, const long long* target, float* row_loss, long R, int D, hipStream_t st);
int launch_ct_latent_ce_backward(const float* probs, const long long* target, const float* g_loss, float* d_probs, long R, int D,
                                 hipStream_t st);
int launch_ct_mask_forward(const float* x, const float* action, const float* pe, const float* keep, float scale, const float* W,
                           const float* bias, const float* expo, int B, int S, int D, int A, float* inter, float* p, float* sample,
                           float* soft, hipStream_t st);
int launch_ct_mask_backward(const float* x, const float* action, const float* pe, const float* keep, float scale,
                            const float* inter, const float* p, const float* soft, const float* g, int B, int S, int D, int A,
                            float* dWp, float* dbp, hipStream_t st);
int launch_ladder_forward(const float* mu_e, const float* lv_e, const float* mu_t, const float* lv_t, const float* eps, int B, int D,
                          float* z, float* kl, hipStream_t st);
int launch_ladder_backward(const float* gz, const float* gkl, const float* mu_e, const float* lv_e, const float* mu_t, const float* lv_t,
                           const float* eps, int B, int D, float* g_mu_e, float* g_lv_e, float* g_mu_t, float* g_lv_t, hipStream_t st);
int launch_gamma_reparam_forward(const float* alpha, const float* beta, const float* zhat, float shape_b, float* z, long n, hipStream_t st);
int launch_gamma_reparam_backward(const float* g, const float* alpha, const float* beta, const float* zhat, float shape_b, float* ga,
                                  float* gb, long n, hipStream_t st);
int launch_gamma_kl_forward(const float* alpha, const float* beta, int B, int D, float prior_alpha, float prior_beta, float* out,
                            float* ws, size_t ws_bytes, hipStream_t st);
int launch_gamma_kl_backward(const float* g, const float* alpha, const float* beta, int B, int D, float prior_alpha, float prior_beta,
                             float* ga, float* gb, hipStream_t st);
int launch_sigmoid(const float* x_or_g, const float* y, float* out, long n, int backward, hipStream_t st);
int launch_tc_forward(const float* z, const float* mu, const float* lv, const float* liw, int B, int D, float* out3, float* lse_s,
                      float* lse_d, float* ws, size_t ws_bytes, hipStream_t st);
int launch_tc_backward(const float* z, const float* mu, const float* lv, const float* liw, const float* lse_s, const float* lse_d,
                       const float* g3, int B, int D, float* dz, float* dmu, float* dlv, hipStream_t st);
int launch_vamp_forward(const float* z, const float* mu, const float* lv, const float* pmu, const float* plv, int B, int D, int K,
                        float* out3, float* wgt, float* ws, size_t ws_bytes, hipStream_t st);
int launch_vamp_backward(const float* z, const float* mu, const float* lv, const float* pmu, const float* plv, const float* wgt,
                         const float* g, int B, int D, int K, float* dz, float* dmu, float* dlv, float* dpmu, float* dplv,
                         hipStream_t st);
int launch_swd_forward(const float* z, const float* prior, const float* proj, int N, int D, int S, float p, float weight, float* out,
                       float* grad_z, float* ws, size_t ws_bytes, hipStream_t st);
int launch_ct_blend_forward(const float* s0, const float* s1, const float* mask, float* out, long rows, hipStream_t st);
int launch_ct_blend_backward(const float* g, const float* s0, const float* s1, const float* mask, float* g0, float* g1, float* gm,
                             long rows, hipStream_t st);
int launch_group_rowsum(const float* parts, long mat_stride, int nmat, int rows, int C, int ld, const int* grp, int G, float* out,
                        int accumulate, hipStream_t st);
int launch_ct_posenc_forward(const float* x, const float* pe, const float* keep, float scale, float* out, long n, int sd,
                             hipStream_t st);
int launch_ct_posenc_backward(const float* g, const float* keep, float scale, float* gx, long n, hipStream_t st);
int launch_one_hot(const long long* inds, long n, int N, float* out, hipStream_t st);
int launch_ct_sample_forward(const float* p, const float* expo, float* out, float* soft, float* weighted, long n, hipStream_t st);
int launch_ct_sample_backward(const float* gs, const float* gw, const float* p, const float* soft, const float* sample, float* gp,
                              long n, hipStream_t st);
int launch_pair_mlp_forward(const float* u, const float* v, int ld, const float* w2, const float* b2, float* out, int B, int N,
                            int H, float slope, int per_sample, const int* row_of, hipStream_t st);
int launch_pair_mlp_backward(const float* u, const float* v, int ld, const float* w2, const float* out, const float* g_out,
                             float* dU, float* dV, int ldd, float* dw2_part, float* db2_part, int B, int N, int H, float slope,
                             int per_sample, const int* row_of, hipStream_t st);
int launch_act_bwd(const float* gout, const float* out, float* gin, long n, int act, hipStream_t st);
int launch_act_fwd(const float* in, float* out, long n, int act, hipStream_t st);
int launch_reparam_fwd(const float* mu, long mu_rs, const float* lv, long lv_rs, const float* eps, float* z, int B, int L,
                       hipStream_t st);
int launch_reparam_bwd(const float* gz, const float* lv, long lv_rs, const float* eps, float* gmu, float* glv, int B, int L,
                       hipStream_t st);
int launch_adam(float* p, const float* g, float* m, float* v, float* state, long n, float grad_scale, hipStream_t st);
size_t adam_state_floats();
int launch_ssim_forward(const float* a, const float* b, const float* window, float* part, float* loss, float* coef, int B, int C, int H,
                        int W, const float* weights, hipStream_t st);
int launch_ssim_backward(const float* a, const float* b, const float* window, const float* coef, const float* g_loss, float* g_a, int B,
                         int C, int H, int W, hipStream_t st);
float* defer_wgrad_ws(float* ws, size_t ws_floats);
void defer_wgrad_done();
int defer_begin(float* arena, size_t arena_floats);
int defer_flush(hipStream_t st);
int launch_gumbel_fwd(const float* p, const float* noise, float* out, float* soft, long n, hipStream_t st);
int launch_gumbel_bwd(const float* go, const float* p, const float* soft, float* gp, long n, hipStream_t st);
int launch_gumbel_softmax_fwd(const float* z, const float* u, float* s, long rows, int Q, float temp, float eps, hipStream_t st);
int launch_gumbel_softmax_bwd(const float* gs, const float* s, float* gz, long rows, int Q, float temp, hipStream_t st);
int launch_cat_kl_fwd(const float* logits, long rows, int Q, int B, float eps, float c, float* out, float* ws, size_t ws_bytes,
                      hipStream_t st);
int launch_cat_kl_bwd(const float* logits, const float* go, float* gq, long rows, int Q, int B, float eps, float c, hipStream_t st);
int launch_iw_loss_forward(const float* recons, const float* x, long n, int R, int rep, const float* mu, const float* lv, int L,
                           int S, float M_N, float* lp, float* kld, float* coef, float* out4, hipStream_t st);
int launch_iw_loss_backward(const float* recons, const float* x, long n, int R, int rep, const float* mu, const float* lv, int L,
                            float M_N, const float* coef, const float* go, float* g_recons, float* g_mu, float* g_lv,
                            hipStream_t st);
int launch_mmd_forward(const float* z, const float* p, int N, int D, int kind, float c, float eps, float w_pp, float w_zz,
                       float w_pz, float* out4, float* grad, float* ws, size_t ws_bytes, hipStream_t st);
size_t dip_state_floats(int B, int D);
int launch_dip_forward(const float* mu, long mu_rs, const float* lv, long lv_rs, int B, int D, float l_diag, float l_off,
                       float* state, hipStream_t st);
int launch_dip_backward(const float* state, const float* go, float* g_mu, float* g_lv, int B, int D, hipStream_t st);
int launch_loss_forward(const float* r, const float* x, long n, const float* mu, long mu_rs, const float* lv, long lv_rs,
                        int B, int L, float M_N, const float* extra, float* out4, float* ws, size_t ws_bytes,
                        hipStream_t st, float logcosh_alpha, float* g_r = nullptr, float* g_mu = nullptr, float* g_lv = nullptr,
                        int ract = 0);
int launch_mse_backward(const float* r, const float* x, const float* go, float* gr, long n, hipStream_t st, float logcosh_alpha,
                        int ract = 0);
int launch_loss_backward(const float* r, const float* x, const float* go, float* gr, long n, float logcosh_alpha, const float* mu,
                         long mu_rs, const float* lv, long lv_rs, float* gmu, float* glv, int B, int L, float M_N, hipStream_t st,
                         int ract = 0);
int launch_kl_backward(const float* mu, long mu_rs, const float* lv, long lv_rs, const float* go, float* gmu, float* glv,
                       int B, int L, float M_N, hipStream_t st);
int launch_vq_inds(const float* lat, const float* cb, long long* inds, int B, int HW, int D, int K, int C, hipStream_t st);
int launch_vq_lookup(const float* lat, const float* cb, const long long* inds, float* out, float* vq_loss, float beta, int B,
                     int HW, int D, int K, int C, float* ws, size_t ws_bytes, hipStream_t st);
int launch_vq_backward(const float* gq, const float* gvq, const float* lat, const float* cb, const long long* inds,
                       float* glat, float* dcb, int accumulate, float beta, int B, int HW, int D, int K, int C, float* ws, size_t ws_bytes,
                       hipStream_t st);
}  // namespace ctvae

using namespace ctvae;

// `kind` of the conv entry points: CTVAE_CONV / CTVAE_CONVT, for CTVAE_CONV optionally | CTVAE_W_CI_TAP (weights stored
// [Ci][tap][Co]: the block of a Linear layer over torch.flatten(NCHW), run as a k x k convolution on the NHWC tensor)
static bool conv_kind_ok(int kind) {
  const int base = kind & ~CTVAE_W_CI_TAP;
  return base == CTVAE_CONV || (base == CTVAE_CONVT && !(kind & CTVAE_W_CI_TAP));
}
// gkind0: 0 for the forward / weight-gradient geometry, 2 for the data gradient's
static int conv_geom(ConvGeom& g, int kind, int gkind0, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad) {
  const int rc = build_geom(g, gkind0 + ((kind & ~CTVAE_W_CI_TAP) == CTVAE_CONV ? 0 : 1), B, H, W, Ci, Co, k, stride, pad, out_pad);
  if (rc) return rc;
  if (kind & CTVAE_W_CI_TAP) {
    g.wts = Co;
    g.wrs = k * k * Co;
  }
  return 0;
}

extern "C" {

const char* ctvae_version(void) { return "0.1.0"; }
const char* ctvae_arch(void) { return "gfx950"; }

const char* ctvae_error_string(int code) {
  if (code == 0) return "success";
  if (code == kErrBadArg) return "ctvae: unsupported shape or bad argument";
  if (code == kErrWorkspace) return "ctvae: workspace too small";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "ctvae: unknown error";
}

int ctvae_crop_resize_u8(const uint8_t* images, const int64_t* rows, float* out, int B, int N, int H, int W, int crop, int size,
                         void* stream) {
  if (!images || !rows || !out) return kErrBadArg;
  return launch_crop_resize_u8(images, (const long long*)rows, out, B, N, H, W, crop, size, (hipStream_t)stream);
}

int ctvae_winograd_enable(int on) { return wino_set_enabled(on); }

size_t ctvae_workspace_bytes(void) { return (size_t)256 << 20; }

int ctvae_conv_forward(int kind, const float* x, const float* w, const float* bias, const float* add, float* y, int B,
                       int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad, int act, const float* in_scale,
                       const float* in_shift, int in_act, float* wino_dgrad_filters_out, const float* wino_fwd_filters, float* ws,
                       size_t ws_bytes, void* stream) {
  if (!x || !w || !y || !conv_kind_ok(kind)) return kErrBadArg;
  if ((in_scale != nullptr) != (in_shift != nullptr)) return kErrBadArg;
  ConvGeom g;
  if (conv_geom(g, kind, 0, B, H, W, Ci, Co, k, stride, pad, out_pad)) return kErrBadArg;
  if ((wino_dgrad_filters_out != nullptr || wino_fwd_filters != nullptr) &&
      !ctvae_conv_wino_filter_floats(kind, B, H, W, Ci, Co, k, stride, pad, out_pad, ws_bytes))
    return kErrBadArg;
  const InXform xf{in_scale, in_shift, in_act};
  // wino_fwd_filters: both filter sets were made ahead of time (ctvae_wino_filters_batch): no transform launch here
  const WinoFilters wf{wino_fwd_filters, wino_fwd_filters != nullptr ? nullptr : wino_dgrad_filters_out};
  return launch_tapgemm(g, x, w, bias, add, nullptr, 0, y, act, nullptr, ws, ws_bytes / sizeof(float), (hipStream_t)stream,
                        nullptr, &xf, &wf);
}

int ctvae_linear_pixmajor_supported(int B, int Ci, int C, int P, size_t ws_bytes) {
  ConvGeom g;
  if (B <= 0 || Ci <= 0 || C <= 0 || P <= 1 || conv_geom(g, CTVAE_CONV, 0, B, 1, 1, Ci, C * P, 1, 1, 0, 0)) return 0;
  return out_pix_supported(g, ws_bytes / sizeof(float), P) ? 1 : 0;
}

int ctvae_linear_pixmajor_forward(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int C, int P, int act,
                                  float* ws, size_t ws_bytes, void* stream) {
  if (!x || !w || !y || !ctvae_linear_pixmajor_supported(B, Ci, C, P, ws_bytes)) return kErrBadArg;
  ConvGeom g;
  if (conv_geom(g, CTVAE_CONV, 0, B, 1, 1, Ci, C * P, 1, 1, 0, 0)) return kErrBadArg;
  return launch_tapgemm(g, x, w, bias, nullptr, nullptr, 0, y, act, nullptr, ws, ws_bytes / sizeof(float), (hipStream_t)stream,
                        nullptr, nullptr, nullptr, nullptr, P);
}

int ctvae_wino_filters_batch(int n, const float* const* w, float* const* fwd_filters, float* const* dgrad_filters, const int* Ci,
                             const int* Co, void* stream) {
  return launch_wino_filters_batch(n, w, fwd_filters, dgrad_filters, Ci, Co, (hipStream_t)stream);
}

size_t ctvae_conv_wino_filter_floats(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                                     size_t ws_bytes) {
  if (kind != CTVAE_CONV || !wino_enabled()) return 0;
  ConvGeom gf, gb;
  if (build_geom(gf, 0, B, H, W, Ci, Co, k, stride, pad, out_pad) || build_geom(gb, 2, B, H, W, Ci, Co, k, stride, pad, out_pad))
    return 0;
  const size_t wsf = ws_bytes / sizeof(float);
  return (wino_supported(gf, wsf) && wino_supported(gb, wsf)) ? (size_t)16 * Ci * Co : 0;
}

int ctvae_conv_input_transform_supported(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad,
                                         int out_pad) {
  if (!conv_kind_ok(kind)) return 0;
  ConvGeom g;
  if (conv_geom(g, kind, 0, B, H, W, Ci, Co, k, stride, pad, out_pad)) return 0;
  if (thin_forward_supported(g) && thin_wgrad_supported(g)) return 1;
  // the general tile kernels: forward on the vector path (not one of the dedicated picture-side / transposed-conv kernels, not
  // Winograd) and the weight gradient on the lean 64 x 64 kernel -- both form act(BN(.)) between the global load and the LDS store
  if (upconv_wgrad_supported(g)) return 1;     // 32 -> 32 transposed conv: both kernels stage the input patch once per tile
  if (img_enc_supported(g) || img_conv_supported(g)) return 0;
  if (wino_enabled() && wino_supported(g, (size_t)1 << 26)) return 0;
  TapGemmPlan pl;
  tapgemm_plan(g, (size_t)1 << 26, pl);
  if (pl.thin || pl.BM == 0) return 0;
  return ((g.gC % KC) == 0 && (g.sC % 4) == 0 && g.wT == 0) ? 1 : 0;
}

int ctvae_conv_bn_act_apply_is_separate(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                                        size_t ws_bytes) {
  if (!conv_kind_ok(kind)) return 0;
  ConvGeom g;
  if (conv_geom(g, kind, 0, B, H, W, Ci, Co, k, stride, pad, out_pad)) return 0;
  TapGemmPlan plan;
  tapgemm_plan(g, ws_bytes / sizeof(float), plan);
  return (plan.splitk > 1 && bn_fused_ok(g.B * g.sH * g.sW, Co)) ? 0 : 1;   // 0: the channel-owner launch writes a anyway
}

int ctvae_conv_bn_act_forward(int kind, const float* x, const float* w, const float* bias, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                              int training, int act, float* y, float* a_out, float* save_mean, float* save_invstd,
                              float* scale_shift_out, int64_t* num_batches_tracked, int B, int H, int W, int Ci, int Co, int k,
                              int stride, int pad, int out_pad, const float* in_scale, const float* in_shift, int in_act,
                              float* ws, size_t ws_bytes, void* stream) {
  long long* nbt = (long long*)num_batches_tracked;
  if (!x || !w || !gamma || !beta || !y || !ws || !conv_kind_ok(kind)) return kErrBadArg;
  if ((in_scale != nullptr) != (in_shift != nullptr)) return kErrBadArg;
  if (in_scale != nullptr && !ctvae_conv_input_transform_supported(kind, B, H, W, Ci, Co, k, stride, pad, out_pad)) return kErrBadArg;
  const InXform xf{in_scale, in_shift, in_act};
  const InXform* xfp = in_scale != nullptr ? &xf : nullptr;
  if (!a_out && !scale_shift_out) return kErrBadArg;   // either materialise a or hand out the coefficients
  if (training && (!save_mean || !save_invstd)) return kErrBadArg;
  if (!training && (!running_mean || !running_var)) return kErrBadArg;
  if (Co % 4 != 0) return kErrBadArg;
  ConvGeom g;
  if (conv_geom(g, kind, 0, B, H, W, Ci, Co, k, stride, pad, out_pad)) return kErrBadArg;
  const size_t wsf = ws_bytes / sizeof(float);
  TapGemmPlan plan;
  tapgemm_plan(g, wsf, plan);
  const int R = g.B * g.sH * g.sW;
  if (plan.splitk > 1 || !training) {
    // small layer run split-K (or eval mode).  Training, a few MB at most: the slices stay channel-major in the workspace and ONE
    // channel-owner launch sums them, takes the statistics, finalizes and applies (bn.hip bn_fused_fwd_kernel).  Otherwise:
    // conv (+ split-K finish) first, then statistics from y with the stand-alone kernels
    SplitKRaw raw{ws, 0};
    const bool fused = training && plan.splitk > 1 && bn_fused_ok(R, Co);
    int rc = launch_tapgemm(g, x, w, bias, nullptr, nullptr, 0, y, ACT_NONE, nullptr, ws, wsf, (hipStream_t)stream, nullptr, xfp,
                            nullptr, fused ? &raw : nullptr);
    if (rc) return rc;
    if (raw.splitk > 1) {
      const BnFusedFwd p{ws, row_map_of(g), raw.splitk, R, Co, bias, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd,
                         scale_shift_out, nbt, y, a_out, act};
      return launch_bn_fused_forward(p, (hipStream_t)stream);
    }
    return launch_bn_forward(y, R, Co, gamma, beta, running_mean, running_var, momentum, eps, training, act, a_out, save_mean,
                             save_invstd, ws, ws_bytes, nbt, (hipStream_t)stream, scale_shift_out);
  }
  const int nparts = plan.bn_parts;
  if (wsf < bn_workspace_floats(Co, nparts)) return kErrWorkspace;
  int rc = launch_tapgemm(g, x, w, bias, nullptr, nullptr, 0, y, ACT_NONE, ws, nullptr, 0, (hipStream_t)stream, nullptr, xfp);
  if (rc) return rc;
  return launch_bn_finish_forward(y, R, Co, nparts, gamma, beta, running_mean, running_var, momentum, eps, training, act,
                                  a_out, save_mean, save_invstd, ws, nbt, (hipStream_t)stream, scale_shift_out);
}

int ctvae_conv_dgrad(int kind, const float* dy, const float* w, const float* add, const float* mask, int mask_act,
                     float* dx, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                     const float* wino_filters, float* ws, size_t ws_bytes, void* stream) {
  if (!dy || !w || !dx || !conv_kind_ok(kind)) return kErrBadArg;
  ConvGeom g;
  if (conv_geom(g, kind, 2, B, H, W, Ci, Co, k, stride, pad, out_pad)) return kErrBadArg;
  if (wino_filters != nullptr && !ctvae_conv_wino_filter_floats(kind, B, H, W, Ci, Co, k, stride, pad, out_pad, ws_bytes))
    return kErrBadArg;
  const WinoFilters wf{wino_filters, nullptr};
  return launch_tapgemm(g, dy, w, nullptr, add, mask, mask_act, dx, ACT_NONE, nullptr, ws, ws_bytes / sizeof(float),
                        (hipStream_t)stream, nullptr, nullptr, &wf);
}

int ctvae_conv_dgrad_bn_rows(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                             size_t ws_bytes) {
  if (!conv_kind_ok(kind)) return 0;
  ConvGeom g;
  if (conv_geom(g, kind, 2, B, H, W, Ci, Co, k, stride, pad, out_pad)) return 0;
  return tapgemm_bnb_rows(g, ws_bytes / sizeof(float));
}

int ctvae_conv_dgrad_bn(int kind, const float* dy, const float* w, const float* add, const float* mask, int mask_act,
                        float* dx, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                        const float* bn_y, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                        const float* bn_beta, int bn_act, float* bn_part, int bn_part_rows, float* ws, size_t ws_bytes,
                        void* stream) {
  if (!dy || !w || !dx || !conv_kind_ok(kind)) return kErrBadArg;
  if (!bn_y || !bn_mean || !bn_invstd || !bn_gamma || !bn_beta || !bn_part) return kErrBadArg;
  ConvGeom g;
  if (conv_geom(g, kind, 2, B, H, W, Ci, Co, k, stride, pad, out_pad)) return kErrBadArg;
  const int rows = tapgemm_bnb_rows(g, ws_bytes / sizeof(float));
  if (rows <= 0 || rows != bn_part_rows) return kErrBadArg;
  if (img_dgrad_supported(g) && (add != nullptr || mask != nullptr)) return kErrBadArg;   // image kernel has no add/mask
  const BnBwdFuse f{bn_y, bn_mean, bn_invstd, bn_gamma, bn_beta, bn_act, bn_part};
  return launch_tapgemm(g, dy, w, nullptr, add, mask, mask_act, dx, ACT_NONE, nullptr, ws, ws_bytes / sizeof(float),
                        (hipStream_t)stream, &f);
}

int ctvae_conv_wgrad(int kind, const float* x, const float* dy, float* dw, float* dbias, int B, int H, int W, int Ci,
                     int Co, int k, int stride, int pad, int out_pad, int accumulate, const float* in_scale,
                     const float* in_shift, int in_act, const float* dy_bn_y, const float* dy_bn_coef, int dy_bn_act,
                     float* gy_out, float* bn_dgamma, float* bn_dbeta, int bn_accumulate, float* ws, size_t ws_bytes,
                     void* stream) {
  if (!x || !dy || !dw || !ws || !conv_kind_ok(kind)) return kErrBadArg;
  if ((in_scale != nullptr) != (in_shift != nullptr)) return kErrBadArg;
  if ((dy_bn_y != nullptr) != (dy_bn_coef != nullptr) || (gy_out != nullptr && dy_bn_y == nullptr)) return kErrBadArg;
  if ((bn_dgamma != nullptr) != (bn_dbeta != nullptr) || (bn_dgamma != nullptr && (dy_bn_y == nullptr || gy_out != nullptr)))
    return kErrBadArg;
  ConvGeom g;
  if (conv_geom(g, kind, 0, B, H, W, Ci, Co, k, stride, pad, out_pad)) return kErrBadArg;
  const InXform xf{in_scale, in_shift, in_act};
  const DyXform dyx{dy_bn_y, dy_bn_coef, gy_out, dy_bn_act, bn_dgamma, bn_dbeta, bn_accumulate};
  float* wws = defer_wgrad_ws(ws, ws_bytes / sizeof(float));   // between ctvae_defer_begin / _flush: slabs into the arena
  const int rc = launch_wgrad(g, x, dy, dw, dbias, wws, ws_bytes, accumulate, (hipStream_t)stream, &xf, &dyx);
  defer_wgrad_done();
  return rc;
}

int ctvae_defer_begin(float* arena, size_t arena_bytes) { return defer_begin(arena, arena_bytes / sizeof(float)); }
int ctvae_defer_flush(void* stream) { return defer_flush((hipStream_t)stream); }

int ctvae_conv_backward_bn_rows(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                                size_t ws_bytes) {
  if (!conv_kind_ok(kind)) return 0;
  ConvGeom g;
  if (conv_geom(g, kind, 2, B, H, W, Ci, Co, k, stride, pad, out_pad)) return 0;
  PairCtx ctx;                      // plan as ctvae_conv_backward does
  pair_ctx() = &ctx;
  const int rows = tapgemm_bnb_rows(g, ((ws_bytes / 2) & ~(size_t)255) / sizeof(float));
  pair_ctx() = nullptr;
  return rows;
}

static thread_local int g_lazy_pixel_major = 0;   // layout of the slices the next conv_backward_impl(dx_slices) leaves (set by its caller)

// dx_slices != nullptr: the data gradient must run split-K and leaves its raw slices there, channel-major (SplitKRaw); dx is
// not written (ctvae_conv_backward_lazy)
static int conv_backward_impl(int kind, const float* x, const float* dy, const float* w, float* dw, float* dbias, float* dx, int B,
                        int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad, int accumulate, const float* mask,
                        int mask_act, const float* wino_filters, const float* bn_y, const float* bn_mean,
                        const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int bn_act, float* bn_part,
                        int bn_part_rows, float* bn_coef_out, float* bn_dgamma, float* bn_dbeta, int bn_accumulate,
                        const float* in_scale, const float* in_shift, int in_act,
                        const float* dy_bn_y, const float* dy_bn_coef, int dy_bn_act, float* gy_out, float* ws, size_t ws_bytes,
                        void* stream, float* dx_slices) {
  if (!x || !dy || !w || !dw || (!dx && !dx_slices) || !ws || !conv_kind_ok(kind)) return kErrBadArg;
  const bool bn = bn_part != nullptr;
  if (bn_coef_out != nullptr && !bn) return kErrBadArg;
  if ((bn_dgamma != nullptr) != (bn_dbeta != nullptr) || (bn_dgamma != nullptr && bn_coef_out == nullptr)) return kErrBadArg;
  if ((in_scale != nullptr) != (in_shift != nullptr)) return kErrBadArg;
  if ((dy_bn_y != nullptr) != (dy_bn_coef != nullptr) || (dy_bn_y != nullptr) != (gy_out != nullptr)) return kErrBadArg;
  if (bn && (!bn_y || !bn_mean || !bn_invstd || !bn_gamma || !bn_beta)) return kErrBadArg;
  ConvGeom gw, gd;
  if (conv_geom(gw, kind, 0, B, H, W, Ci, Co, k, stride, pad, out_pad)) return kErrBadArg;
  if (conv_geom(gd, kind, 2, B, H, W, Ci, Co, k, stride, pad, out_pad)) return kErrBadArg;
  // the two GEMMs run concurrently: each gets its own half of the workspace
  const size_t half_bytes = (ws_bytes / 2) & ~(size_t)255, half_floats = half_bytes / sizeof(float);
  float* ws_d = ws + half_floats;
  if (wino_filters != nullptr && !ctvae_conv_wino_filter_floats(kind, B, H, W, Ci, Co, k, stride, pad, out_pad, half_bytes))
    return kErrBadArg;
  const hipStream_t st = (hipStream_t)stream;
  PairCtx ctx;
  pair_ctx() = &ctx;
  if (bn) {
    const int rows = tapgemm_bnb_rows(gd, half_floats);
    if (rows <= 0 || rows != bn_part_rows || (img_dgrad_supported(gd) && mask != nullptr)) {
      pair_ctx() = nullptr;
      return kErrBadArg;
    }
  }
  {   // how much of the chip the data gradient fills on its own: the weight gradient that shares its launch is sized to match
    TapGemmPlan pl;
    tapgemm_plan(gd, half_floats, pl);
    ctx.dgrad_wgs = (long)pl.mtiles * pl.ntiles * gd.ncls * (pl.splitk > 1 ? pl.splitk : 1);
    int kmax = 0;
    for (int c = 0; c < gd.ncls; ++c) kmax = gd.ntaps[c] * gd.gC / KC > kmax ? gd.ntaps[c] * gd.gC / KC : kmax;
    ctx.dgrad_chunks = pl.splitk > 1 ? (kmax + pl.splitk - 1) / pl.splitk : kmax;
  }
  const InXform xf{in_scale, in_shift, in_act};
  const DyXform dyx{dy_bn_y, dy_bn_coef, gy_out, dy_bn_act};
  // diagnostic: CTVAE_IMG_BWD_FUSED=0 keeps the picture-side conv's weight gradient and data gradient as two kernels
  static const int img_fused = [] { const char* e = getenv("CTVAE_IMG_BWD_FUSED"); return e ? atoi(e) : 1; }();
  int rc;
  if (img_fused && bn && bn_act != ACT_TANH && x == bn_y && in_scale != nullptr && in_act == bn_act && mask == nullptr && dy_bn_y == nullptr &&
      img_conv_supported(gw) && img_dgrad_supported(gd) && img_backward_fused_ws_floats(gd) <= half_floats) {
    // picture-side conv behind BatchNorm + activation (final_layer.1-3): the data gradient's pass over y also forms the
    // weight gradient (image.hip img_bwd_fused_kernel) -- x of the weight gradient is act(BN(y)), which that pass computes
    float *part = nullptr, *pb = nullptr;
    int np = 0;
    const BnBwdFuse f{bn_y, bn_mean, bn_invstd, bn_gamma, bn_beta, bn_act, bn_part};
    rc = launch_img_backward_fused(gd, dy, w, dx, &f, ws, &part, &pb, &np, dbias != nullptr, st);
    if (!rc) rc = wgrad_finish_slabs(part, dw, 9L * 32 * 3, np, pb, dbias, 3L, accumulate, st);
  } else {
    // (a finishing launch that carries the BatchNorm finalize of the layer below stays in the chain, without the reduction)
    static const int defer_bn = [] { const char* e = getenv("CTVAE_DEFER_WITH_BN_RIDER"); return e ? atoi(e) : 0; }();   // diagnostic: 1.615 vs 1.611 ms (VanillaVAE)
    float* wws = (bn_coef_out == nullptr || defer_bn) ? defer_wgrad_ws(ws, half_floats) : ws;
    rc = launch_wgrad(gw, x, dy, dw, dbias, wws, half_bytes, accumulate, st, &xf, &dyx);
    defer_wgrad_done();
    if (!rc) {
    const WinoFilters wf{wino_filters, nullptr};
    const BnBwdFuse f{bn_y, bn_mean, bn_invstd, bn_gamma, bn_beta, bn_act, bn_part};
    // dy_bn_*: dy was g_a of the BatchNorm behind this layer; the weight-gradient kernel left g_y in gy_out for the data gradient
      SplitKRaw raw{dx_slices, 0, g_lazy_pixel_major};
      g_lazy_pixel_major = 0;
      rc = launch_tapgemm(gd, gy_out != nullptr ? gy_out : dy, w, nullptr, nullptr, mask, mask_act, dx, ACT_NONE, nullptr, ws_d,
                          half_floats, st, bn ? &f : nullptr, nullptr, &wf, dx_slices != nullptr ? &raw : nullptr);
      if (!rc && dx_slices != nullptr && raw.splitk <= 1) rc = kErrBadArg;   // the caller asked ctvae_conv_backward_lazy_slices first
    }
  }
  pair_ctx() = nullptr;
  if (rc) return rc;
  if (bn_coef_out != nullptr) {   // the finalize of the BatchNorm below rides in the finishing launch
    ctx.haveBF = true;
    ctx.bf = BnFinJob{bn_part, bn_part_rows, Ci, (float)((long)B * H * W), bn_gamma, bn_mean, bn_invstd, bn_beta, bn_coef_out,
                      bn_dgamma, bn_dbeta, bn_accumulate};
  }
  return pair_flush(ctx, st);
}

int ctvae_conv_backward(int kind, const float* x, const float* dy, const float* w, float* dw, float* dbias, float* dx, int B,
                        int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad, int accumulate, const float* mask,
                        int mask_act, const float* wino_filters, const float* bn_y, const float* bn_mean,
                        const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int bn_act, float* bn_part,
                        int bn_part_rows, float* bn_coef_out, float* bn_dgamma, float* bn_dbeta, int bn_accumulate,
                        const float* in_scale, const float* in_shift, int in_act,
                        const float* dy_bn_y, const float* dy_bn_coef, int dy_bn_act, float* gy_out, float* ws, size_t ws_bytes,
                        void* stream) {
  return conv_backward_impl(kind, x, dy, w, dw, dbias, dx, B, H, W, Ci, Co, k, stride, pad, out_pad, accumulate, mask, mask_act,
                            wino_filters, bn_y, bn_mean, bn_invstd, bn_gamma, bn_beta, bn_act, bn_part, bn_part_rows, bn_coef_out,
                            bn_dgamma, bn_dbeta, bn_accumulate, in_scale, in_shift, in_act, dy_bn_y, dy_bn_coef, dy_bn_act, gy_out, ws,
                            ws_bytes, stream, nullptr);
}

int ctvae_conv_backward_lazy_slices(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                                    int for_bn, size_t ws_bytes) {
  if (!conv_kind_ok(kind)) return 0;
  ConvGeom g;
  if (conv_geom(g, kind, 2, B, H, W, Ci, Co, k, stride, pad, out_pad)) return 0;
  const size_t half_floats = ((ws_bytes / 2) & ~(size_t)255) / sizeof(float);
  if (img_dgrad_supported(g) || (wino_enabled() && wino_supported(g, half_floats))) return 0;
  PairCtx ctx;                      // plan as ctvae_conv_backward does
  pair_ctx() = &ctx;
  TapGemmPlan pl;
  tapgemm_plan(g, half_floats, pl);
  pair_ctx() = nullptr;
  const long Mc = (long)g.B * g.Qh * g.Qw;
  if (pl.thin || pl.splitk <= 1) return 0;
  if (for_bn && (Mc % 4 != 0 || !bn_fused_ok(B * H * W, Ci))) return 0;   // channel-major slices for the BatchNorm's channel owners
  return pl.splitk;
}

int ctvae_conv_backward_lazy(int kind, const float* x, const float* dy, const float* w, float* dw, float* dbias, float* dx_slices,
                             int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad, int accumulate,
                             const float* in_scale, const float* in_shift, int in_act, int pixel_major, float* ws, size_t ws_bytes,
                             void* stream) {
  if (!dx_slices) return kErrBadArg;
  g_lazy_pixel_major = pixel_major;
  return conv_backward_impl(kind, x, dy, w, dw, dbias, nullptr, B, H, W, Ci, Co, k, stride, pad, out_pad, accumulate, nullptr, 0,
                            nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, in_scale,
                            in_shift, in_act, nullptr, nullptr, 0, nullptr, ws, ws_bytes, stream, dx_slices);
}

int ctvae_bn_backward_fused(const float* g_a_slices, int slices, int kind, int B, int H, int W, int Ci, int Co, int k, int stride,
                            int pad, int out_pad, const float* y, const float* gamma, const float* beta, const float* save_mean,
                            const float* save_invstd, int act, float* g_y, float* dgamma, float* dbeta, int accumulate, void* stream) {
  if (!g_a_slices || slices < 1 || !y || !gamma || !beta || !save_mean || !save_invstd || !g_y || !dgamma || !dbeta) return kErrBadArg;
  if (!conv_kind_ok(kind)) return kErrBadArg;
  ConvGeom g;   // the data gradient that wrote the slices: its class-major row order -> pixels of y
  if (conv_geom(g, kind, 2, B, H, W, Ci, Co, k, stride, pad, out_pad)) return kErrBadArg;
  const BnFusedBwd p{g_a_slices, row_map_of(g), slices, B * H * W, Ci, y, gamma, beta, save_mean, save_invstd, act, g_y, dgamma,
                     dbeta, accumulate};
  return launch_bn_fused_backward(p, (hipStream_t)stream);
}

int ctvae_conv_wgrad_bn_apply_supported(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad) {
  if (!conv_kind_ok(kind)) return 0;
  ConvGeom g;
  if (conv_geom(g, kind, 0, B, H, W, Ci, Co, k, stride, pad, out_pad)) return 0;
  if (upconv_wgrad_supported(g)) return 1;
  return img_enc_supported(g) ? 2 : 0;
}

int ctvae_bn_forward(const float* y, int R, int C, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float momentum, float eps, int training, int act, float* out,
                     float* save_mean, float* save_invstd, int64_t* num_batches_tracked, float* ws, size_t ws_bytes,
                     void* stream) {
  if (!y || !gamma || !beta || !out || !ws) return kErrBadArg;
  if (training && (!save_mean || !save_invstd)) return kErrBadArg;
  if (!training && (!running_mean || !running_var)) return kErrBadArg;
  return launch_bn_forward(y, R, C, gamma, beta, running_mean, running_var, momentum, eps, training, act, out, save_mean,
                           save_invstd, ws, ws_bytes, (long long*)num_batches_tracked, (hipStream_t)stream);
}

int ctvae_bn_backward(const float* g_a, const float* beta, const float* y, int R, int C, const float* gamma,
                      const float* save_mean, const float* save_invstd, int act, float* g_y, float* dgamma,
                      float* dbeta, int accumulate, const float* part_in, int part_rows, float* coef_out,
                      const float* coef_in, float* ws, size_t ws_bytes, void* stream) {
  if (!g_a || !beta || !y || !gamma || !save_mean || !save_invstd || !dgamma || !dbeta || !ws) return kErrBadArg;
  if (!g_y && !coef_out) return kErrBadArg;   // either apply here or hand the coefficients to the consumer
  if ((part_in != nullptr) != (part_rows > 0)) return kErrBadArg;
  if (coef_in != nullptr && (!g_y || part_in != nullptr || coef_out != nullptr)) return kErrBadArg;
  return launch_bn_backward(g_a, beta, y, R, C, gamma, save_mean, save_invstd, act, g_y, dgamma, dbeta, accumulate, ws,
                            ws_bytes, part_in, part_rows, (hipStream_t)stream, coef_out, coef_in);
}

int ctvae_gat_score(int mode, const float* xl, const float* xr, const float* attr, const float* we, const float* att, float* out,
                    int B, int N, int H, int C, float slope, void* stream) {
  if (!xl || !xr || !attr || !we || !att || !out || (mode != 0 && mode != 1)) return kErrBadArg;
  return launch_gat_score(mode, xl, xr, attr, we, att, out, B, N, H, C, slope, (hipStream_t)stream);
}

int ctvae_gat_score_backward(const float* xl, const float* xr, const float* attr, const float* we, const float* att,
                             const float* g, float* d_xl, float* d_xr, float* d_att_part, float* d_we_part, int B, int N, int H,
                             int C, float slope, void* stream) {
  if (!xl || !xr || !attr || !we || !att || !g || !d_xl || !d_xr || !d_att_part || !d_we_part) return kErrBadArg;
  return launch_gat_score_backward(xl, xr, attr, we, att, g, d_xl, d_xr, d_att_part, d_we_part, B, N, H, C, slope,
                                   (hipStream_t)stream);
}

int ctvae_gat_layer_forward(const float* xl, const float* xr, int ld, const float* adj, const float* we, const float* att,
                            const float* bias, const int32_t* head_map, float* out, int ldo, float* alpha, int B, int Hs, int C,
                            float slope, int act, void* stream) {
  const GatLayerArgs a{xl, xr, ld, adj, we, att, bias, head_map, out, ldo, alpha, B, Hs, C, slope, act};
  return launch_gat_layer_forward(a, (hipStream_t)stream);
}

int ctvae_gat_layer_backward(const float* xl, const float* xr, int ld, const float* adj, const float* we, const float* att,
                             const float* bias, const int32_t* head_map, const float* out, int ldo, const float* alpha,
                             const float* g_out, float* dS, float* dattr, float* d_xl, float* d_xr, int ldd, float* d_bias_part,
                             float* d_att_part, float* d_we_part, float* d_adj, int accumulate_dadj, int B, int Hs, int C,
                             float slope, int act, void* stream) {
  GatBwdArgs p{};
  p.f = GatLayerArgs{xl, xr, ld, adj, we, att, bias, head_map, const_cast<float*>(out), ldo, const_cast<float*>(alpha), B, Hs, C,
                     slope, act};
  p.g_out = g_out; p.dS = dS; p.dattr = dattr; p.dxl = d_xl; p.dxr = d_xr; p.ldd = ldd;
  p.dbias_part = d_bias_part; p.datt_part = d_att_part; p.dwe_part = d_we_part;
  return launch_gat_layer_backward(p, d_adj, accumulate_dadj, (hipStream_t)stream);
}

int ctvae_gauss_latent_forward(float* heads, const float* eps_in, const uint64_t* rng, float* eps_out, float* z, int B, int L,
                               const float* head_slices, int slices, const float* head_bias, void* stream) {
  return launch_gauss_latent_fwd(heads, eps_in, (const unsigned long long*)rng, eps_out, z, B, L, (hipStream_t)stream, head_slices,
                                 slices, head_bias);
}

int ctvae_gauss_latent_backward(const float* g_mu, const float* g_logvar, const float* g_z, const float* heads, const float* eps,
                                float* g_heads, uint64_t* rng_bump, int B, int L, int g_z_slices, void* stream) {
  return launch_gauss_latent_bwd(g_mu, g_logvar, g_z, heads, eps, g_heads, (unsigned long long*)rng_bump, B, L, (hipStream_t)stream,
                                 g_z_slices);
}

int ctvae_splitk_permute(const float* slices, int n_slices, float* out, int B, int C, int P, void* stream) {
  return launch_splitk_permute(slices, n_slices, out, B, C, P, (hipStream_t)stream);
}

int ctvae_conv_forward_lazy_slices(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                                   size_t ws_bytes) {
  if (!conv_kind_ok(kind)) return 0;
  ConvGeom g;
  if (conv_geom(g, kind, 0, B, H, W, Ci, Co, k, stride, pad, out_pad)) return 0;
  const size_t wsf = ws_bytes / sizeof(float);
  if (img_enc_supported(g) || img_conv_supported(g) || upconv_wgrad_supported(g) || (wino_enabled() && wino_supported(g, wsf))) return 0;
  TapGemmPlan pl;
  tapgemm_plan(g, wsf, pl);
  return (pl.thin || pl.splitk <= 1) ? 0 : pl.splitk;
}

int ctvae_conv_forward_lazy(int kind, const float* x, const float* w, float* y_slices, int B, int H, int W, int Ci, int Co, int k,
                            int stride, int pad, int out_pad, float* ws, size_t ws_bytes, void* stream) {
  if (!x || !w || !y_slices || !ws || !conv_kind_ok(kind)) return kErrBadArg;
  ConvGeom g;
  if (conv_geom(g, kind, 0, B, H, W, Ci, Co, k, stride, pad, out_pad)) return kErrBadArg;
  SplitKRaw raw{y_slices, 0, 1};
  const int rc = launch_tapgemm(g, x, w, nullptr, nullptr, nullptr, 0, nullptr, ACT_NONE, nullptr, ws, ws_bytes / sizeof(float),
                                (hipStream_t)stream, nullptr, nullptr, nullptr, &raw);
  if (rc) return rc;
  return raw.splitk > 1 ? 0 : kErrBadArg;   // the caller asked ctvae_conv_forward_lazy_slices first
}

int ctvae_ct_reg_forward(const float* adj, const float* graph, const float* uniform, float* part4, float ckl, float cgs, float cpt,
                         int B, int N, void* stream) {
  return launch_ct_reg_forward(adj, graph, uniform, part4, ckl, cgs, cpt, B, N, (hipStream_t)stream);
}

int ctvae_ct_reg_backward(const float* adj, const float* graph, const float* uniform, const float* part4, const float* g_loss,
                          float ckl, float cgs, float cpt, float* d_adj, float* d_graph, int B, int N, void* stream) {
  return launch_ct_reg_backward(adj, graph, uniform, part4, g_loss, ckl, cgs, cpt, d_adj, d_graph, B, N, (hipStream_t)stream);
}

int ctvae_ct_blend_softmax_forward(const float* y, const float* mask, float* probs, long R, int Hs, int D, void* stream) {
  return launch_ct_blend_softmax_forward(y, mask, probs, R, Hs, D, (hipStream_t)stream);
}

int ctvae_ct_blend_softmax_backward(const float* g, const float* probs, const float* y, const float* mask, float* dy, float* dmask,
                                    long R, int Hs, int D, void* stream) {
  return launch_ct_blend_softmax_backward(g, probs, y, mask, dy, dmask, R, Hs, D, (hipStream_t)stream);
}

int ctvae_ct_latent_ce_forward(const float* probs, const int64_t* target, float* row_loss, long R, int D, void* stream) {
  return launch_ct_latent_ce_forward(probs, (const long long*)target, row_loss, R, D, (hipStream_t)stream);
}

int ctvae_ct_latent_ce_backward(const float* probs, const int64_t* target, const float* g_loss, float* d_probs, long R, int D,
                                void* stream) {
  return launch_ct_latent_ce_backward(probs, (const long long*)target, g_loss, d_probs, R, D, (hipStream_t)stream);
}

int ctvae_ct_mask_forward(const float* x, const float* action, const float* pe, const float* keep, float scale, const float* W,
                          const float* bias, const float* expo, int B, int S, int D, int A, float* inter, float* p, float* sample,
                          float* soft, void* stream) {
  return launch_ct_mask_forward(x, action, pe, keep, scale, W, bias, expo, B, S, D, A, inter, p, sample, soft, (hipStream_t)stream);
}

int ctvae_ct_mask_backward(const float* x, const float* action, const float* pe, const float* keep, float scale, const float* inter,
                           const float* p, const float* soft, const float* g, int B, int S, int D, int A, float* dWp, float* dbp,
                           void* stream) {
  return launch_ct_mask_backward(x, action, pe, keep, scale, inter, p, soft, g, B, S, D, A, dWp, dbp, (hipStream_t)stream);
}

int ctvae_ct_sample_forward(const float* p, const float* expo, float* sample, float* soft, float* weighted, long n, void* stream) {
  return launch_ct_sample_forward(p, expo, sample, soft, weighted, n, (hipStream_t)stream);
}

int ctvae_ct_sample_backward(const float* g_sample, const float* g_weighted, const float* p, const float* soft, const float* sample,
                             float* g_p, long n, void* stream) {
  return launch_ct_sample_backward(g_sample, g_weighted, p, soft, sample, g_p, n, (hipStream_t)stream);
}

int ctvae_ct_blend_forward(const float* s0, const float* s1, const float* mask, float* out, long rows, void* stream) {
  return launch_ct_blend_forward(s0, s1, mask, out, rows, (hipStream_t)stream);
}

int ctvae_ct_blend_backward(const float* g, const float* s0, const float* s1, const float* mask, float* g0, float* g1, float* g_mask,
                            long rows, void* stream) {
  return launch_ct_blend_backward(g, s0, s1, mask, g0, g1, g_mask, rows, (hipStream_t)stream);
}

int ctvae_ct_posenc_forward(const float* x, const float* pe, const float* keep, float scale, float* out, long n, int sd,
                            void* stream) {
  return launch_ct_posenc_forward(x, pe, keep, scale, out, n, sd, (hipStream_t)stream);
}

int ctvae_ct_posenc_backward(const float* g, const float* keep, float scale, float* g_x, long n, void* stream) {
  return launch_ct_posenc_backward(g, keep, scale, g_x, n, (hipStream_t)stream);
}

int ctvae_one_hot(const int64_t* inds, long n, int N, float* out, void* stream) {
  return launch_one_hot((const long long*)inds, n, N, out, (hipStream_t)stream);
}

int ctvae_group_rowsum(const float* parts, long mat_stride, int nmat, int rows, int C, int ld, const int32_t* group, int G, float* out,
                       int accumulate, void* stream) {
  return launch_group_rowsum(parts, mat_stride, nmat, rows, C, ld, group, G, out, accumulate, (hipStream_t)stream);
}

static int glin_fill(GLinArgs& a, int nseg, int N, const float* const* W, const int* ldw, const int64_t* w_gstride,
                     const float* const* bias, const int* b_gstride, const int32_t* const* group) {
  if (nseg < 1 || nseg > kGLinMaxSeg || !W || !ldw || !w_gstride) return kErrBadArg;
  a.nseg = nseg;
  a.N = N;
  for (int s = 0; s < nseg; ++s) {
    a.W[s] = W[s];
    a.ldw[s] = ldw[s];
    a.wgs[s] = (long)w_gstride[s];
    a.bias[s] = bias ? bias[s] : nullptr;
    a.bgs[s] = b_gstride ? b_gstride[s] : 0;
    a.group[s] = group ? group[s] : nullptr;
  }
  return 0;
}

int ctvae_glinear_forward(const float* x, int ldx, int K, int nseg, int N, const float* const* W, const int* ldw,
                          const int64_t* w_gstride, const float* const* bias, const int* b_gstride,
                          const int32_t* const* group, float* y, int ldy, int B, void* stream) {
  GLinArgs a{};
  if (int rc = glin_fill(a, nseg, N, W, ldw, w_gstride, bias, b_gstride, group)) return rc;
  a.x = x; a.ldx = ldx; a.K = K; a.y = y; a.ldy = ldy; a.B = B;
  return launch_glinear_forward(a, (hipStream_t)stream);
}

int ctvae_glinear_dgrad(const float* dy, int ldy, int nseg, int N, const float* const* W, const int* ldw,
                        const int64_t* w_gstride, const int32_t* const* group, float* dx, int ldx, int K, int B, void* stream) {
  GLinArgs a{};
  if (int rc = glin_fill(a, nseg, N, W, ldw, w_gstride, nullptr, nullptr, group)) return rc;
  a.x = nullptr; a.ldx = ldx; a.K = K; a.y = const_cast<float*>(dy); a.ldy = ldy; a.B = B;
  return launch_glinear_dgrad(a, dx, (hipStream_t)stream);
}

size_t ctvae_glinear_wgrad_ws_bytes(int G, int N, int K) { return glinear_wgrad_ws_floats(G, N, K, 1) * sizeof(float); }

int ctvae_glinear_wgrad(const float* x, int ldx, int K, const float* dy, int ldy, int col0, int N, const int32_t* group, int G,
                        int B, float* dW, int ldo, float* dbias, int accumulate, float* ws, size_t ws_bytes, void* stream) {
  return launch_glinear_wgrad(x, ldx, K, dy, ldy, col0, N, group, G, B, dW, ldo, dbias, accumulate, ws, ws_bytes / sizeof(float),
                              (hipStream_t)stream);
}

int ctvae_pair_mlp_forward(const float* u, const float* v, int ld, const float* w2, const float* b2, float* out, int B, int N,
                           int H, float slope, int per_sample, const int32_t* row_of, void* stream) {
  if (!u || !v || !w2 || !out) return kErrBadArg;
  return launch_pair_mlp_forward(u, v, ld, w2, b2, out, B, N, H, slope, per_sample, row_of, (hipStream_t)stream);
}

int ctvae_pair_mlp_backward(const float* u, const float* v, int ld, const float* w2, const float* out, const float* g_out,
                            float* d_u, float* d_v, int ldd, float* d_w2_part, float* d_b2_part, int B, int N, int H, float slope,
                            int per_sample, const int32_t* row_of, void* stream) {
  if (!u || !v || !w2 || !out || !g_out || !d_u || !d_v || !d_w2_part || !d_b2_part) return kErrBadArg;
  return launch_pair_mlp_backward(u, v, ld, w2, out, g_out, d_u, d_v, ldd, d_w2_part, d_b2_part, B, N, H, slope, per_sample,
                                  row_of, (hipStream_t)stream);
}

int ctvae_permute(const float* in, float* out, int B, int C, int P, int to_nhwc, void* stream) {
  if (!in || !out || B <= 0 || C <= 0 || P <= 0) return kErrBadArg;
  return launch_permute(in, out, B, C, P, to_nhwc, (hipStream_t)stream);
}

int ctvae_act_forward(const float* in, float* out, long n, int act, void* stream) {
  if (!in || !out || n <= 0) return kErrBadArg;
  return launch_act_fwd(in, out, n, act, (hipStream_t)stream);
}

int ctvae_act_backward(const float* g_out, const float* out, float* g_in, long n, int act, void* stream) {
  if (!g_out || !out || !g_in || n <= 0) return kErrBadArg;
  return launch_act_bwd(g_out, out, g_in, n, act, (hipStream_t)stream);
}

int ctvae_reparam_forward(const float* mu, long mu_rs, const float* logvar, long lv_rs, const float* eps, float* z, int B,
                          int L, void* stream) {
  if (!mu || !logvar || !eps || !z || B <= 0 || L <= 0) return kErrBadArg;
  return launch_reparam_fwd(mu, mu_rs, logvar, lv_rs, eps, z, B, L, (hipStream_t)stream);
}

int ctvae_reparam_backward(const float* g_z, const float* logvar, long lv_rs, const float* eps, float* g_mu,
                           float* g_logvar, int B, int L, void* stream) {
  if (!g_z || !logvar || !eps || !g_mu || !g_logvar || B <= 0 || L <= 0) return kErrBadArg;
  return launch_reparam_bwd(g_z, logvar, lv_rs, eps, g_mu, g_logvar, B, L, (hipStream_t)stream);
}

int ctvae_loss_forward(const float* recons, const float* x, long n, const float* mu, long mu_rs, const float* logvar,
                       long lv_rs, int B, int L, float M_N, const float* extra, float* out4, float* ws, size_t ws_bytes,
                       void* stream) {
  if (!recons || !x || !out4 || !ws || n <= 0) return kErrBadArg;
  if ((mu == nullptr) != (logvar == nullptr)) return kErrBadArg;
  return launch_loss_forward(recons, x, n, mu, mu_rs, logvar, lv_rs, B, L, M_N, extra, out4, ws, ws_bytes,
                             (hipStream_t)stream, 0.f);
}

int ctvae_loss_forward_grad(const float* recons, const float* x, long n, const float* mu, long mu_rs, const float* logvar,
                            long lv_rs, int B, int L, float M_N, const float* extra, float* out4, float* g_recons, float* g_mu,
                            float* g_logvar, int recons_act, float* ws, size_t ws_bytes, void* stream) {
  if (!recons || !x || !out4 || !ws || n <= 0 || !mu || !logvar || !g_recons || !g_mu || !g_logvar) return kErrBadArg;
  return launch_loss_forward(recons, x, n, mu, mu_rs, logvar, lv_rs, B, L, M_N, extra, out4, ws, ws_bytes,
                             (hipStream_t)stream, 0.f, g_recons, g_mu, g_logvar, recons_act);
}

int ctvae_mse_backward(const float* recons, const float* x, const float* g_loss, float* g_recons, long n, int recons_act,
                       void* stream) {
  if (!recons || !x || !g_loss || !g_recons || n <= 0) return kErrBadArg;
  return launch_mse_backward(recons, x, g_loss, g_recons, n, (hipStream_t)stream, 0.f, recons_act);
}

int ctvae_loss_backward(const float* recons, const float* x, const float* g_loss, float* g_recons, long n, float logcosh_alpha,
                        const float* mu, long mu_rs, const float* logvar, long lv_rs, float* g_mu, float* g_logvar, int B, int L,
                        float M_N, int recons_act, void* stream) {
  if (!recons || !x || !g_loss || !g_recons || n <= 0 || !mu || !logvar || !g_mu || !g_logvar || B <= 0 || L <= 0) return kErrBadArg;
  return launch_loss_backward(recons, x, g_loss, g_recons, n, logcosh_alpha, mu, mu_rs, logvar, lv_rs, g_mu, g_logvar, B, L, M_N,
                              (hipStream_t)stream, recons_act);
}

int ctvae_l2l1_loss_forward(const float* recons, const float* x, long n, const float* extra, float* out4, float* ws, size_t ws_bytes,
                            void* stream) {
  if (!recons || !x || !out4 || !ws || n <= 0) return kErrBadArg;
  return launch_loss_forward(recons, x, n, nullptr, 0, nullptr, 0, 0, 0, 0.f, extra, out4, ws, ws_bytes, (hipStream_t)stream, -1.f);
}

int ctvae_l2l1_backward(const float* recons, const float* x, const float* g_loss, float* g_recons, long n, int recons_act,
                        void* stream) {
  if (!recons || !x || !g_loss || !g_recons || n <= 0) return kErrBadArg;
  return launch_mse_backward(recons, x, g_loss, g_recons, n, (hipStream_t)stream, -1.f, recons_act);
}

int ctvae_ladder_merge_forward(const float* mu_e, const float* logvar_e, const float* mu_t, const float* logvar_t, const float* eps,
                               int B, int D, float* z, float* kl, void* stream) {
  return launch_ladder_forward(mu_e, logvar_e, mu_t, logvar_t, eps, B, D, z, kl, (hipStream_t)stream);
}

int ctvae_ladder_merge_backward(const float* g_z, const float* g_kl, const float* mu_e, const float* logvar_e, const float* mu_t,
                                const float* logvar_t, const float* eps, int B, int D, float* g_mu_e, float* g_logvar_e,
                                float* g_mu_t, float* g_logvar_t, void* stream) {
  return launch_ladder_backward(g_z, g_kl, mu_e, logvar_e, mu_t, logvar_t, eps, B, D, g_mu_e, g_logvar_e, g_mu_t, g_logvar_t,
                                (hipStream_t)stream);
}

int ctvae_gamma_reparam_forward(const float* alpha, const float* beta, const float* zhat, float gamma_shape, float* z, long n,
                                void* stream) {
  return launch_gamma_reparam_forward(alpha, beta, zhat, gamma_shape, z, n, (hipStream_t)stream);
}

int ctvae_gamma_reparam_backward(const float* g_z, const float* alpha, const float* beta, const float* zhat, float gamma_shape,
                                 float* g_alpha, float* g_beta, long n, void* stream) {
  return launch_gamma_reparam_backward(g_z, alpha, beta, zhat, gamma_shape, g_alpha, g_beta, n, (hipStream_t)stream);
}

int ctvae_gamma_kl_forward(const float* alpha, const float* beta, int B, int D, float prior_alpha, float prior_beta, float* out,
                           float* ws, size_t ws_bytes, void* stream) {
  return launch_gamma_kl_forward(alpha, beta, B, D, prior_alpha, prior_beta, out, ws, ws_bytes, (hipStream_t)stream);
}

int ctvae_gamma_kl_backward(const float* g_kld, const float* alpha, const float* beta, int B, int D, float prior_alpha,
                            float prior_beta, float* g_alpha, float* g_beta, void* stream) {
  return launch_gamma_kl_backward(g_kld, alpha, beta, B, D, prior_alpha, prior_beta, g_alpha, g_beta, (hipStream_t)stream);
}

int ctvae_sigmoid_forward(const float* x, float* y, long n, void* stream) { return launch_sigmoid(x, nullptr, y, n, 0, (hipStream_t)stream); }

int ctvae_sigmoid_backward(const float* g, const float* y, float* g_x, long n, void* stream) {
  return launch_sigmoid(g, y, g_x, n, 1, (hipStream_t)stream);
}

int ctvae_tc_forward(const float* z, const float* mu, const float* logvar, const float* log_iw, int B, int D, float* out3, float* lse_s,
                     float* lse_d, float* ws, size_t ws_bytes, void* stream) {
  return launch_tc_forward(z, mu, logvar, log_iw, B, D, out3, lse_s, lse_d, ws, ws_bytes, (hipStream_t)stream);
}

int ctvae_tc_backward(const float* z, const float* mu, const float* logvar, const float* log_iw, const float* lse_s, const float* lse_d,
                      const float* g3, int B, int D, float* g_z, float* g_mu, float* g_logvar, void* stream) {
  return launch_tc_backward(z, mu, logvar, log_iw, lse_s, lse_d, g3, B, D, g_z, g_mu, g_logvar, (hipStream_t)stream);
}

int ctvae_vamp_kl_forward(const float* z, const float* mu, const float* logvar, const float* prior_mu, const float* prior_logvar,
                          int B, int D, int K, float* out3, float* weights, float* ws, size_t ws_bytes, void* stream) {
  return launch_vamp_forward(z, mu, logvar, prior_mu, prior_logvar, B, D, K, out3, weights, ws, ws_bytes, (hipStream_t)stream);
}

int ctvae_vamp_kl_backward(const float* z, const float* mu, const float* logvar, const float* prior_mu, const float* prior_logvar,
                           const float* weights, const float* g_kld, int B, int D, int K, float* g_z, float* g_mu, float* g_logvar,
                           float* g_prior_mu, float* g_prior_logvar, void* stream) {
  return launch_vamp_backward(z, mu, logvar, prior_mu, prior_logvar, weights, g_kld, B, D, K, g_z, g_mu, g_logvar, g_prior_mu,
                              g_prior_logvar, (hipStream_t)stream);
}

int ctvae_swd_forward(const float* z, const float* prior, const float* proj, int N, int D, int S, float p, float weight, float* out,
                      float* grad_z, float* ws, size_t ws_bytes, void* stream) {
  return launch_swd_forward(z, prior, proj, N, D, S, p, weight, out, grad_z, ws, ws_bytes, (hipStream_t)stream);
}

int ctvae_logcosh_loss_forward(const float* recons, const float* x, long n, float alpha, const float* mu, long mu_rs,
                               const float* logvar, long lv_rs, int B, int L, float M_N, float* out4, float* ws,
                               size_t ws_bytes, void* stream) {
  if (!recons || !x || !out4 || !ws || n <= 0 || !(alpha > 0.f)) return kErrBadArg;
  if ((mu == nullptr) != (logvar == nullptr)) return kErrBadArg;
  return launch_loss_forward(recons, x, n, mu, mu_rs, logvar, lv_rs, B, L, M_N, nullptr, out4, ws, ws_bytes,
                             (hipStream_t)stream, alpha);
}

int ctvae_logcosh_backward(const float* recons, const float* x, const float* g_loss, float* g_recons, long n, float alpha,
                           int recons_act, void* stream) {
  if (!recons || !x || !g_loss || !g_recons || n <= 0 || !(alpha > 0.f)) return kErrBadArg;
  return launch_mse_backward(recons, x, g_loss, g_recons, n, (hipStream_t)stream, alpha, recons_act);
}

int ctvae_kl_backward(const float* mu, long mu_rs, const float* logvar, long lv_rs, const float* g_loss, float* g_mu,
                      float* g_logvar, int B, int L, float M_N, void* stream) {
  if (!mu || !logvar || !g_loss || !g_mu || !g_logvar) return kErrBadArg;
  return launch_kl_backward(mu, mu_rs, logvar, lv_rs, g_loss, g_mu, g_logvar, B, L, M_N, (hipStream_t)stream);
}

int ctvae_vq_inds(const float* latents, const float* codebooks, int64_t* inds, int B, int HW, int D, int K, int C,
                  void* stream) {
  if (!latents || !codebooks || !inds || B <= 0 || HW <= 0 || C <= 0) return kErrBadArg;
  return launch_vq_inds(latents, codebooks, (long long*)inds, B, HW, D, K, C, (hipStream_t)stream);
}

int ctvae_vq_lookup(const float* latents, const float* codebooks, const int64_t* inds, float* quantized, float* vq_loss,
                    float beta, int B, int HW, int D, int K, int C, float* ws, size_t ws_bytes, void* stream) {
  if (!latents || !codebooks || !inds || !quantized || !vq_loss || !ws || C <= 0) return kErrBadArg;
  return launch_vq_lookup(latents, codebooks, (const long long*)inds, quantized, vq_loss, beta, B, HW, D, K, C, ws, ws_bytes,
                          (hipStream_t)stream);
}

int ctvae_vq_backward(const float* g_quantized, const float* g_vq_loss, const float* latents, const float* codebooks,
                      const int64_t* inds, float* g_latents, float* d_codebooks, int accumulate, float beta, int B, int HW,
                      int D, int K, int C, float* ws, size_t ws_bytes, void* stream) {
  if (!latents || !codebooks || !inds || C <= 0) return kErrBadArg;
  return launch_vq_backward(g_quantized, g_vq_loss, latents, codebooks, (const long long*)inds, g_latents, d_codebooks,
                            accumulate, beta, B, HW, D, K, C, ws, ws_bytes, (hipStream_t)stream);
}

int ctvae_gumbel_st_forward(const float* p, const float* gumbel_noise, float* sample, float* soft, long n, void* stream) {
  if (!p || !gumbel_noise || !sample || !soft || n <= 0) return kErrBadArg;
  return launch_gumbel_fwd(p, gumbel_noise, sample, soft, n, (hipStream_t)stream);
}

int ctvae_gumbel_st_backward(const float* g_sample, const float* p, const float* soft, float* g_p, long n, void* stream) {
  if (!g_sample || !p || !soft || !g_p || n <= 0) return kErrBadArg;
  return launch_gumbel_bwd(g_sample, p, soft, g_p, n, (hipStream_t)stream);
}

int ctvae_gumbel_softmax_forward(const float* logits, const float* uniform, float* sample, long rows, int Q, float temperature,
                                 float eps, void* stream) {
  if (!logits || !uniform || !sample || rows <= 0 || Q <= 0 || !(temperature > 0.f)) return kErrBadArg;
  return launch_gumbel_softmax_fwd(logits, uniform, sample, rows, Q, temperature, eps, (hipStream_t)stream);
}

int ctvae_gumbel_softmax_backward(const float* g_sample, const float* sample, float* g_logits, long rows, int Q,
                                  float temperature, void* stream) {
  if (!g_sample || !sample || !g_logits || rows <= 0 || Q <= 0 || !(temperature > 0.f)) return kErrBadArg;
  return launch_gumbel_softmax_bwd(g_sample, sample, g_logits, rows, Q, temperature, (hipStream_t)stream);
}

int ctvae_cat_kl_forward(const float* logits, long rows, int Q, int B, float eps, float log_prior, float* kld, float* ws,
                         size_t ws_bytes, void* stream) {
  if (!logits || !kld || !ws || rows <= 0 || Q <= 0 || B <= 0) return kErrBadArg;
  return launch_cat_kl_fwd(logits, rows, Q, B, eps, log_prior, kld, ws, ws_bytes, (hipStream_t)stream);
}

int ctvae_cat_kl_backward(const float* logits, const float* g_kld, float* g_logits, long rows, int Q, int B, float eps,
                          float log_prior, void* stream) {
  if (!logits || !g_kld || !g_logits || rows <= 0 || Q <= 0 || B <= 0) return kErrBadArg;
  return launch_cat_kl_bwd(logits, g_kld, g_logits, rows, Q, B, eps, log_prior, (hipStream_t)stream);
}

int ctvae_iw_loss_forward(const float* recons, const float* x, long n, int R, int rep, const float* mu, const float* logvar,
                          int L, int S, float M_N, float* lp, float* kld, float* coef, float* out4, void* stream) {
  if (!recons || !x || !mu || !logvar || !lp || !kld || !coef || !out4) return kErrBadArg;
  if (n <= 0 || (n & 3) || R <= 0 || rep <= 0 || R % rep || S <= 0 || R % S || L <= 0) return kErrBadArg;
  return launch_iw_loss_forward(recons, x, n, R, rep, mu, logvar, L, S, M_N, lp, kld, coef, out4, (hipStream_t)stream);
}

int ctvae_iw_loss_backward(const float* recons, const float* x, long n, int R, int rep, const float* mu, const float* logvar,
                           int L, float M_N, const float* coef, const float* g_loss, float* g_recons, float* g_mu,
                           float* g_logvar, void* stream) {
  if (!recons || !x || !mu || !logvar || !coef || !g_loss || (g_mu == nullptr) != (g_logvar == nullptr)) return kErrBadArg;
  if (n <= 0 || (n & 3) || R <= 0 || rep <= 0 || R % rep || L <= 0) return kErrBadArg;
  return launch_iw_loss_backward(recons, x, n, R, rep, mu, logvar, L, M_N, coef, g_loss, g_recons, g_mu, g_logvar,
                                 (hipStream_t)stream);
}

int ctvae_mmd_forward(const float* z, const float* prior, int N, int D, int kind, float c, float eps, float w_pp, float w_zz,
                      float w_pz, float* out4, float* grad_z, float* ws, size_t ws_bytes, void* stream) {
  if (!z || !prior || !out4 || !grad_z || !ws || N <= 0 || D <= 0 || (kind != 0 && kind != 1) || !(c > 0.f)) return kErrBadArg;
  return launch_mmd_forward(z, prior, N, D, kind, c, eps, w_pp, w_zz, w_pz, out4, grad_z, ws, ws_bytes, (hipStream_t)stream);
}

size_t ctvae_dip_state_floats(int B, int D) { return dip_state_floats(B, D); }

int ctvae_dip_forward(const float* mu, long mu_rs, const float* logvar, long lv_rs, int B, int D, float lambda_diag,
                      float lambda_offdiag, float* state, void* stream) {
  if (!mu || !logvar || !state || B <= 0 || D <= 0) return kErrBadArg;
  return launch_dip_forward(mu, mu_rs, logvar, lv_rs, B, D, lambda_diag, lambda_offdiag, state, (hipStream_t)stream);
}

int ctvae_dip_backward(const float* state, const float* g_dip, float* g_mu, float* g_logvar, int B, int D, void* stream) {
  if (!state || !g_dip || !g_mu || !g_logvar || B <= 0 || D <= 0) return kErrBadArg;
  return launch_dip_backward(state, g_dip, g_mu, g_logvar, B, D, (hipStream_t)stream);
}

size_t ctvae_adam_state_floats(void) { return adam_state_floats(); }

size_t ctvae_mssim_part_floats(int B, int C) { return (size_t)5 * B * C * 2; }
int ctvae_mssim_forward(const float* a, const float* b, const float* window, const float* weights, float* part, float* loss,
                        float* coef, int B, int C, int H, int W, void* stream) {
  return launch_ssim_forward(a, b, window, part, loss, coef, B, C, H, W, weights, (hipStream_t)stream);
}
int ctvae_mssim_backward(const float* a, const float* b, const float* window, const float* coef, const float* g_loss, float* g_a,
                         int B, int C, int H, int W, void* stream) {
  return launch_ssim_backward(a, b, window, coef, g_loss, g_a, B, C, H, W, (hipStream_t)stream);
}

int ctvae_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* state, long n,
                    float grad_scale, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !state || n <= 0) return kErrBadArg;
  return launch_adam(params, grads, exp_avg, exp_avg_sq, state, n, grad_scale, (hipStream_t)stream);
}

}  // extern "C"
