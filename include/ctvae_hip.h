/* ctvae_hip.h — C ABI of libctvae_hip.so (MI355X / gfx950).
 *
 * The reference (Strong-AI-Lab/ct-vae) is 100 % Python and has no FFI of its own: every kernel is
 * reached implicitly through torch.nn / torch.autograd.  This header therefore
 * declares, per implicit op on the hot path, the entry point a binding for that op calls instead;
 * each declaration cites the reference call site it replaces.  All tensors are fp32, activations are
 * NHWC ("channels_last") in HBM, weights are packed [kh*kw][Cin][Cout] (Conv2d AND ConvTranspose2d,
 * with Cin/Cout the layer's input/output channels; Linear = 1x1 conv on a 1x1 image).  No function
 * allocates, synchronises or touches the host: callers own every buffer, incl. the scratch workspace
 * (ctvae_workspace_bytes()), so every call is hipGraph-capturable.  Return value: 0, a hipError_t
 * (>0) or a negative argument/workspace error; ctvae_error_string() maps either to text.
 * `stream` is a hipStream_t passed as void*.
 */
#ifndef CTVAE_HIP_H
#define CTVAE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* activation codes */
#define CTVAE_ACT_NONE 0
#define CTVAE_ACT_LRELU 1 /* nn.LeakyReLU() slope 0.01, vanilla_vae.py:31 */
#define CTVAE_ACT_RELU 2  /* nn.ReLU(True), vq_vae.py:65 */
#define CTVAE_ACT_TANH 3  /* nn.Tanh(), vanilla_vae.py:75, mcq_vae.py:237 */

/* layer kinds */
#define CTVAE_CONV 0  /* nn.Conv2d / nn.Linear */
#define CTVAE_CONVT 1 /* nn.ConvTranspose2d */
/* CTVAE_CONV | CTVAE_W_CI_TAP: the weight block is [Ci][kh*kw][Co] instead of [kh*kw][Ci][Co].  That is the [in][out] block of
 * an nn.Linear whose input is torch.flatten(h, start_dim=1) of an NCHW tensor h [B,Ci,k,k] (vanilla_vae.py:36-37,87-91: in =
 * Ci*k*k, feature index ci*k*k + ky*k + kx): with this flag the layer runs as a k x k convolution straight on the NHWC tensor
 * (forward, data gradient, weight gradient), without the NHWC <-> NCHW copies around the flatten. */
#define CTVAE_W_CI_TAP 0x100

const char* ctvae_version(void);
const char* ctvae_arch(void); /* "gfx950" */
const char* ctvae_error_string(int code);
size_t ctvae_workspace_bytes(void); /* scratch size that is sufficient for every call below */

/* y = act(conv(x, w) + bias + add)
 * Conv2d:          vanilla_vae.py:28-29,73-74; mcq_vae.py:170-171,178-179,189-190,205-209; vq_vae.py:63-67
 * ConvTranspose2d: vanilla_vae.py:50-55,65-70; mcq_vae.py:223-236        Linear: vanilla_vae.py:36-37,43
 * x [B,H,W,Ci], y [B,Ho,Wo,Co]; bias/add may be NULL (add has y's layout: ResidualLayer skip, vq_vae.py:69-70).
 * ws: scratch for the split-K partial sums of small-grid layers (may be NULL: no split-K).
 * in_scale/in_shift [Ci] (both or neither; NULL = plain x): the layer reads x' = act_in(x*scale[c] + shift[c]) instead
 * of x, i.e. the BatchNorm2d + LeakyReLU in front of it (vanilla_vae.py:71-74) is applied while loading and its
 * output never touches memory.  Only where ctvae_conv_input_transform_supported() says 1 (the 3-output-channel
 * image-side layers); kErrBadArg otherwise.
 * wino_dgrad_filters_out (may be NULL): for a layer whose forward AND data gradient run Winograd
 * (ctvae_conv_wino_filter_floats() > 0, that many floats), the forward's filter-transform launch also leaves the data
 * gradient's transformed filters there; hand them to ctvae_conv_dgrad of the same layer and weights (one transform
 * launch per layer and step instead of two).
 * wino_fwd_filters (may be NULL; same size): the forward's own transformed filters made ahead of time by
 * ctvae_wino_filters_batch for the CURRENT weights -- no transform launch at all in this call. */
int ctvae_conv_forward(int kind, const float* x, const float* w, const float* bias, const float* add, float* y, int B,
                       int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad, int act, const float* in_scale,
                       const float* in_shift, int in_act, float* wino_dgrad_filters_out, const float* wino_fwd_filters, float* ws,
                       size_t ws_bytes, void* stream);
/* nn.Linear(Ci, C*P) followed by .view(-1, C, h, w), h*w = P (decoder_input, vanilla_vae.py:43,101-102): y is written as the
 * NHWC tensor [B, h, w, C] the next layer gathers -- output feature c*P + p goes to column p*C + c in the GEMM's epilogue, no
 * layout launch behind it.  x [B, Ci], w the packed [Ci][C*P] block, bias [C*P] or NULL.  _supported: 1 where the layer runs on
 * the vector tile kernel without a K split (Ci % 32 == 0, enough rows); elsewhere use ctvae_conv_forward + ctvae_permute.
 * Backward is unchanged: ctvae_permute / ctvae_splitk_permute of the gradient, then ctvae_conv_backward of the Linear. */
int ctvae_linear_pixmajor_supported(int B, int Ci, int C, int P, size_t ws_bytes);
int ctvae_linear_pixmajor_forward(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int C, int P, int act,
                                  float* ws, size_t ws_bytes, void* stream);
/* Both Winograd filter sets (forward / data gradient, ctvae_conv_wino_filter_floats() floats each) of n 3x3 stride-1 layers
 * in ONE launch: w[l] = the layer's packed weights [9][Ci][Co], Ci, Co multiples of 32.  The residual stacks of MCQ-VAE /
 * CT-MCQ-VAE have 13 / 14 such layers (vq_vae.py:57-70 via mcq_vae.py:182-216); their weights change once per optimizer step.
 * w, fwd_filters, dgrad_filters, Ci, Co: HOST arrays of n entries (device pointers / ints). */
int ctvae_wino_filters_batch(int n, const float* const* w, float* const* fwd_filters, float* const* dgrad_filters, const int* Ci,
                             const int* Co, void* stream);
size_t ctvae_conv_wino_filter_floats(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                                     size_t ws_bytes);
int ctvae_conv_input_transform_supported(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad,
                                         int out_pad);

/* Conv2d|ConvTranspose2d -> BatchNorm2d -> activation as one call (nn.Sequential blocks vanilla_vae.py:25-35,47-75):
 * y = conv(x)+bias (kept for backward); the conv epilogue emits per-tile (count, mean, M2) so the batch
 * statistics cost no extra pass over y; a_out = act(BN(y)).  Arguments as ctvae_conv_forward / ctvae_bn_forward.
 * scale_shift_out [2][Co] (may be NULL): receives the per-channel affine a = act(y*scale + shift); with a_out == NULL
 * the apply pass is skipped altogether and the consumer applies it on load (ctvae_conv_forward in_scale/in_shift).
 * in_scale / in_shift [Ci] / in_act (both NULL normally): x is itself the raw BatchNorm input of the PREVIOUS block, handed over
 * that way, and this layer reads act(x*in_scale + in_shift) -- where ctvae_conv_input_transform_supported() says 1; the chain
 * Conv -> BN -> LeakyReLU -> Conv -> BN ... (vanilla_vae.py:25-35,47-62) then runs without a stand-alone apply launch. */
/* 1 when a training call of ctvae_conv_bn_act_forward with this geometry runs BatchNorm's apply as a launch of its own (which
 * a_out == NULL + scale_shift_out saves), 0 when the activated tensor comes out of the channel-owner launch that follows a
 * split-K convolution anyway (then handing the consumer the raw tensor only costs it arithmetic). */
int ctvae_conv_bn_act_apply_is_separate(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                                        size_t ws_bytes);
int ctvae_conv_bn_act_forward(int kind, const float* x, const float* w, const float* bias, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                              int training, int act, float* y, float* a_out, float* save_mean, float* save_invstd,
                              float* scale_shift_out, int64_t* num_batches_tracked, int B, int H, int W, int Ci, int Co, int k,
                              int stride, int pad, int out_pad, const float* in_scale, const float* in_shift, int in_act,
                              float* ws, size_t ws_bytes, void* stream);

/* dx = (dgrad(dy, w) + add) * act'(mask)      (autograd of the ops above)
 * add / mask (saved post-activation output of the PREVIOUS layer, layout of dx) may be NULL.
 * wino_filters: NULL, or what ctvae_conv_forward left in wino_dgrad_filters_out for these weights. */
int ctvae_conv_dgrad(int kind, const float* dy, const float* w, const float* add, const float* mask, int mask_act,
                     float* dx, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                     const float* wino_filters, float* ws, size_t ws_bytes, void* stream);

/* ctvae_conv_dgrad whose result dx is the gradient w.r.t. a = act(BN(y)), the output of a train-mode BatchNorm2d
 * (+activation) that fed this layer (autograd of vanilla_vae.py:28-31 chained into the next block).  The epilogue
 * also emits, per output tile, that BatchNorm's backward sums (sum g', sum g'*xhat; g' = dx*act'(gamma*xhat+beta),
 * xhat = (y-mean)*invstd) into bn_part [bn_part_rows][Ci][2], which ctvae_bn_backward takes as part_in: one pass
 * over (g_a, y) less.  bn_part_rows must equal ctvae_conv_dgrad_bn_rows() of the same geometry and workspace; that
 * function returns 0 when the launch configuration cannot fuse (split-K), then use ctvae_conv_dgrad. */
int ctvae_conv_dgrad_bn_rows(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                             size_t ws_bytes);
int ctvae_conv_dgrad_bn(int kind, const float* dy, const float* w, const float* add, const float* mask, int mask_act,
                        float* dx, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                        const float* bn_y, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                        const float* bn_beta, int bn_act, float* bn_part, int bn_part_rows, float* ws, size_t ws_bytes,
                        void* stream);

/* dw (+)= wgrad(x, dy);  dbias (+)= sum over pixels of dy (dbias may be NULL).  Deterministic two-pass. */
/* in_scale/in_shift/in_act: as ctvae_conv_forward (x is then the raw BatchNorm input y of the previous block). */
int ctvae_conv_wgrad(int kind, const float* x, const float* dy, float* dw, float* dbias, int B, int H, int W, int Ci,
                     int Co, int k, int stride, int pad, int out_pad, int accumulate, const float* in_scale,
                     const float* in_shift, int in_act, const float* dy_bn_y, const float* dy_bn_coef, int dy_bn_act,
                     float* gy_out, float* bn_dgamma, float* bn_dbeta, int bn_accumulate, float* ws, size_t ws_bytes,
                     void* stream);
/* A layer's whole backward pass in one call: weight (+ bias) gradient as ctvae_conv_wgrad(x, dy -> dw, dbias) and data
 * gradient as ctvae_conv_dgrad / ctvae_conv_dgrad_bn(dy, w -> dx; optional mask, Winograd filters, fused BatchNorm-backward
 * sums).  The two GEMMs are independent; when both take their 64x64 tile kernels they are issued as ONE launch
 * (conv_bwd_pair_kernel: one kernel boundary instead of two, the weight-gradient workgroups start while the data
 * gradient's stores drain), otherwise as the separate launches of those entry points.  Each GEMM uses one half of ws:
 * query bn_part_rows with ctvae_conv_backward_bn_rows (0: this layer's data gradient cannot emit the sums).
 * bn_part == NULL: no BatchNorm fusion.
 * bn_coef_out [7][Ci] (may be NULL; needs bn_part): that BatchNorm's backward FINALIZE rides as extra blocks of this call's
 * finishing launch (the slab reduction) -- rows 0-4 = k1,k2,k3,scale,shift as ctvae_bn_backward's coef_out, rows 5,6 = this
 * pass's d gamma, d beta.  ctvae_bn_backward(coef_in = bn_coef_out) then only applies and commits them: one launch less per
 * Conv->BN->act->Conv link of the backward pass (vanilla_vae.py:25-35,47-62).  bn_dgamma / bn_dbeta (both or none):
 * the rider commits the parameter gradients itself (+= under bn_accumulate) -- for a caller that applies the coefficients
 * on load (ctvae_conv_wgrad dy_bn_*) and therefore never calls ctvae_bn_backward.
 * in_scale / in_shift / in_act and dy_bn_y / dy_bn_coef / dy_bn_act / gy_out: the weight gradient's options as in
 * ctvae_conv_wgrad (the layers of the final block, vanilla_vae.py:64-75); with gy_out the data gradient reads g_y from it.
 * x == bn_y with in_scale / in_shift given (they must then be that BatchNorm's scale / shift, in_act == bn_act) on the 32 -> 3
 * picture-side conv (vanilla_vae.py:73-74): ONE kernel forms the data gradient, the BatchNorm-backward sums and the weight
 * gradient in a single pass over y (image.hip img_bwd_fused_kernel). */
int ctvae_conv_backward_bn_rows(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                                size_t ws_bytes);
int ctvae_conv_backward(int kind, const float* x, const float* dy, const float* w, float* dw, float* dbias, float* dx, int B,
                        int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad, int accumulate, const float* mask,
                        int mask_act, const float* wino_filters, const float* bn_y, const float* bn_mean,
                        const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int bn_act, float* bn_part,
                        int bn_part_rows, float* bn_coef_out, float* bn_dgamma, float* bn_dbeta, int bn_accumulate,
                        const float* in_scale, const float* in_shift, int in_act,
                        const float* dy_bn_y, const float* dy_bn_coef, int dy_bn_act, float* gy_out, float* ws, size_t ws_bytes,
                        void* stream);

/* Small layers behind a train-mode BatchNorm (the deep encoder / decoder blocks at small batch, vanilla_vae.py:25-35,47-62): the
 * data gradient runs split-K and its consumer is that BatchNorm's backward pass, so the slices are never summed into a tensor of
 * their own.  ctvae_conv_backward_lazy_slices: the number S (>= 2) of K slices such a call would produce for this geometry and
 * workspace, or 0 when this form does not apply (unsplit / Winograd / picture-side data gradient, or a tensor beyond the few MB
 * the channel-owner kernel pays for; for_bn = 0 drops that size condition -- any split data gradient -- for consumers that take
 * the slices pixel-major, see ctvae_gauss_latent_backward / ctvae_splitk_permute).  ctvae_conv_backward_lazy (pixel_major = 0
 * for the BatchNorm consumer): weight (+ bias) gradient as ctvae_conv_backward (in_scale /
 * in_shift / in_act: its x operand read through the previous block's BatchNorm + activation, as there), and the data
 * gradient's raw slices, CHANNEL-MAJOR, in dx_slices [S][Ci][B*H*W] (no mask, no BatchNorm sums, dx itself is not written;
 * the rows of a slice are in the data gradient's class-major order, not pixel order).
 * ctvae_bn_backward_fused (kind ... out_pad: the geometry of the ctvae_conv_backward_lazy call that wrote the slices; the
 * BatchNorm's tensor is that layer's input [B,H,W,Ci]): g_y, d gamma, d beta from those slices in ONE launch -- a workgroup
 * owns 2 or 4 channels and all R = B*H*W rows: slice sum, g' = g_a * act'(gamma*xhat+beta), both reductions, the coefficients and g_y = k1*g' + k2*y + k3.
 * Replaces split-K finish + ctvae_bn_backward's partial / finalize / apply launches (reference: autograd of
 * nn.BatchNorm2d + nn.LeakyReLU). */
int ctvae_conv_backward_lazy_slices(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                                    int for_bn, size_t ws_bytes);
int ctvae_conv_backward_lazy(int kind, const float* x, const float* dy, const float* w, float* dw, float* dbias, float* dx_slices,
                             int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad, int accumulate,
                             const float* in_scale, const float* in_shift, int in_act, int pixel_major, float* ws, size_t ws_bytes,
                             void* stream);
int ctvae_bn_backward_fused(const float* g_a_slices, int slices, int kind, int B, int H, int W, int Ci, int Co, int k, int stride,
                            int pad, int out_pad, const float* y, const float* gamma, const float* beta, const float* save_mean,
                            const float* save_invstd, int act, float* g_y, float* dgamma, float* dbeta, int accumulate, void* stream);

/* dy_bn_y / dy_bn_coef / gy_out (all or none): `dy` is then g_a, the gradient w.r.t. the output of the BatchNorm +
 * activation that follows this layer; the kernel forms g_y = k1*g_a*act'(y*scale+shift) + k2*y + k3 while loading
 * (coef = [5][Co]: k1,k2,k3,scale,shift as written by ctvae_bn_backward coef_out), uses it for dw/dbias and writes it to
 * gy_out for the ctvae_conv_dgrad call that follows: the separate BatchNorm-backward apply pass disappears.
 * Only where ctvae_conv_wgrad_bn_apply_supported() says 1 (the 32->32 transposed conv of the final block).
 * Where it says 2 (encoder.0, the 3 -> 32 picture-side conv, vanilla_vae.py:25-35 with i = 0: no data gradient follows, the
 * input is the picture) gy_out must be NULL -- g_y is used for dw/dbias and never written -- coef is the [7][Co] block of
 * ctvae_conv_backward's bn_coef_out, and bn_dgamma / bn_dbeta (both or none; += under bn_accumulate) receive its rows 5, 6:
 * the BatchNorm-backward apply launch and its 33 MB output disappear. */
int ctvae_conv_wgrad_bn_apply_supported(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad,
                                        int out_pad);

/* Train/eval BatchNorm2d + activation on an [R=B*H*W][C] matrix (vanilla_vae.py:30-31,56-57,71-72).
 * training: batch statistics (biased var, eps), running stats updated with `momentum` and the unbiased
 * variance; save_mean/save_invstd [C] are written for the backward pass; num_batches_tracked (may be NULL) is
 * incremented on the device (nn.BatchNorm2d bookkeeping without an extra launch). */
int ctvae_bn_forward(const float* y, int R, int C, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float momentum, float eps, int training, int act, float* out,
                     float* save_mean, float* save_invstd, int64_t* num_batches_tracked, float* ws, size_t ws_bytes,
                     void* stream);
/* g_y from g_a (grad wrt the activated output); the activation derivative is re-derived from the sign of
 * gamma*invstd*(y-mean)+beta, so the activated tensor is not read; dgamma/dbeta (+)= ...
 * part_in/part_rows: the per-tile sums a ctvae_conv_dgrad_bn launch emitted for this g_a (NULL/0: computed here).
 * coef_out [5][C] (may be NULL): k1,k2,k3,scale,shift of g_y = k1*g_a*act'(y*scale+shift) + k2*y + k3; with
 * g_y == NULL the apply pass is left to ctvae_conv_wgrad (dy_bn_*).
 * coef_in [7][C] (NULL normally; excludes part_in / coef_out, needs g_y): the finalize already ran as a rider of
 * ctvae_conv_backward(bn_coef_out) for exactly this g_a; only the apply pass runs and d gamma / d beta (rows 5,6) are
 * committed to dgamma / dbeta under `accumulate`. */
int ctvae_bn_backward(const float* g_a, const float* beta, const float* y, int R, int C, const float* gamma,
                      const float* save_mean, const float* save_invstd, int act, float* g_y, float* dgamma,
                      float* dbeta, int accumulate, const float* part_in, int part_rows, float* coef_out,
                      const float* coef_in, float* ws, size_t ws_bytes, void* stream);

/* Pairwise edge scorer of CausalTransition.graph_discovers[k] (ct_mcq_vae.py:86-95,147-151), after the separable first
 * Linear: u = x W1[:, :D]^T, v = x W1[:, D:]^T + b1 ([B,N,H] each, computed by the caller):
 *   out[b,i,j] = sigmoid(b2 + sum_h w2[h] * leaky_relu(u[b,i,h] + v[b,j,h], slope))        out [B,N,N]
 * No [B,N,N,H] tensor is ever materialised.  Backward (N <= 64): d_u, d_v [B,N,H]; d_w2_part [B][H] and d_b2_part [B]
 * are per-sample partials the caller sums over B (deterministic, no atomics).
 * per_sample = 1: w2 is [B][H] and b2 [B] -- every sample has its own scorer (the per-action discoverers of one batch in
 * ONE launch, ct_mcq_vae.py:149-151); the partials then ARE the gradients of those rows.
 * row_of [B] (per_sample only, may be NULL = the sample's own row): the row of the w2 / b2 bank sample b uses -- the bank of
 * all discoverers is then read in place, indexed by the sample's action.
 * u / v rows have stride ld (>= H), d_u / d_v rows stride ldd: they may be column blocks of one wider GEMM output / gradient. */
int ctvae_pair_mlp_forward(const float* u, const float* v, int ld, const float* w2, const float* b2, float* out, int B, int N,
                           int H, float slope, int per_sample, const int32_t* row_of, void* stream);
int ctvae_pair_mlp_backward(const float* u, const float* v, int ld, const float* w2, const float* out, const float* g_out,
                            float* d_u, float* d_v, int ldd, float* d_w2_part, float* d_b2_part, int B, int N, int H, float slope,
                            int per_sample, const int32_t* row_of, void* stream);

/* GATv2 attention scores of CausalTransition.graph_transitioner (ct_mcq_vae.py:103-114; torch_geometric GATv2Conv with
 * edge_dim=1 on dense graphs): xl, xr [B,N,H,C] = lin_l(x), lin_r(x); attr [B,N,N] edge attribute of r -> c (self loops
 * included); we, att [H,C].
 *   mode 0: out[b,h,r,c] = sum_k att[h,k] * leaky_relu(xl[b,r,h,k] + xr[b,c,h,k] + attr[b,r,c]*we[h,k], slope)
 *   mode 1: out[b,h,r,c] = sum_k att[h,k] * we[h,k] * leaky_relu'(same argument)          (d out / d attr)
 * N*N <= 4352 (N <= 65), slope < 1.  Backward (C <= 128): d_xl, d_xr [B,N,H,C]; d_att_part, d_we_part [B][H][C] are
 * per-sample partials for the caller to sum; d attr = sum_h g * (mode 1 output). */
int ctvae_gat_score(int mode, const float* xl, const float* xr, const float* attr, const float* we, const float* att, float* out,
                    int B, int N, int H, int C, float slope, void* stream);
int ctvae_gat_score_backward(const float* xl, const float* xr, const float* attr, const float* we, const float* att,
                             const float* g, float* d_xl, float* d_xr, float* d_att_part, float* d_we_part, int B, int N, int H,
                             int C, float slope, void* stream);

/* One whole GATv2Conv layer of CausalTransition.graph_transitioner (ct_mcq_vae.py:103-114; torch_geometric GATv2Conv(in, C,
 * edge_dim=1, heads=H), add_self_loops / fill_value 'mean', negative_slope = slope) on B dense graphs over the 64 LATENT
 * nodes (the appended action node has no outgoing edge and its own output is discarded, ct_mcq_vae.py:203-221, so it never
 * reaches a latent node).  xl / xr: rows [B*64] of stride ld, head slot hs at columns hs*C..; adj [B,64,64] weighted
 * adjacency (adj[b,r,c] != 0: edge r -> c); we, att, bias [H][C].  head_map [B*Hs] (or NULL: slot == head): the head whose
 * parameters slot hs of sample b uses -- _compute_y reads only head 0 and head 1+action of the last layer (:224-226).
 *   out[b,c,hs,:] = act(sum_r alpha[b,hs,r,c] * xl[b,r,hs,:] + bias[head]),  alpha = softmax over the kept sources of
 *   S[r,c] = sum_k att[k] * leaky_relu(xl[r,k] + xr[c,k] + a'[r,c]*we[k], slope);  alpha [B,Hs,64,64] is kept for backward.
 * act: 0 or 1 (nn.LeakyReLU() between the two layers).  16 <= C <= 128.
 * Backward: g_out like out; writes dS, dattr [B,Hs,64,64] (scratch the caller provides), d_xl / d_xr (rows of stride ldd),
 * per-sample partials d_bias_part / d_att_part / d_we_part [B][Hs][C] (the caller sums them per head: deterministic, no
 * atomics) and, when d_adj != NULL, d_adj [B,64,64] (+= when accumulate_dadj). */
int ctvae_gat_layer_forward(const float* xl, const float* xr, int ld, const float* adj, const float* we, const float* att,
                            const float* bias, const int32_t* head_map, float* out, int ldo, float* alpha, int B, int Hs, int C,
                            float slope, int act, void* stream);
int ctvae_gat_layer_backward(const float* xl, const float* xr, int ld, const float* adj, const float* we, const float* att,
                             const float* bias, const int32_t* head_map, const float* out, int ldo, const float* alpha,
                             const float* g_out, float* dS, float* dattr, float* d_xl, float* d_xr, int ldd, float* d_bias_part,
                             float* d_att_part, float* d_we_part, float* d_adj, int accumulate_dadj, int B, int Hs, int C,
                             float slope, int act, void* stream);

/* Grouped Linear on blocks of 64 rows per sample (the 64 latent nodes of CausalTransition): per output segment s < nseg
 * (<= 4) every sample picks one matrix of a stacked weight bank by its group id,
 *   y[b, m, s*N + n] = sum_k x[b, m, k] * W[s][group[s][b]][n][k] + bias[s][group[s][b]][n],
 * element (g, n, k) of bank s at W[s] + g*w_gstride[s] + n*ldw[s] + k, bias (may be NULL) at bias[s] + g*b_gstride[s] + n;
 * group[s] == NULL: everybody uses matrix 0 (a plain nn.Linear).  Replaces the per-group nn.Linear calls + batch splitting
 * of ct_mcq_vae.py:129-154 (graph_discovers[0] and graph_discovers[1+argmax(action)]) and the lin_l / lin_r rows of the two
 * heads of the last GATv2Conv that _compute_y reads (:224-226), without gathering weights per sample.  W / ldw / w_gstride /
 * bias / b_gstride / group are HOST arrays of nseg entries holding device pointers / ints.  K, N, ld*, strides: multiples of 4.
 * dgrad: dx[b,m,k] = sum_s sum_n dy[b,m,s*N+n] * W[s][g][n][k].
 * wgrad (one segment, columns col0..col0+N-1 of dy): dW[(g*N + n)*ldo + k] (+)= sum_{b: group[b]==g} sum_m dy[b,m,col0+n] *
 * x[b,m,k] (ldo >= K: the destination may be a column block of a wider bank), dbias[g*N + n] (+)= column sums (may be NULL);
 * batch slices meet in a fixed-order slab
 * reduction inside ws (>= ctvae_glinear_wgrad_ws_bytes): deterministic, no atomics. */
int ctvae_glinear_forward(const float* x, int ldx, int K, int nseg, int N, const float* const* W, const int* ldw,
                          const int64_t* w_gstride, const float* const* bias, const int* b_gstride,
                          const int32_t* const* group, float* y, int ldy, int B, void* stream);
int ctvae_glinear_dgrad(const float* dy, int ldy, int nseg, int N, const float* const* W, const int* ldw,
                        const int64_t* w_gstride, const int32_t* const* group, float* dx, int ldx, int K, int B, void* stream);
size_t ctvae_glinear_wgrad_ws_bytes(int G, int N, int K);
int ctvae_glinear_wgrad(const float* x, int ldx, int K, const float* dy, int ldy, int col0, int N, const int32_t* group, int G,
                        int B, float* dW, int ldo, float* dbias, int accumulate, float* ws, size_t ws_bytes, void* stream);

/* CausalTransition._compute_adj's blend (ct_mcq_vae.py:153): out[r, j] = s0[r, j] * (1 - mask[r]) + s1[r, j] * mask[r] for
 * `rows` = B*64 rows of 64 targets (s0: discoverer 0's scores, s1: the action's discoverer, mask: the intervention mask per
 * (sample, source node)); backward: g0 = g * (1 - mask), g1 = g * mask, g_mask[r] = sum_j g * (s1 - s0). */
int ctvae_ct_blend_forward(const float* s0, const float* s1, const float* mask, float* out, long rows, void* stream);
int ctvae_ct_blend_backward(const float* g, const float* s0, const float* s1, const float* mask, float* g0, float* g1, float* g_mask,
                            long rows, void* stream);
/* PositionalEncoding.forward (ct_mcq_vae.py:33-38) on n = B*S*D elements, sd = S*D (the table pe [S][D] repeats per sample):
 * out = (x + pe) * keep * scale with keep the dropout mask (NULL: no dropout) and scale = 1/(1-p); backward g * keep * scale. */
int ctvae_ct_posenc_forward(const float* x, const float* pe, const float* keep, float scale, float* out, long n, int sd,
                            void* stream);
int ctvae_ct_posenc_backward(const float* g, const float* keep, float scale, float* g_x, long n, void* stream);
/* F.one_hot(inds, N).float() (CTMCQVAE.ct_preprocess, ct_mcq_vae.py:472-480): out [n][N], N % 4 == 0. */
int ctvae_one_hot(const int64_t* inds, long n, int N, float* out, void* stream);
/* Per-sample partial gradients into the rows of a parameter bank, in row order (deterministic, no atomics):
 * out[z][g][c] (+)= sum_{r < rows, group[r] == g} parts[z*mat_stride + r*ld + c], z < nmat, g < G, c < C; group == NULL: every
 * row belongs to group 0.  Replaces the one_hot(group)^T @ parts products behind the scorer rows of graph_discovers
 * (ct_mcq_vae.py:147-151) and the per-head vectors of the last GATv2Conv (:224-226). */
int ctvae_group_rowsum(const float* parts, long mat_stride, int nmat, int rows, int C, int ld, const int32_t* group, int G, float* out,
                       int accumulate, void* stream);

/* forward_action's regulariser (ct_mcq_vae.py:275): beta * adjacency_KL_loss(adj) + delta * graph_size_loss(graph) + epsilon *
 * positive_trial_loss(adj) (:314-323) on adj, graph [B,64,64] with the uniform draws of the KL target [B,4096] given
 * (torch.rand in :316).  forward writes part4 [B][4] = {KL_b, ||graph_b||_F, ||prod_j(1 - adj_b[i,j])||_2,
 * ckl*KL_b + cgs*||.||_F + cpt*||.||_2}; the regulariser is the sum of the last column over the batch.  backward (g_loss: device scalar) writes
 * d_adj, d_graph for ckl = beta/B, cgs = delta/B, cpt = epsilon/B; the row products' gradient uses exclusive prefix / suffix
 * products (exact when a factor is 0). */
int ctvae_ct_reg_forward(const float* adj, const float* graph, const float* uniform, float* part4, float ckl, float cgs, float cpt,
                         int B, int N, void* stream);
int ctvae_ct_reg_backward(const float* adj, const float* graph, const float* uniform, const float* part4, const float* g_loss,
                          float ckl, float cgs, float cpt, float* d_adj, float* d_graph, int B, int N, void* stream);
/* Tail of _compute_y (ct_mcq_vae.py:226-228) per node row r < R: probs[r,:] = softmax_d(y[r,0,:] * (1 - mask[r]) + y[r,1,:] *
 * mask[r]) (Hs == 2) or softmax_d(y[r,0,:]) (Hs == 1, mask unused); D <= 64.  Backward: dy like y, dmask [R] (may be NULL). */
int ctvae_ct_blend_softmax_forward(const float* y, const float* mask, float* probs, long R, int Hs, int D, void* stream);
int ctvae_ct_blend_softmax_backward(const float* g, const float* probs, const float* y, const float* mask, float* dy, float* dmask,
                                    long R, int Hs, int D, void* stream);
/* latent_CrossEntropy_loss (ct_mcq_vae.py:306-311) per node row: row_loss[r] = logsumexp_d(lp) - lp[target[r]] with
 * lp = log(max(probs, 1e-4)); the loss is the mean of row_loss.  Backward: d_probs for d loss = g_loss[0] (device scalar). */
int ctvae_ct_latent_ce_forward(const float* probs, const int64_t* target, float* row_loss, long R, int D, void* stream);
int ctvae_ct_latent_ce_backward(const float* probs, const int64_t* target, const float* g_loss, float* d_probs, long R, int D,
                                void* stream);

/* CausalTransition._compute_mask (ct_mcq_vae.py:117-127) for S == D == 64: pos = pe [S,D] * keep [B,S,D] * scale (the dropout of
 * PositionalEncoding applied to zeros; keep == NULL: eval mode), inter = sigmoid(W [action_b ; pos] + bias) with W [D][A+D]
 * (mask.0.weight), p[b,s] = sum_d x[b,s,d] * inter[b,s,d], sample = straight-through Bernoulli(p) from the exponential draws
 * expo [B,S,2] (Gumbel = -log E).  Outputs inter [B,S,D], p, sample, soft [B,S].  Backward for g = d loss / d sample: per-sample
 * partials dWp [B][A+D][D] (transposed like W^T) and dbp [B][D], which the caller sums over B (deterministic). */
int ctvae_ct_mask_forward(const float* x, const float* action, const float* pe, const float* keep, float scale, const float* W,
                          const float* bias, const float* expo, int B, int S, int D, int A, float* inter, float* p, float* sample,
                          float* soft, void* stream);
int ctvae_ct_mask_backward(const float* x, const float* action, const float* pe, const float* keep, float scale, const float* inter,
                           const float* p, const float* soft, const float* g, int B, int S, int D, int A, float* dWp, float* dbp,
                           void* stream);
/* Straight-through Bernoulli(p) like ctvae_gumbel_st_forward but from exponential draws expo [n][2] (what F.gumbel_softmax
 * draws: Gumbel = -log E) and, when weighted != NULL, also weighted = p * sample (weighted_graph, ct_mcq_vae.py:244,271).
 * Backward: gp = (g_sample + g_weighted * p) * d sample/d p + g_weighted * sample (either gradient may be NULL). */
int ctvae_ct_sample_forward(const float* p, const float* expo, float* sample, float* soft, float* weighted, long n, void* stream);
int ctvae_ct_sample_backward(const float* g_sample, const float* g_weighted, const float* p, const float* soft, const float* sample,
                             float* g_p, long n, void* stream);

/* layout change at the NCHW API boundary: to_nhwc=1: in [B,C,P] -> out [B,P,C]; 0: the inverse
 * (torch.flatten on NCHW, vanilla_vae.py:85; .view(-1,512,2,2), vanilla_vae.py:102) */
int ctvae_permute(const float* in, float* out, int B, int C, int P, int to_nhwc, void* stream);
int ctvae_act_forward(const float* in, float* out, long n, int act, void* stream);           /* mcq_vae.py:185,216 */
int ctvae_act_backward(const float* g_out, const float* out, float* g_in, long n, int act, void* stream);

/* The Gaussian latent as one node (vanilla_vae.py:85-92,107-122): heads [B][2L] holds mu | logvar (the fused fc_mu / fc_var
 * GEMM output, L % 4 == 0).  forward: z = eps*exp(0.5*logvar) + mu; eps_in given (injected noise), or NULL: N(0,1) drawn in
 * the kernel (Philox4x32-10 keyed by rng[0], stream position rng[1]; rng: two uint64 on the device) -- eps_out [B,L] keeps it
 * for backward.  backward: g_heads [B][2L] = (g_mu + g_z | g_logvar + g_z*eps*0.5*exp(0.5*logvar)) in one launch (any of the
 * three incoming gradients may be NULL = 0); rng_bump != NULL advances rng[1] by one (pass it when eps was drawn in forward). */
int ctvae_gauss_latent_forward(float* heads, const float* eps_in, const uint64_t* rng, float* eps_out, float* z, int B, int L,
                               const float* head_slices, int slices, const float* head_bias, void* stream);
int ctvae_gauss_latent_backward(const float* g_mu, const float* g_logvar, const float* g_z, const float* heads, const float* eps,
                                float* g_heads, uint64_t* rng_bump, int B, int L, int g_z_slices, void* stream);
/* Split-K results handed to an element-wise consumer as raw slices (the small-batch step: one launch less per hand-over).
 * head_slices [slices][B][2L] + head_bias [2L] (NULL normally): the fused heads are still the slices of their GEMM
 * (ctvae_conv_forward_lazy); ctvae_gauss_latent_forward sums them, writes `heads` (then an output) and goes on.
 * g_z_slices > 0: g_z is that many slices [S][B][L] of decoder_input's data gradient (ctvae_conv_backward_lazy, pixel_major).
 * ctvae_conv_forward_lazy_slices / ctvae_conv_forward_lazy: the number of K slices a forward conv of this geometry would run
 * (0: not split / not the general tile kernel) and the launch that leaves them, without bias or activation, in
 * y_slices [S][B*Ho*Wo][Co].  ctvae_splitk_permute: out[b][c][p] = sum_s slices[s][(b*P+p)*C+c] -- the slice sum of an NHWC
 * data gradient and the NHWC -> NCHW change behind it (the .view(-1,512,2,2) of vanilla_vae.py:102, backward) in one launch. */
int ctvae_conv_forward_lazy_slices(int kind, int B, int H, int W, int Ci, int Co, int k, int stride, int pad, int out_pad,
                                   size_t ws_bytes);
int ctvae_conv_forward_lazy(int kind, const float* x, const float* w, float* y_slices, int B, int H, int W, int Ci, int Co, int k,
                            int stride, int pad, int out_pad, float* ws, size_t ws_bytes, void* stream);
int ctvae_splitk_permute(const float* slices, int n_slices, float* out, int B, int C, int P, void* stream);
/* z = eps*exp(0.5*logvar)+mu (vanilla_vae.py:115-117); mu/logvar rows may be strided (slices of one head GEMM) */
int ctvae_reparam_forward(const float* mu, long mu_row_stride, const float* logvar, long lv_row_stride, const float* eps,
                          float* z, int B, int L, void* stream);
int ctvae_reparam_backward(const float* g_z, const float* logvar, long lv_row_stride, const float* eps, float* g_mu,
                           float* g_logvar, int B, int L, void* stream);

/* out4 = {loss, mse, kld, -kld}: mse = mean((r-x)^2) (F.mse_loss, vanilla_vae.py:140, mcq_vae.py:279);
 * kld = mean_b(-0.5*sum_d(1+lv-mu^2-e^lv)) (vanilla_vae.py:143) when mu != NULL; loss = mse + M_N*kld (+ extra[0]) */
int ctvae_loss_forward(const float* recons, const float* x, long n, const float* mu, long mu_row_stride,
                       const float* logvar, long lv_row_stride, int B, int L, float M_N, const float* extra, float* out4,
                       float* ws, size_t ws_bytes, void* stream);
/* The same, and in the same pass the gradients of `loss` for a unit upstream gradient: g_recons [n] (w.r.t. the input of
 * recons_act when the reconstruction is that activation's output, as ctvae_loss_backward), g_mu / g_logvar [B][L] dense.
 * The loss value is not an operand of its own gradient, so a caller whose backward pass starts at `loss` needs no
 * ctvae_loss_backward launch (and no second read of both pictures). */
int ctvae_loss_forward_grad(const float* recons, const float* x, long n, const float* mu, long mu_row_stride, const float* logvar,
                            long lv_row_stride, int B, int L, float M_N, const float* extra, float* out4, float* g_recons,
                            float* g_mu, float* g_logvar, int recons_act, float* ws, size_t ws_bytes, void* stream);
/* recons_act (CTVAE_ACT_*; 0 = none) in the three reconstruction backward calls: recons is the OUTPUT of that activation
 * (the Tanh closing final_layer, vanilla_vae.py:74 / the decoder, mcq_vae.py:236) and g_recons is the gradient w.r.t. the
 * activation's input, g * act'(recons): the producer's activation-backward pass folded into this one. */
int ctvae_mse_backward(const float* recons, const float* x, const float* g_loss, float* g_recons, long n, int recons_act,
                       void* stream);
/* ctvae_mse_backward (or ctvae_logcosh_backward when logcosh_alpha > 0) and ctvae_kl_backward in ONE launch. */
int ctvae_loss_backward(const float* recons, const float* x, const float* g_loss, float* g_recons, long n, float logcosh_alpha,
                        const float* mu, long mu_row_stride, const float* logvar, long lv_row_stride, float* g_mu, float* g_logvar,
                        int B, int L, float M_N, int recons_act, void* stream);
/* LogCoshVAE's objective (logcosh_vae.py:141-155): out4 = {loss, rl, kld, -kld} with
 * rl = 1/alpha * mean(alpha t + log(1 + exp(-2 alpha t)) - log 2), t = recons - x; loss = rl + M_N * kld (the caller passes
 * M_N = beta * kld_weight).  Backward of the reconstruction term: g_recons = g_loss[0] * tanh(alpha t) / n; the KL term
 * uses ctvae_kl_backward. */
int ctvae_logcosh_loss_forward(const float* recons, const float* x, long n, float alpha, const float* mu, long mu_row_stride,
                               const float* logvar, long lv_row_stride, int B, int L, float M_N, float* out4, float* ws,
                               size_t ws_bytes, void* stream);
int ctvae_logcosh_backward(const float* recons, const float* x, const float* g_loss, float* g_recons, long n, float alpha,
                           int recons_act, void* stream);
int ctvae_kl_backward(const float* mu, long mu_row_stride, const float* logvar, long lv_row_stride, const float* g_loss,
                      float* g_mu, float* g_logvar, int B, int L, float M_N, void* stream);

/* Multi-codebook VQ on latents [B*HW][D]; C codebooks [K][D/C] stored back to back at `codebooks`;
 * codebook i reads channels [i, i+D/C) (reference quirk, mcq_vae.py:104,117); inds [B,C,HW] int64.
 * compute_inds: mcq_vae.py:26-39,100-110;  compute_latents: mcq_vae.py:41-64,112-127 */
int ctvae_vq_inds(const float* latents, const float* codebooks, int64_t* inds, int B, int HW, int D, int K, int C,
                  void* stream);
int ctvae_vq_lookup(const float* latents, const float* codebooks, const int64_t* inds, float* quantized, float* vq_loss,
                    float beta, int B, int HW, int D, int K, int C, float* ws, size_t ws_bytes, void* stream);
/* g_latents (may be NULL) = straight-through g_q + commitment term; d_codebooks (may be NULL) (+)= embedding term.
 * ws (may be NULL): scratch that lets the codebook pass split the positions over more workgroups (fixed-order sum). */
int ctvae_vq_backward(const float* g_quantized, const float* g_vq_loss, const float* latents, const float* codebooks,
                      const int64_t* inds, float* g_latents, float* d_codebooks, int accumulate, float beta, int B, int HW,
                      int D, int K, int C, float* ws, size_t ws_bytes, void* stream);

/* Straight-through Bernoulli(p) sample = F.gumbel_softmax(log(clamp([1-p,p],1e-4)), tau=1, hard=True)[...,1]
 * (ct_mcq_vae.py:126,177-183).  gumbel_noise [n][2] standard Gumbel draws (injectable);
 * `soft` [n] keeps the soft probability for the backward pass. */
int ctvae_gumbel_st_forward(const float* p, const float* gumbel_noise, float* sample, float* soft, long n, void* stream);
int ctvae_gumbel_st_backward(const float* g_sample, const float* p, const float* soft, float* g_p, long n, void* stream);

/* Categorical latent of CategoricalVAE (models/cat_vae.py).  rows = B * latent_dim, Q = categorical_dim (<= 256).
 * Gumbel-softmax reparameterisation (cat_vae.py:118-132): sample = softmax((logits + g) / temperature) along Q with
 * g = -log(-log(uniform + eps) + eps); `uniform` [rows][Q] are the U[0,1) draws (injectable).
 * Backward: g_logits = sample * (g_sample - sum_Q g_sample * sample) / temperature. */
int ctvae_gumbel_softmax_forward(const float* logits, const float* uniform, float* sample, long rows, int Q, float temperature,
                                 float eps, void* stream);
int ctvae_gumbel_softmax_backward(const float* g_sample, const float* sample, float* g_logits, long rows, int Q,
                                  float temperature, void* stream);
/* KL between softmax(logits) and the uniform categorical prior (cat_vae.py:147,160-167):
 * kld[0] = 1/B * sum_rows sum_Q p * (log(p + eps) - log_prior), log_prior = log(1/Q + eps) (computed by the caller in
 * double like the reference's np.log).  ws: >= 4 KiB of scratch.  Backward writes g_logits = g_kld[0] * d kld / d logits. */
int ctvae_cat_kl_forward(const float* logits, long rows, int Q, int B, float eps, float log_prior, float* kld, float* ws,
                         size_t ws_bytes, void* stream);
int ctvae_cat_kl_backward(const float* logits, const float* g_kld, float* g_logits, long rows, int Q, int B, float eps,
                          float log_prior, void* stream);

/* Importance-weighted objective of IWAE / MIWAE (models/iwae.py:126-155, models/miwae.py:130-163).  recons [R][n] are the
 * reconstructions of the R = B*M*S latent samples, row r belongs to image r / rep of x [R/rep][n] (rep = M*S; same
 * per-image layout as recons, n % 4 == 0); mu / logvar [R][L] are the (repeated) posterior parameters of each row.
 *   lp[r] = mean_n (recons - x)^2, kld[r] = -0.5 sum_d (1 + lv - mu^2 - e^lv), lw = lp + M_N*kld, w = softmax over the S
 *   consecutive rows of a group, out4 = {loss = mean_groups sum_s w*lw, mean lp, mean kld, -mean kld};
 *   coef[r] = d loss / d lw[r] (weights NOT detached, like the reference) is kept for the backward call, which writes
 *   g_recons (may be NULL) and g_mu / g_logvar (both or neither) scaled by g_loss[0]. */
int ctvae_iw_loss_forward(const float* recons, const float* x, long n, int R, int rep, const float* mu, const float* logvar,
                          int L, int S, float M_N, float* lp, float* kld, float* coef, float* out4, void* stream);
int ctvae_iw_loss_backward(const float* recons, const float* x, long n, int R, int rep, const float* mu, const float* logvar,
                           int L, float M_N, const float* coef, const float* g_loss, float* g_recons, float* g_mu,
                           float* g_logvar, void* stream);

/* SWAE's reconstruction term F.mse_loss + F.l1_loss (swae.py:121-125) in one pass: out4 = {loss, rl, 0, 0} with
 * rl = mean((r-x)^2) + mean(|r-x|), loss = rl (+ extra[0]); backward g_recons = g_loss[0] * (2 t + sign(t)) / n * act'(recons)
 * (recons_act as in ctvae_mse_backward). */
int ctvae_l2l1_loss_forward(const float* recons, const float* x, long n, const float* extra, float* out4, float* ws, size_t ws_bytes,
                            void* stream);
int ctvae_l2l1_backward(const float* recons, const float* x, const float* g_loss, float* g_recons, long n, int recons_act,
                        void* stream);
/* One rung of LVAE's top-down pass (lvae.py:166-204) on [B][D] tensors: merge_gauss of the bottom-up (mu_e, logvar_e) and the
 * top-down (mu_t, logvar_t) Gaussians, z = eps * exp(logvar/2) + mu of the merged one, kl[b] = sum_d compute_kl_divergence(merged,
 * bottom-up) as the reference writes it.  Backward: g_z [B][D] and / or g_kl [B] -> gradients of the four inputs. */
int ctvae_ladder_merge_forward(const float* mu_e, const float* logvar_e, const float* mu_t, const float* logvar_t, const float* eps,
                               int B, int D, float* z, float* kl, void* stream);
int ctvae_ladder_merge_backward(const float* g_z, const float* g_kl, const float* mu_e, const float* logvar_e, const float* mu_t,
                                const float* logvar_t, const float* eps, int B, int D, float* g_mu_e, float* g_logvar_e,
                                float* g_mu_t, float* g_logvar_t, void* stream);
/* GammaVAE (gamma_vae.py:108-193).  Reparameterisation by shape augmentation with the draw zhat ~ Gamma(alpha + gamma_shape, 1)
 * given: z = h(a, h^-1(a, zhat)) / beta, a = alpha + gamma_shape (:108-149); backward g_alpha (both partials, as autograd forms
 * them), g_beta.  KL of the Gamma posteriors to the Gamma(prior_alpha, prior_beta) prior as the reference writes it (:151-171),
 * out[0] = mean_b sum_d; g_kld one float on the device.  The final layer's nn.Sigmoid (:78) as an elementwise pair
 * (n % 4 == 0).  ws: >= 4*B bytes. */
int ctvae_gamma_reparam_forward(const float* alpha, const float* beta, const float* zhat, float gamma_shape, float* z, long n,
                                void* stream);
int ctvae_gamma_reparam_backward(const float* g_z, const float* alpha, const float* beta, const float* zhat, float gamma_shape,
                                 float* g_alpha, float* g_beta, long n, void* stream);
int ctvae_gamma_kl_forward(const float* alpha, const float* beta, int B, int D, float prior_alpha, float prior_beta, float* out,
                           float* ws, size_t ws_bytes, void* stream);
int ctvae_gamma_kl_backward(const float* g_kld, const float* alpha, const float* beta, int B, int D, float prior_alpha,
                            float prior_beta, float* g_alpha, float* g_beta, void* stream);
int ctvae_sigmoid_forward(const float* x, float* y, long n, void* stream);
int ctvae_sigmoid_backward(const float* g, const float* y, float* g_x, long n, void* stream);
/* BetaTCVAE's decomposition of the KL term (betatc_vae.py:128-199) on z, mu, logvar [B][D] (D <= 32, 2 <= B <= 4096) with the
 * log importance weights log_iw [B][B] of minibatch stratified sampling (:176-184): M[i,j,d] = log N(z_i[d]; mu_j[d], e^lv_j[d]) +
 * log_iw[i,j]; log_q_z = logsumexp_j sum_d M, log_prod = sum_d logsumexp_j M, log_q_zx / log_p_z the sample's own and the
 * standard normal log density; out3 = {mi, tc, kld} = means of (log_q_zx - log_q_z, log_q_z - log_prod, log_prod - log_p_z).
 * lse_s [B], lse_d [B][D] keep the logsumexps for the backward call; g3: d loss / d {mi, tc, kld} (3 floats on the device).
 * ws: >= 16*B bytes. */
int ctvae_tc_forward(const float* z, const float* mu, const float* logvar, const float* log_iw, int B, int D, float* out3, float* lse_s,
                     float* lse_d, float* ws, size_t ws_bytes, void* stream);
int ctvae_tc_backward(const float* z, const float* mu, const float* logvar, const float* log_iw, const float* lse_s, const float* lse_d,
                      const float* g3, int B, int D, float* g_z, float* g_mu, float* g_logvar, void* stream);
/* VampPrior KL term of VampVAE (vampvae.py:140-171): z, mu, logvar [B][D] (contiguous), prior_mu / prior_logvar [K][D] (the
 * encoder on the K pseudo-inputs, K <= 512): out3 = {kld, E_log_p, E_log_q} with E_log_q = mean_b sum_d -0.5 (lv + (z-mu)^2) /
 * e^lv, E_log_p = mean_b logsumexp_k (sum_d -0.5 (plv_k + (z - pmu_k)^2) / e^plv_k - log K), kld = -(E_log_p - E_log_q);
 * weights [B][K] receives softmax_k of the component scores for the backward call (g_kld: one float on the device).
 * ws: >= 8*B bytes. */
int ctvae_vamp_kl_forward(const float* z, const float* mu, const float* logvar, const float* prior_mu, const float* prior_logvar,
                          int B, int D, int K, float* out3, float* weights, float* ws, size_t ws_bytes, void* stream);
int ctvae_vamp_kl_backward(const float* z, const float* mu, const float* logvar, const float* prior_mu, const float* prior_logvar,
                           const float* weights, const float* g_kld, int B, int D, int K, float* g_z, float* g_mu, float* g_logvar,
                           float* g_prior_mu, float* g_prior_logvar, void* stream);
/* Sliced Wasserstein distance of SWAE (swae.py:150-178) between z and prior draws [N][D] (N <= 1024, D <= 512, D % 4 == 0) along
 * the S unit directions proj [S][D]: out[0] = weight * mean_{s,r} (sort_r(z . w_s) - sort_r(prior . w_s))^p (torch.sort + pow + mean
 * in the reference); grad_z [N][D] receives d out / d z (the backward pass scales it by the incoming gradient).
 * ws: >= 4 * (S + N*S) bytes. */
int ctvae_swd_forward(const float* z, const float* prior, const float* proj, int N, int D, int S, float p, float weight, float* out,
                      float* grad_z, float* ws, size_t ws_bytes, void* stream);

/* Maximum-mean-discrepancy regulariser of WAE_MMD / InfoVAE (wae_mmd.py:120-203, info_vae.py:150-229) between the latent
 * codes z [N][D] and prior draws [N][D] (D <= 512):  out4 = {mmd, K(p,p), K(z,z), K(p,z)},
 * mmd = w_pp*K(p,p) + w_zz*K(z,z) - 2*w_pz*K(p,z);  kind 0 (imq): K(a,b) = sum_{i != j} c / (eps + c + |a_i - b_j|^2),
 * kind 1 (rbf): K(a,b) = mean_{i,j} exp(-mean_d (a_i - b_j)^2 / c);  c = 2 * D * latent_var is passed by the caller.
 * grad_z [N][D] receives d mmd / d z (the backward pass scales it by the incoming gradient).  ws: >= 12*N bytes. */
int ctvae_mmd_forward(const float* z, const float* prior, int N, int D, int kind, float c, float eps, float w_pp, float w_zz,
                      float w_pz, float* out4, float* grad_z, float* ws, size_t ws_bytes, void* stream);

/* DIP-VAE II regulariser (dip_vae.py:147-159) on the posterior parameters mu / logvar [B][D] (row strides in floats, D <= 1024):
 * c = mu - mean over the latent dimension, cov = c^T c, v = mean of the main diagonal of exp(2*logvar) (one scalar, as the
 * reference computes it), cz = cov + v, dip = lambda_offdiag * sum_{i != j} cz_ij^2 + lambda_diag * sum_i (cz_ii - 1)^2.
 * state: ctvae_dip_state_floats(B, D) floats kept by the caller between the two calls; dip = state[B*D + D*D + 3*D].
 * Backward writes dense g_mu / g_logvar [B][D] scaled by g_dip[0]. */
size_t ctvae_dip_state_floats(int B, int D);
int ctvae_dip_forward(const float* mu, long mu_row_stride, const float* logvar, long lv_row_stride, int B, int D,
                      float lambda_diag, float lambda_offdiag, float* state, void* stream);
int ctvae_dip_backward(const float* state, const float* g_dip, float* g_mu, float* g_logvar, int B, int D, void* stream);

/* MSSIMVAE's reconstruction loss (mssim_vae.py:182-279): 1 - prod_{i<4} (mcs_i^w_i * mssim_4^w_4) over five levels of SSIM with the
 * reference's 11-tap window (2x2 average pooling between levels), for NHWC pictures a (the reconstruction) and b [B,64,64,C].
 * window [11] and weights [5]: HOST arrays (the window as the reference builds it: exp(+(x-5)^2 / 4.5), normalised).
 * part: ctvae_mssim_part_floats(B, C) floats of scratch; loss [1]; coef [10]: d loss / d (element of the ssim / cs map) per
 * level, which ctvae_mssim_backward reads.  Backward: g_a [B,64,64,C] = g_loss[0] * d loss / d a. */
size_t ctvae_mssim_part_floats(int B, int C);
int ctvae_mssim_forward(const float* a, const float* b, const float* window, const float* weights, float* part, float* loss,
                        float* coef, int B, int C, int H, int W, void* stream);
int ctvae_mssim_backward(const float* a, const float* b, const float* window, const float* coef, const float* g_loss, float* g_a,
                         int B, int C, int H, int W, void* stream);

/* Deferred slab reductions.  A parameter gradient is not read before the optimizer step, so the finishing launch behind a
 * weight-gradient kernel (the deterministic reduction of its per-slice partial dW slabs) need not sit in the backward chain.
 * Between ctvae_defer_begin and ctvae_defer_flush (one deferral at a time per process; the calls in between may come from
 * another thread, as autograd's do), ctvae_conv_wgrad / ctvae_conv_backward write their slabs
 * into `arena` (each call behind the previous call's slabs; a call that finds less than its workspace size left, or whose
 * finishing launch carries a BatchNorm finalize, or that writes a gradient an earlier deferred call wrote, runs as usual)
 * and only record the reduction; ctvae_defer_flush issues all of them in one launch per 24 jobs.  dw / dbias of the deferred
 * calls are valid only after the flush.  `arena` must stay untouched in between. */
int ctvae_defer_begin(float* arena, size_t arena_bytes);
int ctvae_defer_flush(void* stream);

/* torch.optim.Adam step over one flat buffer (experiment.py:158-160).  state (device, ctvae_adam_state_floats() floats, 64-byte
 * aligned): [0..7] = {step, lr, beta1, beta2, eps, weight_decay, beta1^step, beta2^step}, the rest scratch (ticket counters),
 * all zero when the caller creates the state.  The call advances step inside its one launch: the workgroup that finishes
 * last writes the advanced state back. */
size_t ctvae_adam_state_floats(void);
int ctvae_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* state, long n,
                    float grad_scale, void* stream);

/* Input side: the reference's per-sample pipeline ToTensor -> CenterCrop(crop) -> Resize(size)
 * (dataset.py:72-80; bilinear, align_corners=False, no antialias -- what transforms.Resize does to a tensor; images smaller
 * than the crop are zero-padded as torchvision's center_crop does) for a batch of rows of a uint8 dataset
 * images[N][H][W][3] resident in HBM.  rows[B] int64 (out-of-range rows give zeros); out[B][size][size][3] fp32 NHWC. */
int ctvae_crop_resize_u8(const uint8_t* images, const int64_t* rows, float* out, int B, int N, int H, int W, int crop, int size,
                         void* stream);

/* 3x3 / stride-1 / pad-1 layers with at least 64 channels run Winograd F(2x2,3x3) (forward, data gradient) and
 * F(3x3,2x2) (weight and bias gradient) instead of the direct tap-GEMM (csrc/wino.hip): 2.25x fewer MFMA
 * operations, results within a few 1e-6 of the direct kernels.  This switch (default on; environment
 * CTVAE_NO_WINOGRAD=1 turns it off) selects the direct kernels for A/B parity checks.  Returns the previous setting. */
int ctvae_winograd_enable(int on);

/* Measurement aid (bench.py roofline leg): when enabled every launcher brackets its kernels with HIP events
 * on the launch stream.  ctvae_prof_report synchronises them and writes one line per kernel name
 * "name\tcount\ttotal_ms\talgorithmic_flops\talgorithmic_bytes"; returns the buffer size needed.  Must be off
 * during hipGraph capture. */
void ctvae_prof_enable(int on);
void ctvae_prof_calibrate(void* stream, int n); /* n empty event pairs, reported as "(empty event pair)" */
size_t ctvae_prof_report(char* buf, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* CTVAE_HIP_H */
