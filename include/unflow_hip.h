/*
 * unflow_hip.h -- C ABI of libunflow_hip.so: UnFlow's per-pair dense hot path
 * (FlowNetC correlation, backward/forward warping, soft ternary-census loss)
 * as hand-written HIP kernels for MI355X (gfx950).
 *
 * This is the drop-in boundary.  Each entry point replaces one launcher of the
 * reference (paths relative to the reference checkout); a TensorFlow OpKernel,
 * a ctypes stub or any other host binds exactly these symbols.
 *
 * Conventions
 *  - All tensors are dense float32 DEVICE buffers owned by the caller, in the
 *    reference's layouts: correlation NCHW (ops/correlation_op.cc:53-56), every
 *    other op NHWC (ops/backward_warp_op.cu.cc:144-147).  Flow channel 0 is the
 *    x displacement (ops/backward_warp_op.cu.cc:27-28).
 *  - `stream` is a hipStream_t (NULL = the default stream).  Work is enqueued
 *    asynchronously; no entry point synchronises, allocates device memory or
 *    keeps state between calls.  The library is re-entrant.  (The only thing
 *    remembered is, per device, that a kernel needing more than 64 KiB of LDS has
 *    been granted it -- hipFuncSetAttribute is a per-device setting.)
 *  - Return value: UNFLOW_OK, or an error code with a message retrievable
 *    through unflow_last_error() (thread local).  UNFLOW_ERR_INVALID_ARGUMENT
 *    messages are the reference's OP_REQUIRES strings.
 *  - Empty tensors (any extent 0) are accepted and launch nothing
 *    (ops/backward_warp_op.cu.cc:150).
 */
#ifndef UNFLOW_HIP_H_
#define UNFLOW_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNFLOW_OK 0
#define UNFLOW_ERR_INVALID_ARGUMENT 1 /* tf errors::InvalidArgument */
#define UNFLOW_ERR_LAUNCH 2           /* hipGetLastError() after a launch */

#define UNFLOW_ABI_VERSION 5

typedef void *unflow_stream_t; /* hipStream_t */

int unflow_abi_version(void);
const char *unflow_last_error(void);
/* "" for a production build; otherwise the UNFLOW_TUNE_* diagnostic knobs the kernels were compiled with
 * (tools/variants.sh: timing-only builds, some of which compute WRONG results).  The Python binding refuses
 * to load a library whose flags are not empty unless UNFLOW_ALLOW_DIAGNOSTIC_BUILD=1. */
const char *unflow_build_flags(void);

/* Measurement helper, not part of the reference's path: dst[0..bytes) = src[0..bytes) with a
 * hand-written 16-byte-per-lane streaming copy.  bench.py prices the HBM-bound kernels
 * (the warps) against the rate of this copy over >= 1 GiB, next to the nominal 8 TB/s. */
int unflow_stream_copy(const void *src, void *dst, size_t bytes, unflow_stream_t stream);

/* ---------------------------------------------------------------- correlation
 * Geometry of CorrelationState (ops/correlation_op.h:28-51).
 * dims[5] = {out_channels, out_height, out_width, padded_height, padded_width}.
 * Errors: "kernel_size must be odd" (ops/correlation_op.h:16-17),
 *         "Invalid correlation settings" (ops/correlation_op.cc:60-61). */
int unflow_correlation_geometry(int H, int W, int kernel_size, int max_displacement, int pad,
                                int stride_1, int stride_2, int *dims);

/* Replaces Correlation() (ops/correlation_op.cu.cc:250-315; op at
 * ops/correlation_op.cc:43-85).  in0,in1 [B,C,H,W]; out [B,D,Ho,Wo].
 * padded0/padded1 [B,H+2p,W+2p,C] are the op's outputs 1 and 2; pass NULL for
 * both to skip materialising them (the kernels never read them).
 * Speed: the matrix-core kernels serve the FlowNetC attributes only (kernel_size 1, stride_1 1,
 * stride_2 2, max_displacement = pad = 20: core/flownet.py:221-222, the only call site of the op
 * in the reference); every other attribute set is computed by one-lane-per-element kernels. */
int unflow_correlation_fwd(const float *in0, const float *in1, float *out, float *padded0,
                           float *padded1, int B, int C, int H, int W, int kernel_size,
                           int max_displacement, int pad, int stride_1, int stride_2,
                           unflow_stream_t stream);

/* Replaces CorrelationGrad() (ops/correlation_op.cu.cc:317-390; op at
 * ops/correlation_op.cc:92-129).  topdiff [B,D,Ho,Wo]; g0,g1 [B,C,H,W].
 * The reference reads the padded NHWC copies; this implementation reads the
 * NCHW inputs in0,in1 directly (padded0/padded1 are accepted for signature
 * parity and ignored; they may be NULL). */
int unflow_correlation_bwd(const float *topdiff, const float *in0, const float *in1,
                           const float *padded0, const float *padded1, float *g0, float *g1, int B,
                           int C, int H, int W, int kernel_size, int max_displacement, int pad,
                           int stride_1, int stride_2, unflow_stream_t stream);

/* ---------------------------------------------------------------- backward warp
 * Replaces BackwardWarp() / BackwardWarpGrad() (ops/backward_warp_op.cu.cc:140-190):
 * bilinear sampling at (x+u, y+v), taps outside the image contribute zero;
 * gradient wrt the flow only (ops.py:80-84 returns None for the image). */
int unflow_backward_warp_fwd(const float *images, const float *flows, float *output, int B, int H,
                             int W, int C, unflow_stream_t stream);
int unflow_backward_warp_bwd(const float *input_grad, const float *images, const float *flows,
                             float *output_grad, int B, int H, int W, int C, unflow_stream_t stream);

/* ---------------------------------------------------------------- image_warp
 * Replaces the TF graph of src/e2eflow/core/image_warp.py:4-76 (what the loss
 * really calls, losses.py:22-23,38-39): floor the flow, clamp the four taps to
 * the image.  bwd: d_im (scatter-add, zeroed by the callee) and d_flow; either
 * may be NULL. */
int unflow_image_warp_fwd(const float *im, const float *flow, float *warped, int B, int H, int W,
                          int C, unflow_stream_t stream);
int unflow_image_warp_bwd(const float *grad, const float *im, const float *flow, float *d_im,
                          float *d_flow, int B, int H, int W, int C, unflow_stream_t stream);

/* ---------------------------------------------------------------- forward warp
 * Replaces ForwardWarp() / ForwardWarpGrad() (ops/forward_warp_op.cu.cc:127-175):
 * Gaussian splat (sigma 1, window +-4) of unit mass; output [B,H,W,1] is zeroed
 * by the callee.  Summation order of the atomics is not deterministic. */
int unflow_forward_warp_fwd(const float *flows, float *output, int B, int H, int W,
                            unflow_stream_t stream);
int unflow_forward_warp_bwd(const float *input_grad, const float *flows, float *output_grad, int B,
                            int H, int W, unflow_stream_t stream);

/* ---------------------------------------------------------------- downsample
 * Replaces Downsample() (ops/downsample_op.cu.cc:51-72).  Error: "Input height
 * and width must be divisible by scale" (ops/downsample_op.cc:37-40). */
int unflow_downsample_fwd(const float *images, float *output, int B, int H, int W, int C, int scale,
                          unflow_stream_t stream);
/* The image chain of the loss pyramid (src/e2eflow/core/unsupervised.py:97-103,143-145: downsample by 4, then
 * by 2 per level) in one launch, bit-identical to the chain of unflow_downsample_fwd calls.  im_a, im_b [B,H,W,C]
 * (C = 1 or 3; im_b may be NULL); out_ab / out_ba: HOST arrays of `levels` (<= 5) device pointers, level l being
 * [B or 2B, H/(4<<l), W/(4<<l), C].  With im_b the outputs are stacked along the batch: out_ab[l] = [a; b] and
 * (if out_ba is given) out_ba[l] = [b; a].  H and W must be divisible by 4 << (levels - 1). */
int unflow_downsample_pyramid(const float *im_a, const float *im_b, float *const *out_ab, float *const *out_ba,
                              int levels, int B, int H, int W, int C, unflow_stream_t stream);

/* ---------------------------------------------------------------- census loss
 * Replaces ternary_loss (src/e2eflow/core/losses.py:90-122), the soft
 * real-valued census + "Hamming" distance + charbonnier reduction, fused.
 *   im1, im2_warped [B,H,W,3] in [0,1]; mask [B,H,W,1].
 *   loss:      device scalar, OVERWRITTEN with  sum / norm_count.
 *   dldd:      optional [B,H,W] map saved for the backward pass
 *              (mask' * d charbonnier / d dist / norm_count); may be NULL.
 *   dist:      optional [B,H,W,1] per-pixel distance (for tests); may be NULL.
 *   workspace: unflow_reduce_workspace_floats(B, H, W) floats (one partial sum per
 *              workgroup of the loss kernels: a few thousand, not one per pixel).
 *   norm_count: the charbonnier normalisation (losses.py:311-312); 0 means the
 *              local B*H*W, a batch shard passes the GLOBAL pixel count. */
size_t unflow_reduce_workspace_floats(int B, int H, int W);

/* Deferred reductions (launch-bound loss pyramids: core/unsupervised.py:114-141 evaluates ~40 scalar losses per
 * step).  Between unflow_reduce_plan_begin(plan) and unflow_reduce_plan_end() the loss forwards of THIS thread
 * (unflow_ternary_loss_fwd, unflow_ternary_loss_warp_fwd, unflow_sym_loss_fwd, unflow_charbonnier_fwd,
 * unflow_smoothness_loss_fwd, unflow_gradient_loss_fwd) do not launch their final one-block reduction and do not
 * write *loss: they append (workspace, number of partial sums, 1 / norm) to the caller-owned plan.
 * unflow_reduce_plan_run then reduces all segments in ONE launch (one block each, the same fixed order and double
 * accumulation as the immediate form: results[i] is bit-identical) and folds them in a second one-block launch:
 *   total = sum_i coeff[i] * results[i],   per_term[t] = sum_{i: term[i] == t} term_weight[i] * results[i]  (t < UNFLOW_PLAN_TERMS; a larger term[i] is an error,
 * a negative one belongs to no term).
 * coeff / term / term_weight are HOST arrays of plan->n entries (copied into the kernel arguments); results
 * (plan->n floats), total (1) and per_term (UNFLOW_PLAN_TERMS) are device buffers.  At most UNFLOW_PLAN_CAPACITY segments; the
 * workspaces must stay untouched until the run.  The plan is plain data: it can be built once while a HIP graph is
 * captured and is not needed for replays. */
#define UNFLOW_PLAN_CAPACITY 128
#define UNFLOW_PLAN_TERMS 8      /* per_term holds this many floats */
typedef struct {
  int n;
  const float *partial[UNFLOW_PLAN_CAPACITY];
  int count[UNFLOW_PLAN_CAPACITY];
  double inv_norm[UNFLOW_PLAN_CAPACITY];
} unflow_reduce_plan;
int unflow_reduce_plan_begin(unflow_reduce_plan *plan);
int unflow_reduce_plan_end(void);
int unflow_reduce_plan_run(const unflow_reduce_plan *plan, const float *coeff, const int *term,
                           const float *term_weight, float *results, float *total, float *per_term,
                           unflow_stream_t stream);
/* out[i] = sum_{k < n} srcs[k][i] for i < count: the gradient contributions of the n consumers of one tensor in ONE
 * launch (an autograd engine issues n - 1 adds).  srcs is a HOST array of n <= 8 device pointers. */
int unflow_sum_tensors(const float *const *srcs, int n, float *out, long long count, unflow_stream_t stream);
/* The flow scaling of a loss pyramid (src/e2eflow/core/unsupervised.py:118) for levels that run both directions as one
 * batch: F[l] = scale[l] * [fw[l]; bw[l]] and G[l] = scale[l] * [bw[l]; fw[l]] (concatenated along the batch) for n <= 8
 * levels in one launch, and its gradient d_fw[l] = scale[l] * (gF[l][first half] + gG[l][second half]),
 * d_bw[l] = scale[l] * (gF[l][second half] + gG[l][first half]) (gG, or single entries of it, may be NULL).  All pointer
 * and count / scale arguments are HOST arrays of n entries; count[l] = floats of ONE direction's [B,h,w,2] tensor. */
int unflow_scale_stack_fwd(const float *const *fw, const float *const *bw, float *const *F, float *const *G,
                           const long long *count, const float *scale, int n, unflow_stream_t stream);
int unflow_scale_stack_bwd(const float *const *gF, const float *const *gG, float *const *d_fw, float *const *d_bw,
                           const long long *count, const float *scale, int n, unflow_stream_t stream);

int unflow_ternary_loss_fwd(const float *im1, const float *im2_warped, const float *mask,
                            float *loss, float *dldd, float *dist, float *workspace, int B, int H,
                            int W, int max_distance, long long norm_count, unflow_stream_t stream);
/* Gradient wrt im2_warped (im1 is stop-gradient'd upstream, core/augment.py:53).
 * gout: device scalar upstream gradient (NULL = 1).  dldd: the map written by
 * unflow_ternary_loss_fwd for the same inputs and max_distance (it is zero within
 * max_distance of every image border, which the kernel relies on). */
int unflow_ternary_loss_bwd(const float *im1, const float *im2_warped, const float *dldd,
                            const float *gout, float *d_im2_warped, int B, int H, int W,
                            int max_distance, unflow_stream_t stream);

/* ternary_loss(im1, image_warp(im2, flow), mask) as one forward and one backward kernel
 * (losses.py:22,82-83 -> core/image_warp.py + losses.py:90-122), for max_distance 2 or 3 and
 * B*H*W < 2^29 (otherwise UNFLOW_ERR_INVALID_ARGUMENT: compose the separate entry points).
 * _fwd: im2 is the UNWARPED image; the warp is evaluated while the census tiles are staged,
 *   the warped RGB image is never materialised.  Outputs: loss, dldd [B,H,W] and the two gray
 *   planes gray1 = 255*gray(im1), gray2_warped = 255*gray(image_warp(im2, flow)) [B,H,W] that
 *   the backward reads.  workspace / norm_count as unflow_ternary_loss_fwd.
 * _bwd: d_flow [B,H,W,2] = d loss / d flow * gout; the [B,H,W,3] gradient image of
 *   unflow_ternary_loss_bwd is not materialised either. */
int unflow_ternary_loss_warp_fwd(const float *im1, const float *im2, const float *flow,
                                 const float *mask, float *loss, float *dldd, float *gray1,
                                 float *gray2_warped, float *workspace, int B, int H, int W,
                                 int max_distance, long long norm_count, unflow_stream_t stream);
int unflow_ternary_loss_warp_bwd(const float *gray1, const float *gray2_warped, const float *dldd,
                                 const float *gout, const float *im2, const float *flow,
                                 float *d_flow, int B, int H, int W, int max_distance,
                                 unflow_stream_t stream);

/* ---------------------------------------------------------------- masks, charbonnier
 * create_outgoing_mask (losses.py:347-366): mask [B,H,W,1]. */
int unflow_outgoing_mask(const float *flow, float *mask, int B, int H, int W, unflow_stream_t stream);
/* create_mask (losses.py:325-335): ones inside a zero border. */
int unflow_create_mask(float *mask, int B, int H, int W, int pad_top, int pad_bottom, int pad_left,
                       int pad_right, unflow_stream_t stream);
/* Forward-backward occlusion of compute_losses (losses.py:38-51):
 * flow_diff = flow + flow_other_warped; occ = |flow_diff|^2 > 0.01*(|flow|^2 +
 * |flow_other_warped|^2) + 0.5.  Either output may be NULL.  If mask_inout is
 * not NULL it is multiplied by (1 - occ) (losses.py:53-55). */
int unflow_fb_occlusion(const float *flow, const float *flow_other_warped, float *flow_diff,
                        float *occ, float *mask_inout, int B, int H, int W, unflow_stream_t stream);
/* The same with the mask product written somewhere else: mask_out = mask_in * (1 - occ) (both NULL or both given;
 * they may be the same buffer).  Saves the caller the copy of a shared border mask per flow direction and level. */
int unflow_fb_occlusion_masked(const float *flow, const float *flow_other_warped, float *flow_diff, float *occ,
                               const float *mask_in, float *mask_out, int B, int H, int W, unflow_stream_t stream);
/* The same pass (flow_diff, mask_in, mask_out required) with the two losses that need nothing else (losses.py:57-58,
 * 62-67):  loss_fb = charbonnier_loss(flow_diff, mask_out) / norm_fb  and  loss_occ = charbonnier_loss(1 - mask_out) /
 * norm_occ  (norm 0 = the element count).  workspace_*: unflow_reduce_workspace_floats(B, H, W) floats each; the scalars
 * follow the convention of the other loss entry points (NULL inside an open reduce plan, which gets two entries: fb, occ). */
int unflow_fb_occlusion_losses(const float *flow, const float *flow_other_warped, float *flow_diff, float *occ,
                               const float *mask_in, float *mask_out, float *loss_fb, float *workspace_fb,
                               long long norm_fb, float *loss_occ, float *workspace_occ, long long norm_occ, int B, int H,
                               int W, unflow_stream_t stream);
/* The same with flow_other_warped = image_warp(flow_other, flow) (losses.py:38-39) computed on
 * the fly; flow_other_warped (optional output), occ and mask_out may be NULL.
 * mask_out = mask_in * (1 - occ) (losses.py:53-55); mask_in may be the same buffer.
 * unwarped_magnitude != 0: the threshold uses |flow_other|^2 at the pixel itself instead of
 * |flow_other_warped|^2 -- the variant of occlusion() (losses.py:125-134). */
int unflow_fb_occlusion_warp(const float *flow, const float *flow_other, float *flow_other_warped,
                             float *occ, const float *mask_in, float *mask_out, int B, int H,
                             int W, int unwarped_magnitude, unflow_stream_t stream);
/* unflow_fb_occlusion_losses with flow_other_warped = image_warp(flow_other, flow) computed on the fly (never written):
 * warp, occlusion test, mask product and the two scalars in one launch. */
int unflow_fb_occlusion_warp_losses(const float *flow, const float *flow_other, float *flow_diff, float *occ,
                                    const float *mask_in, float *mask_out, float *loss_fb, float *workspace_fb,
                                    long long norm_fb, float *loss_occ, float *workspace_occ, long long norm_occ, int B,
                                    int H, int W, unflow_stream_t stream);
/* One direction of losses['sym'] (losses.py:27-29,57-61): charbonnier_loss((1 - mask) -
 * cast(density < thresh)), density = forward_warp of the other flow; mask, density
 * [B,H,W,1].  No gradient (both operands are thresholded maps).  norm_count 0 = B*H*W. */
int unflow_sym_loss_fwd(const float *mask, const float *density, float thresh, float *loss,
                        float *workspace, int B, int H, int W, long long norm_count,
                        unflow_stream_t stream);
/* charbonnier_loss (losses.py:298-322).  x [B,H,W,C]; mask NULL, [B,H,W,1]
 * (mask_channels 1) or [B,H,W,C].  loss overwritten with sum / norm_count
 * (0 = local B*H*W*C).  truncate <= 0 disables truncation. */
int unflow_charbonnier_fwd(const float *x, const float *mask, int mask_channels, float *loss,
                           float *workspace, int B, int H, int W, int C, float alpha, float beta,
                           float epsilon, float truncate, long long norm_count,
                           unflow_stream_t stream);
int unflow_charbonnier_bwd(const float *x, const float *mask, int mask_channels, const float *gout,
                           float *dx, int B, int H, int W, int C, float alpha, float beta,
                           float epsilon, float truncate, long long norm_count,
                           unflow_stream_t stream);

/* ---------------------------------------------------------------- 3x3 stencil terms
 * smoothness_loss (order 1, src/e2eflow/core/losses.py:206-225,250-255) and
 * second_order_loss (order 2, losses.py:258-295): charbonnier of the masked
 * first / second differences of both flow components, fused.  flow [B,H,W,2].
 *   loss: device scalar, overwritten with (loss_u + loss_v) =
 *         sum / norm_count; norm_count 0 = the local B*H*W*K, K = 2 (order 1)
 *         or 4 (order 2).  workspace as for the census loss.
 * _bwd: dflow [B,H,W,2] overwritten with d loss / d flow * gout (NULL = 1). */
int unflow_smoothness_loss_fwd(const float *flow, float *loss, float *workspace, int B, int H, int W,
                               int order, long long norm_count, unflow_stream_t stream);
int unflow_smoothness_loss_bwd(const float *flow, const float *gout, float *dflow, int B, int H, int W,
                               int order, long long norm_count, unflow_stream_t stream);
/* _smoothness_deltas / _second_order_deltas (losses.py:206-225,258-287):
 * delta_u, delta_v, mask [B,H,W,K]; any of them may be NULL. */
int unflow_smoothness_deltas(const float *flow, float *delta_u, float *delta_v, float *mask, int B,
                             int H, int W, int order, unflow_stream_t stream);
/* gradient_loss (losses.py:244-247): charbonnier of the masked difference of the
 * Sobel gradients of im1 and im2_warped ([B,H,W,3]); mask [B,H,W,1].
 * norm_count 0 = local B*H*W*6.  _bwd: gradient wrt im2_warped only (im1 carries
 * no gradient in the reference's graph). */
int unflow_gradient_loss_fwd(const float *im1, const float *im2_warped, const float *mask,
                             float *loss, float *workspace, int B, int H, int W,
                             long long norm_count, unflow_stream_t stream);
int unflow_gradient_loss_bwd(const float *im1, const float *im2_warped, const float *mask,
                             const float *gout, float *d_im2_warped, int B, int H, int W,
                             long long norm_count, unflow_stream_t stream);
/* _gradient_delta (losses.py:228-241): diff [B,H,W,6], channel 2c = d/dx, 2c+1 = d/dy of colour c. */
int unflow_gradient_delta(const float *im1, const float *im2_warped, float *diff, int B, int H, int W,
                          unflow_stream_t stream);

/* ---------------------------------------------------------------- both directions (extension)
 * A bidirectional FlowNetC runs correlation(a, b) and correlation(b, a) on the same pair
 * (core/flownet.py:33-43).  The second volume is a re-indexing of the first,
 *   corr(b, a)[d, p] = corr(a, b)[-d, p + stride_2 * d]   (0 where that leaves the image),
 * bit-identical because the same products are summed in the same order.
 * out = (add ? add : 0) + flip(in); in, add, out [B,D,H,W].  The map is its own inverse, so
 * top_fw + flip(top_bw) is the upstream gradient of the first volume that makes ONE
 * unflow_correlation_bwd call return the summed gradients of both directions.
 * Needs kernel_size 1, stride_1 1, pad == max_displacement; D (the number of displacement
 * planes of the volumes) must be the out_channels the attrs imply. */
int unflow_correlation_flip(const float *in, const float *add, float *out, int B, int D, int H, int W,
                            int kernel_size, int max_displacement, int pad, int stride_1,
                            int stride_2, unflow_stream_t stream);

/* ---------------------------------------------------------------- fp16 correlation (extension)
 * NOT in the reference (float32 only; BASELINE config 5 names "fp16 with fp32 accumulate").
 * Correlation of half-precision tensors on the f16 MATRIX CORES: in0, in1 [B,C,H,W] and out
 * [B,441,H,W] are IEEE binary16; v_mfma_f32_16x16x16_f16 multiplies halves (every product exact in
 * f32) and accumulates in f32; the result is scaled by 1/C and rounded to half once.  Against an
 * fp32 evaluation of the same half-rounded inputs only the f32 summation order and that one
 * rounding differ (rtol 2e-3).  Replaces, for half tensors, the same reference
 * functions as unflow_correlation_fwd / _bwd (ops/correlation_op.cu.cc:250-315, :317-390).
 * FlowNetC attributes only (kernel_size 1, max_displacement = pad = 20, stride_1 1,
 * stride_2 2), W % 4 == 0, 8-byte aligned tensors; otherwise UNFLOW_ERR_INVALID_ARGUMENT. */
int unflow_correlation_fwd_f16(const void *in0, const void *in1, void *out, int B, int C, int H, int W,
                               int kernel_size, int max_displacement, int pad, int stride_1,
                               int stride_2, unflow_stream_t stream);
/* Its gradients: topdiff [B,441,H,W], in0, in1, g0, g1 [B,C,H,W], all binary16; f16 MFMA with
 * f32 accumulation, one rounding to half per output element. */
int unflow_correlation_bwd_f16(const void *topdiff, const void *in0, const void *in1, void *g0,
                               void *g1, int B, int C, int H, int W, int kernel_size,
                               int max_displacement, int pad, int stride_1, int stride_2,
                               unflow_stream_t stream);
/* The same two operations with fp16 STORAGE only: the halves are widened to f32 on the way into
 * LDS and run through the float32 kernels' f32 MFMA loop, i.e. the k-ordered f32 fmaf chain of
 * unflow_correlation_fwd on the widened tensors, rounded to half once (bit-identical to that, up to
 * results on a rounding boundary).  Slower than the float32 op; kept as the exact form. */
int unflow_correlation_fwd_f16_exact(const void *in0, const void *in1, void *out, int B, int C, int H,
                                     int W, int kernel_size, int max_displacement, int pad,
                                     int stride_1, int stride_2, unflow_stream_t stream);
int unflow_correlation_bwd_f16_exact(const void *topdiff, const void *in0, const void *in1, void *g0,
                                     void *g1, int B, int C, int H, int W, int kernel_size,
                                     int max_displacement, int pad, int stride_1, int stride_2,
                                     unflow_stream_t stream);

/* The warps with fp16 storage (extension, BASELINE config 5: "forward_warp + full-resolution
 * image_warp, fp16 with fp32 accumulate").  Same layouts and semantics as unflow_backward_warp_*,
 * unflow_image_warp_* and unflow_forward_warp_fwd above with IEEE binary16 images, flows and
 * gradients in memory: every value is widened to f32 when loaded, the arithmetic is the f32
 * kernels' expression, the result is rounded to half once when stored.  1 <= C <= 3; all
 * tensors 4-byte aligned; otherwise UNFLOW_ERR_INVALID_ARGUMENT.  The image gradient of
 * image_warp has no fp16 form (its scatter-add accumulates in f32: use unflow_image_warp_bwd).
 * forward_warp takes half flows and writes its density in f32 (the accumulator of the splat). */
int unflow_backward_warp_fwd_f16(const void *images, const void *flows, void *output, int B, int H,
                                 int W, int C, unflow_stream_t stream);
int unflow_backward_warp_bwd_f16(const void *input_grad, const void *images, const void *flows,
                                 void *output_grad, int B, int H, int W, int C, unflow_stream_t stream);
int unflow_image_warp_fwd_f16(const void *im, const void *flow, void *warped, int B, int H, int W,
                              int C, unflow_stream_t stream);
int unflow_image_warp_bwd_flow_f16(const void *grad, const void *im, const void *flow, void *d_flow,
                                   int B, int H, int W, int C, unflow_stream_t stream);
int unflow_forward_warp_fwd_f16(const void *flows, float *output, int B, int H, int W,
                                unflow_stream_t stream);

/* ---------------------------------------------------------------- integer census (extension)
 * NOT in the reference (its census is the soft one above).  Classical ternary
 * census on I = 255*gray(im), window (2*max_distance+1)^2, zero padding:
 * signature [B,H,W,2] uint64 -- bit k of word 0: neighbour k (row-major, centre
 * skipped) brighter than the centre by more than eps; word 1: darker by more than
 * eps.  max_distance in [1,3].  Hamming: dist [B,H,W] int32 = number of neighbours
 * whose ternary digit differs, total = sum of dist where mask != 0 (mask NULL = all);
 * dist or total may be NULL.  Bit-exact against the CPU oracle. */
int unflow_census_transform(const float *im, unsigned long long *signature, int B, int H, int W,
                            int max_distance, float eps, unflow_stream_t stream);
int unflow_census_hamming(const unsigned long long *sig1, const unsigned long long *sig2,
                          const float *mask, int *dist, unsigned long long *total, int B, int H, int W,
                          unflow_stream_t stream);

/* ---------------------------------------------------------------- stacked-net warp block
 * The full-resolution input block of the refinement networks
 * (src/e2eflow/core/flownet.py:46-57):
 *   flow = tf.image.resize_bilinear(flow_lr, [H, W]) * mult   (mult = 4 * FLOW_SCALE)
 *   out  = concat([im1, im2, flow, image_warp(im2, flow), |warp - im1|])  [B,H,W,14]
 * im1, im2 [B,H,W,3]; flow_lr [B,h,w,2].  Forward only: the reference stops the
 * gradient here unless train_all is set (flownet.py:51-54); the differentiable
 * form is composed from unflow_resize_bilinear_* and unflow_image_warp_*. */
int unflow_stacked_inputs_fwd(const float *im1, const float *im2, const float *flow_lr, float *out,
                              int B, int H, int W, int h, int w, float mult, unflow_stream_t stream);
/* tf.image.resize_bilinear (TF 1.x, align_corners=False) times mult: in [B,h,w,C] ->
 * out [B,H,W,C]; _bwd: gin [B,h,w,C] overwritten with the gradient wrt in. */
int unflow_resize_bilinear_fwd(const float *in, float *out, int B, int h, int w, int H, int W, int C,
                               float mult, unflow_stream_t stream);
int unflow_resize_bilinear_bwd(const float *gout, float *gin, int B, int h, int w, int H, int W, int C,
                               float mult, unflow_stream_t stream);

/* tf.image.resize_area (TensorFlow 1.x, align_corners=False; third-party): the odd-extent
 * branch of core/util.downsample (src/e2eflow/core/util.py:23-28), which the loss pyramid
 * (core/unsupervised.py:97-99,143-145) reaches at levels whose height or width is odd.
 * in [B,H,W,C] -> out [B,out_h,out_w,C], box average with fractional coverage weights.
 * Not differentiable on the reference's path (inputs are images and masks). */
int unflow_resize_area_fwd(const float *in, float *out, int B, int H, int W, int C, int out_h,
                           int out_w, unflow_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UNFLOW_HIP_H_ */
