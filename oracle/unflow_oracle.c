/*
 * unflow_oracle.c -- CPU restatement of UnFlow's per-pair dense hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the timed CPU
 * baseline.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it; the product package (unflow_amd/) never does.
 *
 * Every function restates, loop for loop and in the reference's own order of
 * floating-point operations, one reference function (paths relative to the
 * reference checkout):
 *
 *   oracle_correlation_*      ops/correlation_op.h:36-51,
 *                             ops/correlation_op.cu.cc:30-49 (rearrange),
 *                             :51-117 (CorrelateData), :119-181 (Backward0),
 *                             :183-248 (Backward1)
 *   oracle_backward_warp_*    ops/backward_warp_op.cu.cc:14-68, :70-138
 *   oracle_forward_warp_*     ops/forward_warp_op.cu.cc:16-65, :67-125
 *   oracle_downsample         ops/downsample_op.cu.cc:15-49
 *   oracle_image_warp_*       src/e2eflow/core/image_warp.py:4-76 (+ the
 *                             gradients TF autodiff derives from it)
 *   oracle_ternary_loss_*     src/e2eflow/core/losses.py:90-122 (soft census)
 *   oracle_charbonnier_*      src/e2eflow/core/losses.py:298-322
 *   oracle_smoothness_*       src/e2eflow/core/losses.py:206-225,250-295
 *                             (_smoothness_deltas / _second_order_deltas,
 *                             smoothness_loss / second_order_loss)
 *   oracle_gradient_*         src/e2eflow/core/losses.py:228-247
 *                             (_gradient_delta, gradient_loss)
 *   oracle_stacked_inputs     src/e2eflow/core/flownet.py:46-57 (resize_bilinear
 *                             -> image_warp -> abs -> concat); tf.image.resize_bilinear
 *                             itself is third-party: TF 1.x align_corners=False
 *                             semantics restated (assumption A2, PARITY UNPINNED)
 *   oracle_resize_area_fwd    src/e2eflow/core/util.py:23-28 (odd-extent branch of
 *                             downsample); tf.image.resize_area is third-party:
 *                             TF 1.x semantics restated (assumption A3, PARITY UNPINNED)
 *   oracle_create_*_mask      src/e2eflow/core/losses.py:325-366
 *   oracle_fb_occlusion       src/e2eflow/core/losses.py:38-51
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   pinned by the reference's own known-answer tests (tests/golden/reference_kat.json):
 *     backward_warp fwd, image_warp fwd, correlation (k=1, md=0), downsample,
 *     create_outgoing_mask, _smoothness_deltas (+ mask), gradient_loss.
 *   PARITY UNPINNED (no reference fixture exists; pinned only by formula, an
 *   independent numpy restatement and numeric-gradient self-consistency):
 *     soft census / ternary_loss, forward_warp forward values, correlation
 *     with max_displacement>0 / stride_2>1 / C>1 and its gradient.
 *
 * Arithmetic notes.  CUDA contracts `s += a*b` into an FMA, so the restated
 * CUDA kernels use fmaf() where the reference source has that shape; the
 * TF-composed functions (image_warp, census) round every product, so they are
 * written with separate operations.  Build with -ffp-contract=off so the
 * compiler adds no contraction of its own.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_OK 0
#define ORACLE_ERR_KERNEL_SIZE_EVEN 1
#define ORACLE_ERR_SHAPES 2
#define ORACLE_ERR_SETTINGS 3
#define ORACLE_ERR_DOWNSAMPLE 4
#define ORACLE_ERR_ARG 5

typedef struct {
  int pad_size, stride_1, stride_2, kernel_radius, max_displacement, kernel_size;
  int grid_radius, grid_width, padded_height, padded_width, border_size;
  int out_height, out_width, out_channels;
} corr_state;

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ops/correlation_op.h:28-51 */
static int corr_state_init(corr_state *st, int in_height, int in_width, int kernel_size,
                           int max_displacement, int pad, int stride_1, int stride_2) {
  if (kernel_size % 2 == 0) return ORACLE_ERR_KERNEL_SIZE_EVEN; /* :16-17 */
  if (stride_1 <= 0 || stride_2 <= 0 || kernel_size <= 0 || max_displacement < 0 || pad < 0)
    return ORACLE_ERR_ARG;
  st->pad_size = pad;
  st->stride_1 = stride_1;
  st->stride_2 = stride_2;
  st->max_displacement = max_displacement;
  st->kernel_size = kernel_size;
  st->padded_height = in_height + 2 * pad;
  st->padded_width = in_width + 2 * pad;
  st->kernel_radius = (kernel_size - 1) / 2;
  st->border_size = max_displacement + st->kernel_radius;
  st->grid_radius = max_displacement / stride_2;
  st->grid_width = st->grid_radius * 2 + 1;
  st->out_width = (int)ceilf((float)(st->padded_width - st->border_size * 2) / (float)stride_1);
  st->out_height = (int)ceilf((float)(st->padded_height - st->border_size * 2) / (float)stride_1);
  st->out_channels = st->grid_width * st->grid_width;
  /* ops/correlation_op.cc:60-61 tests the product only; two negative extents
     would pass it and then fail at allocation, so both signs are tested. */
  if (st->out_width <= 0 || st->out_height <= 0) return ORACLE_ERR_SETTINGS;
  return ORACLE_OK;
}

/* dims[0..4] = out_channels, out_height, out_width, padded_height, padded_width */
int oracle_correlation_geometry(int H, int W, int kernel_size, int max_displacement, int pad,
                                int stride_1, int stride_2, int *dims) {
  corr_state st;
  int rc = corr_state_init(&st, H, W, kernel_size, max_displacement, pad, stride_1, stride_2);
  if (rc) return rc;
  dims[0] = st.out_channels;
  dims[1] = st.out_height;
  dims[2] = st.out_width;
  dims[3] = st.padded_height;
  dims[4] = st.padded_width;
  return ORACLE_OK;
}

/* ops/correlation_op.cu.cc:30-49 + the cudaMemset at :282-283.
   NCHW [B,C,H,W] -> zero padded NHWC [B,H+2p,W+2p,C]. */
static void blob_rearrange(const float *in, float *out, int num, int channels, int width,
                           int height, int padding) {
  const int widthheight = width * height;
  const int pwidth = width + 2 * padding;
  const size_t pwidthheight = (size_t)pwidth * (height + 2 * padding);
  memset(out, 0, sizeof(float) * (size_t)num * pwidthheight * channels);
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < num; n++) {
    for (int ch = 0; ch < channels; ch++) {
      for (int xy = 0; xy < widthheight; xy++) {
        float value = in[((size_t)n * channels + ch) * widthheight + xy];
        int xpad = (xy % width + padding);
        int ypad = (xy / width + padding);
        size_t xypad = (size_t)ypad * pwidth + xpad;
        out[((size_t)n * pwidthheight + xypad) * channels + ch] = value;
      }
    }
  }
}

#define LANES 32 /* WARPS_PER_BLOCK*THREADS_PER_WARP, ops/correlation_op.cu.cc:21-22 */

/* ops/correlation_op.cu.cc:51-117.  One CUDA block (32 lanes) per output pixel:
   lane l accumulates the channels ch = l, l+32, ... of every kernel tap into
   sum[l] (an FMA chain in j,i,ch order), then lane 0 adds sum[0..31] serially. */
static void correlate_data(const corr_state *st, int num, int bottomchannels,
                           const float *bottom0, const float *bottom1, float *top) {
  const int topwidth = st->out_width, topheight = st->out_height, topchannels = st->out_channels;
  const int bottomwidth = st->padded_width, bottomheight = st->padded_height;
  const int kernel_size = st->kernel_size, stride1 = st->stride_1, stride2 = st->stride_2;
  const int max_displacement = st->max_displacement;
  const int ngr = st->grid_radius, ngw = st->grid_width;
  const size_t topcount = (size_t)topwidth * topheight * topchannels;
  const int sumelems = kernel_size * kernel_size * bottomchannels;

#pragma omp parallel for collapse(2) schedule(dynamic, 4)
  for (int item = 0; item < num; item++) {
    for (int by = 0; by < topheight; by++) {
      for (int bx = 0; bx < topwidth; bx++) {
        const int x1 = bx * stride1 + max_displacement;
        const int y1 = by * stride1 + max_displacement;
        for (int top_channel = 0; top_channel < topchannels; top_channel++) {
          float sum[LANES];
          for (int l = 0; l < LANES; l++) sum[l] = 0.0f;
          const int s2o = (top_channel % ngw - ngr) * stride2;
          const int s2p = (top_channel / ngw - ngr) * stride2;
          const int x2 = x1 + s2o;
          const int y2 = y1 + s2p;
          for (int j = 0; j < kernel_size; j++) {
            for (int i = 0; i < kernel_size; i++) {
              const float *p0 =
                  bottom0 + (((size_t)item * bottomheight + y1 + j) * bottomwidth + x1 + i) * bottomchannels;
              const float *p1 =
                  bottom1 + (((size_t)item * bottomheight + y2 + j) * bottomwidth + x2 + i) * bottomchannels;
              int ch = 0;
              for (; ch + LANES <= bottomchannels; ch += LANES)
                for (int l = 0; l < LANES; l++) sum[l] = fmaf(p0[ch + l], p1[ch + l], sum[l]);
              for (int l = 0; ch + l < bottomchannels; l++) sum[l] = fmaf(p0[ch + l], p1[ch + l], sum[l]);
            }
          }
          float total_sum = 0.0f;
          for (int idx = 0; idx < LANES; idx++) total_sum += sum[idx];
          const size_t index = ((size_t)top_channel * topheight + by) * topwidth + bx;
          top[index + item * topcount] = total_sum / (float)sumelems;
        }
      }
    }
  }
}

/* Full forward op as ops/correlation_op.cc:43-85 + ops/correlation_op.cu.cc:250-315.
   in0,in1 NCHW [B,C,H,W]; out NCHW [B,D,Ho,Wo]; padded0/1 NHWC [B,H+2p,W+2p,C]. */
int oracle_correlation_fwd(const float *in0, const float *in1, float *out, float *padded0,
                           float *padded1, int B, int C, int H, int W, int kernel_size,
                           int max_displacement, int pad, int stride_1, int stride_2) {
  corr_state st;
  int rc = corr_state_init(&st, H, W, kernel_size, max_displacement, pad, stride_1, stride_2);
  if (rc) return rc;
  if (B == 0 || C == 0) return ORACLE_OK;
  blob_rearrange(in0, padded0, B, C, W, H, pad);
  blob_rearrange(in1, padded1, B, C, W, H, pad);
  correlate_data(&st, B, C, padded0, padded1, out);
  return ORACLE_OK;
}

#define ROUND_OFF 50000 /* ops/correlation_op.cu.cc:19 */

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* ops/correlation_op.cu.cc:119-181 and :183-248, launched per item at :366-388.
   topdiff NCHW [B,D,Ho,Wo]; padded0/1 NHWC; g0,g1 NCHW [B,C,H,W]. */
int oracle_correlation_bwd(const float *topdiff, const float *padded0, const float *padded1,
                           float *g0, float *g1, int B, int C, int H, int W, int kernel_size,
                           int max_displacement, int pad, int stride_1, int stride_2) {
  corr_state st;
  int rc = corr_state_init(&st, H, W, kernel_size, max_displacement, pad, stride_1, stride_2);
  if (rc) return rc;
  const int topwidth = st.out_width, topheight = st.out_height, topchannels = st.out_channels;
  const int kernel_radius = st.kernel_radius, stride1 = stride_1, stride2 = stride_2;
  const int ngr = st.grid_radius, ngw = st.grid_width;
  const int bottomwidth = W, bottomheight = H, bottomchannels = C;
  const int pbottomwidth = st.padded_width, pbottomheight = st.padded_height;
  const size_t bottomcount = (size_t)C * H * W;
  const int pad_size = pad;
  const int round_off = ROUND_OFF;
  const int round_off_s1 = stride1 * round_off;
  const int sumelems = (kernel_radius * 2 + 1) * (kernel_radius * 2 + 1) * bottomchannels;

#pragma omp parallel for collapse(2) schedule(dynamic, 2)
  for (int item = 0; item < B; item++) {
    for (int my = 0; my < bottomheight; my++) {
      for (int lx = 0; lx < bottomwidth; lx++) {
        const int l = lx + pad_size;
        const int m = my + pad_size;
        for (int n = 0; n < bottomchannels; n++) {
          /* ---- Backward0 (:126-178) ---- */
          {
            int xmin = (l - 2 * kernel_radius - max_displacement + round_off_s1 - 1) / stride1 + 1 - round_off;
            int ymin = (m - 2 * kernel_radius - max_displacement + round_off_s1 - 1) / stride1 + 1 - round_off;
            int xmax = (l - max_displacement + round_off_s1) / stride1 - round_off;
            int ymax = (m - max_displacement + round_off_s1) / stride1 - round_off;
            float sum = 0.0f;
            if (xmax >= 0 && ymax >= 0 && (xmin <= topwidth - 1) && (ymin <= topheight - 1)) {
              xmin = imax(0, xmin);
              xmax = imin(topwidth - 1, xmax);
              ymin = imax(0, ymin);
              ymax = imin(topheight - 1, ymax);
              for (int p = -ngr; p <= ngr; p++) {
                for (int o = -ngr; o <= ngr; o++) {
                  int s2o = stride2 * o;
                  int s2p = stride2 * p;
                  size_t idxbot1 =
                      (((size_t)item * pbottomheight + (m + s2p)) * pbottomwidth + (l + s2o)) * bottomchannels + n;
                  float bot1tmp = padded1[idxbot1];
                  int op = (p + ngr) * ngw + (o + ngr);
                  size_t idxopoffset = ((size_t)item * topchannels + op);
                  for (int y = ymin; y <= ymax; y++)
                    for (int x = xmin; x <= xmax; x++) {
                      size_t idxtopdiff = (idxopoffset * topheight + y) * topwidth + x;
                      sum = fmaf(topdiff[idxtopdiff], bot1tmp, sum);
                    }
                }
              }
            }
            const size_t bot0index = ((size_t)n * bottomheight + (m - pad_size)) * bottomwidth + (l - pad_size);
            g0[bot0index + item * bottomcount] = sum / (float)sumelems;
          }
          /* ---- Backward1 (:193-245) ---- */
          {
            float sum = 0.0f;
            for (int p = -ngr; p <= ngr; p++) {
              for (int o = -ngr; o <= ngr; o++) {
                int s2o = stride2 * o;
                int s2p = stride2 * p;
                int xmin = (l - 2 * kernel_radius - max_displacement - s2o + round_off_s1 - 1) / stride1 + 1 - round_off;
                int ymin = (m - 2 * kernel_radius - max_displacement - s2p + round_off_s1 - 1) / stride1 + 1 - round_off;
                int xmax = (l - max_displacement - s2o + round_off_s1) / stride1 - round_off;
                int ymax = (m - max_displacement - s2p + round_off_s1) / stride1 - round_off;
                if (xmax >= 0 && ymax >= 0 && (xmin <= topwidth - 1) && (ymin <= topheight - 1)) {
                  xmin = imax(0, xmin);
                  xmax = imin(topwidth - 1, xmax);
                  ymin = imax(0, ymin);
                  ymax = imin(topheight - 1, ymax);
                  size_t idxbot0 =
                      (((size_t)item * pbottomheight + (m - s2p)) * pbottomwidth + (l - s2o)) * bottomchannels + n;
                  float bot0tmp = padded0[idxbot0];
                  int op = (p + ngr) * ngw + (o + ngr);
                  size_t idxOpOffset = ((size_t)item * topchannels + op);
                  for (int y = ymin; y <= ymax; y++)
                    for (int x = xmin; x <= xmax; x++) {
                      size_t idxtopdiff = (idxOpOffset * topheight + y) * topwidth + x;
                      sum = fmaf(topdiff[idxtopdiff], bot0tmp, sum);
                    }
                }
              }
            }
            const size_t bot1index = ((size_t)n * bottomheight + (m - pad_size)) * bottomwidth + (l - pad_size);
            g1[bot1index + item * bottomcount] = sum / (float)sumelems;
          }
        }
      }
    }
  }
  return ORACLE_OK;
}

/* ---------------------------------------------------------------------------
 * BackwardWarp: ops/backward_warp_op.cu.cc:14-68 (forward), :70-138 (grad).
 * images NHWC [B,H,W,C], flows [B,H,W,2] (u = x displacement first).
 * ------------------------------------------------------------------------- */
void oracle_backward_warp_fwd(const float *images, const float *flows, float *output, int batch,
                              int height, int width, int channels) {
  const long total = (long)batch * height * width;
#pragma omp parallel for schedule(static)
  for (long out_idx = 0; out_idx < total; out_idx++) {
    long idx = out_idx;
    const int src_x = (int)(idx % width);
    idx /= width;
    const int src_y = (int)(idx % height);
    const int b = (int)(idx / height);
    const long flow_index = out_idx * 2;
    const float x = src_x + flows[flow_index];
    const float y = src_y + flows[flow_index + 1];
    const int x0 = (int)floorf(x);
    const int x1 = x0 + 1;
    const int y0 = (int)floorf(y);
    const int y1 = y0 + 1;
    const float w_right = x - x0;
    const float w_left = x1 - x;
    const float w_bottom = y - y0;
    const float w_top = y1 - y;
    for (int c = 0; c < channels; ++c) {
      float sum = 0.0f;
#define IMG(iy, ix) images[c + (size_t)channels * (ix + (size_t)width * (iy + (size_t)height * b))]
      if (x0 >= 0 && x0 < width && y0 >= 0 && y0 < height) sum = fmaf(w_left * w_top, IMG(y0, x0), sum);
      if (x1 >= 0 && x1 < width && y0 >= 0 && y0 < height) sum = fmaf(w_right * w_top, IMG(y0, x1), sum);
      if (x0 >= 0 && x0 < width && y1 >= 0 && y1 < height) sum = fmaf(w_left * w_bottom, IMG(y1, x0), sum);
      if (x1 >= 0 && x1 < width && y1 >= 0 && y1 < height) sum = fmaf(w_right * w_bottom, IMG(y1, x1), sum);
#undef IMG
      output[out_idx * channels + c] = sum;
    }
  }
}

void oracle_backward_warp_bwd(const float *input_grad, const float *input_images, const float *flows,
                              float *output_grad, int batch, int height, int width, int channels) {
  const long total = (long)batch * height * width;
#pragma omp parallel for schedule(static)
  for (long in_idx = 0; in_idx < total; in_idx++) {
    long idx = in_idx;
    const int src_x = (int)(idx % width);
    idx /= width;
    const int src_y = (int)(idx % height);
    const int b = (int)(idx / height);
    const long flow_index = in_idx * 2;
    const float x = src_x + flows[flow_index];
    const float y = src_y + flows[flow_index + 1];
    const int x0 = (int)floorf(x);
    const int x1 = x0 + 1;
    const int y0 = (int)floorf(y);
    const int y1 = y0 + 1;
    const float w_right = x - x0;
    const float w_left = x1 - x;
    const float w_bottom = y - y0;
    const float w_top = y1 - y;
    float du = 0.0f, dv = 0.0f;
    for (int c = 0; c < channels; ++c) {
      float px;
      float din = input_grad[c + (size_t)channels * in_idx];
#define IMG(iy, ix) input_images[c + (size_t)channels * (ix + (size_t)width * (iy + (size_t)height * b))]
      if (x0 >= 0 && x0 < width && y0 >= 0 && y0 < height) {
        px = IMG(y0, x0) * din;
        du = fmaf(-w_top, px, du);
        dv = fmaf(-w_left, px, dv);
      }
      if (x1 >= 0 && x1 < width && y0 >= 0 && y0 < height) {
        px = IMG(y0, x1) * din;
        du = fmaf(w_top, px, du);
        dv = fmaf(-w_right, px, dv);
      }
      if (x0 >= 0 && x0 < width && y1 >= 0 && y1 < height) {
        px = IMG(y1, x0) * din;
        du = fmaf(-w_bottom, px, du);
        dv = fmaf(w_left, px, dv);
      }
      if (x1 >= 0 && x1 < width && y1 >= 0 && y1 < height) {
        px = IMG(y1, x1) * din;
        du = fmaf(w_bottom, px, du);
        dv = fmaf(w_right, px, dv);
      }
#undef IMG
    }
    output_grad[in_idx * 2] = du;
    output_grad[in_idx * 2 + 1] = dv;
  }
}

/* ---------------------------------------------------------------------------
 * ForwardWarp: ops/forward_warp_op.cu.cc:16-65 (Gaussian splat, sigma=1,
 * window +-4) and :67-125 (gradient wrt flow).  The CUDA kernel's atomicAdd
 * order is nondeterministic; the restatement adds in source-pixel order.
 * ------------------------------------------------------------------------- */
void oracle_forward_warp_fwd(const float *flows, float *output, int batch, int height, int width) {
  const long total = (long)batch * height * width;
  memset(output, 0, sizeof(float) * (size_t)total);
  /* items are independent; within one item the adds stay in source order */
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; b++) {
    for (int src_y = 0; src_y < height; src_y++)
      for (int src_x = 0; src_x < width; src_x++) {
        const long out_idx = src_x + (long)width * (src_y + (long)height * b);
        const long flow_index = out_idx * 2;
        const float target_x = src_x + flows[flow_index];
        const float target_y = src_y + flows[flow_index + 1];
        const float dist = 2.0f;
        const float std = dist * 0.5f;
        const int k = (int)ceilf(dist + 2);
        if (floorf(target_x - k) < width && floorf(target_x + k) >= 0 && floorf(target_y - k) < height &&
            floorf(target_y + k) >= 0) {
          const int min_n_x = target_x - k > 0 ? (int)floorf(target_x - k) : 0;
          const int min_n_y = target_y - k > 0 ? (int)floorf(target_y - k) : 0;
          const int max_n_x = target_x + k < width ? (int)floorf(target_x + k) : width - 1;
          const int max_n_y = target_y + k < height ? (int)floorf(target_y + k) : height - 1;
          const float gauss_divisor = 2 * powf(std, 2);
          for (int n_x = min_n_x; n_x <= max_n_x; ++n_x)
            for (int n_y = min_n_y; n_y <= max_n_y; ++n_y) {
              const float x = n_x - target_x;
              const float y = n_y - target_y;
              const float weight = expf(-(powf(x, 2) + powf(y, 2)) / gauss_divisor);
              output[n_x + (size_t)width * (n_y + (size_t)height * b)] += weight;
            }
        }
      }
  }
}

void oracle_forward_warp_bwd(const float *input_grad, const float *flows, float *output_grad, int batch,
                             int height, int width) {
  const long total = (long)batch * height * width;
#pragma omp parallel for schedule(static)
  for (long in_idx = 0; in_idx < total; in_idx++) {
    long idx = in_idx;
    const int src_x = (int)(idx % width);
    idx /= width;
    const int src_y = (int)(idx % height);
    const int b = (int)(idx / height);
    const long flow_index = in_idx * 2;
    const float target_x = src_x + flows[flow_index];
    const float target_y = src_y + flows[flow_index + 1];
    const float dist = 2.0f;
    const float std = dist * 0.5f;
    const int k = (int)ceilf(dist + 2);
    float du = 0.0f, dv = 0.0f;
    if (floorf(target_x - k) < width && floorf(target_x + k) >= 0 && floorf(target_y - k) < height &&
        floorf(target_y + k) >= 0) {
      const int min_n_x = target_x - k > 0 ? (int)floorf(target_x - k) : 0;
      const int min_n_y = target_y - k > 0 ? (int)floorf(target_y - k) : 0;
      const int max_n_x = target_x + k < width ? (int)floorf(target_x + k) : width - 1;
      const int max_n_y = target_y + k < height ? (int)floorf(target_y + k) : height - 1;
      const float gauss_divisor = 2 * powf(std, 2);
      for (int n_x = min_n_x; n_x <= max_n_x; ++n_x)
        for (int n_y = min_n_y; n_y <= max_n_y; ++n_y) {
          const float x = n_x - target_x;
          const float y = n_y - target_y;
          const float weight = expf(-(powf(x, 2) + powf(y, 2)) / gauss_divisor);
          const float din = input_grad[n_x + (size_t)width * (n_y + (size_t)height * b)];
          const float factor = 2 * din * weight / gauss_divisor;
          du = fmaf(factor, x, du);
          dv = fmaf(factor, y, dv);
        }
    }
    output_grad[flow_index] = du;
    output_grad[flow_index + 1] = dv;
  }
}

/* ops/downsample_op.cu.cc:15-49 (+ the divisibility check ops/downsample_op.cc:37-40) */
int oracle_downsample(const float *images, float *output, int batch, int in_height, int in_width,
                      int channels, int scale) {
  if (scale <= 0) return ORACLE_ERR_ARG;
  if (in_height % scale != 0 || in_width % scale != 0) return ORACLE_ERR_DOWNSAMPLE;
  const int out_height = in_height / scale, out_width = in_width / scale;
  const long total = (long)batch * out_height * out_width * channels;
#pragma omp parallel for schedule(static)
  for (long out_idx = 0; out_idx < total; out_idx++) {
    long idx = out_idx;
    const int c = (int)(idx % channels);
    idx /= channels;
    const int x = (int)(idx % out_width);
    idx /= out_width;
    const int y = (int)(idx % out_height);
    const int b = (int)(idx / out_height);
    const int scale_y = in_height / out_height;
    const int scale_x = in_width / out_width;
    const int min_in_y = y * scale_y, min_in_x = x * scale_x;
    const int max_in_y = min_in_y + scale_y, max_in_x = min_in_x + scale_x;
    float sum = 0.0f;
    for (int in_y = min_in_y; in_y < max_in_y; ++in_y)
      for (int in_x = min_in_x; in_x < max_in_x; ++in_x)
        sum += images[c + (size_t)channels * (in_x + (size_t)in_width * (in_y + (size_t)in_height * b))];
    sum /= scale_x * scale_y;
    output[c + (size_t)channels * (x + (size_t)out_width * (y + (size_t)out_height * b))] = sum;
  }
  return ORACLE_OK;
}

/* ---------------------------------------------------------------------------
 * image_warp: src/e2eflow/core/image_warp.py:4-76.  The FLOW is floored
 * (:26-27), integer base positions are added (:46-49), indices are clamped to
 * the image (:51-54), and the four taps are combined by add_n (:73) of
 * separately rounded products.
 * ------------------------------------------------------------------------- */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void oracle_image_warp_fwd(const float *im, const float *flow, float *warped, int num_batch, int height,
                           int width, int channels) {
  const int max_x = width - 1, max_y = height - 1;
  const long total = (long)num_batch * height * width;
#pragma omp parallel for schedule(static)
  for (long p = 0; p < total; p++) {
    const int pos_x = (int)(p % width);
    const int pos_y = (int)((p / width) % height);
    const long base = (p / ((long)width * height)) * ((long)width * height);
    const float fx = flow[2 * p], fy = flow[2 * p + 1];
    const float flx = floorf(fx), fly = floorf(fy);
    const int x = (int)flx, y = (int)fly;
    const float xw = fx - flx, yw = fy - fly;
    const float wa = (1 - xw) * (1 - yw);
    const float wb = (1 - xw) * yw;
    const float wc = xw * (1 - yw);
    const float wd = xw * yw;
    int x0 = pos_x + x, x1 = x0 + 1, y0 = pos_y + y, y1 = y0 + 1;
    x0 = clampi(x0, 0, max_x);
    x1 = clampi(x1, 0, max_x);
    y0 = clampi(y0, 0, max_y);
    y1 = clampi(y1, 0, max_y);
    const long idx_a = base + (long)y0 * width + x0;
    const long idx_b = base + (long)y1 * width + x0;
    const long idx_c = base + (long)y0 * width + x1;
    const long idx_d = base + (long)y1 * width + x1;
    for (int c = 0; c < channels; c++) {
      const float Ia = im[idx_a * channels + c], Ib = im[idx_b * channels + c];
      const float Ic = im[idx_c * channels + c], Id = im[idx_d * channels + c];
      float s = wa * Ia;
      s = s + wb * Ib;
      s = s + wc * Ic;
      s = s + wd * Id;
      warped[p * channels + c] = s;
    }
  }
}

/* Gradients TF autodiff derives from image_warp.py: the image receives a
   scatter-add of w*g at the four clamped taps (gradient of tf.gather); the
   flow receives d/d(xw), d/d(yw) of the weights (floor has zero gradient).
   d_im and d_flow may each be NULL. */
void oracle_image_warp_bwd(const float *grad, const float *im, const float *flow, float *d_im,
                           float *d_flow, int num_batch, int height, int width, int channels) {
  const int max_x = width - 1, max_y = height - 1;
  const long hw = (long)width * height;
  if (d_im) memset(d_im, 0, sizeof(float) * (size_t)num_batch * hw * channels);
#pragma omp parallel for schedule(static)
  for (int b = 0; b < num_batch; b++) {
    for (long q = 0; q < hw; q++) {
      const long p = b * hw + q;
      const int pos_x = (int)(q % width);
      const int pos_y = (int)(q / width);
      const long base = b * hw;
      const float fx = flow[2 * p], fy = flow[2 * p + 1];
      const float flx = floorf(fx), fly = floorf(fy);
      const int x = (int)flx, y = (int)fly;
      const float xw = fx - flx, yw = fy - fly;
      const float wa = (1 - xw) * (1 - yw);
      const float wb = (1 - xw) * yw;
      const float wc = xw * (1 - yw);
      const float wd = xw * yw;
      int x0 = pos_x + x, x1 = x0 + 1, y0 = pos_y + y, y1 = y0 + 1;
      x0 = clampi(x0, 0, max_x);
      x1 = clampi(x1, 0, max_x);
      y0 = clampi(y0, 0, max_y);
      y1 = clampi(y1, 0, max_y);
      const long idx_a = base + (long)y0 * width + x0;
      const long idx_b = base + (long)y1 * width + x0;
      const long idx_c = base + (long)y0 * width + x1;
      const long idx_d = base + (long)y1 * width + x1;
      float ga = 0.f, gb = 0.f, gc = 0.f, gd = 0.f; /* d/dw_a.. = sum_c g*I */
      for (int c = 0; c < channels; c++) {
        const float g = grad[p * channels + c];
        if (d_im) {
          d_im[idx_a * channels + c] += wa * g;
          d_im[idx_b * channels + c] += wb * g;
          d_im[idx_c * channels + c] += wc * g;
          d_im[idx_d * channels + c] += wd * g;
        }
        ga += g * im[idx_a * channels + c];
        gb += g * im[idx_b * channels + c];
        gc += g * im[idx_c * channels + c];
        gd += g * im[idx_d * channels + c];
      }
      if (d_flow) {
        /* wa=(1-xw)(1-yw) wb=(1-xw)yw wc=xw(1-yw) wd=xw*yw */
        const float dxw = (gc - ga) * (1 - yw) + (gd - gb) * yw;
        const float dyw = (gb - ga) * (1 - xw) + (gd - gc) * xw;
        d_flow[2 * p] = dxw;
        d_flow[2 * p + 1] = dyw;
      }
    }
  }
}

/* ---------------------------------------------------------------------------
 * Masks and elementwise pieces of compute_losses.
 * ------------------------------------------------------------------------- */

/* losses.py:325-335: ones inside, zero border of `pad_h` rows top/bottom
   (paddings[0]) and `pad_w` columns left/right (paddings[1]); [B,H,W,1]. */
void oracle_create_mask(float *mask, int B, int H, int W, int pad_top, int pad_bottom, int pad_left,
                        int pad_right) {
  for (int b = 0; b < B; b++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++)
        mask[((size_t)b * H + y) * W + x] =
            (y >= pad_top && y < H - pad_bottom && x >= pad_left && x < W - pad_right) ? 1.0f : 0.0f;
}

/* losses.py:338-344: sz = ceil(min(H,W)*ratio) computed in float32 */
int oracle_border_size(int H, int W, float border_ratio) {
  float min_dim = (float)(H < W ? H : W);
  return (int)ceilf(min_dim * border_ratio);
}

/* losses.py:347-366 */
void oracle_create_outgoing_mask(const float *flow, float *mask, int B, int H, int W) {
  const long total = (long)B * H * W;
#pragma omp parallel for schedule(static)
  for (long p = 0; p < total; p++) {
    const int gx = (int)(p % W), gy = (int)((p / W) % H);
    const float pos_x = (float)gx + flow[2 * p];
    const float pos_y = (float)gy + flow[2 * p + 1];
    const int inside_x = pos_x <= (float)(W - 1) && pos_x >= 0.0f;
    const int inside_y = pos_y <= (float)(H - 1) && pos_y >= 0.0f;
    mask[p] = (inside_x && inside_y) ? 1.0f : 0.0f;
  }
}

/* losses.py:38-51: forward-backward consistency occlusion.
   flow_diff = flow + flow_other_warped; mag_sq = |flow|^2 + |flow_other_warped|^2;
   occ = |flow_diff|^2 > 0.01*mag_sq + 0.5.  Writes flow_diff [.,2] and occ [.,1]
   (either may be NULL). length_sq = reduce_sum(square(x), 3): u^2 + v^2. */
void oracle_fb_occlusion(const float *flow, const float *flow_other_warped, float *flow_diff, float *occ,
                         int B, int H, int W) {
  const long total = (long)B * H * W;
#pragma omp parallel for schedule(static)
  for (long p = 0; p < total; p++) {
    const float u = flow[2 * p], v = flow[2 * p + 1];
    const float uw = flow_other_warped[2 * p], vw = flow_other_warped[2 * p + 1];
    const float du = u + uw, dv = v + vw;
    const float mag_sq = (u * u + v * v) + (uw * uw + vw * vw);
    const float occ_thresh = 0.01f * mag_sq + 0.5f;
    const float len = du * du + dv * dv;
    if (flow_diff) {
      flow_diff[2 * p] = du;
      flow_diff[2 * p + 1] = dv;
    }
    if (occ) occ[p] = len > occ_thresh ? 1.0f : 0.0f;
  }
}

/* losses.py:298-322.  x [B,H,W,C]; mask NULL, or [B,H,W,1] (mask_channels=1),
   or [B,H,W,C] (mask_channels=C).  error = pow(square(x*beta)+square(eps), alpha)
   (* mask) (min truncate if truncate>0); returns sum/normalization.  The
   reduction is carried in double so the oracle value is the exactly-rounded
   reference for any fp32 summation order; per-element values are fp32. */
double oracle_charbonnier_fwd(const float *x, const float *mask, int mask_channels, int B, int H, int W,
                              int C, float alpha, float beta, float epsilon, float truncate) {
  const long npix = (long)B * H * W;
  double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
  for (long p = 0; p < npix; p++) {
    for (int c = 0; c < C; c++) {
      const float xv = x[p * C + c] * beta;
      float error = powf(xv * xv + epsilon * epsilon, alpha);
      if (mask) error = mask[mask_channels == 1 ? p : p * C + c] * error;
      if (truncate > 0.0f) error = fminf(error, truncate);
      total += (double)error;
    }
  }
  const float normalization = (float)((double)npix * C);
  return total / (double)normalization;
}

/* d loss / d x for the above, scaled by upstream scalar gradient `gout`.
   d/dx pow(b^2 x^2 + e^2, a) = a * pow(.., a-1) * 2 b^2 x. */
void oracle_charbonnier_bwd(const float *x, const float *mask, int mask_channels, float gout, float *dx,
                            int B, int H, int W, int C, float alpha, float beta, float epsilon,
                            float truncate) {
  const long npix = (long)B * H * W;
  const float normalization = (float)((double)npix * C);
#pragma omp parallel for schedule(static)
  for (long p = 0; p < npix; p++) {
    for (int c = 0; c < C; c++) {
      const float xr = x[p * C + c];
      const float xv = xr * beta;
      const float base = xv * xv + epsilon * epsilon;
      float m = mask ? mask[mask_channels == 1 ? p : p * C + c] : 1.0f;
      float g = alpha * powf(base, alpha - 1.0f) * 2.0f * beta * beta * xr * m;
      if (truncate > 0.0f) {
        float error = m * powf(base, alpha);
        if (error > truncate) g = 0.0f;
      }
      dx[p * C + c] = gout * g / normalization;
    }
  }
}

/* ---------------------------------------------------------------------------
 * Soft ternary census loss: losses.py:90-122.
 *   I = rgb_to_grayscale(im)*255      (:94; luma 0.2989/0.5870/0.1140 -- TF's
 *                                       image_ops constants, assumption A1)
 *   patches[k=i*P+j](y,x) = I[y+i-r, x+j-r], zero outside (conv2d SAME, :99-104)
 *   t = (patch - I)/sqrt(0.81 + (patch-I)^2)                        (:106-107)
 *   dist = sum_k (t1-t2)^2 / (0.1 + (t1-t2)^2)                      (:110-114)
 *   loss = charbonnier(dist, mask*create_mask(r))                   (:119-122)
 * For 1-channel input (C==1) grayscale is the identity*... not used by the
 * reference (rgb_to_grayscale needs 3 channels); C must be 3.
 * ------------------------------------------------------------------------- */
static inline float gray255(const float *px) {
  /* tf.image.rgb_to_grayscale: tensordot(image, [0.2989, 0.5870, 0.1140]) */
  float g = px[0] * 0.2989f;
  g = g + px[1] * 0.5870f;
  g = g + px[2] * 0.1140f;
  return g * 255.0f;
}

static void to_gray(const float *im, float *gray, long npix) {
#pragma omp parallel for schedule(static)
  for (long p = 0; p < npix; p++) gray[p] = gray255(im + 3 * p);
}

/* dist_out (optional) receives the per-pixel dist_sum [B,H,W,1]. */
double oracle_ternary_loss_fwd(const float *im1, const float *im2_warped, const float *mask,
                               float *dist_out, int B, int H, int W, int max_distance) {
  const long npix = (long)B * H * W;
  const int r = max_distance, P = 2 * r + 1;
  float *g1 = (float *)malloc(sizeof(float) * npix);
  float *g2 = (float *)malloc(sizeof(float) * npix);
  to_gray(im1, g1, npix);
  to_gray(im2_warped, g2, npix);
  const float alpha = 0.45f, epsilon = 0.001f;
  double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
  for (long p = 0; p < npix; p++) {
    const int x = (int)(p % W), y = (int)((p / W) % H);
    const long base = (p / ((long)W * H)) * ((long)W * H);
    const float c1 = g1[p], c2 = g2[p];
    float dist_sum = 0.0f;
    for (int i = 0; i < P; i++)
      for (int j = 0; j < P; j++) {
        const int yy = y + i - r, xx = x + j - r;
        const int inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float p1 = inside ? g1[base + (long)yy * W + xx] : 0.0f;
        const float p2 = inside ? g2[base + (long)yy * W + xx] : 0.0f;
        const float d1 = p1 - c1, d2 = p2 - c2;
        const float t1 = d1 / sqrtf(0.81f + d1 * d1);
        const float t2 = d2 / sqrtf(0.81f + d2 * d2);
        const float df = t1 - t2;
        const float dist = df * df;
        dist_sum = dist_sum + dist / (0.1f + dist);
      }
    if (dist_out) dist_out[p] = dist_sum;
    const float tm = (y >= r && y < H - r && x >= r && x < W - r) ? 1.0f : 0.0f;
    const float m = mask[p] * tm;
    const float error = m * powf(dist_sum * dist_sum + epsilon * epsilon, alpha);
    total += (double)error;
  }
  free(g1);
  free(g2);
  const float normalization = (float)npix;
  return total / (double)normalization;
}

/* Gradient of the loss above wrt im2_warped [B,H,W,3], times upstream `gout`.
   With e=(t1-t2)^2, h=e/(0.1+e), t=d/sqrt(0.81+d^2):
     dL/dD   = m * alpha * (D^2+eps^2)^(alpha-1) * 2D / N
     dh/dt2  = -2 (t1-t2) * 0.1/(0.1+e)^2
     dt/dd   = 0.81/(0.81+d^2)^1.5
   g_k is added to the gray gradient at the neighbour (if inside the image)
   and subtracted at the centre; d gray/d rgb = 255*(0.2989,0.5870,0.1140).
   Accumulation into the gray-gradient image is a scatter (order differs from
   TF's conv2d-backprop but each target receives the same <=2*P*P terms). */
void oracle_ternary_loss_bwd(const float *im1, const float *im2_warped, const float *mask, float gout,
                             float *d_im2_warped, int B, int H, int W, int max_distance) {
  const long npix = (long)B * H * W;
  const long hw = (long)W * H;
  const int r = max_distance, P = 2 * r + 1;
  float *g1 = (float *)malloc(sizeof(float) * npix);
  float *g2 = (float *)malloc(sizeof(float) * npix);
  double *dgray = (double *)calloc(npix, sizeof(double));
  to_gray(im1, g1, npix);
  to_gray(im2_warped, g2, npix);
  const float alpha = 0.45f, epsilon = 0.001f;
  const float normalization = (float)npix;
  /* gather form: gradient at pixel q = sum over centres p whose window holds q
     (+g) and over q's own window (-g).  Done as a scatter per item. */
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; b++) {
    const long base = b * hw;
    for (long q = 0; q < hw; q++) {
      const long p = base + q;
      const int x = (int)(q % W), y = (int)(q / W);
      const float c1 = g1[p], c2 = g2[p];
      const float tm = (y >= r && y < H - r && x >= r && x < W - r) ? 1.0f : 0.0f;
      const float m = mask[p] * tm;
      if (m == 0.0f) continue;
      /* pass 1: D */
      float dist_sum = 0.0f;
      for (int i = 0; i < P; i++)
        for (int j = 0; j < P; j++) {
          const int yy = y + i - r, xx = x + j - r;
          const int inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
          const float p1 = inside ? g1[base + (long)yy * W + xx] : 0.0f;
          const float p2 = inside ? g2[base + (long)yy * W + xx] : 0.0f;
          const float d1 = p1 - c1, d2 = p2 - c2;
          const float t1 = d1 / sqrtf(0.81f + d1 * d1);
          const float t2 = d2 / sqrtf(0.81f + d2 * d2);
          const float df = t1 - t2;
          const float dist = df * df;
          dist_sum = dist_sum + dist / (0.1f + dist);
        }
      const double dLdD = (double)gout * m * alpha *
                          pow((double)dist_sum * dist_sum + (double)epsilon * epsilon, (double)alpha - 1.0) *
                          2.0 * dist_sum / normalization;
      /* pass 2: scatter */
      for (int i = 0; i < P; i++)
        for (int j = 0; j < P; j++) {
          const int yy = y + i - r, xx = x + j - r;
          const int inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
          const float p1 = inside ? g1[base + (long)yy * W + xx] : 0.0f;
          const float p2 = inside ? g2[base + (long)yy * W + xx] : 0.0f;
          const float d1 = p1 - c1, d2 = p2 - c2;
          const double t1 = d1 / sqrt(0.81 + (double)d1 * d1);
          const double t2 = d2 / sqrt(0.81 + (double)d2 * d2);
          const double df = t1 - t2;
          const double e = df * df;
          const double dh_dt2 = -2.0 * df * 0.1 / ((0.1 + e) * (0.1 + e));
          const double dt_dd = 0.81 / pow(0.81 + (double)d2 * d2, 1.5);
          const double g = dLdD * dh_dt2 * dt_dd;
          if (inside) dgray[base + (long)yy * W + xx] += g;
          dgray[p] -= g;
        }
    }
  }
#pragma omp parallel for schedule(static)
  for (long p = 0; p < npix; p++) {
    d_im2_warped[3 * p + 0] = (float)(255.0 * 0.2989f * dgray[p]);
    d_im2_warped[3 * p + 1] = (float)(255.0 * 0.5870f * dgray[p]);
    d_im2_warped[3 * p + 2] = (float)(255.0 * 0.1140f * dgray[p]);
  }
  free(g1);
  free(g2);
  free(dgray);
}

/* ---------------------------------------------------------------------------
 * 3x3 stencil terms of compute_losses: smoothness_loss, second_order_loss,
 * gradient_loss (losses.py:206-295).  tf.nn.conv2d SAME = cross-correlation
 * with zero padding: out[y,x] = sum_{i,j} in[y+i-1, x+j-1] * w[i][j]; the taps
 * are summed in row-major order, zero weights skipped (weights are 0, +-1, +-2,
 * so every product is exact and only the order of the additions matters).
 * ------------------------------------------------------------------------- */
static const float kFlowFilters1[2][9] = {
    {0, 0, 0, 0, 1, -1, 0, 0, 0}, /* losses.py:212  filter_x */
    {0, 0, 0, 0, 1, 0, 0, -1, 0}, /* losses.py:213  filter_y */
};
static const float kFlowFilters2[4][9] = {
    {0, 0, 0, 1, -2, 1, 0, 0, 0}, /* losses.py:266-268 filter_x */
    {0, 1, 0, 0, -2, 0, 0, 1, 0}, /* :269-271 filter_y */
    {1, 0, 0, 0, -2, 0, 0, 0, 1}, /* :272-274 filter_diag1 */
    {0, 0, 1, 0, -2, 0, 1, 0, 0}, /* :275-277 filter_diag2 */
};
static const float kSobelX[9] = {-1, 0, 1, -2, 0, 2, -1, 0, 1}; /* losses.py:230 */
static const float kSobelY[9] = {-1, -2, -1, 0, 0, 0, 1, 2, 1}; /* its transpose, :231 */

/* one output of conv2d SAME on channel c of an NHWC tensor with `C` channels */
static inline float conv3x3_at(const float *in, int H, int W, int C, int c, int y, int x, const float *w) {
  float s = 0.0f;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      if (w[i * 3 + j] == 0.0f) continue;
      const int yy = y + i - 1, xx = x + j - 1;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      s = s + in[((size_t)yy * W + xx) * C + c] * w[i * 3 + j];
    }
  return s;
}

/* mask channel k of _smoothness_deltas (order 1, :208-210) / _second_order_deltas (order 2, :260-263) */
static inline float flow_delta_mask(int order, int k, int H, int W, int y, int x) {
  if (order == 1) return k == 0 ? (x < W - 1 ? 1.0f : 0.0f) : (y < H - 1 ? 1.0f : 0.0f);
  const int mx = x >= 1 && x < W - 1, my = y >= 1 && y < H - 1;
  return k == 0 ? (float)mx : (k == 1 ? (float)my : (float)(mx && my));
}

/* delta_u, delta_v, mask: [B,H,W,K], K = 2 (order 1) or 4 (order 2) */
int oracle_smoothness_deltas(const float *flow, float *delta_u, float *delta_v, float *mask, int B, int H,
                             int W, int order) {
  if (order != 1 && order != 2) return 1;
  const int K = order == 1 ? 2 : 4;
  for (int b = 0; b < B; b++) {
    const float *f = flow + (size_t)b * H * W * 2;
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++)
        for (int k = 0; k < K; k++) {
          const float *w = order == 1 ? kFlowFilters1[k] : kFlowFilters2[k];
          const size_t o = (((size_t)b * H + y) * W + x) * K + k;
          if (delta_u) delta_u[o] = conv3x3_at(f, H, W, 2, 0, y, x, w);
          if (delta_v) delta_v[o] = conv3x3_at(f, H, W, 2, 1, y, x, w);
          if (mask) mask[o] = flow_delta_mask(order, k, H, W, y, x);
        }
  }
  return 0;
}

/* smoothness_loss (order 1, :250-255) / second_order_loss (order 2, :290-295):
   charbonnier_loss(delta_u, mask) + charbonnier_loss(delta_v, mask).  norm_count
   (0 = B*H*W*K) is the normalisation the GLOBAL batch gives. */
double oracle_smoothness_loss_fwd(const float *flow, int B, int H, int W, int order, double norm_count) {
  const int K = order == 1 ? 2 : 4;
  const size_t n = (size_t)B * H * W * K;
  float *du = (float *)malloc(n * sizeof(float)), *dv = (float *)malloc(n * sizeof(float));
  float *m = (float *)malloc(n * sizeof(float));
  oracle_smoothness_deltas(flow, du, dv, m, B, H, W, order);
  const double scale = norm_count > 0 ? (double)n / norm_count : 1.0;
  const double l = oracle_charbonnier_fwd(du, m, K, B, H, W, K, 0.45f, 1.0f, 0.001f, 0.0f) +
                   oracle_charbonnier_fwd(dv, m, K, B, H, W, K, 0.45f, 1.0f, 0.001f, 0.0f);
  free(du); free(dv); free(m);
  return l * scale;
}

void oracle_smoothness_loss_bwd(const float *flow, float gout, float *dflow, int B, int H, int W, int order,
                                double norm_count) {
  const int K = order == 1 ? 2 : 4;
  const size_t n = (size_t)B * H * W * K;
  float *du = (float *)malloc(n * sizeof(float)), *dv = (float *)malloc(n * sizeof(float));
  float *m = (float *)malloc(n * sizeof(float));
  float *gu = (float *)malloc(n * sizeof(float)), *gv = (float *)malloc(n * sizeof(float));
  oracle_smoothness_deltas(flow, du, dv, m, B, H, W, order);
  const float g = (float)(norm_count > 0 ? (double)gout * (double)n / norm_count : (double)gout);
  oracle_charbonnier_bwd(du, m, K, g, gu, B, H, W, K, 0.45f, 1.0f, 0.001f, 0.0f);
  oracle_charbonnier_bwd(dv, m, K, g, gv, B, H, W, K, 0.45f, 1.0f, 0.001f, 0.0f);
  /* transpose of the convolution: d in[y+i-1, x+j-1] += w[i][j] * d out[y,x] */
  memset(dflow, 0, (size_t)B * H * W * 2 * sizeof(float));
  for (int b = 0; b < B; b++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++)
        for (int k = 0; k < K; k++) {
          const float *w = order == 1 ? kFlowFilters1[k] : kFlowFilters2[k];
          const size_t o = (((size_t)b * H + y) * W + x) * K + k;
          for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
              const int yy = y + i - 1, xx = x + j - 1;
              if (w[i * 3 + j] == 0.0f || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
              float *d = dflow + (((size_t)b * H + yy) * W + xx) * 2;
              d[0] += w[i * 3 + j] * gu[o];
              d[1] += w[i * 3 + j] * gv[o];
            }
        }
  free(du); free(dv); free(m); free(gu); free(gv);
}

/* _gradient_delta (:228-241): conv2d(im1, sobel) - conv2d(im2_warped, sobel), [B,H,W,6],
   channel 2c = d/dx of image channel c, 2c+1 = d/dy. */
void oracle_gradient_delta(const float *im1, const float *im2_warped, float *diff, int B, int H, int W) {
  for (int b = 0; b < B; b++) {
    const float *a = im1 + (size_t)b * H * W * 3, *c2 = im2_warped + (size_t)b * H * W * 3;
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++)
        for (int c = 0; c < 3; c++) {
          float *o = diff + (((size_t)b * H + y) * W + x) * 6 + 2 * c;
          o[0] = conv3x3_at(a, H, W, 3, c, y, x, kSobelX) - conv3x3_at(c2, H, W, 3, c, y, x, kSobelX);
          o[1] = conv3x3_at(a, H, W, 3, c, y, x, kSobelY) - conv3x3_at(c2, H, W, 3, c, y, x, kSobelY);
        }
  }
}

/* mask * tile(concat(mask_x, mask_y), 3)  (:244-247) as a 6-channel mask */
static float *gradient_mask6(const float *mask, int B, int H, int W) {
  float *m6 = (float *)malloc((size_t)B * H * W * 6 * sizeof(float));
  for (int b = 0; b < B; b++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const size_t p = ((size_t)b * H + y) * W + x;
        const float mx = (x >= 1 && x < W - 1) ? 1.0f : 0.0f, my = (y >= 1 && y < H - 1) ? 1.0f : 0.0f;
        for (int k = 0; k < 6; k++) m6[p * 6 + k] = mask[p] * ((k & 1) ? my : mx);
      }
  return m6;
}

double oracle_gradient_loss_fwd(const float *im1, const float *im2_warped, const float *mask, int B, int H,
                                int W, double norm_count) {
  const size_t n = (size_t)B * H * W * 6;
  float *diff = (float *)malloc(n * sizeof(float));
  float *m6 = gradient_mask6(mask, B, H, W);
  oracle_gradient_delta(im1, im2_warped, diff, B, H, W);
  const double l = oracle_charbonnier_fwd(diff, m6, 6, B, H, W, 6, 0.45f, 1.0f, 0.001f, 0.0f);
  free(diff); free(m6);
  return norm_count > 0 ? l * (double)n / norm_count : l;
}

/* d loss / d im2_warped (im1 carries no gradient in the reference's graph) */
void oracle_gradient_loss_bwd(const float *im1, const float *im2_warped, const float *mask, float gout,
                              float *d_im2_warped, int B, int H, int W, double norm_count) {
  const size_t n = (size_t)B * H * W * 6;
  float *diff = (float *)malloc(n * sizeof(float)), *gd = (float *)malloc(n * sizeof(float));
  float *m6 = gradient_mask6(mask, B, H, W);
  oracle_gradient_delta(im1, im2_warped, diff, B, H, W);
  const float g = (float)(norm_count > 0 ? (double)gout * (double)n / norm_count : (double)gout);
  oracle_charbonnier_bwd(diff, m6, 6, g, gd, B, H, W, 6, 0.45f, 1.0f, 0.001f, 0.0f);
  memset(d_im2_warped, 0, (size_t)B * H * W * 3 * sizeof(float));
  for (int b = 0; b < B; b++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++)
        for (int c = 0; c < 3; c++) {
          const size_t o = (((size_t)b * H + y) * W + x) * 6 + 2 * c;
          for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
              const int yy = y + i - 1, xx = x + j - 1;
              if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
              /* diff = conv(im1) - conv(im2_warped) */
              d_im2_warped[(((size_t)b * H + yy) * W + xx) * 3 + c] -=
                  kSobelX[i * 3 + j] * gd[o] + kSobelY[i * 3 + j] * gd[o + 1];
            }
        }
  free(diff); free(gd); free(m6);
}

/* ---------------------------------------------------------------------------
 * tf.image.resize_bilinear, TensorFlow 1.x, align_corners=False (third-party;
 * its published algorithm: scale = in/out in float, in = out_index*scale,
 * lower = floor(in), upper = min(lower+1, in_size-1), lerp = in - lower,
 * top = tl + (tr-tl)*x_lerp, bottom likewise, out = top + (bottom-top)*y_lerp),
 * followed by the reference's `* 4 * FLOW_SCALE` (flownet.py:48) as `mult`.
 * ------------------------------------------------------------------------- */
void oracle_resize_bilinear_fwd(const float *in, float *out, int B, int h, int w, int H, int W, int C,
                                float mult) {
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  for (int b = 0; b < B; b++)
    for (int y = 0; y < H; y++) {
      const float iy = (float)y * sh;
      const float fy = floorf(iy);
      int y0 = (int)fy; if (y0 > h - 1) y0 = h - 1;
      const int y1 = y0 + 1 < h ? y0 + 1 : h - 1;
      const float ly = iy - fy;
      for (int x = 0; x < W; x++) {
        const float ix = (float)x * sw;
        const float fx = floorf(ix);
        int x0 = (int)fx; if (x0 > w - 1) x0 = w - 1;
        const int x1 = x0 + 1 < w ? x0 + 1 : w - 1;
        const float lx = ix - fx;
        for (int c = 0; c < C; c++) {
          const float tl = in[(((size_t)b * h + y0) * w + x0) * C + c], tr = in[(((size_t)b * h + y0) * w + x1) * C + c];
          const float bl = in[(((size_t)b * h + y1) * w + x0) * C + c], br = in[(((size_t)b * h + y1) * w + x1) * C + c];
          const float top = tl + (tr - tl) * lx;
          const float bot = bl + (br - bl) * lx;
          out[(((size_t)b * H + y) * W + x) * C + c] = (top + (bot - top) * ly) * mult;
        }
      }
    }
}

/* ---------------------------------------------------------------------------
 * tf.image.resize_area, TensorFlow 1.x, align_corners=False -- the odd-extent
 * branch of core/util.downsample (src/e2eflow/core/util.py:23-28).  Third-party
 * (TensorFlow is not vendored; README.md:83 "at least version 1.7", unpinned);
 * its published algorithm (resize_area_op.cc), ASSUMPTION A3, PARITY UNPINNED:
 *   scale = in_size / (float)out_size
 *   rows i = floor(y*scale) .. ceil((y+1)*scale)-1, weight
 *     i < in_y ? (i+1 > in_y1 ? scale : i+1 - in_y) : (i+1 > in_y1 ? in_y1 - i : 1)
 *   (columns likewise), indices clamped to the image; per row sum over x
 *   (first and last column weighted, inner columns weight 1), rows combined with
 *   their weights, result times 1/(scale_y*scale_x).
 * ------------------------------------------------------------------------- */
static float area_weight(int i, float lo, float hi, float scale) {
  return (float)i < lo ? ((float)(i + 1) > hi ? scale : (float)(i + 1) - lo) : ((float)(i + 1) > hi ? hi - (float)i : 1.0f);
}
static int bound_index(int v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); }

void oracle_resize_area_fwd(const float *in, float *out, int B, int H, int W, int C, int oh, int ow) {
  const float sy = (float)H / (float)oh, sx = (float)W / (float)ow;
  const float norm = 1.0f / (sy * sx);
  for (int b = 0; b < B; b++)
    for (int y = 0; y < oh; y++) {
      const float in_y = (float)y * sy, in_y1 = (float)(y + 1) * sy;
      const int y0 = (int)floorf(in_y), y1 = (int)ceilf(in_y1);
      for (int x = 0; x < ow; x++) {
        const float in_x = (float)x * sx, in_x1 = (float)(x + 1) * sx;
        const int x0 = (int)floorf(in_x), x1 = (int)ceilf(in_x1);
        for (int c = 0; c < C; c++) {
          float sum = 0.f;
          for (int i = y0; i < y1; i++) {
            const float *row = in + (((size_t)b * H + bound_index(i, H)) * W) * C + c;
            float sum_y = row[(size_t)bound_index(x0, W) * C] * area_weight(x0, in_x, in_x1, sx);
            if (x0 + 1 != x1) {
              for (int j = x0 + 1; j < x1 - 1; j++) sum_y += row[(size_t)bound_index(j, W) * C];
              sum_y += row[(size_t)bound_index(x1 - 1, W) * C] * area_weight(x1 - 1, in_x, in_x1, sx);
            }
            sum += sum_y * area_weight(i, in_y, in_y1, sy);
          }
          out[(((size_t)b * oh + y) * ow + x) * C + c] = sum * norm;
        }
      }
    }
}

/* gradient of the above wrt `in` (double accumulation, then rounded) */
void oracle_resize_bilinear_bwd(const float *gout, float *gin, int B, int h, int w, int H, int W, int C,
                                float mult) {
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  double *acc = (double *)calloc((size_t)B * h * w * C, sizeof(double));
  for (int b = 0; b < B; b++)
    for (int y = 0; y < H; y++) {
      const float iy = (float)y * sh;
      const float fy = floorf(iy);
      int y0 = (int)fy; if (y0 > h - 1) y0 = h - 1;
      const int y1 = y0 + 1 < h ? y0 + 1 : h - 1;
      const double ly = iy - fy;
      for (int x = 0; x < W; x++) {
        const float ix = (float)x * sw;
        const float fx = floorf(ix);
        int x0 = (int)fx; if (x0 > w - 1) x0 = w - 1;
        const int x1 = x0 + 1 < w ? x0 + 1 : w - 1;
        const double lx = ix - fx;
        for (int c = 0; c < C; c++) {
          const double g = (double)gout[(((size_t)b * H + y) * W + x) * C + c] * mult;
          acc[(((size_t)b * h + y0) * w + x0) * C + c] += g * (1 - ly) * (1 - lx);
          acc[(((size_t)b * h + y0) * w + x1) * C + c] += g * (1 - ly) * lx;
          acc[(((size_t)b * h + y1) * w + x0) * C + c] += g * ly * (1 - lx);
          acc[(((size_t)b * h + y1) * w + x1) * C + c] += g * ly * lx;
        }
      }
    }
  for (size_t i = 0; i < (size_t)B * h * w * C; i++) gin[i] = (float)acc[i];
  free(acc);
}

/* flownet.py:46-57: inputs = concat([im1, im2, flow, warp, diff]) -> [B,H,W,14] */
void oracle_stacked_inputs(const float *im1, const float *im2, const float *flow_lr, float *out, int B, int H,
                           int W, int h, int w, float mult) {
  const size_t npix = (size_t)B * H * W;
  float *flow = (float *)malloc(npix * 2 * sizeof(float));
  float *warp = (float *)malloc(npix * 3 * sizeof(float));
  oracle_resize_bilinear_fwd(flow_lr, flow, B, h, w, H, W, 2, mult);
  oracle_image_warp_fwd(im2, flow, warp, B, H, W, 3);
  for (size_t p = 0; p < npix; p++) {
    float *o = out + p * 14;
    for (int c = 0; c < 3; c++) {
      o[c] = im1[p * 3 + c];
      o[3 + c] = im2[p * 3 + c];
      o[8 + c] = warp[p * 3 + c];
      o[11 + c] = fabsf(warp[p * 3 + c] - im1[p * 3 + c]);
    }
    o[6] = flow[p * 2];
    o[7] = flow[p * 2 + 1];
  }
  free(flow);
  free(warp);
}

/* ---------------------------------------------------------------------------
 * Integer ternary census + Hamming distance.  NOT a restatement of reference code
 * (the reference's census is the soft one, D1): the definition of
 * include/unflow_hip.h, written independently for the CPU.
 * ------------------------------------------------------------------------- */
void oracle_census_transform(const float *im, uint64_t *sig, int B, int H, int W, int r, float eps) {
  const long npix = (long)B * H * W;
  float *g = (float *)malloc((size_t)npix * sizeof(float));
  to_gray(im, g, npix);
#pragma omp parallel for schedule(static)
  for (long p = 0; p < npix; p++) {
    const int x = (int)(p % W), y = (int)((p / W) % H);
    const long base = p - ((long)y * W + x);
    uint64_t brighter = 0, darker = 0;
    int k = 0;
    for (int i = -r; i <= r; i++)
      for (int j = -r; j <= r; j++) {
        if (i == 0 && j == 0) continue;
        const int yy = y + i, xx = x + j;
        const float q = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? g[base + (long)yy * W + xx] : 0.0f;
        const float d = q - g[p];
        if (d > eps) brighter |= (uint64_t)1 << k;
        if (d < -eps) darker |= (uint64_t)1 << k;
        k++;
      }
    sig[2 * p] = brighter;
    sig[2 * p + 1] = darker;
  }
  free(g);
}

uint64_t oracle_census_hamming(const uint64_t *s1, const uint64_t *s2, const float *mask, int32_t *dist, long n) {
  uint64_t total = 0;
  for (long p = 0; p < n; p++) {
    const int d = __builtin_popcountll((s1[2 * p] ^ s2[2 * p]) | (s1[2 * p + 1] ^ s2[2 * p + 1]));
    if (dist) dist[p] = d;
    if (!mask || mask[p] != 0.0f) total += (uint64_t)d;
  }
  return total;
}
