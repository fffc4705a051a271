// FlowNetC correlation cost volume, forward and backward, for gfx950.
//
// Reference (paths relative to the reference checkout):
//   geometry   ops/correlation_op.h:28-51
//   forward    ops/correlation_op.cu.cc:30-117, host :250-315
//   backward   ops/correlation_op.cu.cc:119-248, host :317-390
//
// This file holds the entry points, the attribute-generic kernels (any odd
// kernel_size, any strides, any pad) and the optional materialisation of the
// op's padded NHWC side outputs.  The FlowNetC-shaped problem (kernel_size 1,
// stride_1 1) is dispatched to the MFMA kernels in correlation_mfma.hip.
//
// Unlike the reference, no kernel reads the padded NHWC copies: zero padding is
// a bounds test on the NCHW inputs, which removes 2 x 36 MB/sample of writes
// and reads at 48x160x256.
#include "common.h"
#include "correlation.h"

#include <cmath>

namespace unflow {

int corr_geometry_init(CorrGeometry *g, int H, int W, int kernel_size, int max_displacement, int pad,
                       int stride_1, int stride_2) {
  if (kernel_size % 2 == 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "kernel_size must be odd");
  if (kernel_size <= 0 || stride_1 <= 0 || stride_2 <= 0 || max_displacement < 0 || pad < 0 || H < 0 || W < 0)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT,
                "Invalid correlation settings (kernel_size, strides must be positive; max_displacement, pad >= 0)");
  g->H = H;
  g->W = W;
  g->pad = pad;
  g->stride_1 = stride_1;
  g->stride_2 = stride_2;
  g->max_displacement = max_displacement;
  g->kernel_size = kernel_size;
  g->padded_height = H + 2 * pad;
  g->padded_width = W + 2 * pad;
  g->kernel_radius = (kernel_size - 1) / 2;
  g->border_size = max_displacement + g->kernel_radius;
  g->grid_radius = max_displacement / stride_2;
  g->grid_width = g->grid_radius * 2 + 1;
  g->out_width = (int)std::ceil((float)(g->padded_width - g->border_size * 2) / (float)stride_1);
  g->out_height = (int)std::ceil((float)(g->padded_height - g->border_size * 2) / (float)stride_1);
  g->out_channels = g->grid_width * g->grid_width;
  // ops/correlation_op.cc:60-61 tests the product only; two negative extents
  // would pass it and fail at allocation, so each sign is tested.
  if (g->out_width <= 0 || g->out_height <= 0)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "Invalid correlation settings");
  return UNFLOW_OK;
}

namespace {

// NCHW -> zero padded NHWC (ops/correlation_op.cu.cc:30-49 + memset :282-283),
// written with the channel index on the lane so the stores coalesce.
__global__ __launch_bounds__(256) void pad_nhwc_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                       int C, int H, int W, int pad, long long total) {
  const int PH = H + 2 * pad, PW = W + 2 * pad;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long t = i / C;
    const int xp = (int)(t % PW);
    t /= PW;
    const int yp = (int)(t % PH);
    const long long b = t / PH;
    const int x = xp - pad, y = yp - pad;
    float v = 0.f;
    if (x >= 0 && x < W && y >= 0 && y < H) v = in[((b * C + c) * H + y) * W + x];
    out[i] = v;
  }
}

__device__ inline float load_padded(const float *__restrict__ plane, int yp, int xp, int pad, int H, int W) {
  const int y = yp - pad, x = xp - pad;
  return (y >= 0 && y < H && x >= 0 && x < W) ? plane[(long long)y * W + x] : 0.f;
}

// One lane per output element, x fastest.  out[b,tc,y,x] =
//   1/(k*k*C) * sum_{j,i,c} P0[c, y*s1+md+j, x*s1+md+i] * P1[c, same + (s2p, s2o)]
__global__ __launch_bounds__(256) void corr_fwd_generic_kernel(const float *__restrict__ in0,
                                                               const float *__restrict__ in1,
                                                               float *__restrict__ out, CorrGeometry g, int C,
                                                               long long total) {
  const int Wo = g.out_width, Ho = g.out_height, D = g.out_channels;
  const long long plane = (long long)g.H * g.W;
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int x = (int)(idx % Wo);
    long long t = idx / Wo;
    const int y = (int)(t % Ho);
    t /= Ho;
    const int tc = (int)(t % D);
    const long long b = t / D;
    const int x1 = x * g.stride_1 + g.max_displacement;
    const int y1 = y * g.stride_1 + g.max_displacement;
    const int s2o = (tc % g.grid_width - g.grid_radius) * g.stride_2;
    const int s2p = (tc / g.grid_width - g.grid_radius) * g.stride_2;
    const float *p0 = in0 + b * C * plane;
    const float *p1 = in1 + b * C * plane;
    float sum = 0.f;
    for (int j = 0; j < g.kernel_size; ++j)
      for (int i = 0; i < g.kernel_size; ++i)
        for (int c = 0; c < C; ++c) {
          const float a = load_padded(p0 + c * plane, y1 + j, x1 + i, g.pad, g.H, g.W);
          const float v = load_padded(p1 + c * plane, y1 + s2p + j, x1 + s2o + i, g.pad, g.H, g.W);
          sum = fmaf(a, v, sum);
        }
    out[idx] = sum / (float)(g.kernel_size * g.kernel_size * C);
  }
}

// Integer division that rounds towards -inf / +inf for a positive divisor (the numerators below go negative
// inside the padding, where C's truncation towards zero would be wrong).
__device__ inline int div_floor(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
__device__ inline int div_ceil(int a, int b) { return -div_floor(-a, b); }

// Output positions whose k-wide patch (origin o * stride_1 + max_displacement in padded coordinates) covers
// the padded coordinate q along one axis:  o * stride_1 + j == q - max_displacement  for a tap j in [0, k).
struct Span {
  int lo, hi;   // inclusive, clamped to [0, n_out); empty when lo > hi
};
__device__ inline Span covering_outputs(int q_minus_md, int k, int stride_1, int n_out) {
  Span s;
  s.lo = max(0, div_ceil(q_minus_md - (k - 1), stride_1));
  s.hi = min(n_out - 1, div_floor(q_minus_md, stride_1));
  return s;
}

// sum of one displacement plane of the upstream gradient over a rectangle of output positions
__device__ inline float box_sum(const float *__restrict__ dplane, int out_w, Span ys, Span xs, float partner, float acc) {
  for (int y = ys.lo; y <= ys.hi; ++y)
    for (int x = xs.lo; x <= xs.hi; ++x) acc = fmaf(dplane[(long long)y * out_w + x], partner, acc);
  return acc;
}

// Both gradients for attributes the matrix-core path does not take (kernel_size > 1, stride_1 > 1, ...): one lane
// per input element (b, c, y, x), x fastest.  With (yq, xq) the element's padded coordinates and
// t = (dy, dx) * stride_2 a displacement, the forward (ops/correlation_op.cu.cc:51-117) pairs first[yq, xq] with
// second[yq + t] in every output position whose patch covers (yq, xq), and second[yq, xq] with first[yq - t] in
// every output position whose patch covers (yq, xq) - t.  Hence
//   d first[yq, xq]  = 1/(k*k*C) * sum_t second[(yq, xq) + t] * sum_{o covers (yq, xq)}     top[t, o]
//   d second[yq, xq] = 1/(k*k*C) * sum_t first[(yq, xq) - t]  * sum_{o covers (yq, xq) - t} top[t, o]
// -- the sums the reference's CorrelateDataBackward0 / 1 evaluate (ops/correlation_op.cu.cc:119-248).
__global__ __launch_bounds__(256) void corr_bwd_generic_kernel(const float *__restrict__ topdiff,
                                                               const float *__restrict__ in0,
                                                               const float *__restrict__ in1,
                                                               float *__restrict__ g0, float *__restrict__ g1,
                                                               CorrGeometry g, int C, long long total) {
  const int out_w = g.out_width, out_h = g.out_height;
  const int k = g.kernel_size, s1 = g.stride_1, s2 = g.stride_2, md = g.max_displacement;
  const int radius = g.grid_radius, gw = g.grid_width;
  const long long plane = (long long)g.H * g.W, out_plane = (long long)out_h * out_w;
  const float scale = 1.f / (float)(k * k * C);
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int x = (int)(idx % g.W);
    long long rest = idx / g.W;
    const int y = (int)(rest % g.H);
    rest /= g.H;
    const int c = (int)(rest % C);
    const long long item = rest / C;
    const int xq = x + g.pad, yq = y + g.pad;
    const float *top = topdiff + item * g.out_channels * out_plane;
    const float *first = in0 + (item * C + c) * plane;
    const float *second = in1 + (item * C + c) * plane;

    // wrt the first input: the covering rectangle does not depend on the displacement
    float d_first = 0.f;
    const Span ys0 = covering_outputs(yq - md, k, s1, out_h), xs0 = covering_outputs(xq - md, k, s1, out_w);
    if (ys0.lo <= ys0.hi && xs0.lo <= xs0.hi) {
      for (int dy = -radius; dy <= radius; ++dy)
        for (int dx = -radius; dx <= radius; ++dx) {
          const float partner = load_padded(second, yq + s2 * dy, xq + s2 * dx, g.pad, g.H, g.W);
          d_first = box_sum(top + (long long)((dy + radius) * gw + (dx + radius)) * out_plane, out_w, ys0, xs0, partner, d_first);
        }
    }
    // wrt the second input: the rectangle moves with the displacement
    float d_second = 0.f;
    for (int dy = -radius; dy <= radius; ++dy) {
      const Span ys = covering_outputs(yq - md - s2 * dy, k, s1, out_h);
      if (ys.lo > ys.hi) continue;
      for (int dx = -radius; dx <= radius; ++dx) {
        const Span xs = covering_outputs(xq - md - s2 * dx, k, s1, out_w);
        if (xs.lo > xs.hi) continue;
        const float partner = load_padded(first, yq - s2 * dy, xq - s2 * dx, g.pad, g.H, g.W);
        d_second = box_sum(top + (long long)((dy + radius) * gw + (dx + radius)) * out_plane, out_w, ys, xs, partner, d_second);
      }
    }
    g0[idx] = d_first * scale;
    g1[idx] = d_second * scale;
  }
}

inline unsigned grid_for(long long total) {
  long long gsz = (total + 255) / 256;
  return (unsigned)(gsz < 65536 ? gsz : 65536);
}

}  // namespace
}  // namespace unflow

using namespace unflow;

extern "C" int unflow_correlation_geometry(int H, int W, int kernel_size, int max_displacement, int pad,
                                           int stride_1, int stride_2, int *dims) {
  CorrGeometry g;
  if (int rc = corr_geometry_init(&g, H, W, kernel_size, max_displacement, pad, stride_1, stride_2)) return rc;
  dims[0] = g.out_channels;
  dims[1] = g.out_height;
  dims[2] = g.out_width;
  dims[3] = g.padded_height;
  dims[4] = g.padded_width;
  return UNFLOW_OK;
}

extern "C" int unflow_correlation_fwd(const float *in0, const float *in1, float *out, float *padded0,
                                      float *padded1, int B, int C, int H, int W, int kernel_size,
                                      int max_displacement, int pad, int stride_1, int stride_2,
                                      unflow_stream_t stream) {
  CorrGeometry g;
  if (int rc = corr_geometry_init(&g, H, W, kernel_size, max_displacement, pad, stride_1, stride_2)) return rc;
  if (B < 0 || C < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "correlation: negative dimension");
  if (B == 0) return UNFLOW_OK;
  hipStream_t s = as_stream(stream);
  if (padded0 && padded1 && C > 0) {
    const long long ptotal = (long long)B * g.padded_height * g.padded_width * C;
    hipLaunchKernelGGL(pad_nhwc_kernel, dim3(grid_for(ptotal)), dim3(256), 0, s, in0, padded0, C, H, W, pad, ptotal);
    hipLaunchKernelGGL(pad_nhwc_kernel, dim3(grid_for(ptotal)), dim3(256), 0, s, in1, padded1, C, H, W, pad, ptotal);
    if (int rc = launch_status("correlation pad")) return rc;
  }
  const long long total = (long long)B * g.out_channels * g.out_height * g.out_width;
  if (C == 0) {
    // sum over zero channels divided by zero elements: the reference would emit NaN (0/0)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "correlation: inputs must have at least one channel");
  }
  if (corr_mfma_supported(g, C)) return corr_mfma_fwd(in0, in1, out, B, C, g, s);
  hipLaunchKernelGGL(corr_fwd_generic_kernel, dim3(grid_for(total)), dim3(256), 0, s, in0, in1, out, g, C, total);
  return launch_status("correlation_fwd");
}

namespace {
// shared argument checks of the four fp16 entry points; returns UNFLOW_OK with *skip set for empty tensors
int f16_entry(CorrGeometry *g, int B, int C, int H, int W, int kernel_size, int max_displacement, int pad, int stride_1,
              int stride_2, bool backward, bool *skip) {
  if (int rc = corr_geometry_init(g, H, W, kernel_size, max_displacement, pad, stride_1, stride_2)) return rc;
  const char *what = backward ? "correlation_grad" : "correlation";
  if (B < 0 || C < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "%s: negative dimension", what);
  *skip = backward ? (long long)B * C * H * W == 0 : B == 0;
  if (*skip) return UNFLOW_OK;
  if (C == 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "correlation: inputs must have at least one channel");
  if (!(backward ? corr_mfma_bwd_supported(*g, C) : corr_mfma_supported(*g, C)))
    return fail(UNFLOW_ERR_INVALID_ARGUMENT,
                "%s (fp16) is implemented for the FlowNetC attributes only "
                "(kernel_size 1, max_displacement 20, pad 20, stride_1 1, stride_2 2)", what);
  return UNFLOW_OK;
}
}  // namespace

extern "C" int unflow_correlation_fwd_f16(const void *in0, const void *in1, void *out, int B, int C, int H, int W,
                                          int kernel_size, int max_displacement, int pad, int stride_1,
                                          int stride_2, unflow_stream_t stream) {
  CorrGeometry g;
  bool skip = false;
  if (int rc = f16_entry(&g, B, C, H, W, kernel_size, max_displacement, pad, stride_1, stride_2, false, &skip)) return rc;
  return skip ? UNFLOW_OK : corr16_fwd(in0, in1, out, B, C, g, as_stream(stream));
}

extern "C" int unflow_correlation_bwd_f16(const void *topdiff, const void *in0, const void *in1, void *g0, void *g1,
                                          int B, int C, int H, int W, int kernel_size, int max_displacement,
                                          int pad, int stride_1, int stride_2, unflow_stream_t stream) {
  CorrGeometry g;
  bool skip = false;
  if (int rc = f16_entry(&g, B, C, H, W, kernel_size, max_displacement, pad, stride_1, stride_2, true, &skip)) return rc;
  return skip ? UNFLOW_OK : corr16_bwd(topdiff, in0, in1, g0, g1, B, C, g, as_stream(stream));
}

extern "C" int unflow_correlation_fwd_f16_exact(const void *in0, const void *in1, void *out, int B, int C, int H, int W,
                                                int kernel_size, int max_displacement, int pad, int stride_1,
                                                int stride_2, unflow_stream_t stream) {
  CorrGeometry g;
  bool skip = false;
  if (int rc = f16_entry(&g, B, C, H, W, kernel_size, max_displacement, pad, stride_1, stride_2, false, &skip)) return rc;
  return skip ? UNFLOW_OK : corr_mfma_fwd_f16(in0, in1, out, B, C, g, as_stream(stream));
}

extern "C" int unflow_correlation_bwd_f16_exact(const void *topdiff, const void *in0, const void *in1, void *g0,
                                                void *g1, int B, int C, int H, int W, int kernel_size,
                                                int max_displacement, int pad, int stride_1, int stride_2,
                                                unflow_stream_t stream) {
  CorrGeometry g;
  bool skip = false;
  if (int rc = f16_entry(&g, B, C, H, W, kernel_size, max_displacement, pad, stride_1, stride_2, true, &skip)) return rc;
  return skip ? UNFLOW_OK : corr_mfma_bwd_f16(topdiff, in0, in1, g0, g1, B, C, g, as_stream(stream));
}

extern "C" int unflow_correlation_bwd(const float *topdiff, const float *in0, const float *in1,
                                      const float *padded0, const float *padded1, float *g0, float *g1, int B,
                                      int C, int H, int W, int kernel_size, int max_displacement, int pad,
                                      int stride_1, int stride_2, unflow_stream_t stream) {
  (void)padded0;
  (void)padded1;
  CorrGeometry g;
  if (int rc = corr_geometry_init(&g, H, W, kernel_size, max_displacement, pad, stride_1, stride_2)) return rc;
  if (B < 0 || C < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "correlation_grad: negative dimension");
  const long long total = (long long)B * C * H * W;
  if (total == 0) return UNFLOW_OK;
  hipStream_t s = as_stream(stream);
  if (corr_mfma_bwd_supported(g, C)) return corr_mfma_bwd(topdiff, in0, in1, g0, g1, B, C, g, s);
  hipLaunchKernelGGL(corr_bwd_generic_kernel, dim3(grid_for(total)), dim3(256), 0, s, topdiff, in0, in1, g0, g1, g,
                     C, total);
  return launch_status("correlation_bwd");
}

// ---------------------------------------------------------------------------
// Both directions of a bidirectional FlowNetC (core/flownet.py:33-43 runs
// correlation(a, b) and correlation(b, a) on the same feature pair).  The second cost
// volume is a re-indexing of the first:
//   corr(b, a)[d, p] = 1/C sum_c b[c, p] a[c, p + s2*d] = corr(a, b)[-d, p + s2*d]
// (0 where p + s2*d leaves the image: the zero padding), the very same products summed
// in the very same order, hence bit-identical.  This kernel applies that map, which
// is its own inverse; with `add` it computes add + flip(in), which folds the upstream
// gradient of the second volume into that of the first so that ONE backward call
// yields the summed gradients.
// ---------------------------------------------------------------------------
namespace unflow {
namespace {
__global__ __launch_bounds__(256) void corr_flip_kernel(const float *__restrict__ in, const float *__restrict__ add,
                                                        float *__restrict__ out, int gw, int s2, int H, int W,
                                                        long long total) {
  const int r = gw / 2, D = gw * gw;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W);
    long long t = i / W;
    const int y = (int)(t % H);
    t /= H;
    const int d = (int)(t % D);
    const long long b = t / D;
    const int dy = d / gw - r, dx = d % gw - r;
    const int yy = y + s2 * dy, xx = x + s2 * dx;
    float v = add ? add[i] : 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      const float f = in[((b * D + (D - 1 - d)) * H + yy) * W + xx];
      v = add ? v + f : f;
    }
    out[i] = v;
  }
}
}  // namespace
}  // namespace unflow

extern "C" int unflow_correlation_flip(const float *in, const float *add, float *out, int B, int D, int H, int W,
                                       int kernel_size, int max_displacement, int pad, int stride_1, int stride_2,
                                       unflow_stream_t stream) {
  CorrGeometry g;
  if (int rc = corr_geometry_init(&g, H, W, kernel_size, max_displacement, pad, stride_1, stride_2)) return rc;
  if (D != g.out_channels)   // the launch is sized from the attrs: a volume built with other attrs would be overrun
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "correlation_flip: volume has %d displacement planes, the attrs imply %d", D,
                g.out_channels);
  if (kernel_size != 1 || stride_1 != 1 || pad != max_displacement || g.out_height != H || g.out_width != W)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT,
                "correlation_flip needs kernel_size 1, stride_1 1 and pad == max_displacement (output grid == input grid)");
  if (B < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "correlation_flip: negative dimension");
  const long long total = (long long)B * g.out_channels * H * W;
  if (total == 0) return UNFLOW_OK;
  hipLaunchKernelGGL(corr_flip_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), in, add, out,
                     g.grid_width, stride_2, H, W, total);
  return launch_status("correlation_flip");
}
