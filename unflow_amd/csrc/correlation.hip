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

#define ROUND_OFF 50000  // ops/correlation_op.cu.cc:19

// One lane per input element (b,c,y,x), x fastest; computes both gradients with
// the reference's output-range arithmetic (ops/correlation_op.cu.cc:133-142,
// :211-224).
__global__ __launch_bounds__(256) void corr_bwd_generic_kernel(const float *__restrict__ topdiff,
                                                               const float *__restrict__ in0,
                                                               const float *__restrict__ in1,
                                                               float *__restrict__ g0, float *__restrict__ g1,
                                                               CorrGeometry g, int C, long long total) {
  const int topwidth = g.out_width, topheight = g.out_height, topchannels = g.out_channels;
  const int kr = g.kernel_radius, s1 = g.stride_1, s2 = g.stride_2, md = g.max_displacement;
  const int ngr = g.grid_radius, ngw = g.grid_width;
  const int round_off = ROUND_OFF, round_off_s1 = s1 * round_off;
  const long long plane = (long long)g.H * g.W;
  const long long topplane = (long long)topheight * topwidth;
  const float inv = 1.f / (float)((kr * 2 + 1) * (kr * 2 + 1) * C);
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int lx = (int)(idx % g.W);
    long long t = idx / g.W;
    const int my = (int)(t % g.H);
    t /= g.H;
    const int n = (int)(t % C);
    const long long item = t / C;
    const int l = lx + g.pad, m = my + g.pad;
    const float *td = topdiff + item * topchannels * topplane;
    const float *p0 = in0 + (item * C + n) * plane;
    const float *p1 = in1 + (item * C + n) * plane;

    float sum0 = 0.f;
    {
      int xmin = (l - 2 * kr - md + round_off_s1 - 1) / s1 + 1 - round_off;
      int ymin = (m - 2 * kr - md + round_off_s1 - 1) / s1 + 1 - round_off;
      int xmax = (l - md + round_off_s1) / s1 - round_off;
      int ymax = (m - md + round_off_s1) / s1 - round_off;
      if (xmax >= 0 && ymax >= 0 && xmin <= topwidth - 1 && ymin <= topheight - 1) {
        xmin = max(0, xmin);
        xmax = min(topwidth - 1, xmax);
        ymin = max(0, ymin);
        ymax = min(topheight - 1, ymax);
        for (int p = -ngr; p <= ngr; ++p)
          for (int o = -ngr; o <= ngr; ++o) {
            const float bot1tmp = load_padded(p1, m + s2 * p, l + s2 * o, g.pad, g.H, g.W);
            const float *tp = td + (long long)((p + ngr) * ngw + (o + ngr)) * topplane;
            for (int y = ymin; y <= ymax; ++y)
              for (int x = xmin; x <= xmax; ++x) sum0 = fmaf(tp[(long long)y * topwidth + x], bot1tmp, sum0);
          }
      }
    }
    float sum1 = 0.f;
    for (int p = -ngr; p <= ngr; ++p)
      for (int o = -ngr; o <= ngr; ++o) {
        const int s2o = s2 * o, s2p = s2 * p;
        int xmin = (l - 2 * kr - md - s2o + round_off_s1 - 1) / s1 + 1 - round_off;
        int ymin = (m - 2 * kr - md - s2p + round_off_s1 - 1) / s1 + 1 - round_off;
        int xmax = (l - md - s2o + round_off_s1) / s1 - round_off;
        int ymax = (m - md - s2p + round_off_s1) / s1 - round_off;
        if (xmax >= 0 && ymax >= 0 && xmin <= topwidth - 1 && ymin <= topheight - 1) {
          xmin = max(0, xmin);
          xmax = min(topwidth - 1, xmax);
          ymin = max(0, ymin);
          ymax = min(topheight - 1, ymax);
          const float bot0tmp = load_padded(p0, m - s2p, l - s2o, g.pad, g.H, g.W);
          const float *tp = td + (long long)((p + ngr) * ngw + (o + ngr)) * topplane;
          for (int y = ymin; y <= ymax; ++y)
            for (int x = xmin; x <= xmax; ++x) sum1 = fmaf(tp[(long long)y * topwidth + x], bot0tmp, sum1);
        }
      }
    g0[idx] = sum0 * inv;
    g1[idx] = sum1 * inv;
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
