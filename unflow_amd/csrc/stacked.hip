// Full-resolution warp block of the stacked networks, for gfx950:
//   flow = tf.image.resize_bilinear(flow, [H, W]) * 4 * FLOW_SCALE
//   warp = image_warp(im2, flow);  diff = |warp - im1|
//   inputs = concat([im1, im2, flow, warp, diff], axis=3)        -> [B,H,W,14]
// (src/e2eflow/core/flownet.py:46-57).  The reference runs resize, ~25 image_warp
// graph nodes, abs and concat, each a full pass over HBM; here one kernel reads
// im1 (12 B/px), gathers im2 (12 B/px, 4 taps served by L2) and the low-
// resolution flow, and writes the 56 B/px of the result through LDS so that every
// global store is a 16-byte piece of one contiguous span.
//
// tf.image.resize_bilinear is third-party arithmetic (assumption A2 in DESIGN.md):
// TensorFlow 1.x semantics with align_corners=False -- scale = in/out in float,
// in = out_index * scale, lower = floor(in), upper = min(lower + 1, in_size - 1),
// lerp = in - lower; top = tl + (tr - tl)*x_lerp, bottom likewise,
// out = top + (bottom - top)*y_lerp.
#include "common.h"
#include "warp_taps.h"

#include <cstdint>

namespace unflow {
namespace {

constexpr int kBlock = 256;

struct ResizeTaps {
  int y0, y1, x0, x1;
  float ly, lx;
};

__device__ inline ResizeTaps resize_taps(int y, int x, int h, int w, float sh, float sw) {
  ResizeTaps t;
  const float iy = mul_rn((float)y, sh), ix = mul_rn((float)x, sw);
  const float fy = floorf(iy), fx = floorf(ix);
  t.y0 = min((int)fy, h - 1);
  t.x0 = min((int)fx, w - 1);
  t.y1 = min(t.y0 + 1, h - 1);
  t.x1 = min(t.x0 + 1, w - 1);
  t.ly = iy - fy;
  t.lx = ix - fx;
  return t;
}

__device__ inline float lerp_rn(float a, float b, float t) { return add_rn(a, mul_rn(add_rn(b, -a), t)); }

template <int C>
__global__ __launch_bounds__(kBlock) void resize_bilinear_fwd_kernel(const float *__restrict__ in,
                                                                     float *__restrict__ out, int h, int w,
                                                                     int H, int W, int channels, float sh,
                                                                     float sw, float mult, long long total) {
  const int nc = C > 0 ? C : channels;
  for (long long idx = blockIdx.x * (long long)kBlock + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * kBlock) {
    const int x = (int)(idx % W);
    const long long row = idx / W;
    const int y = (int)(row % H);
    const long long b = row / H;
    const ResizeTaps t = resize_taps(y, x, h, w, sh, sw);
    const float *base = in + b * h * w * nc;
    const float *tl = base + ((long long)t.y0 * w + t.x0) * nc, *tr = base + ((long long)t.y0 * w + t.x1) * nc;
    const float *bl = base + ((long long)t.y1 * w + t.x0) * nc, *br = base + ((long long)t.y1 * w + t.x1) * nc;
    for (int c = 0; c < nc; ++c) {
      const float top = lerp_rn(tl[c], tr[c], t.lx), bot = lerp_rn(bl[c], br[c], t.lx);
      out[idx * nc + c] = mul_rn(lerp_rn(top, bot, t.ly), mult);
    }
  }
}

// d in[tap] += weight * mult * gout  (float atomics; the low-resolution tensor is small)
__global__ __launch_bounds__(kBlock) void resize_bilinear_bwd_kernel(const float *__restrict__ gout,
                                                                     float *__restrict__ gin, int h, int w, int H,
                                                                     int W, int nc, float sh, float sw, float mult,
                                                                     long long total) {
  for (long long idx = blockIdx.x * (long long)kBlock + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * kBlock) {
    const int x = (int)(idx % W);
    const long long row = idx / W;
    const int y = (int)(row % H);
    const long long b = row / H;
    const ResizeTaps t = resize_taps(y, x, h, w, sh, sw);
    float *base = gin + b * h * w * nc;
    for (int c = 0; c < nc; ++c) {
      const float g = gout[idx * nc + c] * mult;
      atomicAdd(base + ((long long)t.y0 * w + t.x0) * nc + c, g * (1.f - t.ly) * (1.f - t.lx));
      atomicAdd(base + ((long long)t.y0 * w + t.x1) * nc + c, g * (1.f - t.ly) * t.lx);
      atomicAdd(base + ((long long)t.y1 * w + t.x0) * nc + c, g * t.ly * (1.f - t.lx));
      atomicAdd(base + ((long long)t.y1 * w + t.x1) * nc + c, g * t.ly * t.lx);
    }
  }
}

// One block = one 64x4 pixel tile; the 14 output floats of its pixels are assembled
// in LDS and stored as float4 pieces of four contiguous 3.5 KB row spans.
__global__ __launch_bounds__(kBlock) void stacked_inputs_kernel(const float *__restrict__ im1,
                                                                const float *__restrict__ im2,
                                                                const float2 *__restrict__ flow_lr,
                                                                float *__restrict__ out, int h, int w, int H,
                                                                int W, float sh, float sw, float mult,
                                                                long long total) {
  __shared__ __attribute__((aligned(16))) float tile[kBlock * 14];
  // one block = one 64x4 pixel tile (see tiled_pixel): 5 rows of im2 for 4 rows of taps
  const TiledPixel tp = tiled_pixel(blockIdx.x * (long long)kBlock + threadIdx.x, H, W);
  if (tp.ok) {
    const long long idx = tp.idx, b = tp.b;
    const int x = tp.x, y = tp.y;
    const ResizeTaps r = resize_taps(y, x, h, w, sh, sw);
    const float2 *fb = flow_lr + b * h * w;
    const float2 tl = fb[(long long)r.y0 * w + r.x0], tr = fb[(long long)r.y0 * w + r.x1];
    const float2 bl = fb[(long long)r.y1 * w + r.x0], br = fb[(long long)r.y1 * w + r.x1];
    float2 f;
    f.x = mul_rn(lerp_rn(lerp_rn(tl.x, tr.x, r.lx), lerp_rn(bl.x, br.x, r.lx), r.ly), mult);
    f.y = mul_rn(lerp_rn(lerp_rn(tl.y, tr.y, r.lx), lerp_rn(bl.y, br.y, r.lx), r.ly), mult);
    const ClampTaps t = clamp_taps(x, y, f, H, W);
    const float *base = im2 + b * H * W * 3;
    const float *pa = base + t.ia * 3, *pb = base + t.ib * 3, *pc = base + t.ic * 3, *pd = base + t.id * 3;
    float la[3], lb[3], lc[3], ld[3], a1[3], a2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      la[c] = pa[c]; lb[c] = pb[c]; lc[c] = pc[c]; ld[c] = pd[c];
      a1[c] = im1[idx * 3 + c];
      a2[c] = im2[idx * 3 + c];
    }
    const float wa = (1.f - t.xw) * (1.f - t.yw), wb = (1.f - t.xw) * t.yw;
    const float wc = t.xw * (1.f - t.yw), wd = t.xw * t.yw;
    float *o = tile + threadIdx.x * 14;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float s = mul_rn(wa, la[c]);
      s = add_rn(s, mul_rn(wb, lb[c]));
      s = add_rn(s, mul_rn(wc, lc[c]));
      s = add_rn(s, mul_rn(wd, ld[c]));
      o[c] = a1[c];
      o[3 + c] = a2[c];
      o[8 + c] = s;
      o[11 + c] = fabsf(s - a1[c]);
    }
    o[6] = f.x;
    o[7] = f.y;
  }
  __syncthreads();
  // each tile row is one contiguous span of (valid pixels) * 14 floats in the output
  const TiledPixel t0 = tiled_pixel(blockIdx.x * (long long)kBlock, H, W);   // first pixel of the tile
  const int nvx = min(kTileW, W - t0.x) * 14;                                 // floats per tile row
  for (int rrow = 0; rrow < kTileH; ++rrow) {
    if (t0.y + rrow >= H) break;
    float *dst = out + ((t0.b * H + t0.y + rrow) * W + t0.x) * 14;            // 3584-byte aligned when W % 64 == 0
    const float *src = tile + rrow * kTileW * 14;
    const bool vec = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    for (int i = threadIdx.x * 4; i < nvx; i += kBlock * 4) {
      if (vec && i + 3 < nvx) {
        *reinterpret_cast<float4 *>(dst + i) = *reinterpret_cast<const float4 *>(src + i);
      } else {
        for (int j = i; j < min(i + 4, nvx); ++j) dst[j] = src[j];
      }
    }
  }
}

// tf.image.resize_area (TensorFlow 1.x, align_corners=False; third-party, its published
// algorithm, assumption A3): output pixel (y, x) is the box [y*sy, (y+1)*sy) x [x*sx, (x+1)*sx)
// of the input averaged with fractional coverage weights; scale = in / out in float, the
// box rows i = floor(in_y) .. ceil(in_y1)-1 weigh  i < in_y ? (i+1 > in_y1 ? sy : i+1-in_y)
// : (i+1 > in_y1 ? in_y1-i : 1), indices clamped to the image; each row is summed over x
// first, rows are then combined, the result is multiplied by 1/(sy*sx).  One lane per output
// element; the dispatcher only sends odd-extent pyramid levels here (core/util.py:23-28).
__device__ inline float area_weight(int i, float lo, float hi, float scale) {
  return (float)i < lo ? ((float)(i + 1) > hi ? scale : (float)(i + 1) - lo) : ((float)(i + 1) > hi ? hi - (float)i : 1.0f);
}

__global__ __launch_bounds__(kBlock) void resize_area_fwd_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                                 int H, int W, int C, int oh, int ow, float sy,
                                                                 float sx, long long total) {
  const float norm = 1.0f / (sy * sx);
  for (long long idx = blockIdx.x * (long long)kBlock + threadIdx.x; idx < total; idx += (long long)gridDim.x * kBlock) {
    const int c = (int)(idx % C);
    long long r = idx / C;
    const int x = (int)(r % ow);
    r /= ow;
    const int y = (int)(r % oh);
    const long long b = r / oh;
    // box edges are ROUNDED products (TF stores them in floats): without mul_rn the compiler
    // contracts them into the weight subtractions as FMAs and the edge weights move by an ulp of
    // the coordinate (1e-6 of the result)
    const float in_y = mul_rn((float)y, sy), in_y1 = mul_rn((float)(y + 1), sy);
    const float in_x = mul_rn((float)x, sx), in_x1 = mul_rn((float)(x + 1), sx);
    const int y0 = (int)floorf(in_y), y1 = (int)ceilf(in_y1);
    const int x0 = (int)floorf(in_x), x1 = (int)ceilf(in_x1);
    float sum = 0.f;
    for (int i = y0; i < y1; ++i) {
      const float *row = in + ((b * H + min(max(i, 0), H - 1)) * (long long)W) * C + c;
      float sum_y = mul_rn(row[(long long)min(max(x0, 0), W - 1) * C], area_weight(x0, in_x, in_x1, sx));
      if (x0 + 1 != x1) {
        for (int j = x0 + 1; j < x1 - 1; ++j) sum_y = add_rn(sum_y, row[(long long)min(max(j, 0), W - 1) * C]);
        sum_y = add_rn(sum_y, mul_rn(row[(long long)min(max(x1 - 1, 0), W - 1) * C], area_weight(x1 - 1, in_x, in_x1, sx)));
      }
      sum = add_rn(sum, mul_rn(sum_y, area_weight(i, in_y, in_y1, sy)));
    }
    out[idx] = mul_rn(sum, norm);
  }
}

inline unsigned grid_for(long long total) {
  const long long g = (total + kBlock - 1) / kBlock;
  return (unsigned)(g < 1 ? 1 : (g > 65535LL * 16 ? 65535LL * 16 : g));
}

inline int check_resize(const char *what, int B, int h, int w, int H, int W, int C) {
  if (B < 0 || h < 0 || w < 0 || H < 0 || W < 0 || C < 0)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "%s: negative dimension", what);
  if ((long long)B * H * W > 0 && (h == 0 || w == 0))
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "%s: input images must have positive height and width", what);
  return UNFLOW_OK;
}

}  // namespace
}  // namespace unflow

using namespace unflow;

extern "C" int unflow_resize_bilinear_fwd(const float *in, float *out, int B, int h, int w, int H, int W, int C,
                                          float mult, unflow_stream_t stream) {
  if (int rc = check_resize("resize_bilinear", B, h, w, H, W, C)) return rc;
  const long long total = (long long)B * H * W;
  if (total == 0 || C == 0) return UNFLOW_OK;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  hipStream_t s = as_stream(stream);
  if (C == 2)
    hipLaunchKernelGGL(resize_bilinear_fwd_kernel<2>, dim3(grid_for(total)), dim3(kBlock), 0, s, in, out, h, w, H, W, C,
                       sh, sw, mult, total);
  else
    hipLaunchKernelGGL(resize_bilinear_fwd_kernel<0>, dim3(grid_for(total)), dim3(kBlock), 0, s, in, out, h, w, H, W, C,
                       sh, sw, mult, total);
  return launch_status("resize_bilinear_fwd");
}

extern "C" int unflow_resize_area_fwd(const float *in, float *out, int B, int H, int W, int C, int out_h, int out_w,
                                      unflow_stream_t stream) {
  if (int rc = check_resize("resize_area", B, H, W, out_h, out_w, C)) return rc;
  const long long total = (long long)B * out_h * out_w * C;
  if (total == 0) return UNFLOW_OK;
  const float sy = (float)H / (float)out_h, sx = (float)W / (float)out_w;
  hipLaunchKernelGGL(resize_area_fwd_kernel, dim3(grid_for(total)), dim3(kBlock), 0, as_stream(stream), in, out, H, W, C,
                     out_h, out_w, sy, sx, total);
  return launch_status("resize_area_fwd");
}

extern "C" int unflow_resize_bilinear_bwd(const float *gout, float *gin, int B, int h, int w, int H, int W, int C,
                                          float mult, unflow_stream_t stream) {
  if (int rc = check_resize("resize_bilinear_grad", B, h, w, H, W, C)) return rc;
  hipStream_t s = as_stream(stream);
  const size_t nin = (size_t)B * h * w * C;
  if (nin && hipMemsetAsync(gin, 0, sizeof(float) * nin, s) != hipSuccess)
    return launch_status("resize_bilinear_bwd memset");
  const long long total = (long long)B * H * W;
  if (total == 0 || C == 0) return UNFLOW_OK;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(grid_for(total)), dim3(kBlock), 0, s, gout, gin, h, w, H, W, C,
                     sh, sw, mult, total);
  return launch_status("resize_bilinear_bwd");
}

extern "C" int unflow_stacked_inputs_fwd(const float *im1, const float *im2, const float *flow_lr, float *out,
                                         int B, int H, int W, int h, int w, float mult, unflow_stream_t stream) {
  if (int rc = check_resize("stacked_inputs", B, h, w, H, W, 2)) return rc;
  const long long total = (long long)B * H * W;
  if (total == 0) return UNFLOW_OK;
  const long long blocks = tiled_total(B, H, W) / kBlock;
  if (blocks > 0x7fffffffLL) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "stacked_inputs: too many pixels for one launch");
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  hipLaunchKernelGGL(stacked_inputs_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), im1, im2,
                     reinterpret_cast<const float2 *>(flow_lr), out, h, w, H, W, sh, sw, mult, total);
  return launch_status("stacked_inputs");
}
