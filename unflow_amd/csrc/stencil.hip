// The 3x3-stencil terms of compute_losses, fused with their charbonnier
// reductions, for gfx950:
//   smoothness_loss / _smoothness_deltas        src/e2eflow/core/losses.py:206-225,250-255
//   second_order_loss / _second_order_deltas    src/e2eflow/core/losses.py:258-295
//   gradient_loss / _gradient_delta             src/e2eflow/core/losses.py:228-247
//
// The reference builds each term from conv2d (SAME, one-hot 3x3 filters), a
// create_mask per output channel, concat / tile and charbonnier_loss: about ten
// full-size tensors per term.  Here a workgroup stages its 32x8 tile (+ halo)
// of the flow / the two images in LDS once; the forward kernels read 8 or 28
// B/pixel, the backward kernels recompute the per-channel charbonnier
// derivatives for tile + 1 halo in LDS and gather the transposed stencil from
// there (no atomics, no intermediate tensor in HBM).
//
// All filter weights are 0, +-1, +-2: every product is exact, so a delta is
// reproduced bit for bit by adding its taps in the reference's (row-major) order.
#include "common.h"

namespace unflow {
namespace {

constexpr int kTW = 32, kTH = 8, kNT = kTW * kTH;
constexpr float kAlpha = 0.45f, kEps = 0.001f;

struct Tap { int dy, dx; float w; };

// _smoothness_deltas (losses.py:212-213) / _second_order_deltas (:266-277); taps in row-major order
template <int ORDER> struct FlowFilters;
template <> struct FlowFilters<1> {
  static constexpr int K = 2, T = 2;
  static constexpr Tap taps[K][T] = {{{0, 0, 1.f}, {0, 1, -1.f}}, {{0, 0, 1.f}, {1, 0, -1.f}}};
};
template <> struct FlowFilters<2> {
  static constexpr int K = 4, T = 3;
  static constexpr Tap taps[K][T] = {{{0, -1, 1.f}, {0, 0, -2.f}, {0, 1, 1.f}},
                                     {{-1, 0, 1.f}, {0, 0, -2.f}, {1, 0, 1.f}},
                                     {{-1, -1, 1.f}, {0, 0, -2.f}, {1, 1, 1.f}},
                                     {{-1, 1, 1.f}, {0, 0, -2.f}, {1, -1, 1.f}}};
};

// create_mask of channel k (losses.py:208-210 / :260-263)
template <int ORDER>
__device__ inline bool flow_mask(int k, int H, int W, int y, int x) {
  if (ORDER == 1) return k == 0 ? x < W - 1 : y < H - 1;
  const bool mx = x >= 1 && x < W - 1, my = y >= 1 && y < H - 1;
  return k == 0 ? mx : (k == 1 ? my : (mx && my));
}

// (x^2 + eps^2)^alpha and its derivative, beta = 1 (pow_pos: common.h)
__device__ inline float charb(float x) { return pow_pos(add_rn(mul_rn(x, x), kEps * kEps), kAlpha); }
__device__ inline float dcharb(float x) {
  return kAlpha * pow_pos(add_rn(mul_rn(x, x), kEps * kEps), kAlpha - 1.f) * 2.f * x;
}

// Stage a (kTH + 2*HALO) x (kTW + 2*HALO) tile of an NHWC tensor with C floats
// per pixel; zero outside the image (conv2d SAME).
template <int C, int HALO>
__device__ inline void stage_tile(const float *__restrict__ src, float *dst, long long item_px, int H, int W,
                                  int x0, int y0) {
  constexpr int LW = kTW + 2 * HALO, LH = kTH + 2 * HALO;
  for (int i = threadIdx.x; i < LW * LH * C; i += kNT) {
    const int ly = i / (LW * C), r = i - ly * (LW * C);
    const int lx = r / C, c = r - lx * C;
    const int y = y0 + ly - HALO, x = x0 + lx - HALO;
    dst[i] = (y >= 0 && y < H && x >= 0 && x < W) ? src[(item_px + (long long)y * W + x) * C + c] : 0.f;
  }
}

template <int ORDER, int LW>
__device__ inline float flow_delta(const float *f /* at the centre pixel, component */, int k) {
  using F = FlowFilters<ORDER>;
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < F::T; ++t) s = add_rn(s, f[(F::taps[k][t].dy * LW + F::taps[k][t].dx) * 2] * F::taps[k][t].w);
  return s;
}

// ------------------------------------------------------------------ flow smoothness
template <int ORDER>
__global__ __launch_bounds__(kNT) void flow_deltas_kernel(const float *__restrict__ flow,
                                                          float *__restrict__ delta_u,
                                                          float *__restrict__ delta_v,
                                                          float *__restrict__ mask, int H, int W) {
  constexpr int K = FlowFilters<ORDER>::K, LW = kTW + 2;
  __shared__ float tile[(kTH + 2) * LW * 2];
  const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
  const long long item_px = (long long)blockIdx.z * H * W;
  stage_tile<2, 1>(flow, tile, item_px, H, W, x0, y0);
  __syncthreads();
  const int tx = threadIdx.x % kTW, ty = threadIdx.x / kTW;
  const int x = x0 + tx, y = y0 + ty;
  if (x >= W || y >= H) return;
  const float *f = tile + ((ty + 1) * LW + tx + 1) * 2;
  const long long o = (item_px + (long long)y * W + x) * K;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (delta_u) delta_u[o + k] = flow_delta<ORDER, LW>(f, k);
    if (delta_v) delta_v[o + k] = flow_delta<ORDER, LW>(f + 1, k);
    if (mask) mask[o + k] = flow_mask<ORDER>(k, H, W, y, x) ? 1.f : 0.f;
  }
}

template <int ORDER>
__global__ __launch_bounds__(kNT) void flow_smooth_fwd_kernel(const float *__restrict__ flow,
                                                              float *__restrict__ partial, int H, int W) {
  constexpr int K = FlowFilters<ORDER>::K, LW = kTW + 2;
  __shared__ float tile[(kTH + 2) * LW * 2];
  __shared__ float red[kNT / 64];
  const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
  const long long item_px = (long long)blockIdx.z * H * W;
  stage_tile<2, 1>(flow, tile, item_px, H, W, x0, y0);
  __syncthreads();
  const int tx = threadIdx.x % kTW, ty = threadIdx.x / kTW;
  const int x = x0 + tx, y = y0 + ty;
  float err = 0.f;
  if (x < W && y < H) {
    const float *f = tile + ((ty + 1) * LW + tx + 1) * 2;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (flow_mask<ORDER>(k, H, W, y, x))
        err += charb(flow_delta<ORDER, LW>(f, k)) + charb(flow_delta<ORDER, LW>(f + 1, k));
    }
  }
  const float s = block_sum<kNT>(err, red);
  if (threadIdx.x == 0) partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
}

// d loss / d flow.  g_k(q) = mask_k(q) * dcharb(delta_k(q)) * scale is computed once for
// the tile + 1 halo (both components) and kept in LDS; then
//   dflow(p) = sum_k sum_t w_kt * g_k(p - o_kt)        (transpose of the stencil).
template <int ORDER>
__global__ __launch_bounds__(kNT) void flow_smooth_bwd_kernel(const float *__restrict__ flow,
                                                              const float *__restrict__ gout,
                                                              float *__restrict__ dflow, int H, int W,
                                                              float inv_norm) {
  using F = FlowFilters<ORDER>;
  constexpr int K = F::K, LW = kTW + 4, GW = kTW + 2, GH = kTH + 2;
  __shared__ float tile[(kTH + 4) * LW * 2];
  __shared__ float g[GH * GW * K * 2];
  const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
  const long long item_px = (long long)blockIdx.z * H * W;
  stage_tile<2, 2>(flow, tile, item_px, H, W, x0, y0);
  __syncthreads();
  const float scale = (gout ? *gout : 1.f) * inv_norm;
  for (int i = threadIdx.x; i < GH * GW; i += kNT) {
    const int gy = i / GW, gx = i - gy * GW;
    const int y = y0 + gy - 1, x = x0 + gx - 1;
    const bool in = y >= 0 && y < H && x >= 0 && x < W;
    const float *f = tile + ((gy + 1) * LW + gx + 1) * 2;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const bool m = in && flow_mask<ORDER>(k, H, W, y, x);
      g[(i * K + k) * 2] = m ? dcharb(flow_delta<ORDER, LW>(f, k)) * scale : 0.f;
      g[(i * K + k) * 2 + 1] = m ? dcharb(flow_delta<ORDER, LW>(f + 1, k)) * scale : 0.f;
    }
  }
  __syncthreads();
  const int tx = threadIdx.x % kTW, ty = threadIdx.x / kTW;
  const int x = x0 + tx, y = y0 + ty;
  if (x >= W || y >= H) return;
  float du = 0.f, dv = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int t = 0; t < F::T; ++t) {
      const int q = ((ty + 1 - F::taps[k][t].dy) * GW + tx + 1 - F::taps[k][t].dx) * K + k;
      du = fmaf(F::taps[k][t].w, g[q * 2], du);
      dv = fmaf(F::taps[k][t].w, g[q * 2 + 1], dv);
    }
  }
  reinterpret_cast<float2 *>(dflow)[item_px + (long long)y * W + x] = make_float2(du, dv);
}

// ------------------------------------------------------------------ gradient loss
// Sobel taps in row-major order with the zero weights skipped (losses.py:230-231)
constexpr Tap kSobelX[6] = {{-1, -1, -1.f}, {-1, 1, 1.f}, {0, -1, -2.f}, {0, 1, 2.f}, {1, -1, -1.f}, {1, 1, 1.f}};
constexpr Tap kSobelY[6] = {{-1, -1, -1.f}, {-1, 0, -2.f}, {-1, 1, -1.f}, {1, -1, 1.f}, {1, 0, 2.f}, {1, 1, 1.f}};

template <int LW>
__device__ inline float sobel(const float *p /* centre pixel, channel */, const Tap (&taps)[6]) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 6; ++t) s = add_rn(s, p[(taps[t].dy * LW + taps[t].dx) * 3] * taps[t].w);
  return s;
}

// one pixel's six channels of  conv2d(im1, sobel) - conv2d(im2_warped, sobel)
template <int LW>
__device__ inline void gradient_delta6(const float *a, const float *b, float (&d)[6]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    d[2 * c] = add_rn(sobel<LW>(a + c, kSobelX), -sobel<LW>(b + c, kSobelX));
    d[2 * c + 1] = add_rn(sobel<LW>(a + c, kSobelY), -sobel<LW>(b + c, kSobelY));
  }
}

__global__ __launch_bounds__(kNT) void gradient_delta_kernel(const float *__restrict__ im1,
                                                             const float *__restrict__ im2,
                                                             float *__restrict__ diff, int H, int W) {
  constexpr int LW = kTW + 2;
  __shared__ float t1[(kTH + 2) * LW * 3], t2[(kTH + 2) * LW * 3];
  const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
  const long long item_px = (long long)blockIdx.z * H * W;
  stage_tile<3, 1>(im1, t1, item_px, H, W, x0, y0);
  stage_tile<3, 1>(im2, t2, item_px, H, W, x0, y0);
  __syncthreads();
  const int tx = threadIdx.x % kTW, ty = threadIdx.x / kTW;
  const int x = x0 + tx, y = y0 + ty;
  if (x >= W || y >= H) return;
  const int c = ((ty + 1) * LW + tx + 1) * 3;
  float d[6];
  gradient_delta6<LW>(t1 + c, t2 + c, d);
  float *o = diff + (item_px + (long long)y * W + x) * 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) o[k] = d[k];
}

__global__ __launch_bounds__(kNT) void gradient_loss_fwd_kernel(const float *__restrict__ im1,
                                                                const float *__restrict__ im2,
                                                                const float *__restrict__ mask,
                                                                float *__restrict__ partial, int H, int W) {
  constexpr int LW = kTW + 2;
  __shared__ float t1[(kTH + 2) * LW * 3], t2[(kTH + 2) * LW * 3];
  __shared__ float red[kNT / 64];
  const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
  const long long item_px = (long long)blockIdx.z * H * W;
  stage_tile<3, 1>(im1, t1, item_px, H, W, x0, y0);
  stage_tile<3, 1>(im2, t2, item_px, H, W, x0, y0);
  __syncthreads();
  const int tx = threadIdx.x % kTW, ty = threadIdx.x / kTW;
  const int x = x0 + tx, y = y0 + ty;
  float err = 0.f;
  if (x < W && y < H) {
    const int c = ((ty + 1) * LW + tx + 1) * 3;
    float d[6];
    gradient_delta6<LW>(t1 + c, t2 + c, d);
    const float m = mask[item_px + (long long)y * W + x];
    // mask * tile(concat(mask_x, mask_y), 3)  (losses.py:244-247)
    const float mx = (x >= 1 && x < W - 1) ? m : 0.f, my = (y >= 1 && y < H - 1) ? m : 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) err += charb(d[k]) * ((k & 1) ? my : mx);
  }
  const float s = block_sum<kNT>(err, red);
  if (threadIdx.x == 0) partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
}

// d loss / d im2_warped = - sum over taps of  sobel weight * g(p - o)
__global__ __launch_bounds__(kNT) void gradient_loss_bwd_kernel(const float *__restrict__ im1,
                                                                const float *__restrict__ im2,
                                                                const float *__restrict__ mask,
                                                                const float *__restrict__ gout,
                                                                float *__restrict__ d_im2, int H, int W,
                                                                float inv_norm) {
  constexpr int LW = kTW + 4, GW = kTW + 2, GH = kTH + 2;
  __shared__ float t1[(kTH + 4) * LW * 3], t2[(kTH + 4) * LW * 3];
  __shared__ float g[GH * GW * 6];
  const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
  const long long item_px = (long long)blockIdx.z * H * W;
  stage_tile<3, 2>(im1, t1, item_px, H, W, x0, y0);
  stage_tile<3, 2>(im2, t2, item_px, H, W, x0, y0);
  __syncthreads();
  const float scale = (gout ? *gout : 1.f) * inv_norm;
  for (int i = threadIdx.x; i < GH * GW; i += kNT) {
    const int gy = i / GW, gx = i - gy * GW;
    const int y = y0 + gy - 1, x = x0 + gx - 1;
    const bool in = y >= 0 && y < H && x >= 0 && x < W;
    float d[6];
    const int c = ((gy + 1) * LW + gx + 1) * 3;
    gradient_delta6<LW>(t1 + c, t2 + c, d);
    const float m = in ? mask[item_px + (long long)y * W + x] * scale : 0.f;
    const float mx = (x >= 1 && x < W - 1) ? m : 0.f, my = (y >= 1 && y < H - 1) ? m : 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) g[i * 6 + k] = dcharb(d[k]) * ((k & 1) ? my : mx);
  }
  __syncthreads();
  const int tx = threadIdx.x % kTW, ty = threadIdx.x / kTW;
  const int x = x0 + tx, y = y0 + ty;
  if (x >= W || y >= H) return;
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const float *gx = g + ((ty + 1 - kSobelX[t].dy) * GW + tx + 1 - kSobelX[t].dx) * 6;
    const float *gy = g + ((ty + 1 - kSobelY[t].dy) * GW + tx + 1 - kSobelY[t].dx) * 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      acc[c] = fmaf(-kSobelX[t].w, gx[2 * c], acc[c]);
      acc[c] = fmaf(-kSobelY[t].w, gy[2 * c + 1], acc[c]);
    }
  }
  float *o = d_im2 + (item_px + (long long)y * W + x) * 3;
  o[0] = acc[0];
  o[1] = acc[1];
  o[2] = acc[2];
}

inline dim3 tile_grid(int B, int H, int W) { return dim3((W + kTW - 1) / kTW, (H + kTH - 1) / kTH, B); }
inline int n_tiles(int B, int H, int W) { return B * ((H + kTH - 1) / kTH) * ((W + kTW - 1) / kTW); }

inline int check_dims(const char *what, int B, int H, int W) {
  if (B < 0 || H < 0 || W < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "%s: negative dimension", what);
  if ((long long)B * ((H + kTH - 1) / kTH) > 65535LL * 65535LL || B > 65535)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "%s: batch too large for one launch", what);
  return UNFLOW_OK;
}

}  // namespace
}  // namespace unflow

using namespace unflow;

extern "C" int unflow_smoothness_deltas(const float *flow, float *delta_u, float *delta_v, float *mask, int B,
                                        int H, int W, int order, unflow_stream_t stream) {
  if (int rc = check_dims("smoothness_deltas", B, H, W)) return rc;
  if (order != 1 && order != 2) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "smoothness_deltas: order must be 1 or 2");
  if ((long long)B * H * W == 0) return UNFLOW_OK;
  hipStream_t s = as_stream(stream);
  if (order == 1)
    hipLaunchKernelGGL(flow_deltas_kernel<1>, tile_grid(B, H, W), dim3(kNT), 0, s, flow, delta_u, delta_v, mask, H, W);
  else
    hipLaunchKernelGGL(flow_deltas_kernel<2>, tile_grid(B, H, W), dim3(kNT), 0, s, flow, delta_u, delta_v, mask, H, W);
  return launch_status("smoothness_deltas");
}

extern "C" int unflow_smoothness_loss_fwd(const float *flow, float *loss, float *workspace, int B, int H, int W,
                                          int order, long long norm_count, unflow_stream_t stream) {
  if (int rc = check_dims("smoothness_loss", B, H, W)) return rc;
  if (order != 1 && order != 2) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "smoothness_loss: order must be 1 or 2");
  hipStream_t s = as_stream(stream);
  const long long npix = (long long)B * H * W;
  if (npix == 0) {
    return zero_loss(loss, "smoothness_loss", s);
  }
  const int K = order == 1 ? 2 : 4;
  const double norm = norm_count > 0 ? (double)norm_count : (double)npix * K;
  if (order == 1)
    hipLaunchKernelGGL(flow_smooth_fwd_kernel<1>, tile_grid(B, H, W), dim3(kNT), 0, s, flow, workspace, H, W);
  else
    hipLaunchKernelGGL(flow_smooth_fwd_kernel<2>, tile_grid(B, H, W), dim3(kNT), 0, s, flow, workspace, H, W);
  if (int rc = launch_status("smoothness_loss_fwd")) return rc;
  return reduce_partials(workspace, n_tiles(B, H, W), 1.0 / norm, loss, s);
}

extern "C" int unflow_smoothness_loss_bwd(const float *flow, const float *gout, float *dflow, int B, int H, int W,
                                          int order, long long norm_count, unflow_stream_t stream) {
  if (int rc = check_dims("smoothness_loss_grad", B, H, W)) return rc;
  if (order != 1 && order != 2) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "smoothness_loss: order must be 1 or 2");
  const long long npix = (long long)B * H * W;
  if (npix == 0) return UNFLOW_OK;
  const int K = order == 1 ? 2 : 4;
  const double norm = norm_count > 0 ? (double)norm_count : (double)npix * K;
  hipStream_t s = as_stream(stream);
  if (order == 1)
    hipLaunchKernelGGL(flow_smooth_bwd_kernel<1>, tile_grid(B, H, W), dim3(kNT), 0, s, flow, gout, dflow, H, W,
                       (float)(1.0 / norm));
  else
    hipLaunchKernelGGL(flow_smooth_bwd_kernel<2>, tile_grid(B, H, W), dim3(kNT), 0, s, flow, gout, dflow, H, W,
                       (float)(1.0 / norm));
  return launch_status("smoothness_loss_bwd");
}

extern "C" int unflow_gradient_delta(const float *im1, const float *im2_warped, float *diff, int B, int H, int W,
                                     unflow_stream_t stream) {
  if (int rc = check_dims("gradient_delta", B, H, W)) return rc;
  if ((long long)B * H * W == 0) return UNFLOW_OK;
  hipLaunchKernelGGL(gradient_delta_kernel, tile_grid(B, H, W), dim3(kNT), 0, as_stream(stream), im1, im2_warped,
                     diff, H, W);
  return launch_status("gradient_delta");
}

extern "C" int unflow_gradient_loss_fwd(const float *im1, const float *im2_warped, const float *mask, float *loss,
                                        float *workspace, int B, int H, int W, long long norm_count,
                                        unflow_stream_t stream) {
  if (int rc = check_dims("gradient_loss", B, H, W)) return rc;
  hipStream_t s = as_stream(stream);
  const long long npix = (long long)B * H * W;
  if (npix == 0) {
    return zero_loss(loss, "gradient_loss", s);
  }
  const double norm = norm_count > 0 ? (double)norm_count : (double)npix * 6;
  hipLaunchKernelGGL(gradient_loss_fwd_kernel, tile_grid(B, H, W), dim3(kNT), 0, s, im1, im2_warped, mask, workspace,
                     H, W);
  if (int rc = launch_status("gradient_loss_fwd")) return rc;
  return reduce_partials(workspace, n_tiles(B, H, W), 1.0 / norm, loss, s);
}

extern "C" int unflow_gradient_loss_bwd(const float *im1, const float *im2_warped, const float *mask,
                                        const float *gout, float *d_im2_warped, int B, int H, int W,
                                        long long norm_count, unflow_stream_t stream) {
  if (int rc = check_dims("gradient_loss_grad", B, H, W)) return rc;
  const long long npix = (long long)B * H * W;
  if (npix == 0) return UNFLOW_OK;
  const double norm = norm_count > 0 ? (double)norm_count : (double)npix * 6;
  hipLaunchKernelGGL(gradient_loss_bwd_kernel, tile_grid(B, H, W), dim3(kNT), 0, as_stream(stream), im1, im2_warped,
                     mask, gout, d_im2_warped, H, W, (float)(1.0 / norm));
  return launch_status("gradient_loss_bwd");
}
