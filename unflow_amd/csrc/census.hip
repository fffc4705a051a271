// Soft ternary-census loss (fused), charbonnier and the mask helpers of
// compute_losses, for gfx950.
//
// Reference (paths relative to the reference checkout):
//   ternary_loss        src/e2eflow/core/losses.py:90-122
//   charbonnier_loss    src/e2eflow/core/losses.py:298-322
//   create_mask         src/e2eflow/core/losses.py:325-335
//   create_outgoing_mask src/e2eflow/core/losses.py:347-366
//   fb occlusion        src/e2eflow/core/losses.py:38-55
//
// The reference materialises [B,H,W,P*P] tensors about eight times per call
// (patches, transf, norm, diff, dist ...).  Here one workgroup stages the two
// grayscale tiles (with an r-pixel halo) in LDS once and every lane walks its
// (2r+1)^2 window out of LDS, so HBM traffic is the 28 B/pixel of the inputs.
// The kernel is VALU-issue bound: per tap 9 f32 ops + 2 v_rsq_f32 + 1 v_rcp_f32.
#include "common.h"
#include "warp_taps.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace unflow {
namespace {

constexpr int kTW = 32, kTH = 8, kNT = kTW * kTH;  // one lane per pixel of a 32x8 tile
constexpr float kAlpha = 0.45f, kEps = 0.001f;     // charbonnier defaults used by ternary_loss

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ inline v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// tf.image.rgb_to_grayscale * 255 (losses.py:94): separately rounded terms.
__device__ inline float gray255(const float *px) {
  float g = mul_rn(px[0], 0.2989f);
  g = add_rn(g, mul_rn(px[1], 0.5870f));
  g = add_rn(g, mul_rn(px[2], 0.1140f));
  return mul_rn(g, 255.0f);
}

// Stage gray tiles of both images (zero outside the image = conv2d SAME).
__device__ inline void stage_gray(const float *__restrict__ im1, const float *__restrict__ im2, float *g1,
                                  float *g2, long long item_px, int H, int W, int x0, int y0, int r, int lw,
                                  int lh) {
  for (int i = threadIdx.x; i < lw * lh; i += kNT) {
    const int ly = i / lw, lx = i - ly * lw;
    const int y = y0 + ly - r, x = x0 + lx - r;
    float a = 0.f, b = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const long long p = (item_px + (long long)y * W + x) * 3;
      a = gray255(im1 + p);
      b = gray255(im2 + p);
    }
    g1[i] = a;
    g2[i] = b;
  }
}

template <int R>
__global__ __launch_bounds__(kNT) void ternary_fwd_kernel(const float *__restrict__ im1,
                                                          const float *__restrict__ im2,
                                                          const float *__restrict__ mask,
                                                          float *__restrict__ partial,
                                                          float *__restrict__ dldd, float *__restrict__ dist_out,
                                                          int H, int W, int r_rt, float inv_norm) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int r = R > 0 ? R : r_rt;
  const int lw = kTW + 2 * r, lh = kTH + 2 * r;
  float *g1 = lds, *g2 = lds + lw * lh;
  const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
  const long long item_px = (long long)blockIdx.z * H * W;
  stage_gray(im1, im2, g1, g2, item_px, H, W, x0, y0, r, lw, lh);
  __syncthreads();

  const int tx = threadIdx.x % kTW, ty = threadIdx.x / kTW;
  const int x = x0 + tx, y = y0 + ty;
  float error = 0.f;
  if (x < W && y < H) {
    const float c1 = g1[(ty + r) * lw + tx + r], c2 = g2[(ty + r) * lw + tx + r];
    float dist_sum;
    if (R > 0) {
      // Two taps per lane-instruction (v_pk_{add,mul,fma}_f32); the three
      // transcendentals per tap stay scalar.  Measured 0.415 -> 0.39 ms (fwd) and
      // 0.47 -> 0.39 ms (bwd) at 32x384x1280: a packed op issues at about the cost
      // of two scalar ones here, the gain is the shorter instruction stream.
      const v2f c1v = {c1, c1}, c2v = {c2, c2};
      v2f acc2 = {0.f, 0.f};
      auto tap2 = [&](v2f a1, v2f a2) {
        const v2f d1 = a1 - c1v, d2 = a2 - c2v;
        const v2f s1 = pk_fma(d1, d1, v2f{0.81f, 0.81f}), s2 = pk_fma(d2, d2, v2f{0.81f, 0.81f});
        const v2f r1 = {__builtin_amdgcn_rsqf(s1.x), __builtin_amdgcn_rsqf(s1.y)};
        const v2f r2 = {__builtin_amdgcn_rsqf(s2.x), __builtin_amdgcn_rsqf(s2.y)};
        const v2f df = pk_fma(d1, r1, -(d2 * r2));
        const v2f e = df * df;
        const v2f den = e + v2f{0.1f, 0.1f};
        const v2f inv = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
        acc2 = pk_fma(e, inv, acc2);
      };
      constexpr int P = 2 * R + 1;
#pragma unroll
      for (int i = 0; i < P; ++i) {
        const float *p1 = g1 + (ty + i) * lw + tx, *p2 = g2 + (ty + i) * lw + tx;
#pragma unroll
        for (int j = 0; j + 1 < P; j += 2) tap2(v2f{p1[j], p1[j + 1]}, v2f{p2[j], p2[j + 1]});
        __builtin_amdgcn_sched_barrier(0);
      }
      // last window column: rows in pairs; the odd one out goes with a zero tap (h(0,0) = 0)
#pragma unroll
      for (int i = 0; i < P; i += 2) {
        const float *p1 = g1 + (ty + i) * lw + tx + P - 1, *p2 = g2 + (ty + i) * lw + tx + P - 1;
        if (i + 1 < P) tap2(v2f{p1[0], p1[lw]}, v2f{p2[0], p2[lw]});
        else tap2(v2f{p1[0], c1}, v2f{p2[0], c2});
      }
      dist_sum = acc2.x + acc2.y;
    } else {
    dist_sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * r + 1; ++i) {
#pragma unroll
      for (int j = 0; j < 2 * r + 1; ++j) {
        const float d1 = g1[(ty + i) * lw + tx + j] - c1;
        const float d2 = g2[(ty + i) * lw + tx + j] - c2;
        const float t1 = d1 * __builtin_amdgcn_rsqf(fmaf(d1, d1, 0.81f));
        const float t2 = d2 * __builtin_amdgcn_rsqf(fmaf(d2, d2, 0.81f));
        const float df = t1 - t2;
        const float e = df * df;
        dist_sum = fmaf(e, __builtin_amdgcn_rcpf(0.1f + e), dist_sum);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads of one window row together (register pressure)
    }
    }
    const long long p = item_px + (long long)y * W + x;
    if (dist_out) dist_out[p] = dist_sum;
    const bool inner = y >= r && y < H - r && x >= r && x < W - r;  // create_mask (losses.py:119-120)
    const float m = inner ? mask[p] : 0.f;
    const float base = add_rn(mul_rn(dist_sum, dist_sum), kEps * kEps);   // tf.square(x) + tf.square(eps)
    const float pw = powf(base, kAlpha);
    error = m * pw;
    if (dldd) dldd[p] = m * kAlpha * (pw / base) * 2.f * dist_sum * inv_norm;
  }
  __shared__ float red[kNT / 64];
  const float s = block_sum<kNT>(error, red);
  if (threadIdx.x == 0) partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
}

// d loss / d im2_warped.  With G(d1,d2) = dh/dt2 * dt2/dd2, the gradient of the
// gray image at q is  -sum_off (dldd[q] + dldd[q+off]) * G(I1[q+off]-I1[q], I2[q+off]-I2[q])
// (the "q is the centre" and "q is a neighbour of q+off" terms share G up to
// sign because the window is symmetric and t is odd in d).
template <int R>
__global__ __launch_bounds__(kNT) void ternary_bwd_kernel(const float *__restrict__ im1,
                                                          const float *__restrict__ im2,
                                                          const float *__restrict__ dldd,
                                                          const float *__restrict__ gout,
                                                          float *__restrict__ d_im2, int H, int W, int r_rt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int r = R > 0 ? R : r_rt;
  const int lw = kTW + 2 * r, lh = kTH + 2 * r;
  float *g1 = lds, *g2 = lds + lw * lh, *gd = lds + 2 * lw * lh;
  const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
  const long long item_px = (long long)blockIdx.z * H * W;
  stage_gray(im1, im2, g1, g2, item_px, H, W, x0, y0, r, lw, lh);
  for (int i = threadIdx.x; i < lw * lh; i += kNT) {
    const int ly = i / lw, lx = i - ly * lw;
    const int y = y0 + ly - r, x = x0 + lx - r;
    gd[i] = (y >= 0 && y < H && x >= 0 && x < W) ? dldd[item_px + (long long)y * W + x] : 0.f;
  }
  __syncthreads();

  const int tx = threadIdx.x % kTW, ty = threadIdx.x / kTW;
  const int x = x0 + tx, y = y0 + ty;
  if (x >= W || y >= H) return;
  const int cidx = (ty + r) * lw + tx + r;
  const float c1 = g1[cidx], c2 = g2[cidx], cd = gd[cidx];
  float acc;
  if (R > 0) {
    const v2f c1v = {c1, c1}, c2v = {c2, c2}, cdv = {cd, cd};
    v2f acc2 = {0.f, 0.f};
    auto tap2 = [&](v2f a1, v2f a2, v2f ad) {
      const v2f w = cdv + ad;
      const v2f d1 = a1 - c1v, d2 = a2 - c2v;
      const v2f s1 = pk_fma(d1, d1, v2f{0.81f, 0.81f}), s2 = pk_fma(d2, d2, v2f{0.81f, 0.81f});
      const v2f r1 = {__builtin_amdgcn_rsqf(s1.x), __builtin_amdgcn_rsqf(s1.y)};
      const v2f r2 = {__builtin_amdgcn_rsqf(s2.x), __builtin_amdgcn_rsqf(s2.y)};
      const v2f df = pk_fma(d1, r1, -(d2 * r2));
      const v2f den = pk_fma(df, df, v2f{0.1f, 0.1f});
      const v2f inv = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
      const v2f q = inv * r2;
      acc2 = pk_fma(w * df, (q * q) * r2, acc2);
    };
    constexpr int P = 2 * R + 1;
#pragma unroll 1
    for (int i = 0; i < P; ++i) {
      const int n = (ty + i) * lw + tx;
#pragma unroll
      for (int j = 0; j + 1 < P; j += 2)
        tap2(v2f{g1[n + j], g1[n + j + 1]}, v2f{g2[n + j], g2[n + j + 1]}, v2f{gd[n + j], gd[n + j + 1]});
    }
#pragma unroll 1
    for (int i = 0; i + 1 < P; i += 2) {
      const int n = (ty + i) * lw + tx + P - 1;
      tap2(v2f{g1[n], g1[n + lw]}, v2f{g2[n], g2[n + lw]}, v2f{gd[n], gd[n + lw]});
    }
    {
      const int n = (ty + P - 1) * lw + tx + P - 1;
      tap2(v2f{g1[n], c1}, v2f{g2[n], c2}, v2f{gd[n], 0.f});   // second tap: d = 0 -> df = 0 -> no contribution
    }
    acc = acc2.x + acc2.y;
  } else {
  acc = 0.f;
  // The row loop is kept rolled: fully unrolled, all 147 LDS reads of the window
  // are hoisted to the top (157 VGPRs, 3 waves per SIMD instead of 8).
#pragma unroll 1
  for (int i = 0; i < 2 * r + 1; ++i) {
#pragma unroll
    for (int j = 0; j < 2 * r + 1; ++j) {
      const int n = (ty + i) * lw + tx + j;
      const float w = cd + gd[n];
      const float d1 = g1[n] - c1;
      const float d2 = g2[n] - c2;
      const float r1 = __builtin_amdgcn_rsqf(fmaf(d1, d1, 0.81f));
      const float r2 = __builtin_amdgcn_rsqf(fmaf(d2, d2, 0.81f));
      const float df = d1 * r1 - d2 * r2;
      const float inv = __builtin_amdgcn_rcpf(fmaf(df, df, 0.1f));
      // G = -0.2*df*inv^2 * 0.81*r2^3 ; acc -= w*G
      const float q = inv * r2;
      acc = fmaf(w * df, (q * q) * r2, acc);
    }
  }
  }
  const float go = gout ? *gout : 1.f;
  const float dgray = acc * (0.2f * 0.81f) * go;
  float *out = d_im2 + (item_px + (long long)y * W + x) * 3;
  out[0] = 255.f * 0.2989f * dgray;
  out[1] = 255.f * 0.5870f * dgray;
  out[2] = 255.f * 0.1140f * dgray;
}

// ---------------------------------------------------------------------------
// Pair-symmetric kernels.  The tap (p, off) and the tap (p+off, -off) see the
// same two pixels with the difference negated, and t is odd in d, so
//   h(p, off) = h(p+off, -off)          (forward,  e = (t1-t2)^2 is even)
//   w*G(p, off) = -w*G(p+off, -off)     (backward, w = dldd[p]+dldd[p+off], G odd)
// -- also across the image border, because the zero padding of conv2d SAME makes
// an outside position an ordinary pixel of intensity 0.  Each unordered pair is
// therefore evaluated ONCE, by its upper/left pixel, for the half plane of
// offsets (dy = 0, dx > 0) and (dy > 0): half the transcendentals and half the
// arithmetic.  A lane owns one image column and walks down the rows of its
// wave's strip; the value of a pair is added to the lane's own pixel and handed
// to the partner pixel (x+dx, y+dy): across columns with a lane permute
// (ds_bpermute, no memory), across rows through R+1 running sums in registers.
// Each wave spends R warm-up rows above its strip and R halo lanes on either
// side (they only hand values over), 1.3x the pixels for half the taps.
// Measured at 32x384x1280, 7x7: forward 0.45 -> 0.35 ms, backward 0.42 -> 0.38 ms.
// Used for max_distance 2 and 3; the one-lane-per-pixel kernels above serve the rest.
// ---------------------------------------------------------------------------
constexpr int kPairWaves = 4;                       // waves of a workgroup, stacked vertically

template <int R, int ROWS>
struct HalfPlane {
  static constexpr int N = 2 * R * R + 2 * R;       // offsets in the half plane (even)
  static constexpr int dy(int k) { return k < R ? 0 : 1 + (k - R) / (2 * R + 1); }
  static constexpr int dx(int k) { return k < R ? k + 1 : (k - R) % (2 * R + 1) - R; }
  static constexpr int LW = 64 + 2 * R;             // LDS tile: columns x0-2R .. x0+63
  static constexpr int LH = kPairWaves * ROWS + 2 * R;   // rows y0-R .. y0+4*ROWS+R-1
  static constexpr int COLS = 64 - 2 * R;           // image columns a workgroup produces
};

// value held by lane (lane - dx); lanes without a source get their own (never used)
__device__ inline float lane_shift(float v, int lane, int dx) {
  const int src = min(max(lane - dx, 0), 63);
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, v)));
}

template <int R, int ROWS>
__device__ inline void stage_gray_pair(const float *__restrict__ im1, const float *__restrict__ im2, float *g1,
                                       float *g2, long long item_px, int H, int W, int x0, int y0) {
  using HP = HalfPlane<R, ROWS>;
  for (int i = threadIdx.x; i < HP::LW * HP::LH; i += 64 * kPairWaves) {
    const int ly = i / HP::LW, lx = i - ly * HP::LW;
    const int y = y0 + ly - R, x = x0 + lx - 2 * R;
    float a = 0.f, b = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const long long p = (item_px + (long long)y * W + x) * 3;
      a = gray255(im1 + p);
      b = gray255(im2 + p);
    }
    g1[i] = a;
    g2[i] = b;
  }
}

// Staging for ternary_loss(im1, image_warp(im2, flow)): the warp (core/image_warp.py order of
// operations, as image_warp_fwd_kernel<3>) is evaluated per tile cell and goes straight into
// the gray LDS image; the warped RGB image never exists.  The two gray values of the block's
// OWN pixels are also written out (2 x 4 B/pixel) for the backward pass.
template <int R, int ROWS>
__device__ inline void stage_gray_pair_warped(const float *__restrict__ im1, const float *__restrict__ im2,
                                              const float2 *__restrict__ flow, float *g1, float *g2,
                                              float *__restrict__ gray1_out, float *__restrict__ gray2_out,
                                              long long item_px, int H, int W, int x0, int y0) {
  using HP = HalfPlane<R, ROWS>;
  const float *base = im2 + item_px * 3;
  for (int i = threadIdx.x; i < HP::LW * HP::LH; i += 64 * kPairWaves) {
    const int ly = i / HP::LW, lx = i - ly * HP::LW;
    const int y = y0 + ly - R, x = x0 + lx - 2 * R;
    float a = 0.f, b = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const long long p = item_px + (long long)y * W + x;
      a = gray255(im1 + p * 3);
      const ClampTaps t = clamp_taps(x, y, flow[p], H, W);
      const float *pa = base + t.ia * 3, *pb = base + t.ib * 3, *pc = base + t.ic * 3, *pd = base + t.id * 3;
      float la[3], lb[3], lc[3], ld[3], wpx[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { la[c] = pa[c]; lb[c] = pb[c]; lc[c] = pc[c]; ld[c] = pd[c]; }
      const float wa = (1.f - t.xw) * (1.f - t.yw), wb = (1.f - t.xw) * t.yw;
      const float wc = t.xw * (1.f - t.yw), wd = t.xw * t.yw;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = mul_rn(wa, la[c]);
        v = add_rn(v, mul_rn(wb, lb[c]));
        v = add_rn(v, mul_rn(wc, lc[c]));
        v = add_rn(v, mul_rn(wd, ld[c]));
        wpx[c] = v;
      }
      b = gray255(wpx);
      if (lx >= 2 * R && lx < 2 * R + HP::COLS && ly >= R && ly < R + kPairWaves * ROWS) {   // the block's own pixel
        gray1_out[p] = a;
        gray2_out[p] = b;
      }
    }
    g1[i] = a;
    g2[i] = b;
  }
}

// Staging from the two gray planes the warped forward wrote (zero outside the image)
template <int R, int ROWS>
__device__ inline void stage_gray_planes(const float *__restrict__ gray1, const float *__restrict__ gray2, float *g1,
                                         float *g2, long long item_px, int H, int W, int x0, int y0) {
  using HP = HalfPlane<R, ROWS>;
  for (int i = threadIdx.x; i < HP::LW * HP::LH; i += 64 * kPairWaves) {
    const int ly = i / HP::LW, lx = i - ly * HP::LW;
    const int y = y0 + ly - R, x = x0 + lx - 2 * R;
    const bool in = y >= 0 && y < H && x >= 0 && x < W;
    const long long p = item_px + (long long)y * W + x;
    g1[i] = in ? gray1[p] : 0.f;
    g2[i] = in ? gray2[p] : 0.f;
  }
}

template <int K, int N, class F>
__device__ inline void static_for_pairs(F &&f) {
  if constexpr (K < N) {
    f(std::integral_constant<int, K>{});
    static_for_pairs<K + 2, N>(f);
  }
}

// WARP: im2 is the UNWARPED image and flow is given: im2_warped = image_warp(im2, flow) is
// evaluated while staging (see stage_gray_pair_warped) and the gray planes are written out.
template <int R, int ROWS, bool WARP = false>
__global__ __launch_bounds__(64 * kPairWaves) void ternary_fwd_pair_kernel(
    const float *__restrict__ im1, const float *__restrict__ im2, const float *__restrict__ mask,
    float *__restrict__ partial, float *__restrict__ dldd, float *__restrict__ dist_out, int H, int W,
    float inv_norm, const float2 *__restrict__ flow = nullptr, float *__restrict__ gray1_out = nullptr,
    float *__restrict__ gray2_out = nullptr) {
  using HP = HalfPlane<R, ROWS>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *g1 = lds, *g2 = lds + HP::LW * HP::LH;
  const int x0 = blockIdx.x * HP::COLS, y0 = blockIdx.y * (kPairWaves * ROWS);
  const long long item_px = (long long)blockIdx.z * H * W;
  if constexpr (WARP) stage_gray_pair_warped<R, ROWS>(im1, im2, flow, g1, g2, gray1_out, gray2_out, item_px, H, W, x0, y0);
  else stage_gray_pair<R, ROWS>(im1, im2, g1, g2, item_px, H, W, x0, y0);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = x0 - R + lane;                       // this lane's image column
  const bool col_ok = lane >= R && lane < 64 - R && x < W;
  float ring[R + 1];
#pragma unroll
  for (int i = 0; i <= R; ++i) ring[i] = 0.f;
  float error = 0.f;
#pragma unroll 1
  for (int it = 0; it < ROWS + R; ++it) {
    const int ly = wave * ROWS + it;            // tile row of this pixel (row 0 = y0 - R)
    const int y = y0 - R + ly;
    const float *p1 = g1 + ly * HP::LW + lane + R, *p2 = g2 + ly * HP::LW + lane + R;
    const float c1 = p1[0], c2 = p2[0];
    const v2f c1v = {c1, c1}, c2v = {c2, c2};
    v2f own = {0.f, 0.f};
    static_for_pairs<0, HP::N>([&](auto K) {
      constexpr int ka = decltype(K)::value, kb = ka + 1;
      constexpr int oa = HP::dy(ka) * HP::LW + HP::dx(ka), ob = HP::dy(kb) * HP::LW + HP::dx(kb);
      const v2f d1 = v2f{p1[oa], p1[ob]} - c1v, d2 = v2f{p2[oa], p2[ob]} - c2v;
      const v2f s1 = pk_fma(d1, d1, v2f{0.81f, 0.81f}), s2 = pk_fma(d2, d2, v2f{0.81f, 0.81f});
      const v2f r1 = {__builtin_amdgcn_rsqf(s1.x), __builtin_amdgcn_rsqf(s1.y)};
      const v2f r2 = {__builtin_amdgcn_rsqf(s2.x), __builtin_amdgcn_rsqf(s2.y)};
      const v2f df = pk_fma(d1, r1, -(d2 * r2));
      const v2f e = df * df;
      const v2f den = e + v2f{0.1f, 0.1f};
      const v2f h = e * v2f{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
      own += h;
      ring[HP::dy(ka)] += HP::dx(ka) == 0 ? h.x : lane_shift(h.x, lane, HP::dx(ka));
      ring[HP::dy(kb)] += HP::dx(kb) == 0 ? h.y : lane_shift(h.y, lane, HP::dx(kb));
    });
    const float dist_sum = ring[0] + (own.x + own.y);
#pragma unroll
    for (int i = 0; i < R; ++i) ring[i] = ring[i + 1];
    ring[R] = 0.f;
    if (it >= R && col_ok && y < H) {
      const long long p = item_px + (long long)y * W + x;
      if (dist_out) dist_out[p] = dist_sum;
      const bool inner = y >= R && y < H - R && x >= R && x < W - R;  // create_mask (losses.py:119-120)
      const float m = inner ? mask[p] : 0.f;
      const float base = add_rn(mul_rn(dist_sum, dist_sum), kEps * kEps);
      // hardware log2 / exp2 / rcp: powf and an IEEE division here were a third of the
      // kernel's VALU instructions (once per row and lane, against 12 packed pairs)
      const float pw = pow_pos(base, kAlpha);
      error += m * pw;
      if (dldd) dldd[p] = m * kAlpha * (pw * __builtin_amdgcn_rcpf(base)) * 2.f * dist_sum * inv_norm;
    }
  }
  __shared__ float red[kPairWaves];
  const float s = block_sum<64 * kPairWaves>(error, red);
  if (threadIdx.x == 0) partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
}

// FUSE: im2 is image_warp(im2_src, flow) and only d loss / d flow is wanted: the gradient
// of the warp (core/image_warp.py autodiff, as image_warp_bwd_kernel) is applied to each
// pixel's d_im2 = 255 * luma * dgray while it is still in registers, so the [B,H,W,3]
// gradient image is neither written nor read back and one launch disappears.
template <int R, int ROWS, bool FUSE = false>
__global__ __launch_bounds__(64 * kPairWaves) void ternary_bwd_pair_kernel(
    const float *__restrict__ im1, const float *__restrict__ im2, const float *__restrict__ dldd,
    const float *__restrict__ gout, float *__restrict__ d_im2, int H, int W,
    const float *__restrict__ im2_src = nullptr, const float2 *__restrict__ flow = nullptr,
    float2 *__restrict__ d_flow = nullptr) {
  using HP = HalfPlane<R, ROWS>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *g1 = lds, *g2 = lds + HP::LW * HP::LH;
  const int x0 = blockIdx.x * HP::COLS, y0 = blockIdx.y * (kPairWaves * ROWS);
  const long long item_px = (long long)blockIdx.z * H * W;
  // FUSE: im1 / im2 are the two GRAY planes written by the warped forward
  if constexpr (FUSE) stage_gray_planes<R, ROWS>(im1, im2, g1, g2, item_px, H, W, x0, y0);
  else stage_gray_pair<R, ROWS>(im1, im2, g1, g2, item_px, H, W, x0, y0);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = x0 - R + lane;
  const bool col_ok = lane >= R && lane < 64 - R && x < W;
  const float go = (gout ? *gout : 1.f) * (0.2f * 0.81f);
  // dldd is read straight from HBM/L2 with LINEAR addressing: it is zero within R of
  // every image border (create_mask of the forward), so a tap that leaves the image
  // lands, through the row / item wrap-around, on a zero, and reads outside the
  // buffer return 0 through the descriptor's range check.  Only the centre value of
  // a lane that is itself outside the image has to be forced to zero.
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(dldd), (short)0, (int)((long long)gridDim.z * H * W * 4), 0x00020000);
  const int rowb = W * 4;
  float ring[R + 1];
#pragma unroll
  for (int i = 0; i <= R; ++i) ring[i] = 0.f;
#pragma unroll 1
  for (int it = 0; it < ROWS + R; ++it) {
    const int ly = wave * ROWS + it;
    const int y = y0 - R + ly;
    const int c = ly * HP::LW + lane + R;
    const float *p1 = g1 + c, *p2 = g2 + c;
    const int pb = (int)((item_px + (long long)y * W + x) * 4);     // byte offset of (y, x); may be negative
    // the whole offset goes into the VGPR operand: the descriptor's range check covers the
    // vector and immediate offsets only, not an SGPR offset
    auto gdl = [&](int dy, int dx) {
      return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, pb + dy * rowb + dx * 4, 0, 0));
    };
    const bool inside = x >= 0 && x < W && y >= 0 && y < H;
    const float c1 = p1[0], c2 = p2[0], cd = inside ? gdl(0, 0) : 0.f;
    const v2f c1v = {c1, c1}, c2v = {c2, c2}, cdv = {cd, cd};
    v2f own = {0.f, 0.f};
    static_for_pairs<0, HP::N>([&](auto K) {
      constexpr int ka = decltype(K)::value, kb = ka + 1;
      constexpr int oa = HP::dy(ka) * HP::LW + HP::dx(ka), ob = HP::dy(kb) * HP::LW + HP::dx(kb);
      const v2f w = cdv + v2f{gdl(HP::dy(ka), HP::dx(ka)), gdl(HP::dy(kb), HP::dx(kb))};
      const v2f d1 = v2f{p1[oa], p1[ob]} - c1v, d2 = v2f{p2[oa], p2[ob]} - c2v;
      const v2f s1 = pk_fma(d1, d1, v2f{0.81f, 0.81f}), s2 = pk_fma(d2, d2, v2f{0.81f, 0.81f});
      const v2f r1 = {__builtin_amdgcn_rsqf(s1.x), __builtin_amdgcn_rsqf(s1.y)};
      const v2f r2 = {__builtin_amdgcn_rsqf(s2.x), __builtin_amdgcn_rsqf(s2.y)};
      const v2f df = pk_fma(d1, r1, -(d2 * r2));
      const v2f den = pk_fma(df, df, v2f{0.1f, 0.1f});
      const v2f inv = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
      const v2f q = inv * r2;
      const v2f t = (w * df) * ((q * q) * r2);       // = -w*G / (0.2*0.81): added at p, subtracted at p+off
      own += t;
      ring[HP::dy(ka)] -= HP::dx(ka) == 0 ? t.x : lane_shift(t.x, lane, HP::dx(ka));
      ring[HP::dy(kb)] -= HP::dx(kb) == 0 ? t.y : lane_shift(t.y, lane, HP::dx(kb));
    });
    const float acc = ring[0] + (own.x + own.y);
#pragma unroll
    for (int i = 0; i < R; ++i) ring[i] = ring[i + 1];
    ring[R] = 0.f;
    if (it >= R && col_ok && y < H) {
      const float dgray = acc * go;
      const float gv[3] = {255.f * 0.2989f * dgray, 255.f * 0.5870f * dgray, 255.f * 0.1140f * dgray};
      const long long p = item_px + (long long)y * W + x;
      if constexpr (FUSE) {
        const ClampTaps t = clamp_taps(x, y, flow[p], H, W);
        const float *base = im2_src + item_px * 3;
        const float *pa = base + t.ia * 3, *pb = base + t.ib * 3, *pc = base + t.ic * 3, *pd = base + t.id * 3;
        float la[3], lb[3], lc[3], ld[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { la[c] = pa[c]; lb[c] = pb[c]; lc[c] = pc[c]; ld[c] = pd[c]; }
        float ga = 0.f, gb = 0.f, gc = 0.f, gd = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          ga = fmaf(gv[c], la[c], ga);
          gb = fmaf(gv[c], lb[c], gb);
          gc = fmaf(gv[c], lc[c], gc);
          gd = fmaf(gv[c], ld[c], gd);
        }
        d_flow[p] = make_float2((gc - ga) * (1.f - t.yw) + (gd - gb) * t.yw,
                                (gb - ga) * (1.f - t.xw) + (gd - gc) * t.xw);
      } else {
        float *out = d_im2 + p * 3;
        out[0] = gv[0];
        out[1] = gv[1];
        out[2] = gv[2];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Integer ternary census + Hamming distance.  NOT in the
// reference, whose census is the soft real-valued one above (D1); defined here as
// the classical ternary transform on the same gray image I = 255*gray(im), same
// (2r+1)^2 window, same zero padding: neighbour k (row-major, centre skipped) is
//   brighter  if I[q] - I[p] >  eps      -> bit k of word 0
//   darker    if I[q] - I[p] < -eps      -> bit k of word 1
// and the distance of two signatures is the number of neighbours whose ternary
// digit differs: popcount((b1^b2) | (d1^d2)).  All of it is compares and integer
// arithmetic on bit-exact gray values (separately rounded products), so the
// results are bit-exact against the CPU oracle.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kNT) void census_transform_kernel(const float *__restrict__ im,
                                                               unsigned long long *__restrict__ sig, int H, int W,
                                                               int r, float eps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lw = kTW + 2 * r, lh = kTH + 2 * r;
  const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
  const long long item_px = (long long)blockIdx.z * H * W;
  for (int i = threadIdx.x; i < lw * lh; i += kNT) {
    const int ly = i / lw, lx = i - ly * lw;
    const int y = y0 + ly - r, x = x0 + lx - r;
    lds[i] = (y >= 0 && y < H && x >= 0 && x < W) ? gray255(im + (item_px + (long long)y * W + x) * 3) : 0.f;
  }
  __syncthreads();
  const int tx = threadIdx.x % kTW, ty = threadIdx.x / kTW;
  const int x = x0 + tx, y = y0 + ty;
  if (x >= W || y >= H) return;
  const float c = lds[(ty + r) * lw + tx + r];
  unsigned long long brighter = 0, darker = 0;
  int k = 0;
  for (int i = 0; i < 2 * r + 1; ++i)
    for (int j = 0; j < 2 * r + 1; ++j) {
      if (i == r && j == r) continue;
      const float d = add_rn(lds[(ty + i) * lw + tx + j], -c);
      brighter |= (unsigned long long)(d > eps) << k;
      darker |= (unsigned long long)(d < -eps) << k;
      ++k;
    }
  const long long p = item_px + (long long)y * W + x;
  sig[2 * p] = brighter;
  sig[2 * p + 1] = darker;
}

// per-pixel distance + the masked total; the total is integer, so the wave / block /
// grid reduction (DPP-free shuffles, then one 64-bit atomic per block) is exact in any order
__global__ __launch_bounds__(256) void census_hamming_kernel(const ulonglong2 *__restrict__ s1,
                                                             const ulonglong2 *__restrict__ s2,
                                                             const float *__restrict__ mask,
                                                             int *__restrict__ dist,
                                                             unsigned long long *__restrict__ total, long long n) {
  unsigned sum = 0;
  for (long long p = blockIdx.x * 256LL + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
    const ulonglong2 a = s1[p], b = s2[p];
    const int d = __popcll((a.x ^ b.x) | (a.y ^ b.y));
    if (dist) dist[p] = d;
    if (!mask || mask[p] != 0.f) sum += (unsigned)d;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, kWave);   // 64-lane reduction
  __shared__ unsigned red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0 && total) atomicAdd(total, (unsigned long long)red[0] + red[1] + red[2] + red[3]);
}

// Final pass of every loss reduction: one block, fixed order, double accumulate.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float *__restrict__ partial, int n,
                                                               double inv_norm, float *__restrict__ loss) {
  __shared__ double sh[1024];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += (double)partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = (float)(sh[0] * inv_norm);
}

__device__ inline float charbonnier_term(float xr, float m, float alpha, float beta, float eps, float truncate) {
  const float xv = xr * beta;
  float e = pow_pos(add_rn(mul_rn(xv, xv), eps * eps), alpha) * m;
  if (truncate > 0.f) e = fminf(e, truncate);
  return e;
}

__global__ __launch_bounds__(256) void charbonnier_fwd_kernel(const float *__restrict__ x,
                                                              const float *__restrict__ mask, int mask_channels,
                                                              float *__restrict__ partial, long long n, int C,
                                                              float alpha, float beta, float eps, float truncate) {
  float s = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float m = mask ? mask[mask_channels == 1 ? i / C : i] : 1.f;
    s += charbonnier_term(x[i], m, alpha, beta, eps, truncate);
  }
  __shared__ float red[4];
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void charbonnier_bwd_kernel(const float *__restrict__ x,
                                                              const float *__restrict__ mask, int mask_channels,
                                                              const float *__restrict__ gout,
                                                              float *__restrict__ dx, long long n, int C, float alpha,
                                                              float beta, float eps, float truncate, float inv_norm) {
  const float go = (gout ? *gout : 1.f) * inv_norm;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float m = mask ? mask[mask_channels == 1 ? i / C : i] : 1.f;
    const float xr = x[i];
    const float xv = xr * beta;
    const float base = add_rn(mul_rn(xv, xv), eps * eps);
    float g = alpha * pow_pos(base, alpha - 1.f) * 2.f * beta * beta * xr * m;
    if (truncate > 0.f && m * pow_pos(base, alpha) > truncate) g = 0.f;
    dx[i] = go * g;
  }
}

__global__ __launch_bounds__(256) void outgoing_mask_kernel(const float2 *__restrict__ flow,
                                                            float *__restrict__ mask, int H, int W, long long n) {
  for (long long p = blockIdx.x * 256LL + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
    const int gx = (int)(p % W), gy = (int)((p / W) % H);
    const float2 f = flow[p];
    const float pos_x = (float)gx + f.x, pos_y = (float)gy + f.y;
    const bool inside = pos_x <= (float)(W - 1) && pos_x >= 0.f && pos_y <= (float)(H - 1) && pos_y >= 0.f;
    mask[p] = inside ? 1.f : 0.f;
  }
}

__global__ __launch_bounds__(256) void create_mask_kernel(float *__restrict__ mask, int H, int W, int pt, int pb,
                                                          int pl, int pr, long long n) {
  for (long long p = blockIdx.x * 256LL + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
    const int x = (int)(p % W), y = (int)((p / W) % H);
    mask[p] = (y >= pt && y < H - pb && x >= pl && x < W - pr) ? 1.f : 0.f;
  }
}

// length_sq test of compute_losses (losses.py:43-51) on one pixel; m = the flow whose
// magnitude enters the threshold (the warped one there, the unwarped one in occlusion(), :126)
__device__ inline float fb_occluded(float2 f, float2 w, float2 m, float &du, float &dv) {
  du = f.x + w.x;
  dv = f.y + w.y;
  // length_sq = reduce_sum(square(x), 3): separately rounded squares (losses.py:12-13)
  const float mag_sq = add_rn(add_rn(mul_rn(f.x, f.x), mul_rn(f.y, f.y)), add_rn(mul_rn(m.x, m.x), mul_rn(m.y, m.y)));
  const float thresh = add_rn(mul_rn(0.01f, mag_sq), 0.5f);
  const float len = add_rn(mul_rn(du, du), mul_rn(dv, dv));
  return len > thresh ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void fb_occlusion_kernel(const float2 *__restrict__ flow,
                                                           const float2 *__restrict__ other,
                                                           float2 *__restrict__ flow_diff, float *__restrict__ occ,
                                                           const float *mask_in, float *mask_out, long long n) {
  for (long long p = blockIdx.x * 256LL + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
    float du, dv;
    const float o = fb_occluded(flow[p], other[p], other[p], du, dv);
    if (flow_diff) flow_diff[p] = make_float2(du, dv);
    if (occ) occ[p] = o;
    if (mask_out) mask_out[p] = mask_in[p] * (1.f - o);      // (mask_in may be mask_out: read, then written, by the same lane)
  }
}

// The same pass with the two scalars that only need what it has in registers (losses.py:57-58, 62-67):
//   'fb'  = charbonnier_loss(flow_diff, mask)   and   'occ' = charbonnier_loss(1 - mask)
// with mask = mask_in * (1 - occ) -- the per-workgroup partial sums of both, instead of two more passes over the
// difference and the mask (a loss pyramid's coarse levels are bound by their launch count).
__global__ __launch_bounds__(256) void fb_occlusion_losses_kernel(const float2 *__restrict__ flow,
                                                                  const float2 *__restrict__ other,
                                                                  float2 *__restrict__ flow_diff, float *__restrict__ occ,
                                                                  const float *mask_in, float *mask_out,
                                                                  float *__restrict__ partial_fb,
                                                                  float *__restrict__ partial_occ, long long n) {
  float s_fb = 0.f, s_occ = 0.f;
  for (long long p = blockIdx.x * 256LL + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
    float du, dv;
    const float o = fb_occluded(flow[p], other[p], other[p], du, dv);
    const float m = mask_in[p] * (1.f - o);
    flow_diff[p] = make_float2(du, dv);
    if (occ) occ[p] = o;
    mask_out[p] = m;
    s_fb += charbonnier_term(du, m, kAlpha, 1.f, kEps, 0.f);
    s_fb += charbonnier_term(dv, m, kAlpha, 1.f, kEps, 0.f);
    s_occ += charbonnier_term(1.f - m, 1.f, kAlpha, 1.f, kEps, 0.f);
  }
  __shared__ float red[4];
  s_fb = block_sum<256>(s_fb, red);
  __syncthreads();
  s_occ = block_sum<256>(s_occ, red);
  if (threadIdx.x == 0) {
    partial_fb[blockIdx.x] = s_fb;
    partial_occ[blockIdx.x] = s_occ;
  }
}

// The same with flow_other_warped = image_warp(flow_other, flow) computed on the fly
// (core/image_warp.py order of operations, as image_warp_fwd_kernel<2>): when only the
// occlusion mask is wanted the warped flow never goes to HBM (52 -> 24 B/pixel).
// LOSSES: the flow difference is written too and the 'fb' / 'occ' partial sums of fb_occlusion_losses_kernel are left behind.
template <bool LOSSES>
__global__ __launch_bounds__(256) void fb_occlusion_warp_kernel(const float2 *__restrict__ flow,
                                                                const float2 *__restrict__ flow_other,
                                                                float2 *__restrict__ warped_out,
                                                                float *__restrict__ occ,
                                                                const float *mask_in, float *mask_out, int H,
                                                                int W, int unwarped_magnitude, long long n,
                                                                float2 *__restrict__ diff_out,
                                                                float *__restrict__ partial_fb,
                                                                float *__restrict__ partial_occ) {
  float s_fb = 0.f, s_occ = 0.f;
  for (long long v = blockIdx.x * 256LL + threadIdx.x; v < n; v += (long long)gridDim.x * 256) {
    const TiledPixel tp = tiled_pixel(v, H, W);   // n = tiled_total(B, H, W): 64x4 pixel tiles
    if (!tp.ok) continue;
    const long long p = tp.idx;
    const int x = tp.x, y = tp.y;
    const float2 f = flow[p];
    const ClampTaps t = clamp_taps(x, y, f, H, W);
    const float2 *base = flow_other + tp.b * H * W;
    const float2 a = base[t.ia], b = base[t.ib], c = base[t.ic], d = base[t.id];
    const float wa = (1.f - t.xw) * (1.f - t.yw), wb = (1.f - t.xw) * t.yw;
    const float wc = t.xw * (1.f - t.yw), wd = t.xw * t.yw;
    float2 w;
    w.x = add_rn(add_rn(add_rn(mul_rn(wa, a.x), mul_rn(wb, b.x)), mul_rn(wc, c.x)), mul_rn(wd, d.x));
    w.y = add_rn(add_rn(add_rn(mul_rn(wa, a.y), mul_rn(wb, b.y)), mul_rn(wc, c.y)), mul_rn(wd, d.y));
    float du, dv;
    const float o = fb_occluded(f, w, unwarped_magnitude ? flow_other[p] : w, du, dv);
    if (warped_out) warped_out[p] = w;
    if (occ) occ[p] = o;
    if constexpr (LOSSES) {
      const float m = mask_in[p] * (1.f - o);
      mask_out[p] = m;
      diff_out[p] = make_float2(du, dv);
      s_fb += charbonnier_term(du, m, kAlpha, 1.f, kEps, 0.f);
      s_fb += charbonnier_term(dv, m, kAlpha, 1.f, kEps, 0.f);
      s_occ += charbonnier_term(1.f - m, 1.f, kAlpha, 1.f, kEps, 0.f);
    } else {
      if (mask_out) mask_out[p] = mask_in[p] * (1.f - o);     // mask_in may be mask_out (in place)
    }
  }
  if constexpr (LOSSES) {
    __shared__ float red[4];
    s_fb = block_sum<256>(s_fb, red);
    __syncthreads();
    s_occ = block_sum<256>(s_occ, red);
    if (threadIdx.x == 0) {
      partial_fb[blockIdx.x] = s_fb;
      partial_occ[blockIdx.x] = s_occ;
    }
  }
}

// Tile form of the kernel above (see warp.hip, "Fast paths"): a 256-thread block owns a 64 x 8 pixel
// tile, a lane two pixels of one column; both flows are loaded together, then all eight taps, and the
// tile decomposition is done once per block in 32-bit arithmetic.  Same arithmetic per pixel.
constexpr int kOccPPL = 2, kOccTileH = kTileH * kOccPPL;
__global__ __launch_bounds__(256) void fb_occlusion_warp_tile_kernel(const float2 *__restrict__ flow,
                                                                     const float2 *__restrict__ flow_other,
                                                                     float2 *__restrict__ warped_out,
                                                                     float *__restrict__ occ, const float *mask_in,
                                                                     float *mask_out, int H, int W, int ntx, int nty,
                                                                     int unwarped_magnitude) {
  const int tile = blockIdx.x;
  const int tx = tile % ntx, ty = (tile / ntx) % nty, b = tile / (ntx * nty);
  const int x = tx * kTileW + (threadIdx.x & (kTileW - 1));
  const int y0 = ty * kOccTileH + (threadIdx.x >> 6);
  const long long item = (long long)b * H * W;
  const float2 *fl = flow + item, *base = flow_other + item;
  bool ok[kOccPPL];
  unsigned pix[kOccPPL];
  float2 f[kOccPPL];
#pragma unroll
  for (int i = 0; i < kOccPPL; ++i) {
    const int y = y0 + kTileH * i;
    ok[i] = x < W && y < H;
    pix[i] = (unsigned)(min(y, H - 1) * W + min(x, W - 1));
    f[i] = fl[pix[i]];
  }
  float2 ta[kOccPPL], tb[kOccPPL], tc[kOccPPL], td[kOccPPL], fo[kOccPPL];
  float xw[kOccPPL], yw[kOccPPL];
#pragma unroll
  for (int i = 0; i < kOccPPL; ++i) {
    const ClampTaps t = clamp_taps(min(x, W - 1), min(y0 + kTileH * i, H - 1), f[i], H, W);
    xw[i] = t.xw;
    yw[i] = t.yw;
    ta[i] = base[(unsigned)t.ia];
    tb[i] = base[(unsigned)t.ib];
    tc[i] = base[(unsigned)t.ic];
    td[i] = base[(unsigned)t.id];
    if (unwarped_magnitude) fo[i] = base[pix[i]];
  }
#pragma unroll
  for (int i = 0; i < kOccPPL; ++i) {
    const float wa = (1.f - xw[i]) * (1.f - yw[i]), wb = (1.f - xw[i]) * yw[i];
    const float wc = xw[i] * (1.f - yw[i]), wd = xw[i] * yw[i];
    float2 w;
    w.x = add_rn(add_rn(add_rn(mul_rn(wa, ta[i].x), mul_rn(wb, tb[i].x)), mul_rn(wc, tc[i].x)), mul_rn(wd, td[i].x));
    w.y = add_rn(add_rn(add_rn(mul_rn(wa, ta[i].y), mul_rn(wb, tb[i].y)), mul_rn(wc, tc[i].y)), mul_rn(wd, td[i].y));
    float du, dv;
    const float o = fb_occluded(f[i], w, unwarped_magnitude ? fo[i] : w, du, dv);
    if (ok[i]) {
      const long long p = item + pix[i];
      if (warped_out) warped_out[p] = w;
      if (occ) occ[p] = o;
      if (mask_out) mask_out[p] = mask_in[p] * (1.f - o);     // mask_in may be mask_out (in place)
    }
  }
}

// losses['sym'] of one direction (losses.py:60-61): charbonnier_loss(occ - disocc) with
// occ = 1 - mask (:57-58) and disocc = cast(density < thresh) (:27-29), one pass
// instead of compare, cast, two subtractions and the reduction.
template <bool VEC4>
__global__ __launch_bounds__(256) void sym_loss_kernel(const float *__restrict__ mask,
                                                       const float *__restrict__ density, float thresh,
                                                       float *__restrict__ partial, long long n) {
  auto term = [thresh](float m, float d) {
    const float occ = 1.f - m;
    const float disocc = d < thresh ? 1.f : 0.f;
    return charbonnier_term(occ - disocc, 1.f, kAlpha, 1.f, kEps, 0.f);
  };
  float s = 0.f;
  const long long stride = (long long)gridDim.x * 256;
  if constexpr (VEC4) {  // n % 4 == 0, both arrays 16-byte aligned (host-checked): 16 bytes per lane and load
    const float4 *m4 = reinterpret_cast<const float4 *>(mask);
    const float4 *d4 = reinterpret_cast<const float4 *>(density);
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n / 4; i += stride) {
      const float4 m = m4[i], d = d4[i];
      s += (term(m.x, d.x) + term(m.y, d.y)) + (term(m.z, d.z) + term(m.w, d.w));
    }
  } else {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += stride) s += term(mask[i], density[i]);
  }
  __shared__ float red[4];
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

inline unsigned ew_grid(long long n) {
  long long g = (n + 255) / 256;
  return (unsigned)(g < kReduceBlocks ? (g > 0 ? g : 1) : kReduceBlocks);
}

// Rows per wave strip: two gray LDS images of 70 x 70 floats, 39 KB per workgroup.  (A
// backward with dldd as a third LDS image fits only two workgroups per CU at 16 rows and
// is slower than the unpaired kernel, 0.47 vs 0.42 ms; 0.38 ms at 8 rows; 0.31 ms with
// dldd read in place.)
constexpr int kPairRowsFwd = 16;   // 12 measures the same, 8 and 24+ are slower (halo rows / occupancy)
// Rows per wave strip.  A launch that fills the chip (full resolution) takes the strip that minimises the total work,
// tiles x (strip + warm-up rows): 16.  The coarse levels of the loss pyramid do not fill it -- 96 x 320 x 32 pixels are
// 384 workgroups of 64 rows -- and there the time is a workgroup's own latency, which falls with the strip: the
// smallest strip whose launch still fits the chip at once (8 workgroups of four waves per CU) wins (KITTI-config
// pyramid as one graph: 1.156 ms with 16 everywhere, 1.075 / 1.013 / 0.952 / 0.958 ms with 12 / 8 / 4 / 2 at the coarse
// levels; profiles/r04_variants.txt).  The result does not depend on the choice: the same pair sums in the same
// order per pixel.
inline int census_cus() {
  static std::atomic<int> cached{0};
  int n = cached.load(std::memory_order_relaxed);
  if (n > 0) return n;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  cached.store(n, std::memory_order_relaxed);
  return n;
}
inline int pair_rows(int B, int H, int W, int r) {
#ifdef UNFLOW_TUNE_PAIR_ROWS      // experiment: UNFLOW_PAIR_ROWS=<16|12|8|4|2> forces the strip height for images below 200 rows
  if (const char *e = getenv("UNFLOW_PAIR_ROWS"))
    if (H < 200) return atoi(e);
#endif
  const long long xt = (W + (64 - 2 * r) - 1) / (64 - 2 * r), cap = 8LL * census_cus();
  for (int rows : {4, 8, 12, 16})
    if (xt * ((H + kPairWaves * rows - 1) / (kPairWaves * rows)) * B <= cap) return rows;
  int best = kPairRowsFwd;
  long long best_cost = -1;
  for (int rows : {16, 12, 8}) {
    const long long tiles = (H + kPairWaves * rows - 1) / (kPairWaves * rows);
    const long long cost = tiles * (rows + r);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = rows;
    }
  }
  return best;
}
// ROWS as a compile-time constant of the statement that follows (M is a macro taking ROWS)
#define UNFLOW_PAIR_ROWS_DISPATCH(rows, M) \
  do {                                     \
    if ((rows) == 16) M(16);               \
    else if ((rows) == 12) M(12);          \
    else if ((rows) == 8) M(8);            \
    else M(4);                             \
  } while (0)
inline dim3 pair_grid(int r, int rows_per_wave, int B, int H, int W) {
  const int cols = 64 - 2 * r, rows = kPairWaves * rows_per_wave;
  return dim3((W + cols - 1) / cols, (H + rows - 1) / rows, B);
}
template <int R, int ROWS>
inline size_t pair_smem(int images) { return sizeof(float) * images * HalfPlane<R, ROWS>::LW * HalfPlane<R, ROWS>::LH; }
inline long long tile_blocks(int B, int H, int W) {
  return (long long)B * ((H + kTH - 1) / kTH) * ((W + kTW - 1) / kTW);
}

}  // namespace

// the plan of this thread between unflow_reduce_plan_begin() and _end(): caller-owned, plain data
static thread_local unflow_reduce_plan *t_plan = nullptr;

int reduce_partials(const float *partial, int n, double inv_norm, float *loss, hipStream_t s) {
  if (t_plan) {                      // deferred: the caller reduces every segment of the plan in one launch later
    if (t_plan->n >= UNFLOW_PLAN_CAPACITY)
      return fail(UNFLOW_ERR_INVALID_ARGUMENT, "reduce plan: more than %d deferred reductions", UNFLOW_PLAN_CAPACITY);
    const int i = t_plan->n++;
    t_plan->partial[i] = partial;
    t_plan->count[i] = n;
    t_plan->inv_norm[i] = inv_norm;
    return UNFLOW_OK;
  }
  // (a NULL scalar is the deferred mode's calling convention: without an open plan on THIS thread it would be
  // written through)
  if (!loss) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "loss: the output scalar is NULL and no reduce plan is open on this thread");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, s, partial, n, inv_norm, loss);
  return launch_status("reduce_partials");
}

int zero_loss(float *loss, const char *what, hipStream_t s) {
  if (t_plan) return reduce_partials(nullptr, 0, 0.0, loss, s);
  if (!loss) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "%s: the output scalar is NULL and no reduce plan is open on this thread", what);
  if (hipMemsetAsync(loss, 0, sizeof(float), s) != hipSuccess) return launch_status(what);
  return UNFLOW_OK;
}

namespace {
struct PlanArgs {
  const float *partial[UNFLOW_PLAN_CAPACITY];
  double inv_norm[UNFLOW_PLAN_CAPACITY];
  int count[UNFLOW_PLAN_CAPACITY];
};
struct FoldArgs {
  float coeff[UNFLOW_PLAN_CAPACITY], term_weight[UNFLOW_PLAN_CAPACITY];
  int term[UNFLOW_PLAN_CAPACITY];
};
// one block per segment: the arithmetic of reduce_partials_kernel (fixed order, double accumulation)
__global__ __launch_bounds__(1024) void reduce_segments_kernel(PlanArgs a, float *__restrict__ results) {
  __shared__ double sh[1024];
  const float *partial = a.partial[blockIdx.x];
  const int n = a.count[blockIdx.x];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += (double)partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) results[blockIdx.x] = (float)(sh[0] * a.inv_norm[blockIdx.x]);
}
// one wave: total and the per-term sums, each in the order of the plan (float, like the host-side fold it replaces)
__global__ __launch_bounds__(64) void fold_results_kernel(FoldArgs a, int n, const float *__restrict__ results,
                                                          float *__restrict__ total, float *__restrict__ per_term) {
  const int t = threadIdx.x;
  if (t == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s = fmaf(a.coeff[i], results[i], s);
    *total = s;
  } else if (t <= UNFLOW_PLAN_TERMS && per_term) {
    float s = 0.f;
    for (int i = 0; i < n; ++i)
      if (a.term[i] == t - 1) s = fmaf(a.term_weight[i], results[i], s);
    per_term[t - 1] = s;
  }
}
struct SumArgs {
  const float *src[8];
};
__global__ __launch_bounds__(256) void sum_tensors_kernel(SumArgs a, int n, float *__restrict__ out, long long count) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
    float s = a.src[0][i];
    for (int k = 1; k < n; ++k) s += a.src[k][i];        // the order autograd's chain of adds would take
    out[i] = s;
  }
}
// The flow scaling of a loss pyramid whose levels run both directions as one batch (core/unsupervised.py:118):
//   F_l = s_l * [fw_l; bw_l],  G_l = s_l * [bw_l; fw_l]   ([a; b]: concatenated along the batch)   for all levels in ONE launch,
// and the gradient   d fw_l = s_l * (gF_l[first half] + gG_l[second half]),  d bw_l = s_l * (gF_l[second] + gG_l[first]).
// Work is in float2 units (a flow vector); off[] = prefix sums of the levels' vector counts.
constexpr int kStackLevels = 8;
struct StackArgs {
  const float2 *a[kStackLevels], *b[kStackLevels];     // forward: fw, bw;   backward: gF, gG (gG may be null)
  float2 *x[kStackLevels], *y[kStackLevels];           // forward: F, G;     backward: d_fw, d_bw
  long long off[kStackLevels + 1];
  float scale[kStackLevels];
};
template <bool BACKWARD>
__global__ __launch_bounds__(256) void scale_stack_kernel(StackArgs a, int n) {
  const long long total = a.off[n];
  for (long long t = blockIdx.x * 256LL + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    int l = 0;
    while (l + 1 < n && t >= a.off[l + 1]) ++l;
    const long long i = t - a.off[l], cnt = a.off[l + 1] - a.off[l];
    const float s = a.scale[l];
    if constexpr (!BACKWARD) {
      const float2 f = a.a[l][i], b = a.b[l][i];
      const float2 sf = make_float2(f.x * s, f.y * s), sb = make_float2(b.x * s, b.y * s);
      a.x[l][i] = sf;
      a.x[l][cnt + i] = sb;
      a.y[l][i] = sb;
      a.y[l][cnt + i] = sf;
    } else {
      float2 df = a.a[l][i], db = a.a[l][cnt + i];
      if (a.b[l]) {
        const float2 g2 = a.b[l][cnt + i], g1 = a.b[l][i];
        df = make_float2(df.x + g2.x, df.y + g2.y);
        db = make_float2(db.x + g1.x, db.y + g1.y);
      }
      a.x[l][i] = make_float2(df.x * s, df.y * s);
      a.y[l][i] = make_float2(db.x * s, db.y * s);
    }
  }
}
}  // namespace

}  // namespace unflow

using namespace unflow;

static int scale_stack_launch(bool backward, const float *const *a, const float *const *b, float *const *x, float *const *y,
                              const long long *count, const float *scale, int n, unflow_stream_t stream) {
  if (!a || !x || !y || !count || !scale || n < 1 || n > kStackLevels)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "scale_stack: 1 to 8 levels");
  StackArgs args;
  args.off[0] = 0;
  for (int l = 0; l < kStackLevels; ++l) {
    const bool in = l < n;
    if (in && (count[l] < 0 || (count[l] & 1))) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "scale_stack: counts are floats of [B,h,w,2] tensors");
    if (in && (!a[l] || !x[l] || !y[l] || (!backward && (!b || !b[l]))))
      return fail(UNFLOW_ERR_INVALID_ARGUMENT, "scale_stack: null tensor");
    args.a[l] = in ? reinterpret_cast<const float2 *>(a[l]) : nullptr;
    args.b[l] = (in && b) ? reinterpret_cast<const float2 *>(b[l]) : nullptr;
    args.x[l] = in ? reinterpret_cast<float2 *>(x[l]) : nullptr;
    args.y[l] = in ? reinterpret_cast<float2 *>(y[l]) : nullptr;
    args.scale[l] = in ? scale[l] : 0.f;
    args.off[l + 1] = args.off[l] + (in ? count[l] / 2 : 0);
  }
  const long long total = args.off[n];
  if (total == 0) return UNFLOW_OK;
  const long long blocks = (total + 255) / 256;
  const dim3 g((unsigned)(blocks < 8192 ? blocks : 8192));
  if (backward) hipLaunchKernelGGL(scale_stack_kernel<true>, g, dim3(256), 0, as_stream(stream), args, n);
  else hipLaunchKernelGGL(scale_stack_kernel<false>, g, dim3(256), 0, as_stream(stream), args, n);
  return launch_status("scale_stack");
}

extern "C" int unflow_scale_stack_fwd(const float *const *fw, const float *const *bw, float *const *F, float *const *G,
                                      const long long *count, const float *scale, int n, unflow_stream_t stream) {
  return scale_stack_launch(false, fw, bw, F, G, count, scale, n, stream);
}

extern "C" int unflow_scale_stack_bwd(const float *const *gF, const float *const *gG, float *const *d_fw, float *const *d_bw,
                                      const long long *count, const float *scale, int n, unflow_stream_t stream) {
  return scale_stack_launch(true, gF, gG, d_fw, d_bw, count, scale, n, stream);
}

extern "C" size_t unflow_reduce_workspace_floats(int B, int H, int W) {
  // One partial per workgroup: the 32x8 tiles of the tiled loss kernels (census, stencils)
  // are the finest grid (the pair-symmetric census uses 58..60 x 64 pixel workgroups, the
  // 1-D reductions kReduceBlocks blocks).
  if (B <= 0 || H <= 0 || W <= 0) return (size_t)kReduceBlocks;
  const long long tiles = tile_blocks(B, H, W);
  return (size_t)(tiles > kReduceBlocks ? tiles : kReduceBlocks);
}

extern "C" int unflow_reduce_plan_begin(unflow_reduce_plan *plan) {
  if (!plan) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "reduce plan: NULL plan");
  if (t_plan) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "reduce plan: a plan is already open on this thread");
  plan->n = 0;
  t_plan = plan;
  return UNFLOW_OK;
}

extern "C" int unflow_reduce_plan_end(void) {
  t_plan = nullptr;
  return UNFLOW_OK;
}

extern "C" int unflow_reduce_plan_run(const unflow_reduce_plan *plan, const float *coeff, const int *term,
                                      const float *term_weight, float *results, float *total, float *per_term,
                                      unflow_stream_t stream) {
  if (!plan || plan->n < 0 || plan->n > UNFLOW_PLAN_CAPACITY || !coeff || !results || !total)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "reduce plan: invalid plan or NULL buffer");
  hipStream_t s = as_stream(stream);
  const int n = plan->n;
  if (term)
    for (int i = 0; i < n; ++i)
      if (term[i] >= UNFLOW_PLAN_TERMS)
        return fail(UNFLOW_ERR_INVALID_ARGUMENT, "reduce plan: term %d of segment %d (per_term holds %d sums)", term[i], i, UNFLOW_PLAN_TERMS);
  if (n > 0) {
    PlanArgs pa;
    for (int i = 0; i < UNFLOW_PLAN_CAPACITY; ++i) {
      pa.partial[i] = i < n ? plan->partial[i] : nullptr;
      pa.inv_norm[i] = i < n ? plan->inv_norm[i] : 0.0;
      pa.count[i] = i < n ? plan->count[i] : 0;
    }
    hipLaunchKernelGGL(reduce_segments_kernel, dim3(n), dim3(1024), 0, s, pa, results);
    if (int rc = launch_status("reduce_plan_run")) return rc;
  }
  FoldArgs fa;
  for (int i = 0; i < UNFLOW_PLAN_CAPACITY; ++i) {
    fa.coeff[i] = i < n ? coeff[i] : 0.f;
    fa.term_weight[i] = (i < n && term_weight) ? term_weight[i] : 0.f;
    fa.term[i] = (i < n && term) ? term[i] : -1;
  }
  hipLaunchKernelGGL(fold_results_kernel, dim3(1), dim3(64), 0, s, fa, n, results, total, per_term);
  return launch_status("reduce_plan_run fold");
}

extern "C" int unflow_sum_tensors(const float *const *srcs, int n, float *out, long long count, unflow_stream_t stream) {
  if (!srcs || n < 1 || n > 8 || count < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "sum_tensors: 1 <= n <= 8 sources");
  if (count == 0) return UNFLOW_OK;
  SumArgs a;
  for (int k = 0; k < 8; ++k) a.src[k] = k < n ? srcs[k] : nullptr;
  const long long blocks = (count + 255) / 256;
  hipLaunchKernelGGL(sum_tensors_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, as_stream(stream), a, n,
                     out, count);
  return launch_status("sum_tensors");
}

extern "C" int unflow_ternary_loss_fwd(const float *im1, const float *im2_warped, const float *mask, float *loss,
                                       float *dldd, float *dist, float *workspace, int B, int H, int W,
                                       int max_distance, long long norm_count, unflow_stream_t stream) {
  if (B < 0 || H < 0 || W < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "ternary_loss: negative dimension");
  if (max_distance < 0 || max_distance > 16)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "ternary_loss: max_distance must be in [0,16]");
  hipStream_t s = as_stream(stream);
  const long long npix = (long long)B * H * W;
  if (npix == 0) {
    // reduce_sum of an empty tensor / 0 -> NaN in TF; keep the sum (0) and do not divide.
    return zero_loss(loss, "ternary_loss", s);
  }
  const double norm = norm_count > 0 ? (double)norm_count : (double)npix;
  const int r = max_distance;
  const float inv_norm = (float)(1.0 / norm);
  if (r >= 2 && r <= 3) {
    const int prows = pair_rows(B, H, W, r);
    const dim3 pgrid = pair_grid(r, prows, B, H, W);
#define LAUNCH_PAIR_FWD(R, ROWS)                                                                               \
  hipLaunchKernelGGL((ternary_fwd_pair_kernel<R, ROWS>), pgrid, dim3(64 * kPairWaves),                          \
                     (pair_smem<R, ROWS>(2)), s, im1, im2_warped, mask, workspace, dldd, dist, H, W, inv_norm)
#define LAUNCH_PAIR_FWD2(ROWS) LAUNCH_PAIR_FWD(2, ROWS)
#define LAUNCH_PAIR_FWD3(ROWS) LAUNCH_PAIR_FWD(3, ROWS)
    if (r == 2) UNFLOW_PAIR_ROWS_DISPATCH(prows, LAUNCH_PAIR_FWD2);
    else UNFLOW_PAIR_ROWS_DISPATCH(prows, LAUNCH_PAIR_FWD3);
#undef LAUNCH_PAIR_FWD3
#undef LAUNCH_PAIR_FWD2
#undef LAUNCH_PAIR_FWD
    if (int rc = launch_status("ternary_loss_fwd")) return rc;
    return reduce_partials(workspace, (int)(pgrid.x * pgrid.y * pgrid.z), 1.0 / norm, loss, s);
  }
  const dim3 grid((W + kTW - 1) / kTW, (H + kTH - 1) / kTH, B);
  const size_t smem = sizeof(float) * 2 * (kTW + 2 * r) * (kTH + 2 * r);
#define LAUNCH_FWD(R) \
  hipLaunchKernelGGL(ternary_fwd_kernel<R>, grid, dim3(kNT), smem, s, im1, im2_warped, mask, workspace, dldd, dist, H, W, r, inv_norm)
  switch (r) {
    case 1: LAUNCH_FWD(1); break;
    case 2: LAUNCH_FWD(2); break;
    case 3: LAUNCH_FWD(3); break;
    default: LAUNCH_FWD(0);
  }
#undef LAUNCH_FWD
  if (int rc = launch_status("ternary_loss_fwd")) return rc;
  return reduce_partials(workspace, (int)tile_blocks(B, H, W), 1.0 / norm, loss, s);
}

extern "C" int unflow_ternary_loss_bwd(const float *im1, const float *im2_warped, const float *dldd,
                                       const float *gout, float *d_im2_warped, int B, int H, int W,
                                       int max_distance, unflow_stream_t stream) {
  if (B < 0 || H < 0 || W < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "ternary_loss_grad: negative dimension");
  if (max_distance < 0 || max_distance > 16)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "ternary_loss: max_distance must be in [0,16]");
  if ((long long)B * H * W == 0) return UNFLOW_OK;
  hipStream_t s = as_stream(stream);
  const int r = max_distance;
  if (r >= 2 && r <= 3 && (long long)B * H * W * 4 < (1LL << 31)) {   // 32-bit byte offsets into dldd
    const int prows = pair_rows(B, H, W, r);
#define LAUNCH_PAIR_BWD(R, ROWS)                                                                               \
  hipLaunchKernelGGL((ternary_bwd_pair_kernel<R, ROWS>), pair_grid(r, ROWS, B, H, W),                           \
                     dim3(64 * kPairWaves), (pair_smem<R, ROWS>(2)), s, im1, im2_warped, dldd, gout,            \
                     d_im2_warped, H, W)
#define LAUNCH_PAIR_BWD2(ROWS) LAUNCH_PAIR_BWD(2, ROWS)
#define LAUNCH_PAIR_BWD3(ROWS) LAUNCH_PAIR_BWD(3, ROWS)
    if (r == 2) UNFLOW_PAIR_ROWS_DISPATCH(prows, LAUNCH_PAIR_BWD2);
    else UNFLOW_PAIR_ROWS_DISPATCH(prows, LAUNCH_PAIR_BWD3);
#undef LAUNCH_PAIR_BWD3
#undef LAUNCH_PAIR_BWD2
#undef LAUNCH_PAIR_BWD
    return launch_status("ternary_loss_bwd");
  }
  const dim3 grid((W + kTW - 1) / kTW, (H + kTH - 1) / kTH, B);
  const size_t smem = sizeof(float) * 3 * (kTW + 2 * r) * (kTH + 2 * r);
#define LAUNCH_BWD(R) \
  hipLaunchKernelGGL(ternary_bwd_kernel<R>, grid, dim3(kNT), smem, s, im1, im2_warped, dldd, gout, d_im2_warped, H, W, r)
  switch (r) {
    case 1: LAUNCH_BWD(1); break;
    case 2: LAUNCH_BWD(2); break;
    case 3: LAUNCH_BWD(3); break;
    default: LAUNCH_BWD(0);
  }
#undef LAUNCH_BWD
  return launch_status("ternary_loss_bwd");
}

extern "C" int unflow_outgoing_mask(const float *flow, float *mask, int B, int H, int W, unflow_stream_t stream) {
  const long long n = (long long)B * H * W;
  if (n <= 0) return n == 0 ? UNFLOW_OK : fail(UNFLOW_ERR_INVALID_ARGUMENT, "outgoing_mask: negative dimension");
  hipLaunchKernelGGL(outgoing_mask_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float2 *>(flow), mask, H, W, n);
  return launch_status("outgoing_mask");
}

extern "C" int unflow_create_mask(float *mask, int B, int H, int W, int pad_top, int pad_bottom, int pad_left,
                                  int pad_right, unflow_stream_t stream) {
  const long long n = (long long)B * H * W;
  if (n <= 0) return n == 0 ? UNFLOW_OK : fail(UNFLOW_ERR_INVALID_ARGUMENT, "create_mask: negative dimension");
  hipLaunchKernelGGL(create_mask_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), mask, H, W, pad_top,
                     pad_bottom, pad_left, pad_right, n);
  return launch_status("create_mask");
}

extern "C" int unflow_fb_occlusion(const float *flow, const float *flow_other_warped, float *flow_diff, float *occ,
                                   float *mask_inout, int B, int H, int W, unflow_stream_t stream) {
  const long long n = (long long)B * H * W;
  if (n <= 0) return n == 0 ? UNFLOW_OK : fail(UNFLOW_ERR_INVALID_ARGUMENT, "fb_occlusion: negative dimension");
  hipLaunchKernelGGL(fb_occlusion_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float2 *>(flow), reinterpret_cast<const float2 *>(flow_other_warped),
                     reinterpret_cast<float2 *>(flow_diff), occ, mask_inout, mask_inout, n);
  return launch_status("fb_occlusion");
}

extern "C" int unflow_fb_occlusion_masked(const float *flow, const float *flow_other_warped, float *flow_diff, float *occ,
                                          const float *mask_in, float *mask_out, int B, int H, int W, unflow_stream_t stream) {
  const long long n = (long long)B * H * W;
  if (n <= 0) return n == 0 ? UNFLOW_OK : fail(UNFLOW_ERR_INVALID_ARGUMENT, "fb_occlusion: negative dimension");
  if ((mask_in == nullptr) != (mask_out == nullptr))
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "fb_occlusion: mask_in and mask_out go together");
  hipLaunchKernelGGL(fb_occlusion_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float2 *>(flow), reinterpret_cast<const float2 *>(flow_other_warped),
                     reinterpret_cast<float2 *>(flow_diff), occ, mask_in, mask_out, n);
  return launch_status("fb_occlusion");
}

extern "C" int unflow_fb_occlusion_losses(const float *flow, const float *flow_other_warped, float *flow_diff, float *occ,
                                          const float *mask_in, float *mask_out, float *loss_fb, float *workspace_fb,
                                          long long norm_fb, float *loss_occ, float *workspace_occ, long long norm_occ,
                                          int B, int H, int W, unflow_stream_t stream) {
  const long long n = (long long)B * H * W;
  if (n < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "fb_occlusion: negative dimension");
  if (!flow_diff || !mask_in || !mask_out || !workspace_fb || !workspace_occ)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "fb_occlusion_losses: flow_diff, the masks and the two workspaces are required");
  hipStream_t s = as_stream(stream);
  if (n == 0) {
    if (int rc = zero_loss(loss_fb, "fb_occlusion_losses", s)) return rc;
    return zero_loss(loss_occ, "fb_occlusion_losses", s);
  }
  const unsigned grid = ew_grid(n);
  hipLaunchKernelGGL(fb_occlusion_losses_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float2 *>(flow),
                     reinterpret_cast<const float2 *>(flow_other_warped), reinterpret_cast<float2 *>(flow_diff), occ,
                     mask_in, mask_out, workspace_fb, workspace_occ, n);
  if (int rc = launch_status("fb_occlusion_losses")) return rc;
  // (two entries of an open reduce plan, in this order; otherwise two one-block reductions)
  if (int rc = reduce_partials(workspace_fb, (int)grid, 1.0 / (norm_fb > 0 ? (double)norm_fb : (double)(2 * n)), loss_fb, s)) return rc;
  return reduce_partials(workspace_occ, (int)grid, 1.0 / (norm_occ > 0 ? (double)norm_occ : (double)n), loss_occ, s);
}

extern "C" int unflow_charbonnier_fwd(const float *x, const float *mask, int mask_channels, float *loss,
                                      float *workspace, int B, int H, int W, int C, float alpha, float beta,
                                      float epsilon, float truncate, long long norm_count,
                                      unflow_stream_t stream) {
  const long long n = (long long)B * H * W * C;
  if (n < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "charbonnier_loss: negative dimension");
  if (mask && mask_channels != 1 && mask_channels != C)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "charbonnier_loss: mask channels must be 1 or match x");
  hipStream_t s = as_stream(stream);
  if (n == 0) {
    return zero_loss(loss, "charbonnier", s);
  }
  const unsigned grid = ew_grid(n);
  hipLaunchKernelGGL(charbonnier_fwd_kernel, dim3(grid), dim3(256), 0, s, x, mask, mask_channels, workspace, n, C,
                     alpha, beta, epsilon, truncate);
  if (int rc = launch_status("charbonnier_fwd")) return rc;
  const double norm = norm_count > 0 ? (double)norm_count : (double)n;
  return reduce_partials(workspace, (int)grid, 1.0 / norm, loss, s);
}

extern "C" int unflow_charbonnier_bwd(const float *x, const float *mask, int mask_channels, const float *gout,
                                      float *dx, int B, int H, int W, int C, float alpha, float beta, float epsilon,
                                      float truncate, long long norm_count, unflow_stream_t stream) {
  const long long n = (long long)B * H * W * C;
  if (n < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "charbonnier_loss_grad: negative dimension");
  if (mask && mask_channels != 1 && mask_channels != C)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "charbonnier_loss: mask channels must be 1 or match x");
  if (n == 0) return UNFLOW_OK;
  const double norm = norm_count > 0 ? (double)norm_count : (double)n;
  hipLaunchKernelGGL(charbonnier_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), x, mask,
                     mask_channels, gout, dx, n, C, alpha, beta, epsilon, truncate, (float)(1.0 / norm));
  return launch_status("charbonnier_bwd");
}

extern "C" int unflow_census_transform(const float *im, unsigned long long *signature, int B, int H, int W,
                                       int max_distance, float eps, unflow_stream_t stream) {
  if (B < 0 || H < 0 || W < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "census_transform: negative dimension");
  if (max_distance < 1 || max_distance > 3)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "census_transform: max_distance must be in [1,3] (48 neighbours per 64-bit word)");
  if (!(eps >= 0.f)) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "census_transform: eps must be >= 0");
  if ((long long)B * H * W == 0) return UNFLOW_OK;
  const int r = max_distance;
  const dim3 grid((W + kTW - 1) / kTW, (H + kTH - 1) / kTH, B);
  const size_t smem = sizeof(float) * (kTW + 2 * r) * (kTH + 2 * r);
  hipLaunchKernelGGL(census_transform_kernel, grid, dim3(kNT), smem, as_stream(stream), im, signature, H, W, r, eps);
  return launch_status("census_transform");
}

extern "C" int unflow_census_hamming(const unsigned long long *sig1, const unsigned long long *sig2,
                                     const float *mask, int *dist, unsigned long long *total, int B, int H, int W,
                                     unflow_stream_t stream) {
  const long long n = (long long)B * H * W;
  if (n < 0 || B < 0 || H < 0 || W < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "census_hamming: negative dimension");
  hipStream_t s = as_stream(stream);
  if (total && hipMemsetAsync(total, 0, sizeof(unsigned long long), s) != hipSuccess)
    return launch_status("census_hamming memset");
  if (n == 0) return UNFLOW_OK;
  hipLaunchKernelGGL(census_hamming_kernel, dim3(ew_grid(n)), dim3(256), 0, s,
                     reinterpret_cast<const ulonglong2 *>(sig1), reinterpret_cast<const ulonglong2 *>(sig2), mask, dist,
                     total, n);
  return launch_status("census_hamming");
}

extern "C" int unflow_fb_occlusion_warp(const float *flow, const float *flow_other, float *flow_other_warped,
                                        float *occ, const float *mask_in, float *mask_out, int B, int H,
                                        int W, int unwarped_magnitude, unflow_stream_t stream) {
  const long long n = (long long)B * H * W;
  if (B < 0 || H < 0 || W < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "fb_occlusion_warp: negative dimension");
  if (n == 0) return UNFLOW_OK;
  if (mask_out && !mask_in) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "fb_occlusion_warp: mask_out needs mask_in");
  const int ntx = (W + kTileW - 1) / kTileW, nty = (H + kOccTileH - 1) / kOccTileH;
  const long long tiles = (long long)B * ntx * nty;
  if ((long long)H * W < (1LL << 30) && tiles < (1LL << 31)) {      // 32-bit offsets inside an item, one block per tile
    hipLaunchKernelGGL(fb_occlusion_warp_tile_kernel, dim3((unsigned)tiles), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2 *>(flow), reinterpret_cast<const float2 *>(flow_other),
                       reinterpret_cast<float2 *>(flow_other_warped), occ, mask_in, mask_out, H, W, ntx, nty,
                       unwarped_magnitude);
    return launch_status("fb_occlusion_warp");
  }
  const long long vn = tiled_total(B, H, W);
  hipLaunchKernelGGL(fb_occlusion_warp_kernel<false>, dim3(ew_grid(vn)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float2 *>(flow), reinterpret_cast<const float2 *>(flow_other),
                     reinterpret_cast<float2 *>(flow_other_warped), occ, mask_in, mask_out, H, W, unwarped_magnitude, vn,
                     (float2 *)nullptr, (float *)nullptr, (float *)nullptr);
  return launch_status("fb_occlusion_warp");
}

extern "C" int unflow_fb_occlusion_warp_losses(const float *flow, const float *flow_other, float *flow_diff, float *occ,
                                               const float *mask_in, float *mask_out, float *loss_fb,
                                               float *workspace_fb, long long norm_fb, float *loss_occ,
                                               float *workspace_occ, long long norm_occ, int B, int H, int W,
                                               unflow_stream_t stream) {
  const long long n = (long long)B * H * W;
  if (B < 0 || H < 0 || W < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "fb_occlusion_warp: negative dimension");
  if (!flow_diff || !mask_in || !mask_out || !workspace_fb || !workspace_occ)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "fb_occlusion_warp_losses: flow_diff, the masks and the two workspaces are required");
  hipStream_t s = as_stream(stream);
  if (n == 0) {
    if (int rc = zero_loss(loss_fb, "fb_occlusion_warp_losses", s)) return rc;
    return zero_loss(loss_occ, "fb_occlusion_warp_losses", s);
  }
  const long long vn = tiled_total(B, H, W);
  const unsigned grid = ew_grid(vn);
  hipLaunchKernelGGL(fb_occlusion_warp_kernel<true>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float2 *>(flow),
                     reinterpret_cast<const float2 *>(flow_other), (float2 *)nullptr, occ, mask_in, mask_out, H, W, 0, vn,
                     reinterpret_cast<float2 *>(flow_diff), workspace_fb, workspace_occ);
  if (int rc = launch_status("fb_occlusion_warp_losses")) return rc;
  if (int rc = reduce_partials(workspace_fb, (int)grid, 1.0 / (norm_fb > 0 ? (double)norm_fb : (double)(2 * n)), loss_fb, s)) return rc;
  return reduce_partials(workspace_occ, (int)grid, 1.0 / (norm_occ > 0 ? (double)norm_occ : (double)n), loss_occ, s);
}

extern "C" int unflow_sym_loss_fwd(const float *mask, const float *density, float thresh, float *loss,
                                   float *workspace, int B, int H, int W, long long norm_count,
                                   unflow_stream_t stream) {
  const long long n = (long long)B * H * W;
  if (B < 0 || H < 0 || W < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "sym_loss: negative dimension");
  hipStream_t s = as_stream(stream);
  if (n == 0) {
    return zero_loss(loss, "sym_loss", s);
  }
  const bool vec4 = n % 4 == 0 && ((uintptr_t)mask | (uintptr_t)density) % 16 == 0;
  const unsigned grid = ew_grid(vec4 ? n / 4 : n);
  if (vec4)
    hipLaunchKernelGGL(sym_loss_kernel<true>, dim3(grid), dim3(256), 0, s, mask, density, thresh, workspace, n);
  else
    hipLaunchKernelGGL(sym_loss_kernel<false>, dim3(grid), dim3(256), 0, s, mask, density, thresh, workspace, n);
  if (int rc = launch_status("sym_loss")) return rc;
  const double norm = norm_count > 0 ? (double)norm_count : (double)n;
  return reduce_partials(workspace, (int)grid, 1.0 / norm, loss, s);
}

extern "C" int unflow_ternary_loss_warp_fwd(const float *im1, const float *im2, const float *flow, const float *mask,
                                            float *loss, float *dldd, float *gray1, float *gray2_warped,
                                            float *workspace, int B, int H, int W, int max_distance,
                                            long long norm_count, unflow_stream_t stream) {
  if (B < 0 || H < 0 || W < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "ternary_loss_warp: negative dimension");
  if (max_distance < 2 || max_distance > 3 || (long long)B * H * W * 4 >= (1LL << 31))
    return fail(UNFLOW_ERR_INVALID_ARGUMENT,
                "ternary_loss_warp: fused for max_distance 2 and 3 and fewer than 2^29 pixels; "
                "compose unflow_image_warp_fwd and unflow_ternary_loss_fwd otherwise");
  hipStream_t s = as_stream(stream);
  const long long npix = (long long)B * H * W;
  if (npix == 0) {
    return zero_loss(loss, "ternary_loss_warp", s);
  }
  const double norm = norm_count > 0 ? (double)norm_count : (double)npix;
  const int r = max_distance;
  const int prows = pair_rows(B, H, W, r);
  const dim3 pgrid = pair_grid(r, prows, B, H, W);
#define LAUNCH_PAIR_WARP(R, ROWS)                                                                              \
  hipLaunchKernelGGL((ternary_fwd_pair_kernel<R, ROWS, true>), pgrid, dim3(64 * kPairWaves),                    \
                     (pair_smem<R, ROWS>(2)), s, im1, im2, mask, workspace, dldd,                               \
                     static_cast<float *>(nullptr), H, W, (float)(1.0 / norm),                                  \
                     reinterpret_cast<const float2 *>(flow), gray1, gray2_warped)
#define LAUNCH_PAIR_WARP2(ROWS) LAUNCH_PAIR_WARP(2, ROWS)
#define LAUNCH_PAIR_WARP3(ROWS) LAUNCH_PAIR_WARP(3, ROWS)
  if (r == 2) UNFLOW_PAIR_ROWS_DISPATCH(prows, LAUNCH_PAIR_WARP2);
  else UNFLOW_PAIR_ROWS_DISPATCH(prows, LAUNCH_PAIR_WARP3);
#undef LAUNCH_PAIR_WARP3
#undef LAUNCH_PAIR_WARP2
#undef LAUNCH_PAIR_WARP
  if (int rc = launch_status("ternary_loss_warp_fwd")) return rc;
  return reduce_partials(workspace, (int)(pgrid.x * pgrid.y * pgrid.z), 1.0 / norm, loss, s);
}

extern "C" int unflow_ternary_loss_warp_bwd(const float *gray1, const float *gray2_warped, const float *dldd,
                                            const float *gout, const float *im2, const float *flow, float *d_flow,
                                            int B, int H, int W, int max_distance, unflow_stream_t stream) {
  if (B < 0 || H < 0 || W < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "ternary_loss_warp_grad: negative dimension");
  if (max_distance < 2 || max_distance > 3 || (long long)B * H * W * 4 >= (1LL << 31))
    return fail(UNFLOW_ERR_INVALID_ARGUMENT,
                "ternary_loss_warp_grad: fused for max_distance 2 and 3 and fewer than 2^29 pixels; "
                "compose unflow_ternary_loss_bwd and unflow_image_warp_bwd otherwise");
  if ((long long)B * H * W == 0) return UNFLOW_OK;
  hipStream_t s = as_stream(stream);
  const int r = max_distance;
  const int prows = pair_rows(B, H, W, r);
#define LAUNCH_PAIR_FUSED(R, ROWS)                                                                             \
  hipLaunchKernelGGL((ternary_bwd_pair_kernel<R, ROWS, true>), pair_grid(r, ROWS, B, H, W),                     \
                     dim3(64 * kPairWaves), (pair_smem<R, ROWS>(2)), s, gray1, gray2_warped, dldd, gout,        \
                     static_cast<float *>(nullptr), H, W, im2, reinterpret_cast<const float2 *>(flow),          \
                     reinterpret_cast<float2 *>(d_flow))
#define LAUNCH_PAIR_FUSED2(ROWS) LAUNCH_PAIR_FUSED(2, ROWS)
#define LAUNCH_PAIR_FUSED3(ROWS) LAUNCH_PAIR_FUSED(3, ROWS)
  if (r == 2) UNFLOW_PAIR_ROWS_DISPATCH(prows, LAUNCH_PAIR_FUSED2);
  else UNFLOW_PAIR_ROWS_DISPATCH(prows, LAUNCH_PAIR_FUSED3);
#undef LAUNCH_PAIR_FUSED3
#undef LAUNCH_PAIR_FUSED2
#undef LAUNCH_PAIR_FUSED
  return launch_status("ternary_loss_warp_bwd");
}
