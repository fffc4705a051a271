// Warping kernels for gfx950: backward_warp (zero outside), image_warp (clamp),
// forward_warp (Gaussian splat) and the box downsample.
//
// Reference semantics (paths relative to the reference checkout):
//   backward_warp  ops/backward_warp_op.cu.cc:14-68, :70-138
//   image_warp     src/e2eflow/core/image_warp.py:4-76
//   forward_warp   ops/forward_warp_op.cu.cc:16-65, :67-125
//   downsample     ops/downsample_op.cu.cc:15-49
//
// All four are HBM-bound gathers/scatters over NHWC images; one lane owns one
// pixel so that the flow read (float2) and the index math are done once per
// pixel and the channel loop is unrolled at compile time for C = 1, 2, 3.
#include "common.h"
#include "warp_taps.h"

#include <cstdint>
#include <type_traits>

namespace unflow {
namespace {

constexpr int kBlock = 256;

// Makes a loaded value opaque to the optimiser (no instruction is emitted): a
// load that feeds only a select would otherwise be turned back into a branch
// around the load, one branch and one wait per tap.
__device__ inline void pin(float &v) { asm volatile("" : "+v"(v)); }

struct Taps {
  int x0, x1, y0, y1;
  float w_left, w_right, w_top, w_bottom;
};

// BackwardWarp: floor the POSITION (ops/backward_warp_op.cu.cc:27-38).
__device__ inline Taps position_taps(int src_x, int src_y, float2 f) {
  Taps t;
  const float x = (float)src_x + f.x;
  const float y = (float)src_y + f.y;
  t.x0 = (int)floorf(x);
  t.x1 = t.x0 + 1;
  t.y0 = (int)floorf(y);
  t.y1 = t.y0 + 1;
  t.w_right = x - (float)t.x0;
  t.w_left = (float)t.x1 - x;
  t.w_bottom = y - (float)t.y0;
  t.w_top = (float)t.y1 - y;
  return t;
}

template <int C>
__global__ __launch_bounds__(kBlock) void backward_warp_fwd_kernel(const float *__restrict__ images,
                                                                   const float2 *__restrict__ flows,
                                                                   float *__restrict__ output, int H, int W,
                                                                   int channels, long long total) {
  const int nc = C > 0 ? C : channels;
  for (long long v = blockIdx.x * (long long)kBlock + threadIdx.x; v < total;
       v += (long long)gridDim.x * kBlock) {
    const TiledPixel tp = tiled_pixel(v, H, W);   // total = tiled_total(B, H, W)
    if (!tp.ok) continue;
    const long long idx = tp.idx, b = tp.b;
    const int src_x = tp.x, src_y = tp.y;
    const Taps t = position_taps(src_x, src_y, flows[idx]);
    const bool in_x0 = t.x0 >= 0 && t.x0 < W, in_x1 = t.x1 >= 0 && t.x1 < W;
    const bool in_y0 = t.y0 >= 0 && t.y0 < H, in_y1 = t.y1 >= 0 && t.y1 < H;
    // all four taps are loaded from clamped (always valid) addresses and the
    // values of taps outside the image are replaced by 0: branch-free, so the
    // loads of a pixel are in flight together
    const int xc0 = min(max(t.x0, 0), W - 1), xc1 = min(max(t.x1, 0), W - 1);
    const int yc0 = min(max(t.y0, 0), H - 1), yc1 = min(max(t.y1, 0), H - 1);
    const float *base = images + b * H * W * nc;
    const float *p00 = base + ((long long)yc0 * W + xc0) * nc;
    const float *p01 = base + ((long long)yc0 * W + xc1) * nc;
    const float *p10 = base + ((long long)yc1 * W + xc0) * nc;
    const float *p11 = base + ((long long)yc1 * W + xc1) * nc;
    const bool in00 = in_x0 && in_y0, in01 = in_x1 && in_y0, in10 = in_x0 && in_y1, in11 = in_x1 && in_y1;
    const float w00 = t.w_left * t.w_top, w01 = t.w_right * t.w_top;
    const float w10 = t.w_left * t.w_bottom, w11 = t.w_right * t.w_bottom;
    float *out = output + idx * nc;
    // selects, not branches: a tap outside the image contributes fmaf(w, 0, sum) == sum
    // exactly; pin() keeps the compiler from sinking the loads back under a condition.
    // For the unrolled channel counts all loads of the pixel are issued before the first use.
    constexpr int NC = C > 0 ? C : 1;
    for (int cb = 0; cb < nc; cb += NC) {
      float l00[NC], l01[NC], l10[NC], l11[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        l00[c] = p00[cb + c];
        l01[c] = p01[cb + c];
        l10[c] = p10[cb + c];
        l11[c] = p11[cb + c];
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        pin(l00[c]); pin(l01[c]); pin(l10[c]); pin(l11[c]);
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float sum = fmaf(w00, in00 ? l00[c] : 0.f, 0.f);
        sum = fmaf(w01, in01 ? l01[c] : 0.f, sum);
        sum = fmaf(w10, in10 ? l10[c] : 0.f, sum);
        sum = fmaf(w11, in11 ? l11[c] : 0.f, sum);
        out[cb + c] = sum;
      }
    }
  }
}

template <int C>
__global__ __launch_bounds__(kBlock) void backward_warp_bwd_kernel(const float *__restrict__ input_grad,
                                                                   const float *__restrict__ images,
                                                                   const float2 *__restrict__ flows,
                                                                   float2 *__restrict__ output_grad, int H,
                                                                   int W, int channels, long long total) {
  const int nc = C > 0 ? C : channels;
  for (long long v = blockIdx.x * (long long)kBlock + threadIdx.x; v < total;
       v += (long long)gridDim.x * kBlock) {
    const TiledPixel tp = tiled_pixel(v, H, W);   // total = tiled_total(B, H, W)
    if (!tp.ok) continue;
    const long long idx = tp.idx, b = tp.b;
    const int src_x = tp.x, src_y = tp.y;
    const Taps t = position_taps(src_x, src_y, flows[idx]);
    const bool in_x0 = t.x0 >= 0 && t.x0 < W, in_x1 = t.x1 >= 0 && t.x1 < W;
    const bool in_y0 = t.y0 >= 0 && t.y0 < H, in_y1 = t.y1 >= 0 && t.y1 < H;
    const int xc0 = min(max(t.x0, 0), W - 1), xc1 = min(max(t.x1, 0), W - 1);
    const int yc0 = min(max(t.y0, 0), H - 1), yc1 = min(max(t.y1, 0), H - 1);
    const float *base = images + b * H * W * nc;
    const float *p00 = base + ((long long)yc0 * W + xc0) * nc;
    const float *p01 = base + ((long long)yc0 * W + xc1) * nc;
    const float *p10 = base + ((long long)yc1 * W + xc0) * nc;
    const float *p11 = base + ((long long)yc1 * W + xc1) * nc;
    const bool in00 = in_x0 && in_y0, in01 = in_x1 && in_y0, in10 = in_x0 && in_y1, in11 = in_x1 && in_y1;
    const float *gin = input_grad + idx * nc;
    float du = 0.f, dv = 0.f;
    constexpr int NC = C > 0 ? C : 1;
    for (int cb = 0; cb < nc; cb += NC) {
      float l00[NC], l01[NC], l10[NC], l11[NC], din[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        l00[c] = p00[cb + c];
        l01[c] = p01[cb + c];
        l10[c] = p10[cb + c];
        l11[c] = p11[cb + c];
        din[c] = gin[cb + c];
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        pin(l00[c]); pin(l01[c]); pin(l10[c]); pin(l11[c]);
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        // masked products: an outside tap adds fmaf(w, 0, d) == d exactly
        const float q00 = (in00 ? l00[c] : 0.f) * din[c], q01 = (in01 ? l01[c] : 0.f) * din[c];
        const float q10 = (in10 ? l10[c] : 0.f) * din[c], q11 = (in11 ? l11[c] : 0.f) * din[c];
        du = fmaf(-t.w_top, q00, du);
        dv = fmaf(-t.w_left, q00, dv);
        du = fmaf(t.w_top, q01, du);
        dv = fmaf(-t.w_right, q01, dv);
        du = fmaf(-t.w_bottom, q10, du);
        dv = fmaf(t.w_left, q10, dv);
        du = fmaf(t.w_bottom, q11, du);
        dv = fmaf(t.w_right, q11, dv);
      }
    }
    output_grad[idx] = make_float2(du, dv);
  }
}

template <int C>
__global__ __launch_bounds__(kBlock) void image_warp_fwd_kernel(const float *__restrict__ im,
                                                                const float2 *__restrict__ flow,
                                                                float *__restrict__ warped, int H, int W,
                                                                int channels, long long total) {
  const int nc = C > 0 ? C : channels;
  for (long long v = blockIdx.x * (long long)kBlock + threadIdx.x; v < total;
       v += (long long)gridDim.x * kBlock) {
    const TiledPixel tp = tiled_pixel(v, H, W);   // total = tiled_total(B, H, W)
    if (!tp.ok) continue;
    const long long idx = tp.idx, b = tp.b;
    const int pos_x = tp.x, pos_y = tp.y;
    const ClampTaps t = clamp_taps(pos_x, pos_y, flow[idx], H, W);
    const float *base = im + b * H * W * nc;
    const float *pa = base + t.ia * nc, *pb = base + t.ib * nc;
    const float *pc = base + t.ic * nc, *pd = base + t.id * nc;
    // separately rounded products combined by add_n (image_warp.py:41-44,73)
    const float wa = (1.f - t.xw) * (1.f - t.yw);
    const float wb = (1.f - t.xw) * t.yw;
    const float wc = t.xw * (1.f - t.yw);
    const float wd = t.xw * t.yw;
    float *out = warped + idx * nc;
    constexpr int NC = C > 0 ? C : 1;
    for (int cb = 0; cb < nc; cb += NC) {
      float la[NC], lb[NC], lc[NC], ld[NC];   // all taps of the pixel in flight before the first use
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        la[c] = pa[cb + c];
        lb[c] = pb[cb + c];
        lc[c] = pc[cb + c];
        ld[c] = pd[cb + c];
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float s = mul_rn(wa, la[c]);
        s = add_rn(s, mul_rn(wb, lb[c]));
        s = add_rn(s, mul_rn(wc, lc[c]));
        s = add_rn(s, mul_rn(wd, ld[c]));
        out[cb + c] = s;
      }
    }
  }
}

template <int C, bool WANT_IM, bool WANT_FLOW>
__global__ __launch_bounds__(kBlock) void image_warp_bwd_kernel(const float *__restrict__ grad,
                                                                const float *__restrict__ im,
                                                                const float2 *__restrict__ flow,
                                                                float *__restrict__ d_im,
                                                                float2 *__restrict__ d_flow, int H, int W,
                                                                int channels, long long total) {
  const int nc = C > 0 ? C : channels;
  constexpr int NC = C > 0 ? C : 1;
  for (long long v = blockIdx.x * (long long)kBlock + threadIdx.x; v < total;
       v += (long long)gridDim.x * kBlock) {
    const TiledPixel tp = tiled_pixel(v, H, W);   // total = tiled_total(B, H, W)
    if (!tp.ok) continue;
    const long long idx = tp.idx, b = tp.b;
    const int pos_x = tp.x, pos_y = tp.y;
    const ClampTaps t = clamp_taps(pos_x, pos_y, flow[idx], H, W);
    const long long item = b * H * W * nc;
    const float wa = (1.f - t.xw) * (1.f - t.yw);
    const float wb = (1.f - t.xw) * t.yw;
    const float wc = t.xw * (1.f - t.yw);
    const float wd = t.xw * t.yw;
    const float *g = grad + idx * nc;
    const float *pa = im + item + t.ia * nc, *pb = im + item + t.ib * nc;
    const float *pc = im + item + t.ic * nc, *pd = im + item + t.id * nc;
    float ga = 0.f, gb = 0.f, gc = 0.f, gd = 0.f;
    for (int cb = 0; cb < nc; cb += NC) {
      float gv[NC], la[NC], lb[NC], lc[NC], ld[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        gv[c] = g[cb + c];
        if (WANT_FLOW) {
          la[c] = pa[cb + c];
          lb[c] = pb[cb + c];
          lc[c] = pc[cb + c];
          ld[c] = pd[cb + c];
        }
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (WANT_IM) {
          // gradient of tf.gather: scatter-add at the clamped taps
          unsafeAtomicAdd(d_im + item + t.ia * nc + cb + c, wa * gv[c]);
          unsafeAtomicAdd(d_im + item + t.ib * nc + cb + c, wb * gv[c]);
          unsafeAtomicAdd(d_im + item + t.ic * nc + cb + c, wc * gv[c]);
          unsafeAtomicAdd(d_im + item + t.id * nc + cb + c, wd * gv[c]);
        }
        if (WANT_FLOW) {
          ga = fmaf(gv[c], la[c], ga);
          gb = fmaf(gv[c], lb[c], gb);
          gc = fmaf(gv[c], lc[c], gc);
          gd = fmaf(gv[c], ld[c], gd);
        }
      }
    }
    if (WANT_FLOW) {
      const float dxw = (gc - ga) * (1.f - t.yw) + (gd - gb) * t.yw;
      const float dyw = (gb - ga) * (1.f - t.xw) + (gd - gc) * t.xw;
      d_flow[idx] = make_float2(dxw, dyw);
    }
  }
}

// ---------------------------------------------------------------------------
// Fast paths of the four HBM-bound gathers for C = 1, 2, 3 (the reference's uses: images and
// flows).  The grid-stride kernels above hold ONE pixel per lane: a flow load, a wait, four tap
// loads, a wait, a store -- two dependent memory latencies per 32..40 bytes, and 64-bit tile
// arithmetic per pixel.  Here a 256-thread block owns a 64 x 16 pixel tile and every lane FOUR
// pixels of one column (rows y, y+4, y+8, y+12): the four flows are loaded together, then all
// sixteen taps, then the four results are stored -- four times the bytes in flight per wave, and
// the item / tile decomposition is done once per block in 32-bit arithmetic.  Arithmetic per
// pixel is the same expression as above (bit-identical results).
// ---------------------------------------------------------------------------
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ inline float2 nt_load2(const float2 *p) {
  const f32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x2_t *>(p));
  return make_float2(v.x, v.y);
}
__device__ inline void nt_store2(float2 v, float2 *p) {
  __builtin_nontemporal_store(f32x2_t{v.x, v.y}, reinterpret_cast<f32x2_t *>(p));
}
// one pixel (C floats, 4-byte aligned) as ONE non-temporal vector access
template <int C>
__device__ inline void nt_store_px(const float (&r)[C], float *p) {
  typedef float vec_t __attribute__((ext_vector_type(C == 3 ? 3 : (C == 2 ? 2 : 1)), aligned(4)));
  if constexpr (C == 1) __builtin_nontemporal_store(r[0], p);
  else if constexpr (C == 2) __builtin_nontemporal_store(vec_t{r[0], r[1]}, reinterpret_cast<vec_t *>(p));
  else __builtin_nontemporal_store(vec_t{r[0], r[1], r[2]}, reinterpret_cast<vec_t *>(p));
}
template <int C>
__device__ inline void nt_load_px(float (&r)[C], const float *p) {
  typedef float vec_t __attribute__((ext_vector_type(C == 3 ? 3 : (C == 2 ? 2 : 1)), aligned(4)));
  if constexpr (C == 1) {
    r[0] = __builtin_nontemporal_load(p);
  } else {
    const vec_t v = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(p));
#pragma unroll
    for (int c = 0; c < C; ++c) r[c] = v[c];
  }
}
// ---- fp16 STORAGE (extension, BASELINE config 5; not in the reference): the same kernels with half
// images / flows / gradients in HBM; every value is widened to f32 when loaded, the arithmetic is
// the f32 expression above, and the result is rounded to half once when stored.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <class T> struct FlowOf { typedef float2 type; };
template <> struct FlowOf<_Float16> { typedef f16x2_t type; };
__device__ inline float2 nt_load2(const f16x2_t *p) {
  const f16x2_t v = __builtin_nontemporal_load(p);
  return make_float2((float)v.x, (float)v.y);
}
__device__ inline void nt_store2(float2 v, f16x2_t *p) {
  __builtin_nontemporal_store(f16x2_t{(_Float16)v.x, (_Float16)v.y}, p);
}
// A half pixel of C = 3 is 6 bytes at a 2-byte aligned address: it is read as ONE 4-byte ALIGNED 8-byte
// access that also covers the half in front of or behind it (three 2-byte accesses per pixel and tap
// made these kernels instruction bound: 0.14 ms against 0.086 ms in f32; an 8-byte access at the
// pixel's own, every other time misaligned, address 0.098 ms).  `guard` carries the parity of the
// item's first half in bit 31 and, below it, the offset of the tensor's very last pixel (all ones when
// the block's item is not the last): there the access ends with the pixel, so nothing is read past
// the end of the tensor.
typedef _Float16 f16x4u_t __attribute__((ext_vector_type(4), aligned(2)));
typedef _Float16 f16x2u_t __attribute__((ext_vector_type(2), aligned(2)));
template <int C>
__device__ inline void load_tap(float (&r)[C], const float *base, unsigned off, unsigned) {
  const float *p = base + off;   // pointer first, constant channel index after: the C loads merge into one dwordx2 / x3
#pragma unroll
  for (int c = 0; c < C; ++c) r[c] = p[c];
}
template <int C>
__device__ inline void load_tap(float (&r)[C], const _Float16 *base, unsigned off, unsigned guard) {
  if constexpr (C == 3) {
    const unsigned sh = off == (guard & 0x7fffffffu) ? 1u : ((off + (guard >> 31)) & 1u);
    // (signed: the first pixel of an item that starts on an odd half reads the half in front of it)
    const f16x4u_t v = *reinterpret_cast<const f16x4u_t *>(base + ((int)off - (int)sh));
    r[0] = (float)(sh ? v.y : v.x);
    r[1] = (float)(sh ? v.z : v.y);
    r[2] = (float)(sh ? v.w : v.z);
  } else if constexpr (C == 2) {
    const f16x2_t v = *reinterpret_cast<const f16x2_t *>(base + off);
    r[0] = (float)v.x;
    r[1] = (float)v.y;
  } else {
    r[0] = (float)base[off];
  }
}
template <int C>
__device__ inline void nt_store_px(const float (&r)[C], _Float16 *p) {
  if constexpr (C == 3) {
    __builtin_nontemporal_store(f16x2u_t{(_Float16)r[0], (_Float16)r[1]}, reinterpret_cast<f16x2u_t *>(p));
    __builtin_nontemporal_store((_Float16)r[2], p + 2);
  } else if constexpr (C == 2) {
    __builtin_nontemporal_store(f16x2_t{(_Float16)r[0], (_Float16)r[1]}, reinterpret_cast<f16x2_t *>(p));
  } else {
    __builtin_nontemporal_store((_Float16)r[0], p);
  }
}

// pixels per lane; measured at B=32 384x1280 C=3: 1 -> 5.3, 2 -> 5.9, 4 -> 5.8, 8 -> 1.8 TB/s (backward_warp)
#ifdef UNFLOW_TUNE_PPL
constexpr int kPPL = UNFLOW_TUNE_PPL;            // diagnostic build (tools/run_warp_variants.sh)
#else
constexpr int kPPL = 2;
#endif
constexpr int kFastTileH = kTileH * kPPL;        // 16 rows
struct FastTile {
  int x, y0;          // lane's column, first row (rows y0 + 4 i)
  long long item_px;  // b * H * W
  int b;
  bool xok;
};
__device__ inline FastTile fast_tile(int H, int W, int ntx, int nty) {
  const int tile = blockIdx.x;
  const int tx = tile % ntx, ty = (tile / ntx) % nty, b = tile / (ntx * nty);
  FastTile t;
  t.x = tx * kTileW + (threadIdx.x & (kTileW - 1));
  t.y0 = ty * kFastTileH + (threadIdx.x >> 6);
  t.item_px = (long long)b * H * W;
  t.b = b;
  t.xok = t.x < W;
  return t;
}
inline bool fast_ok(int B, int H, int W, int C) {
#ifdef UNFLOW_TUNE_NOFAST
  return false;
#endif
  // 32-bit element offsets inside an item, one block per tile
  const long long tiles = (long long)B * ((H + kFastTileH - 1) / kFastTileH) * ((W + kTileW - 1) / kTileW);
  return C >= 1 && C <= 3 && (long long)H * W * 3 < (1LL << 30) && tiles < (1LL << 31) && tiles > 0;
}

template <int C, bool CLAMP, class T = float>   // CLAMP: image_warp (floor the flow, clamped taps); else backward_warp (zero outside)
__global__ __launch_bounds__(kBlock) void warp_fwd_fast_kernel(const T *__restrict__ images,
                                                               const typename FlowOf<T>::type *__restrict__ flows,
                                                               T *__restrict__ output, int H, int W, int ntx,
                                                               int nty) {
  const FastTile ft = fast_tile(H, W, ntx, nty);
  const T *base = images + ft.item_px * C;
  const typename FlowOf<T>::type *fl = flows + ft.item_px;
  T *out = output + ft.item_px * C;
  bool ok[kPPL];
  unsigned pix[kPPL];
  float2 f[kPPL];
#pragma unroll
  for (int i = 0; i < kPPL; ++i) {
    const int y = ft.y0 + kTileH * i;
    ok[i] = ft.xok && y < H;
    pix[i] = (unsigned)(min(y, H - 1) * W + min(ft.x, W - 1));     // clamped: always a valid address
    f[i] = nt_load2(fl + pix[i]);
  }
  unsigned o00[kPPL], o01[kPPL], o10[kPPL], o11[kPPL];
  float w00[kPPL], w01[kPPL], w10[kPPL], w11[kPPL];
  bool in00[kPPL], in01[kPPL], in10[kPPL], in11[kPPL];
#pragma unroll
  for (int i = 0; i < kPPL; ++i) {
    const int y = min(ft.y0 + kTileH * i, H - 1), x = min(ft.x, W - 1);
    if constexpr (CLAMP) {
      const ClampTaps t = clamp_taps(x, y, f[i], H, W);
      o00[i] = (unsigned)t.ia * C; o10[i] = (unsigned)t.ib * C; o01[i] = (unsigned)t.ic * C; o11[i] = (unsigned)t.id * C;
      // separately rounded products combined by add_n (image_warp.py:41-44,73)
      w00[i] = (1.f - t.xw) * (1.f - t.yw);   // wa
      w10[i] = (1.f - t.xw) * t.yw;           // wb
      w01[i] = t.xw * (1.f - t.yw);           // wc
      w11[i] = t.xw * t.yw;                   // wd
      in00[i] = in01[i] = in10[i] = in11[i] = true;
    } else {
      const Taps t = position_taps(x, y, f[i]);
      const bool in_x0 = t.x0 >= 0 && t.x0 < W, in_x1 = t.x1 >= 0 && t.x1 < W;
      const bool in_y0 = t.y0 >= 0 && t.y0 < H, in_y1 = t.y1 >= 0 && t.y1 < H;
      const int xc0 = min(max(t.x0, 0), W - 1), xc1 = min(max(t.x1, 0), W - 1);
      const int yc0 = min(max(t.y0, 0), H - 1), yc1 = min(max(t.y1, 0), H - 1);
      o00[i] = (unsigned)(yc0 * W + xc0) * C; o01[i] = (unsigned)(yc0 * W + xc1) * C;
      o10[i] = (unsigned)(yc1 * W + xc0) * C; o11[i] = (unsigned)(yc1 * W + xc1) * C;
      in00[i] = in_x0 && in_y0; in01[i] = in_x1 && in_y0; in10[i] = in_x0 && in_y1; in11[i] = in_x1 && in_y1;
      w00[i] = t.w_left * t.w_top; w01[i] = t.w_right * t.w_top;
      w10[i] = t.w_left * t.w_bottom; w11[i] = t.w_right * t.w_bottom;
    }
  }
  float l00[kPPL][C], l01[kPPL][C], l10[kPPL][C], l11[kPPL][C];
  // (fp16 storage, C = 3: offset of the tensor's last pixel if this block works on the last item)
  const unsigned guard = (ft.b == (int)(gridDim.x / (unsigned)(ntx * nty)) - 1 ? (unsigned)(H * W - 1) * C : 0x7fffffffu) |
                         ((unsigned)(ft.item_px * C) << 31);
#pragma unroll
  for (int i = 0; i < kPPL; ++i) {
    load_tap<C>(l00[i], base, o00[i], guard);
    load_tap<C>(l01[i], base, o01[i], guard);
    load_tap<C>(l10[i], base, o10[i], guard);
    load_tap<C>(l11[i], base, o11[i], guard);
  }
#pragma unroll
  for (int i = 0; i < kPPL; ++i)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      pin(l00[i][c]); pin(l01[i][c]); pin(l10[i][c]); pin(l11[i][c]);
    }
#pragma unroll
  for (int i = 0; i < kPPL; ++i) {
    float r[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if constexpr (CLAMP) {
        float s = mul_rn(w00[i], l00[i][c]);
        s = add_rn(s, mul_rn(w10[i], l10[i][c]));
        s = add_rn(s, mul_rn(w01[i], l01[i][c]));
        s = add_rn(s, mul_rn(w11[i], l11[i][c]));
        r[c] = s;
      } else {
        float sum = fmaf(w00[i], in00[i] ? l00[i][c] : 0.f, 0.f);
        sum = fmaf(w01[i], in01[i] ? l01[i][c] : 0.f, sum);
        sum = fmaf(w10[i], in10[i] ? l10[i][c] : 0.f, sum);
        sum = fmaf(w11[i], in11[i] ? l11[i][c] : 0.f, sum);
        r[c] = sum;
      }
    }
    if (ok[i]) nt_store_px<C>(r, out + (size_t)pix[i] * C);
  }
}

template <int C, bool CLAMP, class T = float>   // gradient with respect to the flow
__global__ __launch_bounds__(kBlock) void warp_bwd_flow_fast_kernel(const T *__restrict__ grad,
                                                                    const T *__restrict__ images,
                                                                    const typename FlowOf<T>::type *__restrict__ flows,
                                                                    typename FlowOf<T>::type *__restrict__ d_flow,
                                                                    float *__restrict__ zero_out, int H, int W, int ntx,
                                                                    int nty) {
  // zero_out (optional, [B,H,W,C] floats): zeroed pixel by pixel on the way -- the image gradient of image_warp is a
  // scatter-add into a zeroed buffer, and this kernel runs in front of it (one launch instead of a memset + this)
  const FastTile ft = fast_tile(H, W, ntx, nty);
  const T *base = images + ft.item_px * C;
  const T *gin = grad + ft.item_px * C;
  const typename FlowOf<T>::type *fl = flows + ft.item_px;
  typename FlowOf<T>::type *dout = d_flow + ft.item_px;
  bool ok[kPPL];
  unsigned pix[kPPL];
  float2 f[kPPL];
#pragma unroll
  for (int i = 0; i < kPPL; ++i) {
    const int y = ft.y0 + kTileH * i;
    ok[i] = ft.xok && y < H;
    pix[i] = (unsigned)(min(y, H - 1) * W + min(ft.x, W - 1));
    f[i] = nt_load2(fl + pix[i]);
  }
  unsigned o00[kPPL], o01[kPPL], o10[kPPL], o11[kPPL];
  bool in00[kPPL], in01[kPPL], in10[kPPL], in11[kPPL];
  float wx[kPPL][2], wy[kPPL][2];      // CLAMP: (xw, yw) fractions; else (w_left, w_right), (w_top, w_bottom)
#pragma unroll
  for (int i = 0; i < kPPL; ++i) {
    const int y = min(ft.y0 + kTileH * i, H - 1), x = min(ft.x, W - 1);
    if constexpr (CLAMP) {
      const ClampTaps t = clamp_taps(x, y, f[i], H, W);
      o00[i] = (unsigned)t.ia * C; o10[i] = (unsigned)t.ib * C; o01[i] = (unsigned)t.ic * C; o11[i] = (unsigned)t.id * C;
      wx[i][0] = t.xw; wy[i][0] = t.yw; wx[i][1] = 0.f; wy[i][1] = 0.f;
      in00[i] = in01[i] = in10[i] = in11[i] = true;
    } else {
      const Taps t = position_taps(x, y, f[i]);
      const bool in_x0 = t.x0 >= 0 && t.x0 < W, in_x1 = t.x1 >= 0 && t.x1 < W;
      const bool in_y0 = t.y0 >= 0 && t.y0 < H, in_y1 = t.y1 >= 0 && t.y1 < H;
      const int xc0 = min(max(t.x0, 0), W - 1), xc1 = min(max(t.x1, 0), W - 1);
      const int yc0 = min(max(t.y0, 0), H - 1), yc1 = min(max(t.y1, 0), H - 1);
      o00[i] = (unsigned)(yc0 * W + xc0) * C; o01[i] = (unsigned)(yc0 * W + xc1) * C;
      o10[i] = (unsigned)(yc1 * W + xc0) * C; o11[i] = (unsigned)(yc1 * W + xc1) * C;
      in00[i] = in_x0 && in_y0; in01[i] = in_x1 && in_y0; in10[i] = in_x0 && in_y1; in11[i] = in_x1 && in_y1;
      wx[i][0] = t.w_left; wx[i][1] = t.w_right; wy[i][0] = t.w_top; wy[i][1] = t.w_bottom;
    }
  }
  float l00[kPPL][C], l01[kPPL][C], l10[kPPL][C], l11[kPPL][C], g[kPPL][C];
  const unsigned guard = (ft.b == (int)(gridDim.x / (unsigned)(ntx * nty)) - 1 ? (unsigned)(H * W - 1) * C : 0x7fffffffu) |
                         ((unsigned)(ft.item_px * C) << 31);
#pragma unroll
  for (int i = 0; i < kPPL; ++i) {
    load_tap<C>(l00[i], base, o00[i], guard);
    load_tap<C>(l01[i], base, o01[i], guard);
    load_tap<C>(l10[i], base, o10[i], guard);
    load_tap<C>(l11[i], base, o11[i], guard);
  }
#pragma unroll
  for (int i = 0; i < kPPL; ++i) {
    if constexpr (std::is_same<T, float>::value) nt_load_px<C>(g[i], gin + (size_t)pix[i] * C);
    else load_tap<C>(g[i], gin, pix[i] * C, guard);
  }
#pragma unroll
  for (int i = 0; i < kPPL; ++i)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      pin(l00[i][c]); pin(l01[i][c]); pin(l10[i][c]); pin(l11[i][c]);
    }
#pragma unroll
  for (int i = 0; i < kPPL; ++i) {
    float2 d;
    if constexpr (CLAMP) {
      // taps a = (y0,x0), b = (y1,x0), c = (y0,x1), d = (y1,x1)   (image_warp_bwd_kernel)
      float ga = 0.f, gb = 0.f, gc = 0.f, gd = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        ga = fmaf(g[i][c], l00[i][c], ga);
        gb = fmaf(g[i][c], l10[i][c], gb);
        gc = fmaf(g[i][c], l01[i][c], gc);
        gd = fmaf(g[i][c], l11[i][c], gd);
      }
      const float xw = wx[i][0], yw = wy[i][0];
      d.x = (gc - ga) * (1.f - yw) + (gd - gb) * yw;
      d.y = (gb - ga) * (1.f - xw) + (gd - gc) * xw;
    } else {
      float du = 0.f, dv = 0.f;
      const float w_left = wx[i][0], w_right = wx[i][1], w_top = wy[i][0], w_bottom = wy[i][1];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        // masked products: an outside tap adds fmaf(w, 0, d) == d exactly   (backward_warp_bwd_kernel)
        const float q00 = (in00[i] ? l00[i][c] : 0.f) * g[i][c], q01 = (in01[i] ? l01[i][c] : 0.f) * g[i][c];
        const float q10 = (in10[i] ? l10[i][c] : 0.f) * g[i][c], q11 = (in11[i] ? l11[i][c] : 0.f) * g[i][c];
        du = fmaf(-w_top, q00, du);
        dv = fmaf(-w_left, q00, dv);
        du = fmaf(w_top, q01, du);
        dv = fmaf(-w_right, q01, dv);
        du = fmaf(-w_bottom, q10, du);
        dv = fmaf(w_left, q10, dv);
        du = fmaf(w_bottom, q11, du);
        dv = fmaf(w_right, q11, dv);
      }
      d = make_float2(du, dv);
    }
    if (ok[i]) nt_store2(d, dout + pix[i]);
    if (zero_out && ok[i]) {
      float *z = zero_out + (ft.item_px + pix[i]) * C;
#pragma unroll
      for (int c = 0; c < C; ++c) z[c] = 0.f;
    }
  }
}

// ForwardWarp window (ops/forward_warp_op.cu.cc:38-51): k = ceil(dist+2) = 4.
struct Window {
  float tx, ty;
  int min_x, max_x, min_y, max_y;
  bool any;
};

__device__ inline Window splat_window(int src_x, int src_y, float2 f, int H, int W) {
  Window w;
  const float k = 4.f;
  w.tx = (float)src_x + f.x;
  w.ty = (float)src_y + f.y;
  w.any = floorf(w.tx - k) < (float)W && floorf(w.tx + k) >= 0.f && floorf(w.ty - k) < (float)H &&
          floorf(w.ty + k) >= 0.f;
  w.min_x = w.tx - k > 0.f ? (int)floorf(w.tx - k) : 0;
  w.min_y = w.ty - k > 0.f ? (int)floorf(w.ty - k) : 0;
  w.max_x = w.tx + k < (float)W ? (int)floorf(w.tx + k) : W - 1;
  w.max_y = w.ty + k < (float)H ? (int)floorf(w.ty + k) : H - 1;
  return w;
}

// Splat, LDS-privatised.  One workgroup owns a 64x16 tile of SOURCE pixels and
// an LDS accumulation window of output pixels sized and placed to the bounding
// box of everything the tile's pixels touch (up to 15360 cells = 60 KB: 96x48 for
// a coherent flow, 133x85 for i.i.d. +-30 px flows).  Taps inside the window are
// accumulated in LDS; if the box does not fit, the window is centred on where the
// tile-centre pixel lands and the rest falls back to global float atomics.  The
// window is then flushed with one global atomic per touched output pixel, in
// contiguous rows -- 3 to 11 global atomics per source pixel instead of 81.
//
// The LDS accumulator is 32-bit FIXED POINT: on gfx950 ds_add_f32 retires about
// one lane per 3 cycles (194 cycles per wave instruction, measured) while
// ds_add_u32 takes 5 cycles per wave instruction.  A tile has 1024 source pixels
// and a weight is at most 1, so with the scale 2^21 a cell holds at most
// 1024 * 2^21 = 2^31 and cannot overflow.  The granule 2^-21 ~ 4.8e-7 equals
// the float32 ulp at the typical density (2*pi), and integer adds make the
// in-window sum independent of the order of the adds.
// The Gaussian is separable: 9 + 9 instead of 81 exponentials per pixel.
constexpr int kSX = 64, kSY = 16;          // source tile
constexpr int kWinCells = 15360;           // LDS window capacity (60 KB)
// image gradient of image_warp: columns and rows of a workgroup's pixel tile and ints of LDS for its two windows,
// for three channels and for one or two
#ifdef UNFLOW_TUNE_IM3
constexpr int kIm3Cols = UNFLOW_TUNE_IM3_COLS, kIm3Rows = UNFLOW_TUNE_IM3_ROWS, kIm3Win = UNFLOW_TUNE_IM3_WIN;
#else
constexpr int kIm3Cols = 32, kIm3Rows = 16, kIm3Win = 10240;
#endif
#ifdef UNFLOW_TUNE_IM2
constexpr int kIm2Cols = UNFLOW_TUNE_IM2_COLS, kIm2Rows = UNFLOW_TUNE_IM2_ROWS, kIm2Win = UNFLOW_TUNE_IM2_WIN;
#else
constexpr int kIm2Cols = 32, kIm2Rows = 16, kIm2Win = 7680;
#endif
constexpr int kWW = 96, kWH = 48;          // fallback window when the bounding box does not fit
constexpr int kSplatThreads = kSX * kSY;   // one lane per source pixel: 16 waves, two workgroups (120 KB LDS) fill a CU
constexpr float kFix = 2097152.f;          // 2^21
constexpr float kUnfix = 1.f / 2097152.f;
static_assert(kSX * kSY <= 1024, "fixed-point headroom assumes <= 1024 source pixels per window");
static_assert(kWW * kWH <= kWinCells, "fallback window must fit");

// float -> fixed point, round to nearest (ties up): ONE instruction, v_cvt_rpi_i32_f32 = floor(x + 0.5)
// (rintf + a conversion are two); x >= 0 here
__device__ inline unsigned fix_round(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return (unsigned)r;
}

__device__ inline float gauss(float d) {   // exp(-d*d/2) through v_exp_f32
  return __builtin_amdgcn_exp2f(d * d * -0.72134752044448170368f);
}

#ifdef UNFLOW_TUNE_FW_STAMPS
// Diagnostic build only (tools/stamps_fwarp.py): shader-clock stamps of the splat's phases for 64
// workgroups (waves 0 and 15), read back through unflow_debug_read_fw_stamps().  [64 blocks][2 waves][8]
__device__ unsigned long long g_fw_stamps[64 * 2 * 8];
#define UNFLOW_FW_STAMP(I)                                                                                  \
  do {                                                                                                      \
    if (fw_slot >= 0 && (threadIdx.x & 63) == 0) g_fw_stamps[fw_slot * 8 + (I)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define UNFLOW_FW_STAMP(I) do {} while (0)
#endif
__device__ inline float2 flow_at(const float2 *p, long long i) { return p[i]; }
__device__ inline float2 flow_at(const f16x2_t *p, long long i) {
  const f16x2_t v = p[i];
  return make_float2((float)v.x, (float)v.y);
}

template <class FT>   // float2, or f16x2_t (fp16 storage of the flows; the density is accumulated and written in f32)
__global__ __launch_bounds__(kSplatThreads) void forward_warp_fwd_kernel(const FT *__restrict__ flows,
                                                                         float *__restrict__ output, int H, int W) {
  __shared__ unsigned win[kWinCells];
  __shared__ int box[4];                     // min_x, min_y, -max_x, -max_y (all reduced with atomicMin)
  const int x0 = blockIdx.x * kSX, y0 = blockIdx.y * kSY;
  const long long item = (long long)blockIdx.z * H * W;
  float *out = output + item;
#ifdef UNFLOW_TUNE_FW_STAMPS
  int fw_slot = -1;
  {
    // item 5, tile rows 8..11, tile columns 4..19: 64 interior tiles in the middle of the launch
    const int wv = threadIdx.x >> 6;
    if (blockIdx.z == 5 && blockIdx.y >= 8 && blockIdx.y < 12 && blockIdx.x >= 4 && blockIdx.x < 20 && (wv == 0 || wv == 15))
      fw_slot = (((int)blockIdx.y - 8) * 16 + ((int)blockIdx.x - 4)) * 2 + (wv == 15);
  }
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz: calibrates the shader clock
#endif
  UNFLOW_FW_STAMP(0);

  const int sx = x0 + (threadIdx.x % kSX), sy = y0 + threadIdx.x / kSX;
  Window w;
  w.any = false;
  if (sx < W && sy < H) w = splat_window(sx, sy, flow_at(flows, item + (long long)sy * W + sx), H, W);

  // First guess (free for a coherent flow): a 96x48 window centred on where the
  // tile-centre pixel lands.  Only if some lane's patch misses it is the tile's
  // bounding box reduced (wave shuffles, then one LDS atomic per wave) and, when it
  // fits the LDS capacity, used as the window instead.
  const int cxp = min(x0 + kSX / 2, W - 1), cyp = min(y0 + kSY / 2, H - 1);
  const float2 fc = flow_at(flows, item + (long long)cyp * W + cxp);
  // clamped so that a NaN/huge centre flow still gives a usable (if useless) window
  int wx0 = (int)fminf(fmaxf(floorf((float)cxp + fc.x), -1e6f), 1e6f) - kWW / 2;
  int wy0 = (int)fminf(fmaxf(floorf((float)cyp + fc.y), -1e6f), 1e6f) - kWH / 2;
  int ww = kWW, wh = kWH;
  const bool miss = w.any && !(w.min_x >= wx0 && w.max_x < wx0 + kWW && w.min_y >= wy0 && w.max_y < wy0 + kWH);
  if (threadIdx.x < 4) box[threadIdx.x] = 0x7fffffff;
  UNFLOW_FW_STAMP(1);                        // flows loaded, window guessed
  const bool any_miss = __syncthreads_or(miss);
  UNFLOW_FW_STAMP(2);
  if (any_miss) {
    int b0 = w.any ? w.min_x : 0x7fffffff, b1 = w.any ? w.min_y : 0x7fffffff;
    int b2 = w.any ? -w.max_x : 0x7fffffff, b3 = w.any ? -w.max_y : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      b0 = min(b0, __shfl_xor(b0, off, 64));
      b1 = min(b1, __shfl_xor(b1, off, 64));
      b2 = min(b2, __shfl_xor(b2, off, 64));
      b3 = min(b3, __shfl_xor(b3, off, 64));
    }
    if ((threadIdx.x & 63) == 0 && b0 != 0x7fffffff) {
      atomicMin(&box[0], b0);
      atomicMin(&box[1], b1);
      atomicMin(&box[2], b2);
      atomicMin(&box[3], b3);
    }
    __syncthreads();
    const int bw = -box[2] - box[0] + 1, bh = -box[3] - box[1] + 1;
    if (box[0] != 0x7fffffff && (long long)bw * bh <= kWinCells) {
      wx0 = box[0];
      wy0 = box[1];
      ww = bw;
      wh = bh;
    }
  }
  const int cells = ww * wh;
  for (int i = threadIdx.x; i < cells; i += kSplatThreads) win[i] = 0u;
  __syncthreads();
  UNFLOW_FW_STAMP(3);                        // window zeroed

  if (w.any) {
    float ex[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) ex[a] = gauss((float)(w.min_x + a) - w.tx);
    const int lx0 = w.min_x - wx0, ly0 = w.min_y - wy0;
    const int nx = w.max_x - w.min_x, ny = w.max_y - w.min_y;   // taps - 1 per axis (<= 8)
    const bool inside = lx0 >= 0 && lx0 + nx < ww && ly0 >= 0 && ly0 + ny < wh;
    // floor(t + 4) - floor(t - 4) = 8 for every t: a patch that the image border does not clip has 9 x 9 taps.
    // When that holds for every lane of the wave (all but the waves at the image border) the taps need no
    // per-tap predicate: multiply, round, add -- 2 vector instructions per tap instead of 8 (profiles/
    // r03_pmc_loss.txt showed 72 % VALU issue for the predicated loop).  The kernel's time did not move
    // (0.377 ms either way): it is bound by the LDS atomics themselves, as DESIGN 4.3 says.
    if (__builtin_amdgcn_ballot_w64(!(inside && nx == 8 && ny == 8)) == 0) {
      unsigned *wp = win + ly0 * ww + lx0;
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const float ey = gauss((float)(w.min_y + r) - w.ty) * kFix;
#pragma unroll
        for (int a = 0; a < 9; ++a) atomicAdd(wp + a, fix_round(ex[a] * ey));
        wp += ww;
      }
    } else if (inside) {
      unsigned *wp = win + ly0 * ww + lx0;
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        if (r > ny) break;
        const float ey = gauss((float)(w.min_y + r) - w.ty) * kFix;
#pragma unroll
        for (int a = 0; a < 9; ++a)
          if (a <= nx) atomicAdd(wp + r * ww + a, fix_round(ex[a] * ey));
      }
    } else {
      for (int r = 0; r <= ny; ++r) {
        const int n_y = w.min_y + r;
        const float ey = gauss((float)n_y - w.ty);
        const int ly = ly0 + r;
        const bool rowin = ly >= 0 && ly < wh;
#pragma unroll
        for (int a = 0; a < 9; ++a) {
          if (a > nx) break;
          const float weight = ex[a] * ey;
          const int lx = lx0 + a;
          if (rowin && lx >= 0 && lx < ww)
            atomicAdd(&win[ly * ww + lx], fix_round(weight * kFix));
          else
            unsafeAtomicAdd(out + (long long)n_y * W + w.min_x + a, weight);
        }
      }
    }
  }
  UNFLOW_FW_STAMP(4);                        // taps issued
  __syncthreads();
  UNFLOW_FW_STAMP(5);
  // flush: wave r takes window rows r, r+16, ...; lanes walk the row (no runtime division)
  for (int ly = threadIdx.x >> 6; ly < wh; ly += kSplatThreads / 64) {
    const int gy = wy0 + ly;
    if (gy < 0 || gy >= H) continue;
    for (int lx = threadIdx.x & 63; lx < ww; lx += 64) {
      const unsigned v = win[ly * ww + lx];
      const int gx = wx0 + lx;
      if (v != 0u && gx >= 0 && gx < W) unsafeAtomicAdd(out + (long long)gy * W + gx, (float)v * kUnfix);
    }
  }
  UNFLOW_FW_STAMP(6);                        // flush issued
#ifdef UNFLOW_TUNE_FW_STAMPS
  if (fw_slot >= 0 && (threadIdx.x & 63) == 0) g_fw_stamps[fw_slot * 8 + 7] = __builtin_amdgcn_s_memrealtime() - rt0;
#endif
}

// Gradient of the splat with the tile's window of input_grad staged in LDS: the same
// 64x16 source tile and window choice as the forward kernel (centre guess, else the
// bounding box when it fits 60 KB); each lane then reads its 9x9 patch from LDS instead
// of issuing 81 global loads (9 dependent rounds of 9).  Patches outside the staged
// window fall back to global loads.
__global__ __launch_bounds__(kSplatThreads, 8) void forward_warp_bwd_tiled_kernel(const float *__restrict__ input_grad,
                                                                               const float2 *__restrict__ flows,
                                                                               float2 *__restrict__ output_grad,
                                                                               int H, int W) {
  __shared__ float win[kWinCells];
  __shared__ int box[4];
  const int x0 = blockIdx.x * kSX, y0 = blockIdx.y * kSY;
  const long long item = (long long)blockIdx.z * H * W;
  const float *gin = input_grad + item;
  const int sx = x0 + (threadIdx.x % kSX), sy = y0 + threadIdx.x / kSX;
  const bool live = sx < W && sy < H;
  Window w;
  w.any = false;
  if (live) w = splat_window(sx, sy, flows[item + (long long)sy * W + sx], H, W);

  const int cxp = min(x0 + kSX / 2, W - 1), cyp = min(y0 + kSY / 2, H - 1);
  const float2 fc = flows[item + (long long)cyp * W + cxp];
  int wx0 = (int)fminf(fmaxf(floorf((float)cxp + fc.x), -1e6f), 1e6f) - kWW / 2;
  int wy0 = (int)fminf(fmaxf(floorf((float)cyp + fc.y), -1e6f), 1e6f) - kWH / 2;
  int ww = kWW, wh = kWH;
  const bool miss = w.any && !(w.min_x >= wx0 && w.max_x < wx0 + kWW && w.min_y >= wy0 && w.max_y < wy0 + kWH);
  if (threadIdx.x < 4) box[threadIdx.x] = 0x7fffffff;
  if (__syncthreads_or(miss)) {
    int b0 = w.any ? w.min_x : 0x7fffffff, b1 = w.any ? w.min_y : 0x7fffffff;
    int b2 = w.any ? -w.max_x : 0x7fffffff, b3 = w.any ? -w.max_y : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      b0 = min(b0, __shfl_xor(b0, off, 64));
      b1 = min(b1, __shfl_xor(b1, off, 64));
      b2 = min(b2, __shfl_xor(b2, off, 64));
      b3 = min(b3, __shfl_xor(b3, off, 64));
    }
    if ((threadIdx.x & 63) == 0 && b0 != 0x7fffffff) {
      atomicMin(&box[0], b0);
      atomicMin(&box[1], b1);
      atomicMin(&box[2], b2);
      atomicMin(&box[3], b3);
    }
    __syncthreads();
    const int bw = -box[2] - box[0] + 1, bh = -box[3] - box[1] + 1;
    if (box[0] != 0x7fffffff && (long long)bw * bh <= kWinCells) {
      wx0 = box[0];
      wy0 = box[1];
      ww = bw;
      wh = bh;
    }
  }
  // stage the window (zero outside the image: such cells are never inside a clamped patch)
  for (int ly = threadIdx.x >> 6; ly < wh; ly += kSplatThreads / 64) {
    const int gy = wy0 + ly;
    for (int lx = threadIdx.x & 63; lx < ww; lx += 64) {
      const int gx = wx0 + lx;
      win[ly * ww + lx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? gin[(long long)gy * W + gx] : 0.f;
    }
  }
  __syncthreads();
  if (!live) return;
  float du = 0.f, dv = 0.f;
  if (w.any) {
    float ex[9], dx[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      dx[a] = (float)(w.min_x + a) - w.tx;
      ex[a] = gauss(dx[a]);
    }
    const int nx = w.max_x - w.min_x, ny = w.max_y - w.min_y;
    const int lx0 = w.min_x - wx0, ly0 = w.min_y - wy0;
    const bool inside = lx0 >= 0 && lx0 + nx < ww && ly0 >= 0 && ly0 + ny < wh;
    if (inside) {
      const float *wp = win + ly0 * ww + lx0;
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        if (r > ny) break;
        const float y = (float)(w.min_y + r) - w.ty;
        const float ey = gauss(y);
        float sxx = 0.f, s0 = 0.f;
#pragma unroll
        for (int a = 0; a < 9; ++a) {
          const float f = a <= nx ? wp[r * ww + min(a, nx)] * ex[a] : 0.f;
          sxx = fmaf(f, dx[a], sxx);
          s0 += f;
        }
        du = fmaf(ey, sxx, du);
        dv = fmaf(ey * y, s0, dv);
      }
    } else {
#pragma unroll 1
      for (int n_y = w.min_y; n_y <= w.max_y; ++n_y) {
        const float y = (float)n_y - w.ty;
        const float ey = gauss(y);
        const float *grow = gin + (long long)n_y * W + w.min_x;
        float gv[9];
#pragma unroll
        for (int a = 0; a < 9; ++a) gv[a] = grow[min(a, nx)];
#pragma unroll
        for (int a = 0; a < 9; ++a) pin(gv[a]);
        float sxx = 0.f, s0 = 0.f;
#pragma unroll
        for (int a = 0; a < 9; ++a) {
          const float f = a <= nx ? gv[a] * ex[a] : 0.f;
          sxx = fmaf(f, dx[a], sxx);
          s0 += f;
        }
        du = fmaf(ey, sxx, du);
        dv = fmaf(ey * y, s0, dv);
      }
    }
  }
  reinterpret_cast<float2 *>(output_grad)[item + (long long)sy * W + sx] = make_float2(du, dv);
}

__global__ __launch_bounds__(kBlock) void forward_warp_bwd_kernel(const float *__restrict__ input_grad,
                                                                  const float2 *__restrict__ flows,
                                                                  float2 *__restrict__ output_grad, int H,
                                                                  int W, long long total) {
  // du = sum_n 2*din[n]*w(n)*(n_x - t_x)/2,  w separable: 9 + 9 exponentials per pixel
  for (long long v = blockIdx.x * (long long)kBlock + threadIdx.x; v < total;
       v += (long long)gridDim.x * kBlock) {
    const TiledPixel tp = tiled_pixel(v, H, W);   // total = tiled_total(B, H, W)
    if (!tp.ok) continue;
    const long long idx = tp.idx, b = tp.b;
    const int src_x = tp.x, src_y = tp.y;
    const Window w = splat_window(src_x, src_y, flows[idx], H, W);
    float du = 0.f, dv = 0.f;
    if (w.any) {
      const float *gin = input_grad + b * H * W;
      float ex[9], dx[9];
#pragma unroll
      for (int a = 0; a < 9; ++a) {
        dx[a] = (float)(w.min_x + a) - w.tx;
        ex[a] = gauss(dx[a]);
      }
      const int nx = w.max_x - w.min_x;
#pragma unroll 1
      for (int n_y = w.min_y; n_y <= w.max_y; ++n_y) {
        const float y = (float)n_y - w.ty;
        const float ey = gauss(y);
        const float *grow = gin + (long long)n_y * W + w.min_x;
        // clamped, unconditional loads (a guarded load is a branch and a wait per tap);
        // taps beyond the window are masked afterwards
        float gv[9];
#pragma unroll
        for (int a = 0; a < 9; ++a) gv[a] = grow[min(a, nx)];
#pragma unroll
        for (int a = 0; a < 9; ++a) pin(gv[a]);
        float sx = 0.f, s0 = 0.f;              // sum_x din*ex*x and sum_x din*ex of this row
#pragma unroll
        for (int a = 0; a < 9; ++a) {
          const float f = a <= nx ? gv[a] * ex[a] : 0.f;
          sx = fmaf(f, dx[a], sx);
          s0 += f;
        }
        du = fmaf(ey, sx, du);
        dv = fmaf(ey * y, s0, dv);
      }
    }
    output_grad[idx] = make_float2(du, dv);
  }
}

__global__ __launch_bounds__(kBlock) void downsample_kernel(const float *__restrict__ images,
                                                            float *__restrict__ output, int in_height,
                                                            int in_width, int channels, int out_height,
                                                            int out_width, int scale, long long total) {
  for (long long out_idx = blockIdx.x * (long long)kBlock + threadIdx.x; out_idx < total;
       out_idx += (long long)gridDim.x * kBlock) {
    long long idx = out_idx;
    const int c = (int)(idx % channels);
    idx /= channels;
    const int x = (int)(idx % out_width);
    idx /= out_width;
    const int y = (int)(idx % out_height);
    const long long b = idx / out_height;
    const float *src = images + ((b * in_height + (long long)y * scale) * in_width + (long long)x * scale) * channels + c;
    float sum = 0.f;
    for (int dy = 0; dy < scale; ++dy)
      for (int dx = 0; dx < scale; ++dx) sum += src[((long long)dy * in_width + dx) * channels];
    output[out_idx] = sum / (float)(scale * scale);
  }
}

// Downsample for the shapes the loss pyramid uses (C <= 3, scale 2 or 4): one lane per OUTPUT PIXEL.
// The scale*C floats a pixel takes from one input row are contiguous and 8- (scale 2) or 16-byte
// (scale 4) aligned, so a row is C vector loads instead of scale*C strided scalar ones, and a wave
// reads whole rows.  Per channel the taps are added in the same row-major order as above.
template <int C, int SCALE>
__global__ __launch_bounds__(kBlock) void downsample_px_kernel(const float *__restrict__ images,
                                                               float *__restrict__ output, int in_width,
                                                               int out_height, int out_width, long long npx) {
  typedef float vec_t __attribute__((ext_vector_type(SCALE)));       // SCALE floats: C of them per row
  const long long p = blockIdx.x * (long long)kBlock + threadIdx.x;
  if (p >= npx) return;
  const int x = (int)(p % out_width);
  const long long by = p / out_width;                                  // b * out_height + y
  const vec_t *src = reinterpret_cast<const vec_t *>(images + (by * SCALE * in_width + (long long)x * SCALE) * C);
  float sum[C];
#pragma unroll
  for (int c = 0; c < C; ++c) sum[c] = 0.f;
  vec_t row[SCALE][C];
#pragma unroll
  for (int dy = 0; dy < SCALE; ++dy)
#pragma unroll
    for (int k = 0; k < C; ++k) row[dy][k] = __builtin_nontemporal_load(src + (long long)dy * (in_width * C / SCALE) + k);   // W % SCALE == 0
#pragma unroll
  for (int dy = 0; dy < SCALE; ++dy)
#pragma unroll
    for (int dx = 0; dx < SCALE; ++dx)
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int e = dx * C + c;                                      // float index inside the row's SCALE*C span
        sum[c] += row[dy][e / SCALE][e % SCALE];
      }
  float *out = output + p * C;
#pragma unroll
  for (int c = 0; c < C; ++c) out[c] = sum[c] / (float)(SCALE * SCALE);
}

// The image chain of the loss pyramid (core/unsupervised.py:97-103,143-145: downsample by 4, then by 2 per level) in ONE
// launch: a workgroup reads a 64 x 64 block of one image (coalesced 16-byte loads, 16 rows at a time through LDS) and
// writes its 16 x 16 pixels of level 0, 8 x 8 of level 1, ... 1 x 1 of level 4.  Every output is computed exactly as the
// chain of downsample_px_kernel launches computes it -- the taps of a window added one after the other in row-major
// order, then the division -- so the results are the chain's bit for bit.  With two source tensors the outputs are
// the stacked batches the pyramid driver wants: ab = [a; b] and (optionally) ba = [b; a] along the batch.
constexpr int kPyrLevels = 5, kPyrBlock = 64, kPyrStrip = 16;
struct PyrArgs {
  const float *src[2];
  float *ab[kPyrLevels], *ba[kPyrLevels];
  int dup;            // the two sources are one tensor: read it once, write both halves of ab
};
template <int C>
__global__ __launch_bounds__(256) void downsample_pyramid_kernel(PyrArgs a, int levels, int B, int H, int W) {
  __shared__ __attribute__((aligned(16))) float strip[kPyrStrip * kPyrBlock * C];
  __shared__ float lv[16 * 16 * C + 8 * 8 * C + 4 * 4 * C + 2 * 2 * C];
  const int which = blockIdx.z / B, b = blockIdx.z % B;      // (a launch over ONE source that is both a and b has z < B)
  const bool twice = a.dup != 0;
  const float *src = a.src[which] + (long long)b * H * W * C;
  const int x0 = blockIdx.x * kPyrBlock, y0 = blockIdx.y * kPyrBlock;
  const int H0 = H / 4, W0 = W / 4;
  const long long item_ab = (long long)which * B + b, item_ba = (long long)(1 - which) * B + b;
  constexpr int kRowF4 = kPyrBlock * C / 4;                  // 16-byte pieces of a block row
  const int row_floats = min(kPyrBlock, W - x0) * C;         // W % 4 == 0: whole pieces
  typedef float f4_t __attribute__((ext_vector_type(4)));
  constexpr int kPieces = kPyrStrip * kRowF4, kPerThread = (kPieces + 255) / 256;
  f4_t held[kPerThread];                                     // the next strip, in flight while this one is summed
  auto fetch = [&](int st) {
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
      const int i = threadIdx.x + k * 256;
      const int r = i / kRowF4, q = i - r * kRowF4;
      const int y = y0 + st * kPyrStrip + r;
      f4_t v = {0.f, 0.f, 0.f, 0.f};
      if (i < kPieces && y < H && q * 4 < row_floats)
        v = __builtin_nontemporal_load(reinterpret_cast<const f4_t *>(src + ((long long)y * W + x0) * C) + q);
      held[k] = v;
    }
  };
  fetch(0);
  for (int st = 0; st < kPyrBlock / kPyrStrip; ++st) {
    __syncthreads();                                         // the previous strip has been consumed
#pragma unroll
    for (int k = 0; k < kPerThread; ++k)
      if (threadIdx.x + k * 256 < kPieces) reinterpret_cast<f4_t *>(strip)[threadIdx.x + k * 256] = held[k];
    __syncthreads();
    if (st + 1 < kPyrBlock / kPyrStrip) fetch(st + 1);
    for (int o = threadIdx.x; o < 4 * 16 * C; o += 256) {    // 4 rows x 16 pixels x C of level 0
      const int c = o % C, px = (o / C) % 16, py = o / (16 * C);
      float sum = 0.f;
#pragma unroll
      for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) sum += strip[((py * 4 + dy) * kPyrBlock + px * 4 + dx) * C + c];
      const float v = sum / 16.f;
      const int ly = st * 4 + py;
      lv[(ly * 16 + px) * C + c] = v;
      const int gx = x0 / 4 + px, gy = y0 / 4 + ly;
      if (gx < W0 && gy < H0) {
        const long long at = ((long long)gy * W0 + gx) * C + c, per = (long long)H0 * W0 * C;
        a.ab[0][item_ab * per + at] = v;
        if (a.ba[0]) a.ba[0][item_ba * per + at] = v;
        if (twice) a.ab[0][(item_ab + B) * per + at] = v;
      }
    }
  }
  int in_off = 0, side = 16;
  for (int l = 1; l < levels; ++l) {
    __syncthreads();
    const int half = side / 2, out_off = in_off + side * side * C;
    const int Hl = H0 >> l, Wl = W0 >> l;
    for (int o = threadIdx.x; o < half * half * C; o += 256) {
      const int c = o % C, px = (o / C) % half, py = o / (half * C);
      const float *p = lv + in_off + ((py * 2) * side + px * 2) * C + c;
      float sum = 0.f;
      sum += p[0];
      sum += p[C];
      sum += p[side * C];
      sum += p[side * C + C];
      const float v = sum / 4.f;
      if (l < 4) lv[out_off + (py * half + px) * C + c] = v;
      const int gx = (x0 >> (2 + l)) + px, gy = (y0 >> (2 + l)) + py;
      if (gx < Wl && gy < Hl) {
        const long long at = ((long long)gy * Wl + gx) * C + c, per = (long long)Hl * Wl * C;
        a.ab[l][item_ab * per + at] = v;
        if (a.ba[l]) a.ba[l][item_ba * per + at] = v;
        if (twice) a.ab[l][(item_ab + B) * per + at] = v;
      }
    }
    in_off = out_off;
    side = half;
  }
}

// Image gradient of image_warp (the scatter-add TF derives from tf.gather), LDS-privatised.
// One workgroup owns a tile of pixels (32 x 16); the cells its 2x2 taps touch form a bounding
// box that -- for a coherent flow -- is little larger than the tile.  The taps are added
// into an LDS window of that box and the window is flushed with global atomics over
// CONTIGUOUS rows of ww*C floats -- instead of 4*C strided global atomics per pixel.  Tiles
// whose box does not fit fall back to those.
// LDS float atomics retire a lane per ~3 cycles on gfx950 (ds_add_f32 and a compare-and-swap
// loop alike: 12 adds per pixel made this kernel 1.0 ms at 32 x 384 x 1280 x 3), integer LDS
// atomics 64 lanes per 5 cycles.  So the window accumulates in FIXED POINT with a scale chosen per
// tile from the largest |grad| of its pixels: a value v becomes two 32-bit integers,
//   hi = rint(v * S1),  lo = rint((v - hi / S1) * S1 * 2^19),   S1 = 2^18 / gmax  (a power of two),
// added to two integer windows.  |hi| <= 2^18 and |lo| <= 2^18 per add and a window cell receives at
// most 4 * 1024 adds, so neither sum leaves 32 bits; a contribution is represented to 2^-38 of the
// tile's largest gradient (f32 itself resolves 2^-24 of a value) and integer addition makes the
// window sum independent of the order of the adds.
// TX x TY: the workgroup's pixel tile, one lane per pixel; WIN: ints of LDS for its two windows.  The box of a tile is
// (TX + d) x (TY + d) pixels with d the spread of the flow inside the tile, and d grows with the tile's diameter: at
// the flow gradients of the KITTI-shaped workload a 64 x 8 tile spreads to ~2.5 thousand cells (5 per pixel), a
// 32 x 16 tile to ~1.2 thousand (2.4 per pixel) -- fewer cells to zero and to flush, and a window that fits in 40 KB
// (three channels) or 30 KB (one, two) instead of 60, so that four workgroups of eight waves fill every wave slot of
// a CU (the kernel needs 30 VGPRs: the window is what set its occupancy).  Round 4, tools/iwim_variant.py at
// 32 x 384 x 1280: three channels 64 x 8 / 60 KB 0.454 ms, 32 x 16 / 40 KB 0.335 ms (32 x 32 / 60 KB and 32 x 8 / 30 KB
// the same; 32 x 8 / 20 KB 0.61: boxes no longer fit); two channels 64 x 16 / 60 KB 0.263, 32 x 16 / 30 KB 0.236 ms.
// A window too small is expensive (the strided float atomics of the fallback: 64 x 8 / 40 KB took 1.49 ms).
// Timing-only builds of the 32 x 16 form: without the LDS atomics 0.205 ms, without the flush's global atomics 0.309.
// The tap coordinates come from clamp_tap_coords: un-flattening clamp_taps' 64-bit indices cost four 64-bit
// divisions per pixel, 40 % of the kernel's vector instructions (0.484 -> 0.454 ms).
template <int C, int TX, int TY, int WIN>
__global__ __launch_bounds__(TX * TY) void image_warp_bwd_im_tiled_kernel(const float *__restrict__ grad,
                                                                                const float2 *__restrict__ flow,
                                                                                float *__restrict__ d_im, int H,
                                                                                int W) {
  __shared__ int win[WIN];                     // [hi window | lo window], WIN / 2 cells each
  __shared__ int box[4];
  __shared__ unsigned gmax_bits;
  constexpr int kCells = WIN / 2;
  constexpr int NT = TX * TY;
  static_assert(NT % 64 == 0 && NT <= 1024, "whole waves; fixed-point headroom assumes <= 1024 source pixels per window");
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
  const long long item = (long long)blockIdx.z * H * W;
  const int sx = x0 + (threadIdx.x % TX), sy = y0 + threadIdx.x / TX;
  const bool live = sx < W && sy < H;
  int tx0 = 0, tx1 = 0, ty0 = 0, ty1 = 0;
  float wa = 0.f, wb = 0.f, wc = 0.f, wd = 0.f, g[C];
  float gabs = 0.f;
  if (live) {
    const long long p = item + (long long)sy * W + sx;
    const ClampCoords t = clamp_tap_coords(sx, sy, flow[p], H, W);
    tx0 = t.x0; ty0 = t.y0;
    tx1 = t.x1; ty1 = t.y1;
    wa = (1.f - t.xw) * (1.f - t.yw); wb = (1.f - t.xw) * t.yw;
    wc = t.xw * (1.f - t.yw); wd = t.xw * t.yw;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      g[c] = grad[p * C + c];
      gabs = fmaxf(gabs, fabsf(g[c]));
    }
  }
  if (threadIdx.x < 4) box[threadIdx.x] = 0x7fffffff;
  if (threadIdx.x == 0) gmax_bits = 0u;
  __syncthreads();
  {
    int b0 = live ? tx0 : 0x7fffffff, b1 = live ? ty0 : 0x7fffffff;
    int b2 = live ? -tx1 : 0x7fffffff, b3 = live ? -ty1 : 0x7fffffff;
    // NaN / inf gradients: the tile takes the float path below (fmaxf drops a NaN, so test it apart)
    bool bad = false;
#pragma unroll
    for (int c = 0; c < C; ++c) bad = bad || (live && !(fabsf(g[c]) <= 3.0e38f));
    unsigned gb = bad ? 0x7f800000u : __float_as_uint(gabs);      // non-negative floats order like their bits
    // wave reductions on the DPP network (row_shr 1, 2, 4, 8: an inclusive scan of each row of 16 lanes; row_bcast 15 and 31
    // carry the rows' results on): six vector instructions per value and no LDS traffic, where the xor butterfly is six
    // ds_bpermute round trips per value (0.330 -> 0.309 ms for three channels, 0.225 -> 0.204 ms for two).  The result is in lane 63.
    auto red = [](int v, auto op) {
      v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
      v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
      v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
      v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
      v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));     // row_bcast:15 into rows 1 and 3
      v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));     // row_bcast:31 into rows 2 and 3
      return v;
    };
    auto imin = [](int a, int b) { return min(a, b); };
    auto umax = [](int a, int b) { return (int)max((unsigned)a, (unsigned)b); };
    b0 = red(b0, imin); b1 = red(b1, imin); b2 = red(b2, imin); b3 = red(b3, imin);
    gb = (unsigned)red((int)gb, umax);
    if ((threadIdx.x & 63) == 63 && b0 != 0x7fffffff) {
      atomicMin(&box[0], b0);
      atomicMin(&box[1], b1);
      atomicMin(&box[2], b2);
      atomicMin(&box[3], b3);
      atomicMax(&gmax_bits, gb);
    }
  }
  __syncthreads();
  const int wx0 = box[0], wy0 = box[1];
  const int ww = -box[2] - wx0 + 1, wh = -box[3] - wy0 + 1;
  float *base = d_im + item * C;
  if (wx0 == 0x7fffffff) return;                       // no live pixel (cannot happen for a launched tile)
  const float gmax = __uint_as_float(gmax_bits);
  if (gmax == 0.f) return;                             // an all-zero gradient tile adds nothing
  // S1 = 2^(18 - e) with 2^(e-1) <= gmax < 2^e: |v * S1| <= 2^18.  Extreme exponents (S1 or its inverse
  // not a normal float) and non-finite gradients keep the float atomics.
  const int e = (int)((gmax_bits >> 23) & 0xff) - 126;
  const bool fixed_ok = e > -100 && e < 100 && gmax_bits < 0x7f800000u;
  if ((long long)ww * wh * C > kCells || !fixed_ok) {  // incoherent flow: the strided global atomics
    if (live) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        unsafeAtomicAdd(base + ((long long)ty0 * W + tx0) * C + c, wa * g[c]);
        unsafeAtomicAdd(base + ((long long)ty1 * W + tx0) * C + c, wb * g[c]);
        unsafeAtomicAdd(base + ((long long)ty0 * W + tx1) * C + c, wc * g[c]);
        unsafeAtomicAdd(base + ((long long)ty1 * W + tx1) * C + c, wd * g[c]);
      }
    }
    return;
  }
  const float s1 = __uint_as_float((unsigned)(127 + 18 - e) << 23), inv_s1 = __uint_as_float((unsigned)(127 - 18 + e) << 23);
  const float s2 = 524288.f;                           // 2^19: lo = residual * S1 * 2^19
  const int cells = ww * wh * C;
  for (int i = threadIdx.x; i < cells; i += NT) {
    win[i] = 0;
    win[kCells + i] = 0;
  }
  __syncthreads();
  {
    // In a coherent flow the right-hand taps of a pixel are the left-hand taps of its right neighbour.  A lane whose
    // left neighbour (DPP row_shr:1, rows of 16 lanes lie inside one tile row) has its taps c, d on this lane's cells
    // a, b adds the neighbour's two integer pairs to its own and the neighbour skips those atomics: 2 cells per pixel
    // and channel instead of 4 wherever floor(flow) does not change between the two.  Integer sums: the window is
    // bit-equal to the unmerged one's.  0.344 -> 0.331 ms (three channels), 0.237 -> 0.224 ms (two).
    static_assert(TX % 16 == 0, "a DPP row must lie inside one tile row");
    const int ka = live ? ((ty0 - wy0) * ww + (tx0 - wx0)) * C : -1, kb = live ? ((ty1 - wy0) * ww + (tx0 - wx0)) * C : -1;
    const int kc = live ? ((ty0 - wy0) * ww + (tx1 - wx0)) * C : -3, kd = live ? ((ty1 - wy0) * ww + (tx1 - wx0)) * C : -3;
    const int lkc = __builtin_amdgcn_update_dpp(-2, kc, 0x111, 0xf, 0xf, false);     // row_shr:1
    const int lkd = __builtin_amdgcn_update_dpp(-2, kd, 0x111, 0xf, 0xf, false);
    const int take = (lkc == ka && lkd == kb) ? 1 : 0;
    const int given = __builtin_amdgcn_update_dpp(0, take, 0x101, 0xf, 0xf, false);  // row_shl:1: my right neighbour takes c, d
    auto split = [&](float v, int &hi_i, int &lo_i) {
      const float hi = rintf(v * s1);                  // exact product (power of two), |hi| <= 2^18
      const float res = fmaf(-hi, inv_s1, v);          // exact residual, |res| <= 0.5 / S1
      hi_i = (int)hi;
      lo_i = (int)rintf(res * s1 * s2);
    };
#pragma unroll
    for (int c = 0; c < C; ++c) {
      int ha, la, hb, lb, hc, lc, hd, ld;
      split(wa * g[c], ha, la);
      split(wb * g[c], hb, lb);
      split(wc * g[c], hc, lc);
      split(wd * g[c], hd, ld);
      const int nhc = __builtin_amdgcn_update_dpp(0, hc, 0x111, 0xf, 0xf, false), nlc = __builtin_amdgcn_update_dpp(0, lc, 0x111, 0xf, 0xf, false);
      const int nhd = __builtin_amdgcn_update_dpp(0, hd, 0x111, 0xf, 0xf, false), nld = __builtin_amdgcn_update_dpp(0, ld, 0x111, 0xf, 0xf, false);
      if (take) { ha += nhc; la += nlc; hb += nhd; lb += nld; }
#ifdef UNFLOW_TUNE_IM_NOSCATTER        // timing only: what the LDS atomics cost
      if (live && wa == 123.f) {
#else
      if (live) {
#endif
        atomicAdd(win + ka + c, ha);
        atomicAdd(win + kCells + ka + c, la);
        atomicAdd(win + kb + c, hb);
        atomicAdd(win + kCells + kb + c, lb);
        if (!given) {
          atomicAdd(win + kc + c, hc);
          atomicAdd(win + kCells + kc + c, lc);
          atomicAdd(win + kd + c, hd);
          atomicAdd(win + kCells + kd + c, ld);
        }
      }
    }
  }
  __syncthreads();
  const int rowf = ww * C;                             // one window row = rowf contiguous floats in d_im
  const float inv_s12 = inv_s1 * (1.f / 524288.f);
  for (int ly = threadIdx.x >> 6; ly < wh; ly += NT / 64) {
    float *drow = base + ((long long)(wy0 + ly) * W + wx0) * C;
    for (int i = threadIdx.x & 63; i < rowf; i += 64) {
      const int hi = win[ly * rowf + i], lo = win[kCells + ly * rowf + i];
#ifdef UNFLOW_TUNE_IM_NOFLUSH          // timing only: what the global atomics of the flush cost
      if ((hi | lo) == 0x7fffffff) unsafeAtomicAdd(drow + i, fmaf((float)lo, inv_s12, (float)hi * inv_s1));
#else
      if ((hi | lo) != 0) unsafeAtomicAdd(drow + i, fmaf((float)lo, inv_s12, (float)hi * inv_s1));
#endif
    }
  }
}

inline unsigned grid_for(long long total) {
  // memory-bound: cap the grid and grid-stride the rest (256 CUs x 8 blocks)
  long long g = (total + kBlock - 1) / kBlock;
  return (unsigned)(g < 8192 ? g : 8192);
}

int check_dims(const char *op, int B, int H, int W, int C) {
  if (B < 0 || H < 0 || W < 0 || C < 0) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "%s: negative dimension", op);
  return UNFLOW_OK;
}

}  // namespace
const char *tune_flags_warp() {
  return ""
#ifdef UNFLOW_TUNE_PPL
         " UNFLOW_TUNE_PPL"
#endif
#ifdef UNFLOW_TUNE_NOFAST
         " UNFLOW_TUNE_NOFAST"
#endif
#ifdef UNFLOW_TUNE_FW_STAMPS
         " UNFLOW_TUNE_FW_STAMPS"
#endif
#ifdef UNFLOW_TUNE_IM3
         " UNFLOW_TUNE_IM3"
#endif
#ifdef UNFLOW_TUNE_IM2
         " UNFLOW_TUNE_IM2"
#endif
#ifdef UNFLOW_TUNE_IM_NOFLUSH
         " UNFLOW_TUNE_IM_NOFLUSH"
#endif
#ifdef UNFLOW_TUNE_IM_NOSCATTER
         " UNFLOW_TUNE_IM_NOSCATTER"
#endif
      ;
}

}  // namespace unflow

using namespace unflow;

// the 4-pixel-per-lane kernels: one 256-thread block per 64 x 16 tile
#define FAST_LAUNCH(KERNEL, CLAMP, ...)                                                                    \
  do {                                                                                                     \
    const int ntx_ = (W + kTileW - 1) / kTileW, nty_ = (H + kFastTileH - 1) / kFastTileH;                  \
    const dim3 g_((unsigned)((long long)B * ntx_ * nty_));                                                  \
    if (C == 1) hipLaunchKernelGGL((KERNEL<1, CLAMP>), g_, dim3(kBlock), 0, s, __VA_ARGS__, H, W, ntx_, nty_);      \
    else if (C == 2) hipLaunchKernelGGL((KERNEL<2, CLAMP>), g_, dim3(kBlock), 0, s, __VA_ARGS__, H, W, ntx_, nty_); \
    else hipLaunchKernelGGL((KERNEL<3, CLAMP>), g_, dim3(kBlock), 0, s, __VA_ARGS__, H, W, ntx_, nty_);             \
  } while (0)

#define FAST_LAUNCH_F16(KERNEL, CLAMP, ...)                                                                          \
  do {                                                                                                               \
    const int ntx_ = (W + kTileW - 1) / kTileW, nty_ = (H + kFastTileH - 1) / kFastTileH;                            \
    const dim3 g_((unsigned)((long long)B * ntx_ * nty_));                                                            \
    if (C == 1) hipLaunchKernelGGL((KERNEL<1, CLAMP, _Float16>), g_, dim3(kBlock), 0, s, __VA_ARGS__, H, W, ntx_, nty_);      \
    else if (C == 2) hipLaunchKernelGGL((KERNEL<2, CLAMP, _Float16>), g_, dim3(kBlock), 0, s, __VA_ARGS__, H, W, ntx_, nty_); \
    else hipLaunchKernelGGL((KERNEL<3, CLAMP, _Float16>), g_, dim3(kBlock), 0, s, __VA_ARGS__, H, W, ntx_, nty_);             \
  } while (0)

#define DISPATCH_C(KERNEL, C, ...)                                                               \
  do {                                                                                           \
    switch (C) {                                                                                 \
      case 1: hipLaunchKernelGGL(KERNEL<1>, dim3(grid), dim3(kBlock), 0, s, __VA_ARGS__); break; \
      case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(grid), dim3(kBlock), 0, s, __VA_ARGS__); break; \
      case 3: hipLaunchKernelGGL(KERNEL<3>, dim3(grid), dim3(kBlock), 0, s, __VA_ARGS__); break; \
      default: hipLaunchKernelGGL(KERNEL<0>, dim3(grid), dim3(kBlock), 0, s, __VA_ARGS__);       \
    }                                                                                            \
  } while (0)

extern "C" int unflow_backward_warp_fwd(const float *images, const float *flows, float *output, int B, int H,
                                        int W, int C, unflow_stream_t stream) {
  if (int rc = check_dims("backward_warp", B, H, W, C)) return rc;
  const long long total = (long long)B * H * W;
  const long long vtotal = tiled_total(B, H, W);   // the kernels walk 64x4 pixel tiles
  if (total == 0 || C == 0) return UNFLOW_OK;
  hipStream_t s = as_stream(stream);
  if (fast_ok(B, H, W, C)) {
    FAST_LAUNCH(warp_fwd_fast_kernel, false, images, reinterpret_cast<const float2 *>(flows), output);
    return launch_status("backward_warp_fwd");
  }
  const unsigned grid = grid_for(vtotal);
  DISPATCH_C(backward_warp_fwd_kernel, C, images, reinterpret_cast<const float2 *>(flows), output, H, W, C, vtotal);
  return launch_status("backward_warp_fwd");
}

extern "C" int unflow_backward_warp_bwd(const float *input_grad, const float *images, const float *flows,
                                        float *output_grad, int B, int H, int W, int C,
                                        unflow_stream_t stream) {
  if (int rc = check_dims("backward_warp_grad", B, H, W, C)) return rc;
  const long long total = (long long)B * H * W;
  const long long vtotal = tiled_total(B, H, W);   // the kernels walk 64x4 pixel tiles
  if (total == 0) return UNFLOW_OK;
  hipStream_t s = as_stream(stream);
  if (fast_ok(B, H, W, C)) {
    FAST_LAUNCH(warp_bwd_flow_fast_kernel, false, input_grad, images, reinterpret_cast<const float2 *>(flows),
                reinterpret_cast<float2 *>(output_grad), (float *)nullptr);
    return launch_status("backward_warp_bwd");
  }
  const unsigned grid = grid_for(vtotal);
  DISPATCH_C(backward_warp_bwd_kernel, C, input_grad, images, reinterpret_cast<const float2 *>(flows),
             reinterpret_cast<float2 *>(output_grad), H, W, C, vtotal);
  return launch_status("backward_warp_bwd");
}

extern "C" int unflow_image_warp_fwd(const float *im, const float *flow, float *warped, int B, int H, int W,
                                     int C, unflow_stream_t stream) {
  if (int rc = check_dims("image_warp", B, H, W, C)) return rc;
  const long long total = (long long)B * H * W;
  const long long vtotal = tiled_total(B, H, W);   // the kernels walk 64x4 pixel tiles
  if (total == 0 || C == 0) return UNFLOW_OK;
  hipStream_t s = as_stream(stream);
  if (fast_ok(B, H, W, C)) {
    FAST_LAUNCH(warp_fwd_fast_kernel, true, im, reinterpret_cast<const float2 *>(flow), warped);
    return launch_status("image_warp_fwd");
  }
  const unsigned grid = grid_for(vtotal);
  DISPATCH_C(image_warp_fwd_kernel, C, im, reinterpret_cast<const float2 *>(flow), warped, H, W, C, vtotal);
  return launch_status("image_warp_fwd");
}

extern "C" int unflow_image_warp_bwd(const float *grad, const float *im, const float *flow, float *d_im,
                                     float *d_flow, int B, int H, int W, int C, unflow_stream_t stream) {
  if (int rc = check_dims("image_warp_grad", B, H, W, C)) return rc;
  const long long total = (long long)B * H * W;
  const long long vtotal = tiled_total(B, H, W);   // the kernels walk 64x4 pixel tiles
  if (total == 0 || (!d_im && !d_flow)) return UNFLOW_OK;
  hipStream_t s = as_stream(stream);
  // The image gradient is a scatter-add into a zeroed buffer.  When the flow gradient is wanted too and goes through
  // the tiled kernel, that kernel runs FIRST and zeroes d_im on the way (it visits every pixel once): one launch
  // less than a memset in front of the pair.
  const bool tiled_im = d_im && C >= 1 && C <= 3 && B <= 65535;
  const bool zero_in_flow_kernel = tiled_im && d_flow && fast_ok(B, H, W, C);
  if (d_im && C > 0 && !zero_in_flow_kernel) {
    if (hipMemsetAsync(d_im, 0, sizeof(float) * (size_t)total * C, s) != hipSuccess)
      return launch_status("image_warp_bwd memset");
  }
  const unsigned grid = grid_for(vtotal);
#define IW_BWD(C_, IM_, FL_)                                                                              \
  hipLaunchKernelGGL((image_warp_bwd_kernel<C_, IM_, FL_>), dim3(grid), dim3(kBlock), 0, s, grad, im,     \
                     reinterpret_cast<const float2 *>(flow), d_im, reinterpret_cast<float2 *>(d_flow), H, W, C, vtotal)
#define IW_BWD_C(IM_, FL_)                   \
  switch (C) {                               \
    case 1: IW_BWD(1, IM_, FL_); break;      \
    case 2: IW_BWD(2, IM_, FL_); break;      \
    case 3: IW_BWD(3, IM_, FL_); break;      \
    default: IW_BWD(0, IM_, FL_);            \
  }
  if (tiled_im) {
    // image gradient through the LDS-privatised scatter; the flow gradient (if wanted) separately
    const dim3 tg((W + kIm2Cols - 1) / kIm2Cols, (H + kIm2Rows - 1) / kIm2Rows, B), tg3((W + kIm3Cols - 1) / kIm3Cols, (H + kIm3Rows - 1) / kIm3Rows, B);
    const float2 *fl = reinterpret_cast<const float2 *>(flow);
    if (zero_in_flow_kernel) FAST_LAUNCH(warp_bwd_flow_fast_kernel, true, grad, im, fl, reinterpret_cast<float2 *>(d_flow), d_im);
    if (C == 1) hipLaunchKernelGGL((image_warp_bwd_im_tiled_kernel<1, kIm2Cols, kIm2Rows, kIm2Win>), tg, dim3(kIm2Cols * kIm2Rows), 0, s, grad, fl, d_im, H, W);
    else if (C == 2) hipLaunchKernelGGL((image_warp_bwd_im_tiled_kernel<2, kIm2Cols, kIm2Rows, kIm2Win>), tg, dim3(kIm2Cols * kIm2Rows), 0, s, grad, fl, d_im, H, W);
    else hipLaunchKernelGGL((image_warp_bwd_im_tiled_kernel<3, kIm3Cols, kIm3Rows, kIm3Win>), tg3, dim3(kIm3Cols * kIm3Rows), 0, s, grad, fl, d_im, H, W);
    if (d_flow && !zero_in_flow_kernel) { IW_BWD_C(false, true) }
  } else if (d_im && d_flow) { IW_BWD_C(true, true) }
  else if (d_im) { IW_BWD_C(true, false) }
  else if (fast_ok(B, H, W, C)) {
    FAST_LAUNCH(warp_bwd_flow_fast_kernel, true, grad, im, reinterpret_cast<const float2 *>(flow), reinterpret_cast<float2 *>(d_flow), (float *)nullptr);
  } else { IW_BWD_C(false, true) }
#undef IW_BWD_C
#undef IW_BWD
  return launch_status("image_warp_bwd");
}

extern "C" int unflow_forward_warp_fwd(const float *flows, float *output, int B, int H, int W,
                                       unflow_stream_t stream) {
  if (int rc = check_dims("forward_warp", B, H, W, 2)) return rc;
  const long long total = (long long)B * H * W;
  if (total == 0) return UNFLOW_OK;
  hipStream_t s = as_stream(stream);
  if (hipMemsetAsync(output, 0, sizeof(float) * (size_t)total, s) != hipSuccess)
    return launch_status("forward_warp_fwd memset");
  hipLaunchKernelGGL(forward_warp_fwd_kernel<float2>, dim3((W + kSX - 1) / kSX, (H + kSY - 1) / kSY, B), dim3(kSplatThreads), 0, s,
                     reinterpret_cast<const float2 *>(flows), output, H, W);
  return launch_status("forward_warp_fwd");
}

extern "C" int unflow_forward_warp_bwd(const float *input_grad, const float *flows, float *output_grad, int B,
                                       int H, int W, unflow_stream_t stream) {
  if (int rc = check_dims("forward_warp_grad", B, H, W, 2)) return rc;
  const long long total = (long long)B * H * W;
  const long long vtotal = tiled_total(B, H, W);   // the kernels walk 64x4 pixel tiles
  if (total == 0) return UNFLOW_OK;
  hipStream_t s = as_stream(stream);
  if (B <= 65535) {
    hipLaunchKernelGGL(forward_warp_bwd_tiled_kernel, dim3((W + kSX - 1) / kSX, (H + kSY - 1) / kSY, B),
                       dim3(kSplatThreads), 0, s, input_grad, reinterpret_cast<const float2 *>(flows),
                       reinterpret_cast<float2 *>(output_grad), H, W);
    return launch_status("forward_warp_bwd");
  }
  hipLaunchKernelGGL(forward_warp_bwd_kernel, dim3(grid_for(vtotal)), dim3(kBlock), 0, s, input_grad,
                     reinterpret_cast<const float2 *>(flows), reinterpret_cast<float2 *>(output_grad), H, W,
                     vtotal);
  return launch_status("forward_warp_bwd");
}

// ---- fp16 storage (extension, BASELINE config 5): half images / flows / gradients, f32 arithmetic.
// C = 1, 2 or 3; 2-byte aligned tensors (4-byte for the flows).
static int check_f16(const char *what, int B, int H, int W, int C) {
  if (int rc = check_dims(what, B, H, W, C)) return rc;
  if ((long long)B * H * W != 0 && !fast_ok(B, H, W, C))
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "%s (fp16 storage): 1 <= channels <= 3 and H*W*3 < 2^30 required", what);
  // C == 3: a pixel is read as a 4-byte aligned 8-byte access that reaches one half beyond it; at the tensor's
  // last pixel it reaches BACKWARDS instead -- which for a one-pixel tensor would be in front of the allocation
  if (C == 3 && (long long)B * H * W == 1)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "%s (fp16 storage): a single-pixel 3-channel tensor is not supported", what);
  return UNFLOW_OK;
}
static bool aligned4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }
#define UNFLOW_F16_ALIGNED(WHAT, ...)                                                               \
  do {                                                                                              \
    const void *ptrs_[] = {__VA_ARGS__};                                                            \
    for (const void *q_ : ptrs_)                                                                    \
      if (!aligned4(q_)) return fail(UNFLOW_ERR_INVALID_ARGUMENT, WHAT " (fp16 storage): tensors must be 4-byte aligned"); \
  } while (0)

extern "C" int unflow_backward_warp_fwd_f16(const void *images, const void *flows, void *output, int B, int H, int W,
                                            int C, unflow_stream_t stream) {
  if (int rc = check_f16("backward_warp", B, H, W, C)) return rc;
  if ((long long)B * H * W == 0) return UNFLOW_OK;
  UNFLOW_F16_ALIGNED("backward_warp", images, flows, output);
  hipStream_t s = as_stream(stream);
  FAST_LAUNCH_F16(warp_fwd_fast_kernel, false, static_cast<const _Float16 *>(images), static_cast<const f16x2_t *>(flows),
                  static_cast<_Float16 *>(output));
  return launch_status("backward_warp_fwd_f16");
}

extern "C" int unflow_backward_warp_bwd_f16(const void *input_grad, const void *images, const void *flows,
                                            void *output_grad, int B, int H, int W, int C, unflow_stream_t stream) {
  if (int rc = check_f16("backward_warp_grad", B, H, W, C)) return rc;
  if ((long long)B * H * W == 0) return UNFLOW_OK;
  UNFLOW_F16_ALIGNED("backward_warp_grad", input_grad, images, flows, output_grad);
  hipStream_t s = as_stream(stream);
  FAST_LAUNCH_F16(warp_bwd_flow_fast_kernel, false, static_cast<const _Float16 *>(input_grad),
                  static_cast<const _Float16 *>(images), static_cast<const f16x2_t *>(flows),
                  static_cast<f16x2_t *>(output_grad), (float *)nullptr);
  return launch_status("backward_warp_bwd_f16");
}

extern "C" int unflow_image_warp_fwd_f16(const void *im, const void *flow, void *warped, int B, int H, int W, int C,
                                         unflow_stream_t stream) {
  if (int rc = check_f16("image_warp", B, H, W, C)) return rc;
  if ((long long)B * H * W == 0) return UNFLOW_OK;
  UNFLOW_F16_ALIGNED("image_warp", im, flow, warped);
  hipStream_t s = as_stream(stream);
  FAST_LAUNCH_F16(warp_fwd_fast_kernel, true, static_cast<const _Float16 *>(im), static_cast<const f16x2_t *>(flow),
                  static_cast<_Float16 *>(warped));
  return launch_status("image_warp_fwd_f16");
}

extern "C" int unflow_image_warp_bwd_flow_f16(const void *grad, const void *im, const void *flow, void *d_flow, int B,
                                              int H, int W, int C, unflow_stream_t stream) {
  if (int rc = check_f16("image_warp_grad", B, H, W, C)) return rc;
  if ((long long)B * H * W == 0) return UNFLOW_OK;
  UNFLOW_F16_ALIGNED("image_warp_grad", grad, im, flow, d_flow);
  hipStream_t s = as_stream(stream);
  FAST_LAUNCH_F16(warp_bwd_flow_fast_kernel, true, static_cast<const _Float16 *>(grad), static_cast<const _Float16 *>(im),
                  static_cast<const f16x2_t *>(flow), static_cast<f16x2_t *>(d_flow), (float *)nullptr);
  return launch_status("image_warp_bwd_flow_f16");
}

extern "C" int unflow_forward_warp_fwd_f16(const void *flows, float *output, int B, int H, int W,
                                           unflow_stream_t stream) {
  if (int rc = check_dims("forward_warp", B, H, W, 2)) return rc;
  const long long total = (long long)B * H * W;
  if (total == 0) return UNFLOW_OK;
  if (!aligned4(flows)) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "forward_warp (fp16 storage): flows must be 4-byte aligned");
  hipStream_t s = as_stream(stream);
  if (hipMemsetAsync(output, 0, sizeof(float) * (size_t)total, s) != hipSuccess)
    return launch_status("forward_warp_fwd_f16 memset");
  hipLaunchKernelGGL(forward_warp_fwd_kernel<f16x2_t>, dim3((W + kSX - 1) / kSX, (H + kSY - 1) / kSY, B),
                     dim3(kSplatThreads), 0, s, static_cast<const f16x2_t *>(flows), output, H, W);
  return launch_status("forward_warp_fwd_f16");
}

#ifdef UNFLOW_TUNE_FW_STAMPS
extern "C" int unflow_debug_read_fw_stamps(void *dst, size_t bytes) {
  const size_t n = sizeof(unflow::g_fw_stamps) < bytes ? sizeof(unflow::g_fw_stamps) : bytes;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(unflow::g_fw_stamps), n, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int unflow_downsample_fwd(const float *images, float *output, int B, int H, int W, int C, int scale,
                                     unflow_stream_t stream) {
  if (int rc = check_dims("downsample", B, H, W, C)) return rc;
  if (scale <= 0 || H % scale != 0 || W % scale != 0)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "Input height and width must be divisible by scale");
  const int oh = H / scale, ow = W / scale;
  const long long total = (long long)B * oh * ow * C;
  if (total == 0) return UNFLOW_OK;
  const long long npx = (long long)B * oh * ow;
  const bool vec_ok = (reinterpret_cast<uintptr_t>(images) & 15) == 0 && npx < (1LL << 31) * kBlock;
  if (vec_ok && C >= 1 && C <= 3 && (scale == 2 || scale == 4)) {
    const dim3 g((unsigned)((npx + kBlock - 1) / kBlock));
    hipStream_t s = as_stream(stream);
#define DS_LAUNCH(C_, S_) hipLaunchKernelGGL((downsample_px_kernel<C_, S_>), g, dim3(kBlock), 0, s, images, output, W, oh, ow, npx)
    if (scale == 2) { if (C == 1) DS_LAUNCH(1, 2); else if (C == 2) DS_LAUNCH(2, 2); else DS_LAUNCH(3, 2); }
    else { if (C == 1) DS_LAUNCH(1, 4); else if (C == 2) DS_LAUNCH(2, 4); else DS_LAUNCH(3, 4); }
#undef DS_LAUNCH
    return launch_status("downsample");
  }
  hipLaunchKernelGGL(downsample_kernel, dim3(grid_for(total)), dim3(kBlock), 0, as_stream(stream), images,
                     output, H, W, C, oh, ow, scale, total);
  return launch_status("downsample");
}

extern "C" int unflow_downsample_pyramid(const float *im_a, const float *im_b, float *const *out_ab,
                                         float *const *out_ba, int levels, int B, int H, int W, int C,
                                         unflow_stream_t stream) {
  if (int rc = check_dims("downsample_pyramid", B, H, W, C)) return rc;
  if (levels < 1 || levels > kPyrLevels) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "downsample_pyramid: 1 to 5 levels");
  if (C != 1 && C != 3) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "downsample_pyramid: 1 or 3 channels");
  const int step = 4 << (levels - 1);
  if (H % step != 0 || W % step != 0)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "Input height and width must be divisible by scale");   // the chain's own error
  if (!im_a || !out_ab) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "downsample_pyramid: null tensor");
  if ((reinterpret_cast<uintptr_t>(im_a) & 15) || (im_b && (reinterpret_cast<uintptr_t>(im_b) & 15)))
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "downsample_pyramid: images must be 16-byte aligned");
  if (out_ba && !im_b) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "downsample_pyramid: the second order needs two tensors");
  const bool dup = im_b == im_a && !out_ba;                  // [a; a]: one pass over the tensor, both halves written
  const long long nimg = (long long)B * ((im_b && !dup) ? 2 : 1);
  if (nimg == 0 || H == 0 || W == 0) return UNFLOW_OK;
  if (nimg > 65535) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "downsample_pyramid: at most 65535 images per launch");
  PyrArgs a;
  a.src[0] = im_a;
  a.src[1] = im_b ? im_b : im_a;
  a.dup = dup ? 1 : 0;
  for (int l = 0; l < kPyrLevels; ++l) {
    a.ab[l] = l < levels ? out_ab[l] : nullptr;
    a.ba[l] = (out_ba && l < levels) ? out_ba[l] : nullptr;
    if (l < levels && !a.ab[l]) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "downsample_pyramid: null output level");
  }
  const dim3 g((W + kPyrBlock - 1) / kPyrBlock, (H + kPyrBlock - 1) / kPyrBlock, (unsigned)nimg);
  if (C == 1) hipLaunchKernelGGL(downsample_pyramid_kernel<1>, g, dim3(256), 0, as_stream(stream), a, levels, B, H, W);
  else hipLaunchKernelGGL(downsample_pyramid_kernel<3>, g, dim3(256), 0, as_stream(stream), a, levels, B, H, W);
  return launch_status("downsample_pyramid");
}
