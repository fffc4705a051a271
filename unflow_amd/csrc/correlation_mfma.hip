// FlowNetC-shaped correlation (kernel_size 1, stride_1 1, stride_2 2, pad ==
// max_displacement) on the f32 matrix cores of gfx950.
//
// Reference semantics: ops/correlation_op.cu.cc:51-117 (forward), :119-248
// (backward) of the reference checkout; see correlation.hip for the generic path.
//
// Formulation.  With stride_2 = 2 a feature position only ever meets positions
// of the same (row, column) parity, so the problem splits into 4 independent
// "classes" (cy, cx); inside a class the displacement window is DENSE:
//     out[d=(dy,dx), p] = 1/C * sum_c f0s[c, p] * f1s[c, p + (dy,dx)],  |dy|,|dx| <= R
// For a 4x4 tile of class positions P and the (4+2R)^2 window Q around it this
// is a genuine dense GEMM  G[P x Q] = f0s[P x C] * f1s[C x Q]  of which
// (2R+1)^2 / (4+2R)^2 = 441/576 = 77 % of the entries are needed (every entry
// with q - p inside the window).  G is computed with v_mfma_f32_16x16x4_f32
// (exact f32: a k-ordered fmaf chain), M = 16 positions, N = 16 window
// positions (a 4x4 sub-block of Q), K = 4 channels.
//
// Forward kernel: one workgroup = 8 waves = {cx 0,1} x 2x2 tiles of one row
// class cy: an 8 (class rows) x 16 (full-resolution columns) output region.
// Channels stream through LDS in chunks of 8 (register-prefetched, double
// buffered); each wave keeps its G tile in 36 x 4 accumulator registers.  The
// epilogue regroups G by displacement through LDS so that every global store is
// a 64-byte row segment of one output plane.
#include "common.h"
#include "correlation.h"

namespace unflow {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int S = 2;            // stride_2
constexpr int R = 10;           // neighbourhood grid radius (max_displacement / stride_2)
constexpr int GW = 2 * R + 1;   // 21
constexpr int TP = 4;           // tile of class positions is TP x TP (= MFMA M 16)
constexpr int WIN = TP + 2 * R; // 24: window of one tile
constexpr int TQ = WIN / 4;     // 6 N-tiles per dimension
constexpr int NT = TQ * TQ;     // 36 N-tiles = accumulators per wave
constexpr int RY = 8, RX = 8;   // region of one class in a workgroup (2x2 tiles)
constexpr int LW = RX + 2 * R;  // 28 staged window columns
constexpr int LH = RY + 2 * R;  // 28 staged window rows
constexpr int PLANE = LW * LH;  // 784 floats per channel
constexpr int KC = 8;           // channels per LDS stage
constexpr int NTHREADS = 512;
static_assert(WIN % 4 == 0, "window must tile by 4");
// ds_read_b32 of a B fragment touches, per 32-lane half, 2 channels x 4 rows x
// 4 columns: conflict free iff the row stride is 4 * odd (mod 32) and the
// channel-plane stride is 16 (mod 32).
static_assert(LW % 8 == 4 && PLANE % 32 == 16, "LDS image would bank-conflict");

constexpr int F1_STAGE = S * KC * PLANE;     // floats per buffer (both cx classes)
constexpr int F0_STAGE = S * KC * RY * RX;
constexpr int STAGE = F1_STAGE + F0_STAGE;
constexpr int F1_PAIRS = KC * LH * LW;       // float pairs (cx 0,1) per stage
constexpr int F0_PAIRS = KC * RY * RX;
constexpr int F1_PER_THREAD = (F1_PAIRS + NTHREADS - 1) / NTHREADS;  // 13
constexpr int F0_PER_THREAD = (F0_PAIRS + NTHREADS - 1) / NTHREADS;  // 1
constexpr int SLAB_DY = 7;                    // displacement rows per epilogue slab
constexpr int SLAB = SLAB_DY * GW * RY * (S * RX);  // floats
static_assert(SLAB <= 2 * STAGE, "epilogue slab must fit the staging buffers");
static_assert(GW % SLAB_DY == 0, "slabs must tile the displacement rows");

struct Prefetch {
  float2 f1[F1_PER_THREAD];
  float2 f0[F0_PER_THREAD];
};

__device__ inline float ld_guard(const float *__restrict__ plane, int Y, int X, int H, int W) {
  return (Y >= 0 && Y < H && X >= 0 && X < W) ? plane[(long long)Y * W + X] : 0.f;
}

// Issue the global loads of one channel chunk (both inputs) into registers.
__device__ inline void prefetch_chunk(Prefetch &pf, const float *__restrict__ in0, const float *__restrict__ in1,
                                      long long item, int c0, int C, int H, int W, int cy, int ys0, int xs0) {
  const long long plane = (long long)H * W;
#pragma unroll
  for (int it = 0; it < F1_PER_THREAD; ++it) {
    const int idx = it * NTHREADS + threadIdx.x;
    float2 v = make_float2(0.f, 0.f);
    if (idx < F1_PAIRS) {
      const int c = idx / PLANE, rem = idx - c * PLANE;
      const int row = rem / LW, xp = rem - row * LW;
      if (c0 + c < C) {
        const int Y = S * (ys0 - R + row) + cy, X = S * (xs0 - R + xp);
        const float *p = in1 + (item * C + c0 + c) * plane;
        v.x = ld_guard(p, Y, X, H, W);
        v.y = ld_guard(p, Y, X + 1, H, W);
      }
    }
    pf.f1[it] = v;
  }
#pragma unroll
  for (int it = 0; it < F0_PER_THREAD; ++it) {
    const int idx = it * NTHREADS + threadIdx.x;
    float2 v = make_float2(0.f, 0.f);
    if (idx < F0_PAIRS) {
      const int c = idx / (RY * RX), rem = idx - c * (RY * RX);
      const int row = rem / RX, xp = rem - row * RX;
      if (c0 + c < C) {
        const int Y = S * (ys0 + row) + cy, X = S * (xs0 + xp);
        const float *p = in0 + (item * C + c0 + c) * plane;
        v.x = ld_guard(p, Y, X, H, W);
        v.y = ld_guard(p, Y, X + 1, H, W);
      }
    }
    pf.f0[it] = v;
  }
}

// De-interleave the two column classes into their LDS planes.
__device__ inline void commit_chunk(const Prefetch &pf, float *stage) {
  float *f1s = stage, *f0s = stage + F1_STAGE;
#pragma unroll
  for (int it = 0; it < F1_PER_THREAD; ++it) {
    const int idx = it * NTHREADS + threadIdx.x;
    if (idx < F1_PAIRS) {
      f1s[idx] = pf.f1[it].x;                 // cx = 0 : [c][row][xp] == idx
      f1s[KC * PLANE + idx] = pf.f1[it].y;    // cx = 1
    }
  }
#pragma unroll
  for (int it = 0; it < F0_PER_THREAD; ++it) {
    const int idx = it * NTHREADS + threadIdx.x;
    if (idx < F0_PAIRS) {
      f0s[idx] = pf.f0[it].x;
      f0s[KC * RY * RX + idx] = pf.f0[it].y;
    }
  }
}

__global__ __launch_bounds__(NTHREADS, 2) void corr_mfma_fwd_kernel(const float *__restrict__ in0,
                                                                    const float *__restrict__ in1,
                                                                    float *__restrict__ out, int C, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cx = wave & 1, ty = (wave >> 1) & 1, tx = wave >> 2;
  const int cy = blockIdx.z & 1;
  const long long item = blockIdx.z >> 1;
  const int ys0 = blockIdx.y * RY;           // class-row origin of the region
  const int xs0 = blockIdx.x * RX;           // class-column origin (same for both cx)
  const int hs = (H - cy + S - 1) / S;       // class rows that exist for this cy
  if (ys0 >= hs) return;                     // uniform per block

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane fragment addresses inside one stage (floats)
  const int k = lane >> 4, n = lane & 15;
  const int a_off = F1_STAGE + (cx * KC + k) * (RY * RX) + (ty * TP + (n >> 2)) * RX + tx * TP + (n & 3);
  const int b_off = (cx * KC + k) * PLANE + (ty * TP + (n >> 2)) * LW + tx * TP + (n & 3);

  Prefetch pf;
  const int nchunks = (C + KC - 1) / KC;
  prefetch_chunk(pf, in0, in1, item, 0, C, H, W, cy, ys0, xs0);
  commit_chunk(pf, lds);
  __syncthreads();

  for (int ch = 0; ch < nchunks; ++ch) {
    const float *cur = lds + (ch & 1) * STAGE;
    float *nxt = lds + ((ch + 1) & 1) * STAGE;
    const bool more = ch + 1 < nchunks;
    if (more) prefetch_chunk(pf, in0, in1, item, (ch + 1) * KC, C, H, W, cy, ys0, xs0);
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      const float a = cur[a_off + kk * 4 * (RY * RX)];
      const float *bp = cur + b_off + kk * 4 * PLANE;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float b = bp[(t / TQ) * 4 * LW + (t % TQ) * 4];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
      }
    }
    if (more) commit_chunk(pf, nxt);
    __syncthreads();
  }

  // ---- epilogue: regroup G[p][q] by displacement d = q - p through LDS ----
  // lane holds, for tile t = (tqy,tqx) and register j:
  //   p = (py = lane>>4, px = j),  q = (4*tqy + (n>>2), 4*tqx + (n&3))
  const float scale = 1.f / (float)C;
  const bool pow2 = (C & (C - 1)) == 0;
  const int py = lane >> 4;
  const int row = ty * TP + py;                       // class row inside the region
  const int D = GW * GW;
  const long long oplane = (long long)H * W;
  float *ob = lds;
  constexpr int RW = S * RX;                          // 16 full-resolution columns
#pragma unroll
  for (int s = 0; s < GW / SLAB_DY; ++s) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int tqy = t / TQ, tqx = t % TQ;
      // dd = 4*tqy + (n>>2) - py lies in [4*tqy-3, 4*tqy+3]; skip tiles that cannot hit the slab
      if (4 * tqy + 3 < s * SLAB_DY || 4 * tqy - 3 >= (s + 1) * SLAB_DY) continue;
      const int dd = 4 * tqy + (n >> 2) - py;
      const int ddl = dd - s * SLAB_DY;
      if (ddl < 0 || ddl >= SLAB_DY || dd > 2 * R) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int dxx = 4 * tqx + (n & 3) - j;
        if (dxx < 0 || dxx > 2 * R) continue;
        const int col = S * (tx * TP + j) + cx;
        const float v = acc[t][j];
        ob[((ddl * GW + dxx) * RY + row) * RW + col] = pow2 ? v * scale : v / (float)C;
      }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < SLAB / 4; idx += NTHREADS) {
      const int c4 = idx % (RW / 4);
      int rest = idx / (RW / 4);
      const int r = rest % RY;
      rest /= RY;
      const int dxx = rest % GW, ddl = rest / GW;
      const int d = (s * SLAB_DY + ddl) * GW + dxx;
      const int y = S * (ys0 + r) + cy;
      const int x = S * xs0 + c4 * 4;
      if (y >= H || x >= W) continue;
      const float4 v = *reinterpret_cast<const float4 *>(ob + idx * 4);
      float *dst = out + (item * D + d) * oplane + (long long)y * W + x;
      if (x + 3 < W && (W & 3) == 0) {
        *reinterpret_cast<float4 *>(dst) = v;
      } else {
        dst[0] = v.x;
        if (x + 1 < W) dst[1] = v.y;
        if (x + 2 < W) dst[2] = v.z;
        if (x + 3 < W) dst[3] = v.w;
      }
    }
    __syncthreads();
  }
}

}  // namespace

bool corr_mfma_supported(const CorrGeometry &g, int C) {
  return C > 0 && g.kernel_size == 1 && g.stride_1 == 1 && g.stride_2 == S && g.max_displacement == S * R &&
         g.pad == g.max_displacement;
}

int corr_mfma_fwd(const float *in0, const float *in1, float *out, int B, int C, const CorrGeometry &g,
                  hipStream_t s) {
  const int H = g.H, W = g.W;
  const int hs = (H + S - 1) / S, ws = (W + S - 1) / S;
  const dim3 grid((ws + RX - 1) / RX, (hs + RY - 1) / RY, B * S);
  const size_t smem = sizeof(float) * 2 * STAGE;
  // > 64 KiB of dynamic LDS needs an opt-in; thread-safe one-time initialisation
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(corr_mfma_fwd_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (attr != hipSuccess) return fail(UNFLOW_ERR_LAUNCH, "correlation_fwd (mfma): cannot reserve %zu B of LDS", smem);
  hipLaunchKernelGGL(corr_mfma_fwd_kernel, grid, dim3(NTHREADS), smem, s, in0, in1, out, C, H, W);
  return launch_status("correlation_fwd (mfma)");
}

bool corr_mfma_bwd_supported(const CorrGeometry &, int) { return false; }

int corr_mfma_bwd(const float *, const float *, const float *, float *, float *, int, int, const CorrGeometry &,
                  hipStream_t) {
  return fail(UNFLOW_ERR_LAUNCH, "correlation_bwd (mfma) not built");
}

}  // namespace unflow
