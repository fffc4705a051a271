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

#include <cstdlib>

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
constexpr int SLAB_DY = 7;                    // displacement rows per epilogue slab
constexpr int SLAB = SLAB_DY * GW * RY * (S * RX);  // floats
static_assert(SLAB <= 2 * STAGE, "epilogue slab must fit the staging buffers");
static_assert(GW % SLAB_DY == 0, "slabs must tile the displacement rows");

// Staging is done by LDS-DMA (buffer_load_dword ... lds): one wave instruction
// fills 64 consecutive LDS floats from 64 per-lane buffer offsets, and the
// buffer's range check returns 0 for the zero padding and for channels >= C,
// so no data register, no ds_write and no branch is spent on staging.
constexpr int PIECE = 64;
static_assert(F1_STAGE % PIECE == 0 && F0_STAGE % PIECE == 0, "stage must split into wave pieces");
constexpr int F1_PIECES = F1_STAGE / PIECE;            // 196
constexpr int PIECES = STAGE / PIECE;                  // 212
constexpr int NWAVES = NTHREADS / 64;
constexpr int PIECES_PER_WAVE = (PIECES + NWAVES - 1) / NWAVES;  // 27
constexpr unsigned OOB = 0x80000000u;                  // beyond any num_records (< 2 GiB)

__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const float *base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), (short)0, (int)bytes, 0x00020000);
}

// Byte offset (relative to channel c0 of the item) of the element that LDS slot
// `slot` of a stage holds, or OOB for the zero padding.
__device__ inline unsigned slot_offset(int slot, int H, int W, int cy, int ys0, int xs0) {
  int cx, c, Y, X;
  if (slot < F1_STAGE) {
    cx = slot / (KC * PLANE);
    const int r = slot - cx * (KC * PLANE);
    c = r / PLANE;
    const int r2 = r - c * PLANE;
    const int row = r2 / LW, xp = r2 - row * LW;
    Y = S * (ys0 - R + row) + cy;
    X = S * (xs0 - R + xp) + cx;
  } else {
    const int s0 = slot - F1_STAGE;
    cx = s0 / (KC * RY * RX);
    const int r = s0 - cx * (KC * RY * RX);
    c = r / (RY * RX);
    const int r2 = r - c * (RY * RX);
    const int row = r2 / RX, xp = r2 - row * RX;
    Y = S * (ys0 + row) + cy;
    X = S * (xs0 + xp) + cx;
  }
  const bool in = Y >= 0 && Y < H && X >= 0 && X < W;
  return in ? (unsigned)(((long long)c * H * W + (long long)Y * W + X) * 4) : OOB;
}

struct Stager {
  unsigned voff[PIECES_PER_WAVE];
  int wave;
  __device__ inline void init(int H, int W, int cy, int ys0, int xs0) {
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __builtin_assume(wave >= 0 && wave < NWAVES);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < PIECES_PER_WAVE; ++j) {
      const int p = wave + NWAVES * j;
      voff[j] = p < PIECES ? slot_offset(p * PIECE + lane, H, W, cy, ys0, xs0) : OOB;
    }
  }
  // Enqueue the DMA of channels [c0, c0+KC) into `stage`.
  __device__ inline void issue(float *stage, const float *in0, const float *in1, long long item, int c0, int C,
                               long long plane, int dbg = 0) const {
    if (dbg & 12) {  // timing experiments only (results are garbage)
      const unsigned bytes = (unsigned)((long long)(C - c0) * plane * 4);
      const __amdgpu_buffer_rsrc_t r1 = make_rsrc(in1 + (item * C + c0) * plane, bytes);
      const int lane = threadIdx.x & 63;
      if (dbg & 4) {
#pragma unroll
        for (int j = 0; j < PIECES_PER_WAVE; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, stage + (wave + NWAVES * j) * PIECE, 4, (wave + NWAVES * j) * 256 + lane * 4, 0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < (PIECES_PER_WAVE + 3) / 4; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, stage + (wave + NWAVES * j) * PIECE * 4, 16, (wave + NWAVES * j) * 1024 + lane * 16, 0, 0, 0);
      }
      return;
    }
    const unsigned bytes = (unsigned)((long long)(C - c0) * plane * 4);
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(in0 + (item * C + c0) * plane, bytes);
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(in1 + (item * C + c0) * plane, bytes);
    __builtin_assume(wave >= 0 && wave < NWAVES);
#pragma unroll
    for (int j = 0; j < PIECES_PER_WAVE; ++j) {
      const int p = wave + NWAVES * j;
      if (p < F1_PIECES) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, stage + p * PIECE, 4, (int)voff[j], 0, 0, 0);
      } else if (p < PIECES) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, stage + p * PIECE, 4, (int)voff[j], 0, 0, 0);
      }
    }
  }
};

__global__ __launch_bounds__(NTHREADS, 2) void corr_mfma_fwd_kernel(const float *__restrict__ in0,
                                                                    const float *__restrict__ in1,
                                                                    float *__restrict__ out, int C, int H, int W, int nbx, int nby, int dbg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cx = wave & 1, ty = (wave >> 1) & 1, tx = wave >> 2;
  // XCD-aware block order: hardware deals consecutive block ids round-robin to
  // the 8 XCDs; remap so that each XCD walks a contiguous range of regions (one
  // image pair's windows overlap heavily and then share that XCD's L2).
  const int nblk = gridDim.x;
  int vb = blockIdx.x;
  if ((nblk & 7) == 0) vb = (vb & 7) * (nblk >> 3) + (vb >> 3);
  const int bx = vb % nbx;
  const int by = (vb / nbx) % nby;
  const int bz = vb / (nbx * nby);
  const int cy = bz & 1;
  const long long item = bz >> 1;
  const int ys0 = by * RY;                   // class-row origin of the region
  const int xs0 = bx * RX;                   // class-column origin (same for both cx)
  const int hs = (H - cy + S - 1) / S;       // class rows that exist for this cy
  if (ys0 >= hs) return;                     // uniform per block

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane fragment addresses inside one stage (floats)
  const int k = lane >> 4, n = lane & 15;
  const int a_off = F1_STAGE + (cx * KC + k) * (RY * RX) + (ty * TP + (n >> 2)) * RX + tx * TP + (n & 3);
  const int b_off = (cx * KC + k) * PLANE + (ty * TP + (n >> 2)) * LW + tx * TP + (n & 3);

  Stager st;
  st.init(H, W, cy, ys0, xs0);
  const long long plane = (long long)H * W;
  const int nchunks = (C + KC - 1) / KC;
  st.issue(lds, in0, in1, item, 0, C, plane);
  __syncthreads();   // drains the DMA (vmcnt(0)) and publishes the stage

  for (int ch = 0; ch < nchunks; ++ch) {
    const float *cur = lds + (ch & 1) * STAGE;
    if (ch + 1 < nchunks && !(dbg & 1)) st.issue(lds + ((ch + 1) & 1) * STAGE, in0, in1, item, (ch + 1) * KC, C, plane, dbg);
    if (!(dbg & 2))
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      // all fragment reads of the K-step are issued first; the MFMAs then wait
      // with a decreasing lgkmcnt, so one LDS latency is exposed per 36 MFMAs
      const float a = cur[a_off + kk * 4 * (RY * RX)];
      const float *bp = cur + b_off + kk * 4 * PLANE;
      float b[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = bp[(t / TQ) * 4 * LW + (t % TQ) * 4];
      __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t], acc[t], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: regroup G[p][q] by displacement d = q - p through LDS ----
  // lane holds, for tile t = (tqy,tqx) and register j:
  //   p = (py = lane>>4, px = j),  q = (4*tqy + (n>>2), 4*tqx + (n&3))
  // Slab image [row][col][dy][dx]: with the displacement innermost the 64 lanes
  // of one scatter instruction (which differ in dy, dx and row) spread over the
  // banks (<= 2-way); a [dy][dx][row][col] image would put them all on 2 banks.
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
        ob[(row * RW + col) * (SLAB_DY * GW) + ddl * GW + dxx] = pow2 ? v * scale : v / (float)C;
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
      const float *src = ob + (r * RW + c4 * 4) * (SLAB_DY * GW) + ddl * GW + dxx;
      const float4 v = make_float4(src[0], src[SLAB_DY * GW], src[2 * SLAB_DY * GW], src[3 * SLAB_DY * GW]);
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


// ---------------------------------------------------------------------------
// Backward.  Both gradients are the same GEMM with the roles swapped:
//   g0[c, o] = 1/C * sum_{w in window(o)} top[d = w - o, o] * f1[c, w]
//   g1[c, o] = 1/C * sum_{w in window(o)} top[d = o - w, w] * f0[c, w]
// (ops/correlation_op.cu.cc:119-181 and :183-248 of the reference, restricted to
// kernel_size 1 / stride_1 1, where the output-range tests reduce to "the
// correlation position lies inside the image").  Per 4x4 tile of class positions
// o:  G[16 channels x 16 o] += A[16 channels x 4 w] * B[4 w x 16 o]  with
// A = the other input's window (from LDS) and B = top gathered by displacement.
// One workgroup = 8 waves = {cx} x 2x2 tiles (as in the forward kernel) and all
// channels (16 M-tiles of 16, 64 accumulator registers per wave); the K loop
// walks the 28 window rows, one row (28 columns x 256 channels x 2 classes, plus
// the 8 x 21 top segments that pair with it) per LDS stage, filled by LDS-DMA.
constexpr int MT = 16;                         // M-tiles (of 16 channels) per pass
constexpr int CW = MT * 16;                    // 256 channels per pass
constexpr int ACS = 30;                        // A row stride: c*30 + k is conflict-free (30 = 2*odd)
static_assert(ACS >= LW && (ACS % 4) == 2 && ((ACS / 2) & 1), "A image would bank-conflict");
constexpr int A_STAGE = S * CW * ACS;          // 15360 floats
constexpr int T_STAGE = RY * GW * (S * RX);    // 2688 floats: [row 8][dx 21][16 columns]
constexpr int BW_STAGE = A_STAGE + T_STAGE;    // 18048 floats = 72 KB
static_assert(A_STAGE % PIECE == 0 && T_STAGE % PIECE == 0, "stage must split into wave pieces");
constexpr int A_PIECES = A_STAGE / PIECE;      // 240
constexpr int T_PIECES = T_STAGE / PIECE;      // 42
constexpr int A_PER_WAVE = A_PIECES / NWAVES;  // 30
constexpr int T_PER_WAVE = (T_PIECES + NWAVES - 1) / NWAVES;  // 6
static_assert(A_PIECES % NWAVES == 0, "A pieces must split evenly");
constexpr int OUT_HALF = (CW / 2) * RY * (S * RX);  // epilogue image of 128 channels
static_assert(OUT_HALF <= 2 * BW_STAGE, "epilogue image must fit the staging buffers");

__global__ __launch_bounds__(NTHREADS, 2) void corr_mfma_bwd_kernel(const float *__restrict__ topdiff,
                                                                    const float *__restrict__ in0,
                                                                    const float *__restrict__ in1,
                                                                    float *__restrict__ g0, float *__restrict__ g1,
                                                                    int C, int H, int W, int nbx, int nby) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_assume(wave >= 0 && wave < NWAVES);
  const int cx = wave & 1, ty = (wave >> 1) & 1, tx = wave >> 2;
  const int nblk = gridDim.x;
  int vb = blockIdx.x;
  if ((nblk & 7) == 0) vb = (vb & 7) * (nblk >> 3) + (vb >> 3);
  const int bx = vb % nbx;
  const int by = (vb / nbx) % nby;
  int bz = vb / (nbx * nby);
  const int cy = bz & 1;
  bz >>= 1;
  const bool wrt1 = bz & 1;                    // false: gradient wrt input 0, true: wrt input 1
  const long long item = bz >> 1;
  const int ys0 = by * RY, xs0 = bx * RX;
  const int hs = (H - cy + S - 1) / S;
  if (ys0 >= hs) return;
  const long long plane = (long long)H * W;
  const float *fin = wrt1 ? in0 : in1;         // the input the window is read from
  float *gout = wrt1 ? g1 : g0;
  const __amdgpu_buffer_rsrc_t rtop = make_rsrc(topdiff + item * (GW * GW) * plane, (unsigned)(GW * GW * plane * 4));

  const int k = lane >> 4, n = lane & 15;
  const int oyl = n >> 2, oxl = n & 3;
  const int row8 = ty * TP + oyl;
  // B fragment: column term of the LDS address and the dx' of K-step 0 (k-th window column)
  const int b_col = 2 * (tx * TP + oxl) + cx;
  const int dx0 = wrt1 ? (oxl + 2 * R - k) : (k - oxl);   // dx' at K-step s is dx0 -/+ 4*s
  const int a_off = (cx * CW + n) * ACS + tx * TP + k;    // + mt*16*ACS + 4*s

  for (int cg = 0; cg < C; cg += CW) {
    const int nmt = min(MT, (C - cg + 15) / 16);
    const __amdgpu_buffer_rsrc_t rin =
        make_rsrc(fin + (item * C + cg) * plane, (unsigned)((long long)(C - cg) * plane * 4));
    // loop-invariant part of the A-stage offsets: channel plane + column, or OOB
    unsigned a_voff[A_PER_WAVE];
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
      const int slot = (wave + NWAVES * j) * PIECE + lane;
      const int scx = slot / (CW * ACS), r = slot - scx * (CW * ACS);
      const int c = r / ACS, col = r - c * ACS;
      const int X = S * (xs0 - R + col) + scx;
      a_voff[j] = (col < LW && X >= 0 && X < W && cg + c < C) ? (unsigned)((c * plane + X) * 4) : OOB;
    }

    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int r, float *stage) {
      // ---- A: window row r of the other input, all channels, both column classes
      const int Y = S * (ys0 - R + r) + cy;
      const bool rowok = Y >= 0 && Y < H;
      const int soff = rowok ? Y * W * 4 : 0;
#pragma unroll
      for (int j = 0; j < A_PER_WAVE; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, stage + (wave + NWAVES * j) * PIECE, 4,
                                                 (int)(rowok ? a_voff[j] : OOB), soff, 0, 0);
      // ---- top segments that pair with window row r: [row8][dx'][16 columns]
#pragma unroll
      for (int j = 0; j < T_PER_WAVE; ++j) {
        const int pc = wave + NWAVES * j;
        if (pc < T_PIECES) {
          const int slot = pc * PIECE + lane;
          const int tr = slot / (GW * S * RX), rem = slot - tr * (GW * S * RX);
          const int dxp = rem / (S * RX), col = rem - dxp * (S * RX);
          int dyp, Yt, Xt;
          if (wrt1) {   // position = window position w: row r, column shifted by -dx
            dyp = tr - r + 2 * R;
            Yt = Y;
            Xt = S * xs0 - S * (dxp - R) + col;
          } else {      // position = output position o: region row tr
            dyp = r - tr;
            Yt = S * (ys0 + tr) + cy;
            Xt = S * xs0 + col;
          }
          const bool ok = dyp >= 0 && dyp <= 2 * R && Yt >= 0 && Yt < H && Xt >= 0 && Xt < W;
          const unsigned voff = ok ? (unsigned)((((long long)(dyp * GW + dxp) * H + Yt) * W + Xt) * 4) : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rtop, stage + A_STAGE + pc * PIECE, 4, (int)voff, 0, 0, 0);
        }
      }
    };

    issue(0, lds);
    __syncthreads();
    for (int r = 0; r < LH; ++r) {
      const float *cur = lds + (r & 1) * BW_STAGE;
      if (r + 1 < LH) issue(r + 1, lds + ((r + 1) & 1) * BW_STAGE);
      // this wave's tile sees window rows [4*ty, 4*ty + 24)
      if (r >= ty * TP && r < ty * TP + WIN) {
        const int dyp = wrt1 ? (row8 - r + 2 * R) : (r - row8);
        const bool dyok = dyp >= 0 && dyp <= 2 * R;
        const float *tp = cur + A_STAGE + row8 * (GW * S * RX) + b_col;
#pragma unroll
        for (int s4 = 0; s4 < WIN / 4; ++s4) {
          const int dxp = wrt1 ? dx0 - 4 * s4 : dx0 + 4 * s4;
          const bool ok = dyok && dxp >= 0 && dxp <= 2 * R;
          const float bv = tp[min(max(dxp, 0), 2 * R) * (S * RX)];
          const float b = ok ? bv : 0.f;
          const float *ap = cur + a_off + 4 * s4;
          float a[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) a[m] = ap[m * 16 * ACS];
#pragma unroll
          for (int m = 0; m < MT; ++m)
            if (m < nmt) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b, acc[m], 0, 0, 0);
        }
      }
      __syncthreads();
    }

    // ---- epilogue: acc[m][j] = G[c = 16m + 4*(lane>>4) + j][o = n]; regroup through
    // LDS to [channel][row 8][16 columns] and store 64-byte row segments.
    const float inv = 1.f / (float)C;
    const bool pow2 = (C & (C - 1)) == 0;
    float *ob = lds;
    constexpr int RW = S * RX;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int m = h * (MT / 2); m < (h + 1) * (MT / 2); ++m) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cl = (m - h * (MT / 2)) * 16 + 4 * k + j;
          const float v = acc[m][j];
          ob[(cl * RY + row8) * RW + b_col] = pow2 ? v * inv : v / (float)C;
        }
      }
      __syncthreads();
      for (int idx = threadIdx.x; idx < OUT_HALF / 4; idx += NTHREADS) {
        const int c4 = idx % (RW / 4);
        const int rr = (idx / (RW / 4)) % RY;
        const int cl = idx / (RW / 4 * RY);
        const int c = cg + h * (CW / 2) + cl;
        const int y = S * (ys0 + rr) + cy, x = S * xs0 + c4 * 4;
        if (c >= C || y >= H || x >= W) continue;
        const float4 v = *reinterpret_cast<const float4 *>(ob + idx * 4);
        float *dst = gout + (item * C + c) * plane + (long long)y * W + x;
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
  const int nbx = (ws + RX - 1) / RX, nby = (hs + RY - 1) / RY;
  const dim3 grid(nbx * nby * B * S);
  const size_t smem = sizeof(float) * 2 * STAGE;
  // > 64 KiB of dynamic LDS needs an opt-in; thread-safe one-time initialisation
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(corr_mfma_fwd_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (attr != hipSuccess) return fail(UNFLOW_ERR_LAUNCH, "correlation_fwd (mfma): cannot reserve %zu B of LDS", smem);
  static const int dbg = getenv("UNFLOW_CORR_DBG") ? atoi(getenv("UNFLOW_CORR_DBG")) : 0;
  hipLaunchKernelGGL(corr_mfma_fwd_kernel, grid, dim3(NTHREADS), smem, s, in0, in1, out, C, H, W, nbx, nby, dbg);
  return launch_status("correlation_fwd (mfma)");
}

bool corr_mfma_bwd_supported(const CorrGeometry &g, int C) {
  // buffer descriptors address one image pair with 32-bit offsets
  return corr_mfma_supported(g, C) && (long long)(GW * GW) * g.H * g.W * 4 < (1LL << 31) &&
         (long long)C * g.H * g.W * 4 < (1LL << 31);
}

int corr_mfma_bwd(const float *topdiff, const float *in0, const float *in1, float *g0, float *g1, int B, int C,
                  const CorrGeometry &g, hipStream_t s) {
  const int H = g.H, W = g.W;
  const int hs = (H + S - 1) / S, ws = (W + S - 1) / S;
  const int nbx = (ws + RX - 1) / RX, nby = (hs + RY - 1) / RY;
  const dim3 grid(nbx * nby * B * S * 2);   // x2: gradient wrt input 0 and wrt input 1
  const size_t smem = sizeof(float) * 2 * BW_STAGE;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(corr_mfma_bwd_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (attr != hipSuccess) return fail(UNFLOW_ERR_LAUNCH, "correlation_bwd (mfma): cannot reserve %zu B of LDS", smem);
  hipLaunchKernelGGL(corr_mfma_bwd_kernel, grid, dim3(NTHREADS), smem, s, topdiff, in0, in1, g0, g1, C, H, W, nbx,
                     nby);
  return launch_status("correlation_bwd (mfma)");
}

}  // namespace unflow
