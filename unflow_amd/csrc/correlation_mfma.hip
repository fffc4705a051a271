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
// Staging.  Operands reach LDS by LDS-DMA (buffer_load_dwordx4 ... lds): one
// wave instruction fills 1 KiB of LDS from 64 per-lane 16-byte segments, and the
// buffer's range check returns 0 for the zero padding and for channels >= C, so
// no data register, no ds_write and no branch is spent on staging.  The LDS
// images keep the NCHW row order (both column classes interleaved) because only
// whole 16-byte row segments stream at full rate (measured, profiles/
// ubench_lds_dma.txt: 10.3 TB/s for 224-byte row segments with dwordx4 against
// 5-6 TB/s for any 4-byte gather that de-interleaves the classes); the price is
// a 2-way bank conflict on the fragment reads, which the LDS has room for.  A
// DMA instruction holds the CU's texture addresser for several cycles, so the
// DMA of stage i+1 is issued in slices between the MFMA groups of stage i.
//
// THIS FILE holds the first-generation kernels.  The float32 FlowNetC shape with whole 16-channel chunks and 16-byte
// aligned rows runs correlation_fwd2.hip (forward) and correlation_bwd4.hip (backward); what stays here serves every other
// case: channel counts that are not multiples of 16 / 128, rows that are not 16-byte aligned (4-byte LDS-DMA), and fp16
// STORAGE on the f32 matrix cores (unflow_correlation_*_f16_exact).
#include "corr_mfma_common.h"

namespace unflow {
namespace {

// ===========================================================================
// Forward
// ===========================================================================
constexpr int KC = 8;                        // channels per LDS stage (= waves: wave w stages channel w)
static_assert(KC == NWAVES, "each wave stages one channel plane of the chunk");
constexpr int PLANE = LH * ROWF;             // 1568 floats: [row 28][x 56] of one channel
constexpr int PLS = PLANE + 4;               // plane stride; odd planes start ONE FLOAT later (see below)
constexpr int F0PL = RY * RW;                // 128 floats:  [row 8][x 16]
constexpr int F1_STAGE = (KC * PLS + 1 + 63) / 64 * 64;   // 12608
constexpr int F0_STAGE = KC * F0PL;          // 1024
constexpr int STAGE = F1_STAGE + F0_STAGE;   // 13632 floats = 53 KB
constexpr int SLAB_DY = 7;                   // displacement rows per epilogue slab
constexpr int SLAB = SLAB_DY * GW * RY * RW; // floats
static_assert(SLAB <= 2 * STAGE, "epilogue slab must fit the staging buffers");
static_assert(GW % SLAB_DY == 0 && GW / SLAB_DY == 3, "three slabs tile the displacement rows");
// B fragment (ds_read_b32; per 32-lane half: 2 channels x 4 rows x 4 columns of
// one class): rows land on distinct multiples of 8 (56 = 24 mod 32), columns on
// 2*col, so the 16 positions of ONE channel take the 16 banks of one parity.
// The half's second channel must therefore sit at an odd offset: the image of an
// odd channel starts one float later (plane offset c*1572 + (c&1), 1573 = 5 mod
// 32), which makes the read conflict free.  16-byte LDS-DMA into a destination
// that is only 4-byte aligned works (tools/ubench/lds_dma_unaligned.hip).
static_assert(ROWF % 32 == 24 && (PLS + 1) % 2 == 1, "LDS image would bank-conflict");

template <int VEC>
struct FwdStager {
  // wave w stages plane w of f1 (1568 floats) piece by piece, and a share of the f0 image
  static constexpr int PIECE = 64 * VEC;
  static constexpr int F1_PER_WAVE = (PLANE + PIECE - 1) / PIECE;        // 7 (dwordx4) or 25 (dword); last one partial
  static_assert(F0_STAGE % PIECE == 0, "f0 image must split into wave pieces");
  static constexpr int F0_PIECES = F0_STAGE / PIECE;                     // 4 or 16
  static constexpr int F0_PER_WAVE = (F0_PIECES + NWAVES - 1) / NWAVES;  // 1 or 2
  static constexpr int PER_WAVE = F1_PER_WAVE + F0_PER_WAVE;

  unsigned voff[PER_WAVE];
  int wave;
  __amdgpu_buffer_rsrc_t r0, r1;

  __device__ inline void init(int H, int W, int cy, int ys0, int xs0) {
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __builtin_assume(wave >= 0 && wave < NWAVES);
    const int lane = threadIdx.x & 63;
    const long long plane = (long long)H * W;
    // VEC > 1: X and W are multiples of VEC, so a segment is inside or outside as a whole
#pragma unroll
    for (int j = 0; j < F1_PER_WAVE; ++j) {
      const int f = (j * 64 + lane) * VEC;               // float inside the plane
      const int row = f / ROWF, x = f - row * ROWF;
      const int Y = S * (ys0 - R + row) + cy, X = S * (xs0 - R) + x;
      const bool in = f < PLANE && Y >= 0 && Y < H && X >= 0 && X < W;
      voff[j] = in ? (unsigned)((wave * plane + (long long)Y * W + X) * 4) : OOB;
    }
#pragma unroll
    for (int j = 0; j < F0_PER_WAVE; ++j) {
      const int p = wave + NWAVES * j;
      const int f = (p * 64 + lane) * VEC;               // float inside the f0 image [c][row][x]
      const int c = f / F0PL, r = f - c * F0PL, row = r / RW, x = r - row * RW;
      const int Y = S * (ys0 + row) + cy, X = S * xs0 + x;
      const bool in = p < F0_PIECES && Y >= 0 && Y < H && X >= 0 && X < W;
      voff[F1_PER_WAVE + j] = in ? (unsigned)((c * plane + (long long)Y * W + X) * 4) : OOB;
    }
  }
  __device__ inline void bind(const float *in0, const float *in1, long long item, int c0, int C, long long plane) {
    const unsigned bytes = (unsigned)((long long)(C - c0) * plane * 4);
    r0 = make_rsrc(in0 + (item * C + c0) * plane, bytes);
    r1 = make_rsrc(in1 + (item * C + c0) * plane, bytes);
  }
  template <int J>
  __device__ inline void issue_one(float *stage) const {
    if constexpr (J < F1_PER_WAVE) {
      __builtin_assume(wave >= 0 && wave < NWAVES);
      float *dst = stage + wave * PLS + (wave & 1) + J * PIECE;
      if constexpr ((J + 1) * PIECE <= PLANE) {
        dma<VEC>(r1, dst, voff[J], 0);
      } else {                                             // partial last piece: only the lanes inside the plane
        if ((int)((J * 64 + (threadIdx.x & 63)) * VEC) < PLANE) dma<VEC>(r1, dst, voff[J], 0);
      }
    } else if constexpr (J < PER_WAVE) {
      __builtin_assume(wave >= 0 && wave < NWAVES);
      const int p = wave + NWAVES * (J - F1_PER_WAVE);
      if (p < F0_PIECES) dma<VEC>(r0, stage + F1_STAGE + p * PIECE, voff[J], 0);
    }
  }
  template <int J0, int N>
  __device__ inline void issue_range(float *stage) const {
    if constexpr (N > 0) {
      issue_one<J0>(stage);
      issue_range<J0 + 1, N - 1>(stage);
    }
  }
};

template <int VEC>
__global__ __launch_bounds__(NTHREADS, 2) void corr_mfma_fwd_kernel(const float *__restrict__ in0,
                                                                    const float *__restrict__ in1,
                                                                    float *__restrict__ out, int C, int H, int W,
                                                                    int nbx, int nby) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // waves w and w+4 share a SIMD: they get the two tile ROWS, whose numbers of
  // live window rows differ at the top and bottom of the image (rowmask), so that
  // every SIMD carries the same MFMA load between two barriers
  const int cx = wave & 1, tx = (wave >> 1) & 1, ty = wave >> 2;
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = vb % nbx, by = (vb / nbx) % nby, bz = vb / (nbx * nby);
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
  const int a_off = F1_STAGE + k * F0PL + (ty * TP + (n >> 2)) * RW + S * (tx * TP + (n & 3)) + cx;
  const int b_off = k * PLS + (k & 1) + (ty * TP + (n >> 2)) * ROWF + S * (tx * TP + (n & 3)) + cx;

  // Rows of N-tiles (4 window rows) that lie entirely in the zero padding above
  // or below the image have an all-zero B operand: their MFMAs are skipped
  // (wave-uniform mask).  At 48x160 that is 17 % of the tiles.
  unsigned rowmask = 0;
  {
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int tyu = wv >> 2;
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
      const int r0 = ys0 + tyu * TP - R + 4 * q;
      if (r0 + 3 >= 0 && r0 < hs) rowmask |= 1u << q;
    }
  }

  using St = FwdStager<VEC>;
  St st;
  st.init(H, W, cy, ys0, xs0);
  const long long plane = (long long)H * W;
  const int nchunks = (C + KC - 1) / KC;
  st.bind(in0, in1, item, 0, C, plane);
  st.template issue_range<0, St::PER_WAVE>(lds);
  dma_barrier();   // drains the DMA (vmcnt(0)) and publishes the stage

  constexpr int GROUPS = (KC / 4) * TQ;                              // MFMA groups per chunk (12)
  // DMA pieces issued after each group: all of them in the first half of the
  // chunk, so that the last one has half a chunk to land before the barrier
  constexpr int PER_GROUP = (2 * St::PER_WAVE + GROUPS - 1) / GROUPS;
  for (int ch = 0; ch < nchunks; ++ch) {
    const float *cur = lds + (ch & 1) * STAGE;
    float *nxt = lds + ((ch + 1) & 1) * STAGE;
    const bool more = ch + 1 < nchunks;
    if (more) st.bind(in0, in1, item, (ch + 1) * KC, C, plane);
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      // all fragment reads of the K-step are issued first; the MFMAs then wait
      // with a decreasing lgkmcnt, so one LDS latency is exposed per 36 MFMAs
      const float a = cur[a_off + kk * 4 * F0PL];
      const float *bp = cur + b_off + kk * 4 * PLS;
      float b[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = bp[(t / TQ) * 4 * ROWF + (t % TQ) * 4 * S];
      __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs
      auto group = [&](auto G) {
        constexpr int g = decltype(G)::value;
        // one branch per row of 6 N-tiles, not one per MFMA: a taken branch after
        // every MFMA starves the matrix pipe (45 % busy with per-tile column masks)
        if (rowmask & (1u << g)) {
#pragma unroll
          for (int t = g * TQ; t < (g + 1) * TQ; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t], acc[t], 0, 0, 0);
        }
        if (more) {
          if (kk == 0) st.template issue_range<g * PER_GROUP, PER_GROUP>(nxt);
          else st.template issue_range<(TQ + g) * PER_GROUP, PER_GROUP>(nxt);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      group(Int<0>{}); group(Int<1>{}); group(Int<2>{}); group(Int<3>{}); group(Int<4>{}); group(Int<5>{});
    }
    dma_barrier();
  }

  // ---- epilogue: regroup G[p][q] by displacement d = q - p through LDS ----
  // lane holds, for tile t = (tqy,tqx) and register j:
  //   p = (py = lane>>4, px = j),  q = (4*tqy + (n>>2), 4*tqx + (n&3))
  // Slab image [row][col][dy][dx]: with the displacement innermost the 64 lanes
  // of one scatter instruction (which differ in dy, dx and row) spread over the
  // banks (<= 2-way); a [dy][dx][row][col] image would put them all on 2 banks.
  // Everything that depends on the lane is computed once: with u = (n>>2) - py and
  // bq = n&3 an element (t, j) has dy' = 4*tqy + u and dx' = 4*tqx + bq - j, so its
  // slab address is L[j] + a compile-time constant, its dy' test is a range test on
  // u per (slab, tqy), and its dx' test only exists for the first and last tile
  // column.  (Per-element index arithmetic and an inline float division made this
  // epilogue cost a third of the kernel.)
  const float scale = 1.f / (float)C;
  const bool pow2 = (C & (C - 1)) == 0;
  const int py = lane >> 4;
  const int row = ty * TP + py;                       // class row inside the region
  const int D = GW * GW;
  float *ob = lds;
  const int u = (n >> 2) - py, bq = n & 3;
  int L[4];
  bool mx_first[4], mx_last[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    L[j] = (row * RW + S * (tx * TP + j) + cx) * (SLAB_DY * GW) + u * GW + (bq - j);
    mx_first[j] = bq >= j;       // tqx == 0     : dx' = bq - j >= 0
    mx_last[j] = bq <= j;        // tqx == TQ - 1: dx' = 4*(TQ-1) + bq - j <= 2R
  }
  static_assert(4 * (TQ - 1) == 2 * R, "last tile column starts at dx' = 2R");
  auto scatter = [&](auto SLABNO, auto DIVIDE) {
    constexpr int s = decltype(SLABNO)::value;
    constexpr bool divide = decltype(DIVIDE)::value;
#pragma unroll
    for (int tqy = 0; tqy < TQ; ++tqy) {
      // dy' = 4*tqy + u must lie in [7s, min(7s+6, 2R)], u in [-3, 3]
      constexpr int dlo = s * SLAB_DY, dhi = (s * SLAB_DY + SLAB_DY - 1) < 2 * R ? (s * SLAB_DY + SLAB_DY - 1) : 2 * R;
      const int lo = dlo - 4 * tqy, hi = dhi - 4 * tqy;
      if (hi < -3 || lo > 3) continue;          // compile-time: this tile row cannot hit the slab
      const bool my = u >= lo && u <= hi;
#pragma unroll
      for (int tqx = 0; tqx < TQ; ++tqx) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = my && (tqx == 0 ? mx_first[j] : (tqx == TQ - 1 ? mx_last[j] : true));
          if (ok) {
            const float v = acc[tqy * TQ + tqx][j];
            ob[L[j] + (4 * tqy - s * SLAB_DY) * GW + 4 * tqx] = divide ? v / (float)C : v * scale;
          }
        }
      }
    }
  };
#pragma unroll
  for (int s = 0; s < GW / SLAB_DY; ++s) {
    if (pow2) {
      if (s == 0) scatter(Int<0>{}, std::false_type{});
      else if (s == 1) scatter(Int<1>{}, std::false_type{});
      else scatter(Int<2>{}, std::false_type{});
    } else {
      if (s == 0) scatter(Int<0>{}, std::true_type{});
      else if (s == 1) scatter(Int<1>{}, std::true_type{});
      else scatter(Int<2>{}, std::true_type{});
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
      float *dst = out + (item * D + d) * plane + (long long)y * W + x;
      if (VEC == 4 && x + 3 < W && (W & 3) == 0) {   // VEC == 4: the host checked 16-byte alignment of out
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

// ===========================================================================
// Forward, two workgroups per CU.  The kernel above holds 36 accumulator tiles
// per wave (256 VGPRs, one 8-wave workgroup per CU): whenever that workgroup waits
// at a barrier or writes its epilogue the matrix pipe idles (44 % of its time,
// cycle stamps in DESIGN.md).  Here a workgroup computes only HALF of the window
// rows of a region (N-tile rows 3h .. 3h+2): 18 accumulator tiles per wave, 128
// VGPRs, 16 of the 28 window rows staged (64.5 KB of LDS), so two independent
// workgroups share a CU and one's barriers and epilogue hide behind the other's
// MFMAs.  The two halves of a region write disjoint 64-byte output segments:
// element (position row py, displacement row dy') belongs to the half that holds
// window row py + dy' (>= 12: upper half).  DMA volume 16/14 of the kernel above.
// ===========================================================================
constexpr int TQH = TQ / 2;                    // 3 N-tile rows per workgroup
constexpr int NTH = TQH * TQ;                  // 18 accumulator tiles per wave
constexpr int LHH = RY + 4 * TQH - 4;          // 16 staged window rows
constexpr int PLANEH = LHH * ROWF;             // 896
constexpr int PLSH = PLANEH + 4;               // 900 = 4 mod 32 like PLS
constexpr int F1_STAGEH = (KC * PLSH + 1 + 63) / 64 * 64;   // 7232
constexpr int STAGEH = F1_STAGEH + F0_STAGE;   // 8256 floats = 32.25 KB
constexpr int SLAB_DYH = 6;                    // displacement rows per epilogue slab
constexpr int SLABH = SLAB_DYH * GW * RY * RW; // 16128 floats
static_assert(SLABH <= 2 * STAGEH, "epilogue slab must fit the staging buffers");
static_assert(PLSH % 32 == PLS % 32, "same bank pattern as the full-window image");

template <int VEC>
struct FwdStagerH {
  static constexpr int PIECE = 64 * VEC;
  static constexpr int F1_PER_WAVE = (PLANEH + PIECE - 1) / PIECE;       // 4 (dwordx4, last one half) or 14
  static constexpr int F0_PIECES = F0_STAGE / PIECE;
  static constexpr int F0_PER_WAVE = (F0_PIECES + NWAVES - 1) / NWAVES;
  static constexpr int PER_WAVE = F1_PER_WAVE + F0_PER_WAVE;

  unsigned voff[PER_WAVE];
  int wave;
  __amdgpu_buffer_rsrc_t r0, r1;

  // wrow0: first staged window row, relative to the region's window (0 or 12)
  __device__ inline void init(int H, int W, int cy, int ys0, int xs0, int wrow0) {
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __builtin_assume(wave >= 0 && wave < NWAVES);
    const int lane = threadIdx.x & 63;
    const long long plane = (long long)H * W;
#pragma unroll
    for (int j = 0; j < F1_PER_WAVE; ++j) {
      const int f = (j * 64 + lane) * VEC;
      const int row = f / ROWF, x = f - row * ROWF;
      const int Y = S * (ys0 - R + wrow0 + row) + cy, X = S * (xs0 - R) + x;
      const bool in = f < PLANEH && Y >= 0 && Y < H && X >= 0 && X < W;
      voff[j] = in ? (unsigned)((wave * plane + (long long)Y * W + X) * 4) : OOB;
    }
#pragma unroll
    for (int j = 0; j < F0_PER_WAVE; ++j) {
      const int p = wave + NWAVES * j;
      const int f = (p * 64 + lane) * VEC;
      const int c = f / F0PL, r = f - c * F0PL, row = r / RW, x = r - row * RW;
      const int Y = S * (ys0 + row) + cy, X = S * xs0 + x;
      const bool in = p < F0_PIECES && Y >= 0 && Y < H && X >= 0 && X < W;
      voff[F1_PER_WAVE + j] = in ? (unsigned)((c * plane + (long long)Y * W + X) * 4) : OOB;
    }
  }
  __device__ inline void bind(const float *in0, const float *in1, long long item, int c0, int C, long long plane) {
    const unsigned bytes = (unsigned)((long long)(C - c0) * plane * 4);
    r0 = make_rsrc(in0 + (item * C + c0) * plane, bytes);
    r1 = make_rsrc(in1 + (item * C + c0) * plane, bytes);
  }
  template <int J>
  __device__ inline void issue_one(float *stage) const {
    if constexpr (J < F1_PER_WAVE) {
      __builtin_assume(wave >= 0 && wave < NWAVES);
      float *dst = stage + wave * PLSH + (wave & 1) + J * PIECE;
      if constexpr ((J + 1) * PIECE <= PLANEH) {
        dma<VEC>(r1, dst, voff[J], 0);
      } else {
        if ((int)((J * 64 + (threadIdx.x & 63)) * VEC) < PLANEH) dma<VEC>(r1, dst, voff[J], 0);
      }
    } else if constexpr (J < PER_WAVE) {
      __builtin_assume(wave >= 0 && wave < NWAVES);
      const int p = wave + NWAVES * (J - F1_PER_WAVE);
      if (p < F0_PIECES) dma<VEC>(r0, stage + F1_STAGEH + p * PIECE, voff[J], 0);
    }
  }
  template <int J0, int N>
  __device__ inline void issue_range(float *stage) const {
    if constexpr (N > 0) {
      issue_one<J0>(stage);
      issue_range<J0 + 1, N - 1>(stage);
    }
  }
};

// fp16 STORAGE (extension, not in the reference): the operands are half in HBM and are
// converted to f32 on their way into the same LDS images, so the MFMA loop -- f32 inputs,
// f32 accumulation, every product exact -- is unchanged.  LDS-DMA cannot convert, so the
// next chunk is fetched into registers (8-byte loads of 4 halves, range-checked by the
// buffer descriptor) at the start of a chunk and converted + written to LDS at its end.
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct FwdStagerH16 {
  static constexpr int F1_ROUNDS = (PLANEH / 4 + 63) / 64;   // 4 rounds of 64 lanes x 4 halves (last one half)
  unsigned voff1[F1_ROUNDS], voff0;
  i32x2 r1[F1_ROUNDS], r0v;
  int wave, lane;
  __amdgpu_buffer_rsrc_t rs0, rs1;

  __device__ inline void init(int H, int W, int cy, int ys0, int xs0, int wrow0) {
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __builtin_assume(wave >= 0 && wave < NWAVES);
    lane = threadIdx.x & 63;
    const long long plane = (long long)H * W;
#pragma unroll
    for (int j = 0; j < F1_ROUNDS; ++j) {
      const int f = (j * 64 + lane) * 4;
      const int row = f / ROWF, x = f - row * ROWF;
      const int Y = S * (ys0 - R + wrow0 + row) + cy, X = S * (xs0 - R) + x;
      const bool in = f < PLANEH && Y >= 0 && Y < H && X >= 0 && X < W;   // X, W multiples of 4: all or nothing
      voff1[j] = in ? (unsigned)((wave * plane + (long long)Y * W + X) * 2) : OOB;
    }
    const int f = (int)threadIdx.x * 4;                                   // float inside the f0 image [c][row][x]
    const int c = f / F0PL, r = f - c * F0PL, row = r / RW, x = r - row * RW;
    const int Y = S * (ys0 + row) + cy, X = S * xs0 + x;
    const bool in = f < F0_STAGE && Y >= 0 && Y < H && X >= 0 && X < W;
    voff0 = in ? (unsigned)((c * plane + (long long)Y * W + X) * 2) : OOB;
  }
  __device__ inline void bind(const _Float16 *in0, const _Float16 *in1, long long item, int c0, int C, long long plane) {
    const int bytes = (int)((long long)(C - c0) * plane * 2);
    rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(in0 + (item * C + c0) * plane), (short)0, bytes, 0x00020000);
    rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(in1 + (item * C + c0) * plane), (short)0, bytes, 0x00020000);
  }
  __device__ inline void load() {
#pragma unroll
    for (int j = 0; j < F1_ROUNDS; ++j) r1[j] = __builtin_amdgcn_raw_buffer_load_b64(rs1, (int)voff1[j], 0, 0);
    r0v = __builtin_amdgcn_raw_buffer_load_b64(rs0, (int)voff0, 0, 0);
  }
  __device__ inline void store(float *stage) const {
#pragma unroll
    for (int j = 0; j < F1_ROUNDS; ++j) {
      const int f = (j * 64 + lane) * 4;
      if (f < PLANEH) {
        const f16x4 h = __builtin_bit_cast(f16x4, r1[j]);
        float *dst = stage + wave * PLSH + (wave & 1) + f;
        dst[0] = (float)h.x; dst[1] = (float)h.y; dst[2] = (float)h.z; dst[3] = (float)h.w;
      }
    }
    const int f = (int)threadIdx.x * 4;
    if (f < F0_STAGE) {
      const f16x4 h = __builtin_bit_cast(f16x4, r0v);
      float *dst = stage + F1_STAGEH + f;
      dst[0] = (float)h.x; dst[1] = (float)h.y; dst[2] = (float)h.z; dst[3] = (float)h.w;
    }
  }
};

template <int VEC, bool F16>
__global__ __launch_bounds__(NTHREADS, 4) void corr_mfma_fwd_half_kernel(
    const std::conditional_t<F16, _Float16, float> *__restrict__ in0,
    const std::conditional_t<F16, _Float16, float> *__restrict__ in1,
    std::conditional_t<F16, _Float16, float> *__restrict__ out, int C, int H, int W, int nbx, int nby) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cx = wave & 1, tx = (wave >> 1) & 1, ty = wave >> 2;
  int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int half = vb & 1;                     // uniform: which three N-tile rows
  vb >>= 1;
  const int bx = vb % nbx, by = (vb / nbx) % nby, bz = vb / (nbx * nby);
  const int cy = bz & 1;
  const long long item = bz >> 1;
  const int ys0 = by * RY, xs0 = bx * RX;
  const int hs = (H - cy + S - 1) / S;
  if (ys0 >= hs) return;

  f32x4 acc[NTH];
#pragma unroll
  for (int t = 0; t < NTH; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int k = lane >> 4, n = lane & 15;
  const int a_off = F1_STAGEH + k * F0PL + (ty * TP + (n >> 2)) * RW + S * (tx * TP + (n & 3)) + cx;
  const int b_off = k * PLSH + (k & 1) + (ty * TP + (n >> 2)) * ROWF + S * (tx * TP + (n & 3)) + cx;

  unsigned rowmask = 0;
  {
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int tyu = wv >> 2;
#pragma unroll
    for (int q = 0; q < TQH; ++q) {
      const int r0 = ys0 + tyu * TP - R + 4 * (TQH * half + q);
      if (r0 + 3 >= 0 && r0 < hs) rowmask |= 1u << q;
    }
  }

  using St = std::conditional_t<F16, FwdStagerH16, FwdStagerH<VEC>>;
  St st;
  st.init(H, W, cy, ys0, xs0, 4 * TQH * half);
  const long long plane = (long long)H * W;
  const int nchunks = (C + KC - 1) / KC;
  st.bind(in0, in1, item, 0, C, plane);
  if constexpr (F16) {
    st.load();
    st.store(lds);
  } else {
    st.template issue_range<0, St::PER_WAVE>(lds);
    if (nchunks > 1) {                           // f32: the DMA runs two chunks ahead (see the loop)
      st.bind(in0, in1, item, KC, C, plane);
      st.template issue_range<0, St::PER_WAVE>(lds + STAGEH);
    }
  }
  dma_barrier();

#ifdef UNFLOW_TUNE_STAMPS
  int stamp_slot = -1;           // workgroups 0..3 and 1000..1003, waves 0 and 4
  {
    const int bsel = (int)blockIdx.x < 4 ? (int)blockIdx.x : ((int)blockIdx.x >= 1000 && (int)blockIdx.x < 1004 ? (int)blockIdx.x - 996 : -1);
    if (!F16 && bsel >= 0 && (wave == 0 || wave == 4)) stamp_slot = bsel * 2 + (wave >> 2);
  }
  UNFLOW_STAMP(STAMP_ROWS, 0);
#endif
  for (int ch = 0; ch < nchunks; ++ch) {
    UNFLOW_STAMP(ch, 0);
    float *cur = lds + (ch & 1) * STAGEH;
    float *nxt = lds + ((ch + 1) & 1) * STAGEH;
    // f32: the chunk's barrier stands in front of the MFMAs of its SECOND K-step, behind that
    // K-step's fragment reads -- the last reads of this stage.  When it opens, 18 MFMAs are ready to
    // issue (a barrier at the end of the chunk was followed by 19 LDS reads with an empty matrix
    // pipe), chunk ch + 1 has landed in the other stage, and this stage takes the DMA of chunk
    // ch + 2, issued between the MFMA groups that follow.
    // fp16 storage: the next chunk is loaded into registers at the top and written to the other
    // stage at the end of the chunk, with the barrier behind it.
    const bool more = F16 ? ch + 1 < nchunks : ch + 2 < nchunks;
    if constexpr (F16) {
      if (more) {
        st.bind(in0, in1, item, (ch + 1) * KC, C, plane);
        st.load();                               // lands in registers while this chunk computes
      }
    }
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      const float a = cur[a_off + kk * 4 * F0PL];
      const float *bp = cur + b_off + kk * 4 * PLSH;
      float b[NTH];
#pragma unroll
      for (int t = 0; t < NTH; ++t) b[t] = bp[(t / TQ) * 4 * ROWF + (t % TQ) * 4 * S];
      if constexpr (!F16) {
        if (kk == KC / 4 - 1) {
          UNFLOW_STAMP(ch, 2);
          dma_barrier();   // (a plain __syncthreads() came out as lgkmcnt(0) alone here)
          UNFLOW_STAMP(ch, 3);
          if (more) st.bind(in0, in1, item, (ch + 2) * KC, C, plane);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      auto group = [&](auto G) {
        constexpr int g = decltype(G)::value;
        if (rowmask & (1u << g)) {
#pragma unroll
          for (int t = g * TQ; t < (g + 1) * TQ; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t], acc[t], 0, 0, 0);
        }
        if constexpr (!F16) {
          constexpr int PER_GROUP = (St::PER_WAVE + TQH - 1) / TQH;   // all DMA behind the barrier, in the second K-step
#ifdef UNFLOW_TUNE_FWD_NODMA
          if (more && C < 0) {
#else
          if (more) {
#endif
            if (kk == KC / 4 - 1) st.template issue_range<g * PER_GROUP, PER_GROUP>(cur);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      group(Int<0>{}); group(Int<1>{}); group(Int<2>{});
    }
    if constexpr (F16) {
      if (more) st.store(nxt);
      UNFLOW_STAMP(ch, 2);
      __syncthreads();
      UNFLOW_STAMP(ch, 3);
    }
  }
  UNFLOW_STAMP(STAMP_ROWS, 1);   // epilogue starts

  // ---- epilogue (see the full-window kernel): two slabs of 6 displacement rows,
  // dy' = 9*half + 6*s + ddl; local tile row tl holds dy' = 4*(3*half + tl) + u.
  const float scale = 1.f / (float)C;
  const bool pow2 = (C & (C - 1)) == 0;
  const int py = lane >> 4;
  const int row = ty * TP + py;
  const int D = GW * GW;
  float *ob = lds;
  const int u = (n >> 2) - py, bq = n & 3;
  int L[4];
  bool mx_first[4], mx_last[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    L[j] = (row * RW + S * (tx * TP + j) + cx) * (SLAB_DYH * GW) + u * GW + (bq - j);
    mx_first[j] = bq >= j;
    mx_last[j] = bq <= j;
  }
  auto epilogue = [&](auto HALF) {
    constexpr int hf = decltype(HALF)::value;
    auto scatter = [&](auto SLABNO, auto DIVIDE) {
      constexpr int s = decltype(SLABNO)::value;
      constexpr bool divide = decltype(DIVIDE)::value;
      constexpr int base = 9 * hf + SLAB_DYH * s;
#pragma unroll
      for (int tl = 0; tl < TQH; ++tl) {
        constexpr int dlo = base, dhi = (base + SLAB_DYH - 1) < 2 * R ? (base + SLAB_DYH - 1) : 2 * R;
        const int tqy = TQH * hf + tl;
        const int lo = dlo - 4 * tqy, hi = dhi - 4 * tqy;
        if (hi < -3 || lo > 3) continue;
        const bool my = u >= lo && u <= hi;
#pragma unroll
        for (int tqx = 0; tqx < TQ; ++tqx) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool ok = my && (tqx == 0 ? mx_first[j] : (tqx == TQ - 1 ? mx_last[j] : true));
            if (ok) {
              const float v = acc[tl * TQ + tqx][j];
              ob[L[j] + (4 * tqy - base) * GW + 4 * tqx] = divide ? v / (float)C : v * scale;
            }
          }
        }
      }
    };
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (pow2) {
        if (s == 0) scatter(Int<0>{}, std::false_type{});
        else scatter(Int<1>{}, std::false_type{});
      } else {
        if (s == 0) scatter(Int<0>{}, std::true_type{});
        else scatter(Int<1>{}, std::true_type{});
      }
      __syncthreads();
      const int base = 9 * hf + SLAB_DYH * s;
      for (int idx = threadIdx.x; idx < SLABH / 4; idx += NTHREADS) {
        const int c4 = idx % (RW / 4);
        int rest = idx / (RW / 4);
        const int r = rest % RY;
        rest /= RY;
        const int dxx = rest % GW, ddl = rest / GW;
        const int dyp = base + ddl;
        // this half owns (row, dy') iff the window row (r & 3) + dy' lies in its three tile rows
        if (dyp > 2 * R || (((r & 3) + dyp >= 4 * TQH) != (hf == 1))) continue;
        const int d = dyp * GW + dxx;
        const int y = S * (ys0 + r) + cy;
        const int x = S * xs0 + c4 * 4;
        if (y >= H || x >= W) continue;
#ifdef UNFLOW_TUNE_FWD_NOSTORE
        if (C > 0 && idx > 0) continue;
#endif
        const float *src = ob + (r * RW + c4 * 4) * (SLAB_DYH * GW) + ddl * GW + dxx;
        const float4 v = make_float4(src[0], src[SLAB_DYH * GW], src[2 * SLAB_DYH * GW], src[3 * SLAB_DYH * GW]);
        auto *dst = out + (item * D + d) * plane + (long long)y * W + x;
        if constexpr (F16) {   // W % 4 == 0 and an 8-byte aligned base are preconditions of this variant
          *reinterpret_cast<f16x4 *>(dst) = f16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        } else if (x + 3 < W && (W & 3) == 0) {
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
  };
  if (half == 0) epilogue(Int<0>{});
  else epilogue(Int<1>{});
  UNFLOW_STAMP(STAMP_ROWS, 2);   // epilogue done
}

// ===========================================================================
// Backward.  Both gradients are the same GEMM with the roles swapped:
//   g0[c, o] = 1/C * sum_{w in window(o)} top[d = w - o, o] * f1[c, w]
//   g1[c, o] = 1/C * sum_{w in window(o)} top[d = o - w, w] * f0[c, w]
// (ops/correlation_op.cu.cc:119-181 and :183-248 of the reference, restricted to
// kernel_size 1 / stride_1 1, where the output-range tests reduce to "the
// correlation position lies inside the image").  Per 4x4 tile of class positions
// o:  G[16 channels x 16 o] += A[16 channels x 4 w] * B[4 w x 16 o]  with
// A = the other input's window (from LDS) and B = top gathered by displacement.
//
// Row sweep.  A workgroup owns a strip of 8 class columns (16 full-resolution
// columns, both column classes) of one (pair, row class, gradient, 128-channel
// group) and walks the class rows w = 0 .. hs-1 of the OTHER input top to bottom.
// Row w is staged once (LDS-DMA, double buffered): A = 128 channels x 56 floats,
// plus the top segments that pair it with every output row within +-10 rows
// ([21 rows][21 dx][16 columns]).  It contributes to the <= 6 tiles whose rows
// lie within +-10 of w; those tiles' accumulators stay in registers -- a ring
// of 3 tiles per wave (8 waves = {cx} x {2 column tiles} x {tile parity}), 8
// M-tiles of 16 channels each, 96 accumulator VGPRs -- and a finished tile is
// regrouped through LDS and stored as 64-byte row segments.  Compared with
// staging the whole window per 8-row region this moves 3.5x less data into LDS,
// never touches window rows outside the image, and reads each A fragment once
// for up to three tiles.
// ===========================================================================
// (MTS .. RING: corr_mfma_common.h)

// First-generation row sweep: kept for fp16 storage, C % 128 != 0 and rows that are not 16-byte
// aligned (the float32 FlowNetC shape runs corr_mfma_bwd4_kernel below).
template <int VA, int VT, bool WRT1, bool FULL, bool F16 = false>
__global__ __launch_bounds__(NTHREADS, 2) void corr_mfma_bwd_kernel(
    const std::conditional_t<F16, _Float16, float> *__restrict__ topdiff,
    const std::conditional_t<F16, _Float16, float> *__restrict__ fin,
    std::conditional_t<F16, _Float16, float> *__restrict__ gout, int C, int H, int W, int nbx, int nch) {
  // F16 (extension): half tensors in HBM.  The row is fetched into registers (8-byte loads of
  // 4 halves, same piece layout as the DMA) while the previous row computes, and widened to
  // f32 into the LDS stage before the barrier; the MFMA loop is the f32 one.
  static_assert(!F16 || (VA == 4 && VT == 4), "fp16 storage uses the 4-element pieces");
  constexpr int ES = F16 ? 2 : 4;              // bytes per element in HBM
  // WRT1 = false: gradient wrt input 0 (fin = input 1); true: wrt input 1 (fin = input 0).
  // FULL: every channel group holds 128 channels (C % 128 == 0), so no MFMA needs a
  // guard (a guard per MFMA costs three scalar instructions, 432 per window row).
  // VT: DMA width of the top segments.  For the gradient wrt input 0 they are
  // whole, aligned 64-byte segments.  Wrt input 1 they are shifted by the
  // displacement: 8-byte aligned, and a 16-byte piece may straddle the end of an
  // image row; the floats it then picks up from the neighbouring row belong to
  // window positions outside the image and are masked when the fragment is read
  // (b_dxok also tests the window column), so they are never used.
  constexpr int A_PIECE = 64 * VA, T_PIECE = 64 * VT;
  static_assert(SA_STAGE % A_PIECE == 0 && ST_STAGE % T_PIECE == 0, "stage must split into wave pieces");
  constexpr int A_PIECES = SA_STAGE / A_PIECE, T_PIECES = ST_STAGE / T_PIECE;
  constexpr int A_PER_WAVE = (A_PIECES + NWAVES - 1) / NWAVES;
  constexpr int T_PER_WAVE = (T_PIECES + NWAVES - 1) / NWAVES;
  // the DMA of the next row is issued during the first DSL K-steps of this row,
  // so that it has the remaining K-steps to land before the barrier
  constexpr int DSL = (VA == 4 && VT == 4) ? 3 : KSTEPS;
  constexpr int APS = (A_PER_WAVE + DSL - 1) / DSL;   // A pieces issued per K-step
  constexpr int TPS = (T_PER_WAVE + DSL - 1) / DSL;   // top pieces issued per K-step

  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_assume(wave >= 0 && wave < NWAVES);
  const int cx = wave & 1, tx = (wave >> 1) & 1, ph = wave >> 2;
  int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = vb % nbx;
  vb /= nbx;
  const int chg = vb % nch;
  vb /= nch;
  const int cy = vb & 1;
  const long long item = vb >> 1;
  const int xs0 = bx * RX;
  const int hs = (H - cy + S - 1) / S;         // class rows of this row class
  if (hs <= 0) return;
  const int ntiles = (hs + TP - 1) / TP;
  const int c0 = chg * CWS;
  const int nmt = min(MTS, (C - c0 + 15) / 16);
  const long long plane = (long long)H * W;
  const __amdgpu_buffer_rsrc_t rtop = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<std::conditional_t<F16, _Float16, float> *>(topdiff + item * (GW * GW) * plane), (short)0,
      (int)(GW * GW * plane * ES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<std::conditional_t<F16, _Float16, float> *>(fin + (item * C + c0) * plane), (short)0,
      (int)((long long)(C - c0) * plane * ES), 0x00020000);
  const int row_bytes = S * W * ES;            // one class row further down

  const int k = lane >> 4, n = lane & 15;
  const int oyl = n >> 2, oxl = n & 3;
  const int b_col = S * (tx * TP + oxl) + cx;  // column of this lane's output position in the region
  const int a_off = n * ACS + S * (tx * TP + k) + cx;     // + mt*16*ACS + 8*s

  // Everything that does not depend on the sweep row is computed once:
  //  * A pieces: channel plane + column (or OOB); the row is the SGPR offset
  //  * top pieces: offset at row 0 (linear in w), the top-image row they fill
  //  * B fragments: dx' of each K-step as an LDS offset plus its validity
  unsigned a_voff[A_PER_WAVE];
#pragma unroll
  for (int j = 0; j < A_PER_WAVE; ++j) {
    const int f = ((wave + NWAVES * j) * 64 + lane) * VA;
    const int c = f / ACS, x = f - c * ACS;
    const int X = S * (xs0 - R) + x;
    a_voff[j] = (f < SA_STAGE && x < ROWF && X >= 0 && X < W && c0 + c < C) ? (unsigned)((c * plane + X) * ES) : OOB;
  }
  int t_base[T_PER_WAVE];      // byte offset of the segment at sweep row 0
  int t_orow[T_PER_WAVE];      // row of the top image (output row = w - R + orow); huge = never valid
#pragma unroll
  for (int j = 0; j < T_PER_WAVE; ++j) {
    const int f = ((wave + NWAVES * j) * 64 + lane) * VT;
    const int orow = f / (GW * RW), rem = f - orow * (GW * RW);
    const int dxp = rem / RW, col = rem - dxp * RW;
    int dyp, Y0, Xt;
    if (WRT1) {   // position = window position: row w, column shifted by -dx
      dyp = orow;                              // oy - w + R
      Y0 = cy;
      Xt = S * xs0 - S * (dxp - R) + col;
    } else {      // position = output position: row oy = w - R + orow
      dyp = 2 * R - orow;                      // w - oy + R
      Y0 = S * (orow - R) + cy;
      Xt = S * xs0 + col;
    }
    const bool ok = wave + NWAVES * j < T_PIECES && f < ST_IMAGE && Xt + VT - 1 >= 0 && Xt < W;
    t_base[j] = (int)((((long long)(dyp * GW + dxp) * H + Y0) * W + Xt) * ES);
    t_orow[j] = ok ? orow : (1 << 28);
  }
  int b_dx[KSTEPS];            // LDS offset of dx' for K-step s (clamped), validity in b_dxok
  unsigned b_dxok = 0;
#pragma unroll
  for (int q = 0; q < KSTEPS; ++q) {
    const int dxp = WRT1 ? (oxl + 2 * R - k - 4 * q) : (k - oxl + 4 * q);
    const int Xw = S * (xs0 + tx * TP - R + 4 * q + k) + cx;   // full-resolution column of the window position
    if (dxp >= 0 && dxp <= 2 * R && (!WRT1 || (Xw >= 0 && Xw < W))) b_dxok |= 1u << q;
    b_dx[q] = min(max(dxp, 0), 2 * R) * RW + b_col + SA_STAGE;
  }

  f32x4 acc[RING][MTS];
#pragma unroll
  for (int sl = 0; sl < RING; ++sl)
#pragma unroll
    for (int m = 0; m < MTS; ++m) acc[sl][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // DMA of window row w into `stage`, in KSTEPS slices so that it can be
  // interleaved with the K-steps of the previous row.
  auto issue_slice = [&](auto SL, int w, float *stage) {
    constexpr int sl = decltype(SL)::value;
    const int soff = (S * w + cy) * W * ES;    // row w is always inside the image
#pragma unroll
    for (int j = sl * APS; j < (sl + 1) * APS; ++j) {
      if (j < A_PER_WAVE && wave + NWAVES * j < A_PIECES)
        dma<VA>(rin, stage + (wave + NWAVES * j) * A_PIECE, a_voff[j < A_PER_WAVE ? j : 0], soff);
    }
#pragma unroll
    for (int j = sl * TPS; j < (sl + 1) * TPS; ++j) {
      if (j < T_PER_WAVE && wave + NWAVES * j < T_PIECES) {
        const int jj = j < T_PER_WAVE ? j : 0;
        const int oy = w - R + t_orow[jj];
        const unsigned voff = (oy >= 0 && oy < hs) ? (unsigned)(t_base[jj] + w * row_bytes) : OOB;
        dma<VT>(rtop, stage + SA_STAGE + (wave + NWAVES * j) * T_PIECE, voff, 0);
      }
    }
  };
  auto issue_row = [&](int w, float *stage) {
    issue_slice(Int<0>{}, w, stage); issue_slice(Int<1>{}, w, stage); issue_slice(Int<2>{}, w, stage);
    issue_slice(Int<3>{}, w, stage); issue_slice(Int<4>{}, w, stage); issue_slice(Int<5>{}, w, stage);
  };

  // fp16 storage: the same pieces through registers
  i32x2 ra[F16 ? A_PER_WAVE : 1], rt[F16 ? T_PER_WAVE : 1];
  auto load_row = [&](int w) {
    if constexpr (F16) {
      const int soff = (S * w + cy) * W * ES;
#pragma unroll
      for (int j = 0; j < A_PER_WAVE; ++j)
        ra[j] = (wave + NWAVES * j < A_PIECES) ? __builtin_amdgcn_raw_buffer_load_b64(rin, (int)a_voff[j], soff, 0) : i32x2{0, 0};
#pragma unroll
      for (int j = 0; j < T_PER_WAVE; ++j) {
        const int oy = w - R + t_orow[j];
        const unsigned voff = (oy >= 0 && oy < hs) ? (unsigned)(t_base[j] + w * row_bytes) : OOB;
        rt[j] = (wave + NWAVES * j < T_PIECES) ? __builtin_amdgcn_raw_buffer_load_b64(rtop, (int)voff, 0, 0) : i32x2{0, 0};
      }
    }
  };
  auto store_row = [&](float *stage) {
    if constexpr (F16) {
      auto put = [&](float *dst, i32x2 raw) {
        const f16x4 h = __builtin_bit_cast(f16x4, raw);
        *reinterpret_cast<f32x4 *>(dst) = f32x4{(float)h.x, (float)h.y, (float)h.z, (float)h.w};
      };
#pragma unroll
      for (int j = 0; j < A_PER_WAVE; ++j)
        if (wave + NWAVES * j < A_PIECES) put(stage + (wave + NWAVES * j) * A_PIECE + lane * VA, ra[j]);
#pragma unroll
      for (int j = 0; j < T_PER_WAVE; ++j)
        if (wave + NWAVES * j < T_PIECES) put(stage + SA_STAGE + (wave + NWAVES * j) * T_PIECE + lane * VT, rt[j]);
    }
  };

  // Tile t = 2u + ph of this wave's parity lives in ring slot u % 3 while
  // 4t - 10 <= w <= 4t + 13.
  auto tile_in_slot = [&](int slot, int w) -> int {
    const int u_hi = (w - 4 * ph + R) / 8;     // >= 0 because w >= 0 and ph <= 1
    const int back = ((u_hi - slot) % RING + RING) % RING;
    const int u = u_hi - back;
    const int t = 2 * u + ph;
    return (u >= 0 && t < ntiles && w <= 4 * t + TP - 1 + R) ? t : -1;
  };

  if constexpr (F16) {
    load_row(0);
    store_row(lds);
  } else {
    issue_row(0, lds);
  }
  dma_barrier();
  for (int w = 0; w < hs; ++w) {
    const float *cur = lds + (w & 1) * SW_STAGE;
    float *nxt = lds + ((w + 1) & 1) * SW_STAGE;
    const bool more = w + 1 < hs;
    if (F16 && more) load_row(w + 1);
    // per ring slot: live tile, LDS row offset of the top image row that pairs
    // (w, this lane's output row), validity
    int tile[RING], b_row[RING];
    bool b_rowok[RING];
#pragma unroll
    for (int sl = 0; sl < RING; ++sl) {
      tile[sl] = tile_in_slot(sl, w);
      const int orow = 4 * tile[sl] + oyl - w + R;
      b_rowok[sl] = tile[sl] >= 0 && orow >= 0 && orow <= 2 * R;
      b_row[sl] = min(max(orow, 0), 2 * R) * (GW * RW);
    }

    // A and B fragments are read one K-step ahead (two register sets), so the
    // LDS latency of step s+1 hides behind the MFMAs of step s.
    float abuf[2][MTS], bbuf[2][RING];
    auto load_ab = [&](auto S4) {
      constexpr int s4 = decltype(S4)::value;
      const float *ap = cur + a_off + 4 * S * s4;
#pragma unroll
      for (int m = 0; m < MTS; ++m) abuf[s4 & 1][m] = ap[m * 16 * ACS];
#pragma unroll
      for (int sl = 0; sl < RING; ++sl) bbuf[s4 & 1][sl] = cur[b_row[sl] + b_dx[s4]];   // masked at use
    };
    auto kstep = [&](auto S4) {
      constexpr int s4 = decltype(S4)::value;
      if constexpr (s4 + 1 < KSTEPS) load_ab(Int<s4 + 1>{});
      const bool dxok = b_dxok & (1u << s4);
#pragma unroll
      for (int sl = 0; sl < RING; ++sl) {
        if (tile[sl] >= 0) {                       // wave-uniform
          const float b = (dxok && b_rowok[sl]) ? bbuf[s4 & 1][sl] : 0.f;
#pragma unroll
          for (int m = 0; m < MTS; ++m)
            if (FULL || m < nmt)
              acc[sl][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[s4 & 1][m], b, acc[sl][m], 0, 0, 0);
        }
      }
      if (!F16 && more) issue_slice(S4, w + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
    };
    load_ab(Int<0>{});
    kstep(Int<0>{}); kstep(Int<1>{}); kstep(Int<2>{}); kstep(Int<3>{}); kstep(Int<4>{}); kstep(Int<5>{});
    if (F16 && more) store_row(nxt);
    dma_barrier();

    // ---- tiles whose last window row was w: regroup through LDS to
    // [channel][row 4][16 columns] and store 64-byte row segments.
    const int t_lo = max(0, (w - R) / TP);         // first tile that can still be live
    for (int t = t_lo; t < ntiles && 4 * t - R <= w; ++t) {
      const int fin_row = min(hs - 1, 4 * t + TP - 1 + R);
      if (fin_row != w) continue;                  // uniform over the workgroup
      float *ob = lds + 2 * SW_STAGE;
      const float inv = 1.f / (float)C;
      const bool pow2 = (C & (C - 1)) == 0;
      if ((t & 1) == ph) {
        const int slot = (t >> 1) % RING;
#pragma unroll
        for (int sl = 0; sl < RING; ++sl) {
          if (sl == slot) {
#pragma unroll
            for (int m = 0; m < MTS; ++m) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int cl = m * 16 + 4 * k + j;
                const float v = acc[sl][m][j];
                ob[(cl * TP + oyl) * RW + b_col] = v;      // scaled by 1/C when the image is stored
                acc[sl][m][j] = 0.f;
              }
            }
          }
        }
      }
      __syncthreads();
      for (int idx = threadIdx.x; idx < SW_OUT / 4; idx += NTHREADS) {
        const int c4 = idx % (RW / 4);
        const int rr = (idx / (RW / 4)) % TP;
        const int cl = idx / (RW / 4 * TP);
        const int c = c0 + cl;
        const int oy = 4 * t + rr;
        const int y = S * oy + cy, x = S * xs0 + c4 * 4;
        if (c >= C || oy >= hs || x >= W) continue;
        float4 v = *reinterpret_cast<const float4 *>(ob + idx * 4);
        if (pow2) {
          v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        } else {
          v.x /= (float)C; v.y /= (float)C; v.z /= (float)C; v.w /= (float)C;
        }
        auto *dst = gout + (item * C + c) * plane + (long long)y * W + x;
        if constexpr (F16) {   // W % 4 == 0 and 8-byte aligned tensors are preconditions of this variant
          *reinterpret_cast<f16x4 *>(dst) = f16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        } else if (VA == 4 && x + 3 < W && (W & 3) == 0) {   // VA == 4: the host checked 16-byte alignment of gout
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

#ifdef UNFLOW_TUNE_STAMPS
extern "C" int unflow_debug_read_fwd_stamps(void *dst, size_t bytes) {
  const size_t n = sizeof(g_stamps) < bytes ? sizeof(g_stamps) : bytes;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), n, 0, hipMemcpyDeviceToHost);
}
#endif

bool corr_mfma_supported(const CorrGeometry &g, int C) {
  // buffer descriptors address one image pair with 32-bit offsets
  return C > 0 && g.kernel_size == 1 && g.stride_1 == 1 && g.stride_2 == S && g.max_displacement == S * R &&
         g.pad == g.max_displacement && (long long)C * g.H * g.W * 4 < (1LL << 31);
}

bool corr_mfma_bwd_supported(const CorrGeometry &g, int C) {
  return corr_mfma_supported(g, C) && (long long)(GW * GW) * g.H * g.W * 4 < (1LL << 31);
}

int corr_mfma_fwd(const float *in0, const float *in1, float *out, int B, int C, const CorrGeometry &g,
                  hipStream_t s) {
#ifndef UNFLOW_TUNE_OLD_FWD
  if (corr_fwd2_supported(g, C, in0, in1, out)) return corr_fwd2(in0, in1, out, B, C, g, s);
#endif
  const int H = g.H, W = g.W;
  const int hs = (H + S - 1) / S, ws = (W + S - 1) / S;
  const int nbx = (ws + RX - 1) / RX, nby = (hs + RY - 1) / RY;
  const dim3 grid(nbx * nby * B * S);
  const size_t smem = sizeof(float) * 2 * STAGE;
  // 16-byte DMA segments need 16-byte aligned rows
  const bool wide = (W % 4) == 0 && aligned16(in0) && aligned16(in1) && aligned16(out);   // out is stored as float4
  if (wide) {   // two half-window workgroups per region, two workgroups per CU: 0.76 -> 0.66 ms at B=32
    const size_t smem2 = sizeof(float) * 2 * STAGEH;
    if (int rc = reserve_lds<corr_mfma_fwd_half_kernel<4, false>>(smem2, "correlation_fwd (mfma, half windows)")) return rc;
    hipLaunchKernelGGL((corr_mfma_fwd_half_kernel<4, false>), dim3(grid.x * 2), dim3(NTHREADS), smem2, s, in0, in1, out, C, H, W,
                       nbx, nby);
    return launch_status("correlation_fwd (mfma, half windows)");
  }
  // rows that are not 16-byte aligned: 4-byte DMA, full-window workgroups
  if (int rc = reserve_lds<corr_mfma_fwd_kernel<1>>(smem, "correlation_fwd (mfma)")) return rc;
  hipLaunchKernelGGL(corr_mfma_fwd_kernel<1>, grid, dim3(NTHREADS), smem, s, in0, in1, out, C, H, W, nbx, nby);
  return launch_status("correlation_fwd (mfma)");
}

int corr_mfma_fwd_f16(const void *in0, const void *in1, void *out, int B, int C, const CorrGeometry &g,
                      hipStream_t s) {
  const int H = g.H, W = g.W;
  const int hs = (H + S - 1) / S, ws = (W + S - 1) / S;
  const int nbx = (ws + RX - 1) / RX, nby = (hs + RY - 1) / RY;
  auto al8 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
  if ((W % 4) != 0 || !al8(in0) || !al8(in1) || !al8(out))
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "correlation (fp16 storage): W must be a multiple of 4 and the tensors 8-byte aligned");
  const size_t smem = sizeof(float) * 2 * STAGEH;
  if (int rc = reserve_lds<corr_mfma_fwd_half_kernel<4, true>>(smem, "correlation_fwd (fp16 storage)")) return rc;
  hipLaunchKernelGGL((corr_mfma_fwd_half_kernel<4, true>), dim3(nbx * nby * B * S * 2), dim3(NTHREADS), smem, s,
                     static_cast<const _Float16 *>(in0), static_cast<const _Float16 *>(in1), static_cast<_Float16 *>(out), C,
                     H, W, nbx, nby);
  return launch_status("correlation_fwd (fp16 storage)");
}

int corr_mfma_bwd_r1(const float *topdiff, const float *in0, const float *in1, float *g0, float *g1, int B, int C,
                     const CorrGeometry &g, hipStream_t s) {
  const int H = g.H, W = g.W;
  const int ws = (W + S - 1) / S;
  const int nbx = (ws + RX - 1) / RX, nch = (C + CWS - 1) / CWS;
  // one workgroup per (pair, row class, channel group, column strip); one launch per gradient
  const dim3 grid(nbx * nch * S * B);
  const size_t smem = sizeof(float) * SW_LDS;
  const bool wide = (W % 4) == 0 && aligned16(in0) && aligned16(in1) && aligned16(topdiff) && aligned16(g0) && aligned16(g1);
#define UNFLOW_LAUNCH(VA, VT0, FULL)                                                                               \
  do {                                                                                                              \
    if (int rc = reserve_lds<corr_mfma_bwd_kernel<VA, VT0, false, FULL>>(smem, "correlation_bwd (mfma)")) return rc; \
    if (int rc = reserve_lds<corr_mfma_bwd_kernel<VA, VT0, true, FULL>>(smem, "correlation_bwd (mfma)")) return rc;  \
    hipLaunchKernelGGL((corr_mfma_bwd_kernel<VA, VT0, false, FULL>), grid, dim3(NTHREADS), smem, s, topdiff, in1, g0, \
                       C, H, W, nbx, nch);                                                                          \
    hipLaunchKernelGGL((corr_mfma_bwd_kernel<VA, VT0, true, FULL>), grid, dim3(NTHREADS), smem, s, topdiff, in0, g1,  \
                       C, H, W, nbx, nch);                                                                          \
  } while (0)
  const bool full = (C % CWS) == 0;
  if (wide && full) UNFLOW_LAUNCH(4, 4, true);
  else if (wide) UNFLOW_LAUNCH(4, 4, false);
  else UNFLOW_LAUNCH(1, 1, false);
#undef UNFLOW_LAUNCH
  return launch_status("correlation_bwd (mfma)");
}

int corr_mfma_bwd_f16(const void *topdiff, const void *in0, const void *in1, void *g0, void *g1, int B, int C,
                      const CorrGeometry &g, hipStream_t s) {
  const int H = g.H, W = g.W;
  const int ws = (W + S - 1) / S;
  const int nbx = (ws + RX - 1) / RX, nch = (C + CWS - 1) / CWS;
  auto al8 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
  if ((W % 4) != 0 || !al8(topdiff) || !al8(in0) || !al8(in1) || !al8(g0) || !al8(g1))
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "correlation_grad (fp16 storage): W must be a multiple of 4 and the tensors 8-byte aligned");
  const dim3 grid(nbx * nch * S * B);
  const size_t smem = sizeof(float) * SW_LDS;
  using H16 = _Float16;
#define UNFLOW_LAUNCH16(FULL)                                                                                       \
  do {                                                                                                              \
    if (int rc = reserve_lds<corr_mfma_bwd_kernel<4, 4, false, FULL, true>>(smem, "correlation_bwd (fp16 storage)")) return rc; \
    if (int rc = reserve_lds<corr_mfma_bwd_kernel<4, 4, true, FULL, true>>(smem, "correlation_bwd (fp16 storage)")) return rc;  \
    hipLaunchKernelGGL((corr_mfma_bwd_kernel<4, 4, false, FULL, true>), grid, dim3(NTHREADS), smem, s,               \
                       static_cast<const H16 *>(topdiff), static_cast<const H16 *>(in1), static_cast<H16 *>(g0), C, H, W, nbx, nch); \
    hipLaunchKernelGGL((corr_mfma_bwd_kernel<4, 4, true, FULL, true>), grid, dim3(NTHREADS), smem, s,                \
                       static_cast<const H16 *>(topdiff), static_cast<const H16 *>(in0), static_cast<H16 *>(g1), C, H, W, nbx, nch); \
  } while (0)
  if ((C % CWS) == 0) UNFLOW_LAUNCH16(true);
  else UNFLOW_LAUNCH16(false);
#undef UNFLOW_LAUNCH16
  return launch_status("correlation_bwd (fp16 storage)");
}

const char *tune_flags_correlation_mfma() {
  return ""
#ifdef UNFLOW_TUNE_STAMPS
         " UNFLOW_TUNE_STAMPS"
#endif
#ifdef UNFLOW_TUNE_FWD_NODMA
         " UNFLOW_TUNE_FWD_NODMA"
#endif
#ifdef UNFLOW_TUNE_FWD_NOSTORE
         " UNFLOW_TUNE_FWD_NOSTORE"
#endif
#ifdef UNFLOW_TUNE_OLD_FWD
         " UNFLOW_TUNE_OLD_FWD"
#endif
      ;
}

}  // namespace unflow
