// FlowNetC correlation on the f16 MATRIX CORES of gfx950 (extension: BASELINE config 5,
// "fp16 with fp32 accumulate"; the reference is float32 only).
//
// Reference semantics: ops/correlation_op.cu.cc:51-117 (forward), :119-248 (backward) of the
// reference checkout, FlowNetC attributes (core/flownet.py:221-222).  Operands and results are
// IEEE binary16 in HBM; every product of two halves is exact in f32 and the sums run in f32 inside
// v_mfma_f32_16x16x16_f16, so against the fp32 oracle on the same half-rounded inputs the only
// differences are the summation order and the one rounding of the result (rtol 2e-3).
//
// Same dense-GEMM formulation as correlation_mfma.hip (four parity classes, 4x4 position tiles,
// 24x24 windows), but a 16-bit MFMA wants K CONTIGUOUS in a lane's registers (4 halves per lane
// per operand), which an NCHW image does not give for K = channel:
//   * forward (K = channels): the LDS images stay channel-major ([channel][row][column], the NCHW
//     order) and the fragments are read with ds_read_b64_tr_b16, the transposing LDS read of gfx950 --
//     a 16-lane group reads 4 channels x (4 rows x 4 columns) and every lane receives the 4
//     channels of its position.  What it needs is 4 CONTIGUOUS columns of one class, so the two
//     column classes are de-interleaved on the way into LDS (one v_perm_b32 per class and 8-byte
//     load: the operands go through registers, LDS-DMA cannot permute).
//   * backward (K = window positions): the other input's rows are K-contiguous as they are (plain
//     ds_read_b64 after the same de-interleave).  The upstream gradient is not: B[q][o] =
//     top[o, q - o] runs along the DISPLACEMENT for a fixed position, a stride of H*W in HBM.  It is
//     transposed while it is staged: every staged value is written to its (position, k) slot of a
//     k-group-major image with one ds_write_b16.
// At 16x the f32 matrix rate the MFMAs are a few percent of the time: both kernels are bound by
// their staging (L1 -> registers -> LDS), so the layouts are chosen for conflict-free LDS traffic.
#include "common.h"
#include "correlation.h"

#include <atomic>
#include <cstdint>
#include <type_traits>

namespace unflow {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 p16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr int S = 2;            // stride_2
constexpr int R = 10;           // grid radius
constexpr int GW = 2 * R + 1;   // 21
constexpr int TP = 4;           // 4x4 position tiles (MFMA M or N = 16)
constexpr int WIN = TP + 2 * R; // 24
constexpr int TQ = WIN / 4;     // 6 N-tiles per window dimension
constexpr int NTHREADS = 512, NWAVES = 8;
constexpr unsigned OOB = 0x80000000u;   // beyond any num_records: reads as 0
constexpr int KC = 16;          // channels per stage = K of v_mfma_f32_16x16x16_f16

template <int N>
using Int = std::integral_constant<int, N>;

__device__ inline int xcd_remap(int b, int nblk) { return (nblk & 7) == 0 ? (b & 7) * (nblk >> 3) + (b >> 3) : b; }

__device__ inline __amdgpu_buffer_rsrc_t make_rsrc16(const _Float16 *base, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(base), (short)0, (int)bytes, 0x00020000);
}

// ds_read_b64_tr_b16: lane 4q+p of a 16-lane group passes the address of 4 contiguous halves of
// block row q (here: channel), column group p; lane i receives column i of the four rows.
__device__ inline f16x4 tr_read(const _Float16 *p) {
  return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) p16x4 *)(p)));
}

// An 8-byte piece holds 4 consecutive full-resolution columns X .. X+3 (X even): classes 0 1 0 1.
__device__ inline int class0_of(i32x2 d) { return (int)__builtin_amdgcn_perm((unsigned)d.y, (unsigned)d.x, 0x05040100u); }
__device__ inline int class1_of(i32x2 d) { return (int)__builtin_amdgcn_perm((unsigned)d.y, (unsigned)d.x, 0x07060302u); }

// ===========================================================================
// Forward.  Workgroup = 8 waves = {column class} x 2x2 position tiles of one row class and HALF of
// the window rows (N-tile rows 3h .. 3h+2), exactly the decomposition of corr_mfma_fwd_half_kernel
// (two workgroups share a CU; the halves of a region write disjoint output segments; same
// accumulator layout, same epilogue).  A stage holds 16 channels:
//   f1 image, per column class: [16 channels][16 window rows][28 columns] halves,
//   f0 image, per column class: [16 channels][8 rows][8 columns (+4 pad)].
// Transposed fragment read of a 32-lane half: 8 channels x 4 tile rows, 8 bytes each; the slot
// (8-byte unit mod 32) of lane (channel c, row p) is c*PLANE + p*ROW: with ROW = 7 (f1) or 3 (f0)
// slots, odd, and PLANE = 4*odd slots the 32 lanes take the 32 slots once: conflict free.
// ===========================================================================
constexpr int RY = 8, RX = 8;                  // region of one class in a workgroup
constexpr int LW = RX + 2 * R;                 // 28 staged window columns per class
constexpr int RW = S * RX;                     // 16 full-resolution columns of the region
constexpr int TQH = TQ / 2;                    // 3 N-tile rows per workgroup
constexpr int NTH = TQH * TQ;                  // 18 accumulator tiles per wave
constexpr int LHH = RY + 4 * TQH - 4;          // 16 staged window rows
constexpr int F1_ROW = LW;                     // 28 halves = 56 B = 7 slots
constexpr int F1_PL = LHH * F1_ROW + 16;       // 464 halves = 928 B = 116 slots = 4 * 29
constexpr int F1_CLS = KC * F1_PL;             // 7424 halves
constexpr int F0_ROW = RX + 4;                 // 12 halves = 24 B = 3 slots
constexpr int F0_PL = RY * F0_ROW + 16;        // 112 halves = 224 B = 28 slots = 4 * 7
constexpr int F0_CLS = KC * F0_PL;             // 1792 halves
constexpr int F0_BASE = 2 * F1_CLS;            // [f1 class 0][f1 class 1][f0 class 0][f0 class 1]
constexpr int STAGE_H = 2 * (F1_CLS + F0_CLS); // 18432 halves = 36 KB
constexpr int SLAB_DYH = 6;                    // displacement rows per epilogue slab
constexpr int SLABH = SLAB_DYH * GW * RY * RW; // 16128 floats
constexpr int FWD_LDS_BYTES = 2 * STAGE_H * 2; // 73728 B: two workgroups per CU
static_assert(SLABH * 4 <= FWD_LDS_BYTES, "epilogue slab must fit the staging buffers");
static_assert((F1_ROW * 2 / 8) % 2 == 1 && (F1_PL * 2 / 8) % 8 == 4, "f1 image would bank-conflict");
static_assert((F0_ROW * 2 / 8) % 2 == 1 && (F0_PL * 2 / 8) % 8 == 4, "f0 image would bank-conflict");
static_assert(F1_ROW % 4 == 0 && F1_PL % 4 == 0 && F0_ROW % 4 == 0 && F0_PL % 4 == 0, "transposed reads need 8-byte aligned rows");

struct FwdStager16 {
  // wave w stages channels w and w + 8 of the chunk: per channel 16 rows x 14 pieces of 4 full-resolution
  // columns (one 8-byte load each), de-interleaved into the two class images; the f0 image is 512 pieces.
  static constexpr int ROW_PIECES = S * LW / 4;                  // 14
  static constexpr int F1_PIECES = LHH * ROW_PIECES;             // 224
  static constexpr int F1_ROUNDS = (F1_PIECES + 63) / 64;        // 4 (the last one half)
  unsigned voff1[F1_ROUNDS], voff0;
  i32x2 r1[2][F1_ROUNDS], r0v;
  int wave, lane;
  long long plane;
  __amdgpu_buffer_rsrc_t rs0, rs1;

  __device__ inline void init(int H, int W, int cy, int ys0, int xs0, int wrow0) {
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __builtin_assume(wave >= 0 && wave < NWAVES);
    lane = threadIdx.x & 63;
    plane = (long long)H * W;
#pragma unroll
    for (int j = 0; j < F1_ROUNDS; ++j) {
      const int f = j * 64 + lane;
      const int row = f / ROW_PIECES, pc = f - row * ROW_PIECES;
      const int Y = S * (ys0 - R + wrow0 + row) + cy, X = S * (xs0 - R) + 4 * pc;
      const bool in = f < F1_PIECES && Y >= 0 && Y < H && X >= 0 && X < W;   // X, W multiples of 4: all or nothing
      voff1[j] = in ? (unsigned)((wave * plane + (long long)Y * W + X) * 2) : OOB;
    }
    const int t = (int)threadIdx.x;                                          // f0: [16 channels][8 rows][4 pieces]
    const int c = t >> 5, row = (t >> 2) & 7, pc = t & 3;
    const int Y = S * (ys0 + row) + cy, X = S * xs0 + 4 * pc;
    voff0 = (Y < H && X < W) ? (unsigned)((c * plane + (long long)Y * W + X) * 2) : OOB;
  }
  __device__ inline void bind(const _Float16 *in0, const _Float16 *in1, long long item, int c0, int C) {
    const long long bytes = (long long)(C - c0) * plane * 2;
    rs0 = make_rsrc16(in0 + (item * C + c0) * plane, bytes);
    rs1 = make_rsrc16(in1 + (item * C + c0) * plane, bytes);
  }
  __device__ inline void load() {
    // channel wave + 8: in the VECTOR offset -- the range check that zero-fills the channels beyond the last
    // one looks at the vector offset only, not at the scalar one (OOB + this stays out of range: C*H*W*2 < 2^31)
    const unsigned second = (unsigned)(8 * plane * 2);
#pragma unroll
    for (int j = 0; j < F1_ROUNDS; ++j) {
      r1[0][j] = __builtin_amdgcn_raw_buffer_load_b64(rs1, (int)voff1[j], 0, 0);
      r1[1][j] = __builtin_amdgcn_raw_buffer_load_b64(rs1, (int)(voff1[j] + second), 0, 0);
    }
    r0v = __builtin_amdgcn_raw_buffer_load_b64(rs0, (int)voff0, 0, 0);
  }
  __device__ inline void store(_Float16 *stage) const {
    // piece f = 64 j + lane of a channel plane sits at halves 2 f of that plane (14 pieces x 2 halves = the
    // 28-half row): lane-linear, one base address and immediate offsets
    _Float16 *base1 = stage + wave * F1_PL + 2 * lane;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int j = 0; j < F1_ROUNDS; ++j) {
        if (j * 64 + 63 < F1_PIECES || j * 64 + lane < F1_PIECES) {
          _Float16 *dst = base1 + 8 * h * F1_PL + 128 * j;
          *reinterpret_cast<int *>(dst) = class0_of(r1[h][j]);
          *reinterpret_cast<int *>(dst + F1_CLS) = class1_of(r1[h][j]);
        }
      }
    }
    const int t = (int)threadIdx.x;                                // f0: [channel t / 32][row][piece]: 32 pieces per plane
    _Float16 *dst = stage + F0_BASE + (t >> 5) * F0_PL + ((t >> 2) & 7) * F0_ROW + 2 * (t & 3);
    *reinterpret_cast<int *>(dst) = class0_of(r0v);
    *reinterpret_cast<int *>(dst + F0_CLS) = class1_of(r0v);
  }
};

__global__ __launch_bounds__(NTHREADS, 4) void corr16_fwd_kernel(const _Float16 *__restrict__ in0,
                                                                 const _Float16 *__restrict__ in1,
                                                                 _Float16 *__restrict__ out, int C, int H, int W,
                                                                 int nbx, int nby) {
  extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
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

  // transposed fragment addresses (halves inside a stage): lane 4q+p of group g -> channel 4g+q, tile row p
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int a_off = F0_BASE + cx * F0_CLS + (4 * g + q) * F0_PL + (ty * TP + p) * F0_ROW + tx * TP;
  const int b_off = cx * F1_CLS + (4 * g + q) * F1_PL + (ty * TP + p) * F1_ROW + tx * TP;

  // N-tile rows that lie wholly in the zero padding above / below the image are skipped (wave-uniform)
  unsigned rowmask = 0;
  {
    const int tyu = __builtin_amdgcn_readfirstlane(wave) >> 2;
#pragma unroll
    for (int tl = 0; tl < TQH; ++tl) {
      const int r0 = ys0 + tyu * TP - R + 4 * (TQH * half + tl);
      if (r0 + 3 >= 0 && r0 < hs) rowmask |= 1u << tl;
    }
  }

  FwdStager16 st;
  st.init(H, W, cy, ys0, xs0, 4 * TQH * half);
  const long long plane = (long long)H * W;
  const int nchunks = (C + KC - 1) / KC;
  st.bind(in0, in1, item, 0, C);
  st.load();
  st.store(ldsh);
  __syncthreads();

  for (int ch = 0; ch < nchunks; ++ch) {
    const _Float16 *cur = ldsh + (ch & 1) * STAGE_H;
    _Float16 *nxt = ldsh + ((ch + 1) & 1) * STAGE_H;
    const bool more = ch + 1 < nchunks;
    if (more) {
      st.bind(in0, in1, item, (ch + 1) * KC, C);
      st.load();                               // lands in registers while this chunk computes
    }
    const f16x4 a = tr_read(cur + a_off);
    // B fragments are read one N-tile row (6 tiles) at a time: 128 VGPRs must hold 72 accumulators, the 18
    // staging registers of the next chunk and the fragments (two workgroups per CU = four waves per SIMD,
    // which also cover the LDS latency that is exposed in front of each group)
    auto group = [&](auto G) {
      constexpr int gi = decltype(G)::value;
      if (rowmask & (1u << gi)) {
        f16x4 b[TQ];
#pragma unroll
        for (int t = 0; t < TQ; ++t) b[t] = tr_read(cur + b_off + gi * 4 * F1_ROW + t * 4);
#pragma unroll
        for (int t = 0; t < TQ; ++t)
          acc[gi * TQ + t] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b[t], acc[gi * TQ + t], 0, 0, 0);
        // all six reads in flight before the first MFMA (left alone, the scheduler serialises read -> wait -> MFMA
        // through one register pair to save registers)
        __builtin_amdgcn_sched_group_barrier(0x100, TQ, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, TQ, 0);
      }
    };
    group(Int<0>{}); group(Int<1>{}); group(Int<2>{});
    if (more) st.store(nxt);
    __syncthreads();
  }

  // ---- epilogue: regroup G[p][q] by displacement through LDS (corr_mfma_fwd_half_kernel's): two slabs of
  // 6 displacement rows, dy' = 9*half + 6*s + ddl; local tile row tl holds dy' = 4*(3*half + tl) + u.
  // lane holds, for tile (tl, tqx) and register j: p = (py = lane>>4, px = j), q = (4 tqy + (n>>2), 4 tqx + (n&3))
  const int n = lane & 15;
  const float scale = 1.f / (float)C;
  const bool pow2 = (C & (C - 1)) == 0;
  const int py = lane >> 4;
  const int row = ty * TP + py;
  const int D = GW * GW;
  float *ob = reinterpret_cast<float *>(ldsh);
  const int u = (n >> 2) - py, bq = n & 3;
  int L[4];
  bool mx_first[4], mx_last[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    L[j] = (row * RW + S * (tx * TP + j) + cx) * (SLAB_DYH * GW) + u * GW + (bq - j);
    mx_first[j] = bq >= j;       // tqx == 0     : dx' = bq - j >= 0
    mx_last[j] = bq <= j;        // tqx == TQ - 1: dx' = 4*(TQ-1) + bq - j <= 2R
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
        const float *src = ob + (r * RW + c4 * 4) * (SLAB_DYH * GW) + ddl * GW + dxx;
        _Float16 *dst = out + (item * D + d) * plane + (long long)y * W + x;   // W % 4 == 0, 8-byte aligned base
        *reinterpret_cast<f16x4 *>(dst) = f16x4{(_Float16)src[0], (_Float16)src[SLAB_DYH * GW],
                                                (_Float16)src[2 * SLAB_DYH * GW], (_Float16)src[3 * SLAB_DYH * GW]};
      }
      __syncthreads();
    }
  };
  if (half == 0) epilogue(Int<0>{});
  else epilogue(Int<1>{});
}

// ===========================================================================
// Backward.  Both gradients are  g[c, o] = 1/C * sum_e T(o, e) * fin[c, o + e],  e in [-R, R]^2 (class
// units), with
//   wrt input 0:  T(o, e) = top[d = e, position o],       fin = input 1
//   wrt input 1:  T(o, e) = top[d = -e, position o + e],  fin = input 0
// (ops/correlation_op.cu.cc:119-181, :183-248 at kernel_size 1 / stride_1 1).  Per 4x4 tile of class
// positions o and 16 channels:  G[16 c x 16 o] += A[16 c x 16 k] * B[16 k x 16 o]  where k runs over
// 4-column groups of TWO window rows (2 x 24 window columns = 12 groups = 3 MFMAs), A = fin's window
// rows (K-contiguous as they are), B[k][o] = T(o, window position k - o) (zero outside the 21 x 21 grid).
//
// Workgroup = (pair, row class, 128 channels, block of 16 class rows x 8 class columns x both column
// classes); 8 waves = {column class} x {2 column tiles} x {2 pairs of tile rows}, 2 tiles x 8 M-tiles of
// accumulators per wave.  It walks the 36 window rows of its block in 18 pairs.  ONE LDS stage (64 KB, so
// that two workgroups share a CU and one's staging hides behind the other's MFMAs):
//   F, per column class: [2 rows][128 channels][28 columns (pitch 40)] halves  (ds_read_b64 of 16 channels x
//      2 k-groups: 80-byte pitch = 10 slots -> the even slots, the second k-group the odd ones: conflict free),
//   T, per column class and tile: [12 k-groups][16 positions][4 halves] (+4: the tile stride shifts the banks
//      by 2), read as 32 consecutive 8-byte slots per half-wave, written value by value (ds_write_b16).
// The next pair is loaded into registers while this one computes; a thread's pieces differ only by constants
// from round to round (F: 32 channels further; T: 4 (wrt 0) or 2 (wrt 1) displacement columns further), so
// the addresses are one register plus immediates and the lanes of a store instruction spread over the banks.
// ===========================================================================
constexpr int BR = 16, BC = 8;                 // block: class rows x class columns (per column class)
constexpr int BTR = BR / TP, BTC = BC / TP;    // 4 x 2 tiles per column class
constexpr int CWB = 128, MTB = CWB / 16;       // channels / M-tiles per workgroup
constexpr int WROWS = BR + 2 * R;              // 36 window rows
constexpr int NPAIRS = WROWS / 2;              // 18
constexpr int FB_COLS = BC + 2 * R;            // 28 window columns per class
constexpr int FB_PITCH = 40;                   // halves: 80 B
constexpr int FB_ROW = CWB * FB_PITCH;         // one window row of one class: 5120 halves
constexpr int FB_CLS = 2 * FB_ROW;             // 10240
constexpr int KG = 2 * TQ;                     // 12 k-groups (of 4 window columns) per pair of window rows
constexpr int TB_TILE = KG * 16 * 4 + 4;       // 772 halves per tile
constexpr int TB_CLS = BTR * BTC * TB_TILE;    // 6176
constexpr int TB_BASE = 2 * FB_CLS;            // [F class 0][F class 1][T class 0][T class 1]
constexpr int BSTAGE_H = 2 * (FB_CLS + TB_CLS);// 32832 halves
constexpr int BWD_LDS_BYTES = BSTAGE_H * 2;    // 65664 B: two workgroups per CU
constexpr int OB_HALVES = CWB * BR * 2 * BC;   // epilogue image [128 channels][16 rows][16 full-res columns] halves = 64 KB
static_assert(OB_HALVES * 2 <= BWD_LDS_BYTES, "epilogue image must fit the stage");
static_assert(WROWS % 2 == 0 && (FB_PITCH * 2 / 8) % 4 == 2 && TB_TILE % 4 == 0 && (TB_TILE / 2) % 32 == 2, "LDS bank patterns");

template <bool WRT1>
__global__ __launch_bounds__(NTHREADS, 4) void corr16_bwd_kernel(const _Float16 *__restrict__ topdiff,
                                                                 const _Float16 *__restrict__ fin,
                                                                 _Float16 *__restrict__ gout, int C, int H, int W,
                                                                 int nbx, int nrb, int nch) {
  extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < NWAVES);
  const int cx = wave & 1, tx = (wave >> 1) & 1, th = wave >> 2;
  int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = vb % nbx;
  vb /= nbx;
  const int rb = vb % nrb;
  vb /= nrb;
  const int chg = vb % nch;
  vb /= nch;
  const int cy = vb & 1;
  const long long item = vb >> 1;
  const int hs = (H - cy + S - 1) / S;         // class rows of this row class
  const int by0 = rb * BR, xs0 = bx * BC;
  if (by0 >= hs) return;
  const int c0 = chg * CWB;
  const long long plane = (long long)H * W;
  const __amdgpu_buffer_rsrc_t rtop = make_rsrc16(topdiff + item * (GW * GW) * plane, (long long)GW * GW * plane * 2);
  const __amdgpu_buffer_rsrc_t rin = make_rsrc16(fin + (item * C + c0) * plane, (long long)(C - c0) * plane * 2);
  const int dplane2 = (int)(plane * 2);        // bytes per displacement / channel plane
  const int row_bytes = S * W * 2;             // one class row further down

  // ---- F pieces: round j = (row r = j / 4 of the pair, channels 32 (j % 4) + tid / 16), piece tid % 16 (< 14) of
  // 4 full-resolution columns
  constexpr int F_ROUNDS = 8;
  const int fpc = tid & 15, fcl = tid >> 4;
  const int fX = S * (xs0 - R) + 4 * fpc;
  const unsigned f_v0 = (fpc < S * FB_COLS / 4 && fX >= 0 && fX < W) ? (unsigned)((fcl * plane + (long long)cy * W + fX) * 2) : OOB;
  const int f_l0 = fcl * FB_PITCH + 2 * fpc;

  // ---- T pieces.  A thread owns ONE (block row oyl, row r of the pair, column piece) and walks the displacement
  // columns ex' = e0 + ESTEP j; per pair only ey' = w - oy + R changes.
  //   wrt 0: top[(ey', ex'), row oy, 4 halves = positions (ox, ox + 1) of both classes];  ESTEP 4, 6 rounds
  //   wrt 1: top[(2R - ey', 2R - ex'), row w, 2 halves = position ox + ex' - R of both classes] (a 4-half piece
  //          could straddle the image edge);  ESTEP 2, 11 rounds
  // Lane bits are chosen so that the 32 lanes of a store instruction hit 32 different banks (wrt 0) / 16 (wrt 1).
  constexpr int ESTEP = WRT1 ? 2 : 4, T_ROUNDS = WRT1 ? 11 : 6;
  int oyl, tr, oxl, e0;
  if (WRT1) {
    oxl = (tid & 3) + 4 * ((tid >> 4) & 1);
    oyl = ((tid >> 2) & 3) + 4 * (tid >> 7);
    e0 = (tid >> 5) & 1;
    tr = (tid >> 6) & 1;
  } else {
    oxl = 2 * (tid & 1) + 4 * ((tid >> 1) & 1);
    oyl = ((tid >> 2) & 3) + 4 * (tid >> 7);
    e0 = 2 * ((tid >> 4) & 1) + ((tid >> 5) & 1);
    tr = (tid >> 6) & 1;
  }
  const int tX0 = S * (xs0 + oxl) + (WRT1 ? S * (e0 - R) : 0);     // full-resolution column of round 0 (wrt 1: + 4 j)
  const bool t_ok = by0 + oyl < hs && (WRT1 || tX0 < W);
  int t_b0;                                                        // byte offset of round 0 without the pair terms
  if (WRT1) t_b0 = (2 * R - e0) * dplane2 + (cy * W + tX0) * 2;    // round j: - 2 j planes + 8 j bytes
  else t_b0 = e0 * dplane2 + ((S * (by0 + oyl) + cy) * W + tX0) * 2;   // round j: + 4 j planes
  auto t_slot = [&](int k, int ox) {   // (position (oyl, ox), k) inside the class-0 T image
    return ((oyl >> 2) * BTC + (ox >> 2)) * TB_TILE + ((k >> 2) * 16 + 4 * (oyl & 3) + (ox & 3)) * 4 + (k & 3);
  };
  const int k0 = e0 + (oxl & 3) + WIN * tr;                         // k of round 0; k = ex' + ox % 4 + 24 r
  // wrt 0: k = k0 + 4 j -> slot + 64 j;  second position (ox + 1): k + 1
  // wrt 1: k = k0 + 2 j -> even j: slot(k0) + 64 (j / 2), odd j: slot(k0 + 2) + 64 (j / 2)
  const int t_s0 = t_slot(k0, oxl);
  const int t_s1 = WRT1 ? t_slot(k0 + 2, oxl) : t_slot(k0 + 1, oxl + 1);

  // ---- fragment addresses (halves inside the stage)
  // A: lane = (channel m = lane & 15, k-group g = lane >> 4); MFMA i takes the k-groups 4i + g
  // B: lane = (position n = lane & 15, k-group g)
  const int m = lane & 15, g = lane >> 4;
  int a_off[3], b_off[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int G = 4 * i + g, r = G / TQ, c4 = G % TQ;
    a_off[i] = cx * FB_CLS + r * FB_ROW + m * FB_PITCH + tx * TP + 4 * c4;               // + mt * 16 * FB_PITCH
    b_off[i] = TB_BASE + cx * TB_CLS + (2 * th * BTC + tx) * TB_TILE + (G * 16 + m) * 4;  // second tile row: + BTC * TB_TILE
  }

  f32x4 acc[2][MTB];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int mt = 0; mt < MTB; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the k slots no displacement maps to (k < ox % 4 or k > 20 + ox % 4) and the positions outside the image
  // are never written: they stay zero for the whole kernel
  for (int i = tid; i < TB_CLS; i += NTHREADS) reinterpret_cast<int *>(ldsh + TB_BASE)[i] = 0;

  i32x2 rf[F_ROUNDS];
  std::conditional_t<WRT1, int, i32x2> rt[T_ROUNDS];
  auto load_pair = [&](int pr) {
    const int w0 = by0 - R + 2 * pr;           // first window row (class row) of the pair
#pragma unroll
    for (int j = 0; j < F_ROUNDS; ++j) {
      const int w = w0 + j / 4;
      // the channel goes into the VECTOR offset (the range check ignores the scalar one); rows are tested here
      const bool ok = w >= 0 && w < hs && f_v0 != OOB && c0 + fcl + 32 * (j & 3) < C;
      rf[j] = __builtin_amdgcn_raw_buffer_load_b64(rin, ok ? (int)(f_v0 + (unsigned)((j & 3) * 32) * (unsigned)dplane2) : (int)OOB,
                                                   ok ? w * row_bytes : 0, 0);
    }
    const int w = w0 + tr;
    const int eyp = w - (by0 + oyl) + R;       // ey' of T(o, e): e_y = w - oy
    const bool ok = t_ok && eyp >= 0 && eyp <= 2 * R && w >= 0 && w < hs;
    const int base = t_b0 + (WRT1 ? (2 * R - eyp) * GW * dplane2 + w * row_bytes : eyp * GW * dplane2);
#pragma unroll
    for (int j = 0; j < T_ROUNDS; ++j) {
      bool okj = ok && e0 + ESTEP * j <= 2 * R;
      if constexpr (WRT1) {
        const int X = tX0 + 4 * j;
        okj = okj && X >= 0 && X < W;
        rt[j] = __builtin_amdgcn_raw_buffer_load_b32(rtop, okj ? base - 2 * j * dplane2 + 8 * j : (int)OOB, 0, 0);
      } else {
        rt[j] = __builtin_amdgcn_raw_buffer_load_b64(rtop, okj ? base + 4 * j * dplane2 : (int)OOB, 0, 0);
      }
    }
  };
  auto store_pair = [&]() {
#pragma unroll
    for (int j = 0; j < F_ROUNDS; ++j) {
      if (fpc < S * FB_COLS / 4) {
        _Float16 *dst = ldsh + f_l0 + (j / 4) * FB_ROW + (j & 3) * 32 * FB_PITCH;
        *reinterpret_cast<int *>(dst) = class0_of(rf[j]);
        *reinterpret_cast<int *>(dst + FB_CLS) = class1_of(rf[j]);
      }
    }
    unsigned short *tb = reinterpret_cast<unsigned short *>(ldsh + TB_BASE);
    if (t_ok) {
#pragma unroll
      for (int j = 0; j < T_ROUNDS; ++j) {
        if (e0 + ESTEP * j <= 2 * R) {
          if constexpr (WRT1) {
            const unsigned v = (unsigned)rt[j];                   // (class 0, class 1) of one position
            const int s = ((j & 1) ? t_s1 : t_s0) + 64 * (j / 2);
            tb[s] = (unsigned short)v;
            tb[TB_CLS + s] = (unsigned short)(v >> 16);
          } else {
            const unsigned v0 = (unsigned)rt[j].x, v1 = (unsigned)rt[j].y;   // positions ox, ox + 1 (ox even)
            tb[t_s0 + 64 * j] = (unsigned short)v0;
            tb[TB_CLS + t_s0 + 64 * j] = (unsigned short)(v0 >> 16);
            tb[t_s1 + 64 * j] = (unsigned short)v1;
            tb[TB_CLS + t_s1 + 64 * j] = (unsigned short)(v1 >> 16);
          }
        }
      }
    }
  };

  // pairs of window rows that touch the image: w in [0, hs)
  const int pr_first = max(0, (R - by0) / 2);                                   // first pair with w0 >= 0 (w0 is even)
  const int pr_last = min(NPAIRS - 1, (hs - 1 - (by0 - R)) / 2);                // last pair with w0 <= hs - 1
  // tile row T (0..3) of the block is live for the window rows [4T - R, 4T + 3 + R] relative to by0, i.e. for
  // the pairs 2T .. 2T + 11
  load_pair(pr_first);
  __syncthreads();                             // the zero fill of the T image is complete
  store_pair();
  __syncthreads();
  for (int pr = pr_first; pr <= pr_last; ++pr) {
    const bool more = pr < pr_last;
    if (more) load_pair(pr + 1);               // lands in registers while this pair computes
    const bool live0 = pr >= 2 * (2 * th) && pr <= 2 * (2 * th) + 11;
    const bool live1 = pr >= 2 * (2 * th + 1) && pr <= 2 * (2 * th + 1) + 11;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const f16x4 b0 = *reinterpret_cast<const f16x4 *>(ldsh + b_off[i]);
      const f16x4 b1 = *reinterpret_cast<const f16x4 *>(ldsh + b_off[i] + BTC * TB_TILE);
#pragma unroll
      for (int hm = 0; hm < 2; ++hm) {         // M-tiles in two halves: 8 fragment registers instead of 16
        f16x4 a[MTB / 2];
#pragma unroll
        for (int mt = 0; mt < MTB / 2; ++mt)
          a[mt] = *reinterpret_cast<const f16x4 *>(ldsh + a_off[i] + (hm * (MTB / 2) + mt) * 16 * FB_PITCH);
        if (live0) {
#pragma unroll
          for (int mt = 0; mt < MTB / 2; ++mt)
            acc[0][hm * (MTB / 2) + mt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a[mt], b0, acc[0][hm * (MTB / 2) + mt], 0, 0, 0);
        }
        if (live1) {
#pragma unroll
          for (int mt = 0; mt < MTB / 2; ++mt)
            acc[1][hm * (MTB / 2) + mt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a[mt], b1, acc[1][hm * (MTB / 2) + mt], 0, 0, 0);
        }
      }
    }
    __syncthreads();                           // every wave has read this pair's images
    if (more) store_pair();
    __syncthreads();
  }

  // ---- epilogue: [128 channels][16 rows][16 full-resolution columns] halves through LDS, 8-byte stores
  _Float16 *ob = ldsh;
  const float inv = 1.f / (float)C;
  const bool pow2 = (C & (C - 1)) == 0;
  const int n = lane & 15, cq = lane >> 4;     // D: row (channel) 4 cq + j, column (position) n
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int orow = (2 * th + t) * TP + (n >> 2);
    const int ocol = S * (tx * TP + (n & 3)) + cx;
#pragma unroll
    for (int mt = 0; mt < MTB; ++mt) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cl = mt * 16 + 4 * cq + j;
        const float v = pow2 ? acc[t][mt][j] * inv : acc[t][mt][j] / (float)C;
        ob[(cl * BR + orow) * (S * BC) + ocol] = (_Float16)v;
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < OB_HALVES / 4; idx += NTHREADS) {
    const int c4 = idx % (S * BC / 4);
    const int rr = (idx / (S * BC / 4)) % BR;
    const int cl = idx / (S * BC / 4 * BR);
    const int c = c0 + cl, oy = by0 + rr;
    const int y = S * oy + cy, x = S * xs0 + c4 * 4;
    if (c >= C || oy >= hs || x >= W) continue;
    *reinterpret_cast<i32x2 *>(gout + (item * C + c) * plane + (long long)y * W + x) =
        *reinterpret_cast<const i32x2 *>(ob + idx * 4);
  }
}

inline bool aligned8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

constexpr int kMaxDevices = 64;
template <auto KERNEL>
int reserve_lds(size_t smem, const char *what) {
  static std::atomic<bool> granted[kMaxDevices];      // zero-initialised
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return launch_status(what);
  const bool tracked = dev >= 0 && dev < kMaxDevices;
  if (tracked && granted[dev].load(std::memory_order_acquire)) return UNFLOW_OK;
  const hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (st != hipSuccess) return fail(UNFLOW_ERR_LAUNCH, "%s: cannot reserve %zu B of LDS (%s)", what, smem, hipGetErrorString(st));
  if (tracked) granted[dev].store(true, std::memory_order_release);
  return UNFLOW_OK;
}

}  // namespace

int corr16_fwd(const void *in0, const void *in1, void *out, int B, int C, const CorrGeometry &g, hipStream_t s) {
  const int H = g.H, W = g.W;
  const int hs = (H + S - 1) / S, ws = (W + S - 1) / S;
  const int nbx = (ws + RX - 1) / RX, nby = (hs + RY - 1) / RY;
  if ((W % 4) != 0 || !aligned8(in0) || !aligned8(in1) || !aligned8(out))
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "correlation (fp16): W must be a multiple of 4 and the tensors 8-byte aligned");
  if (int rc = reserve_lds<corr16_fwd_kernel>(FWD_LDS_BYTES, "correlation_fwd (f16 matrix cores)")) return rc;
  hipLaunchKernelGGL(corr16_fwd_kernel, dim3(nbx * nby * B * S * 2), dim3(NTHREADS), FWD_LDS_BYTES, s,
                     static_cast<const _Float16 *>(in0), static_cast<const _Float16 *>(in1), static_cast<_Float16 *>(out), C, H,
                     W, nbx, nby);
  return launch_status("correlation_fwd (f16 matrix cores)");
}

int corr16_bwd(const void *topdiff, const void *in0, const void *in1, void *g0, void *g1, int B, int C,
               const CorrGeometry &g, hipStream_t s) {
  const int H = g.H, W = g.W;
  const int hs = (H + S - 1) / S, ws = (W + S - 1) / S;
  const int nbx = (ws + BC - 1) / BC, nrb = (hs + BR - 1) / BR, nch = (C + CWB - 1) / CWB;
  if ((W % 4) != 0 || !aligned8(topdiff) || !aligned8(in0) || !aligned8(in1) || !aligned8(g0) || !aligned8(g1))
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "correlation_grad (fp16): W must be a multiple of 4 and the tensors 8-byte aligned");
  const dim3 grid(nbx * nrb * nch * S * B);
  using H16 = _Float16;
  if (int rc = reserve_lds<corr16_bwd_kernel<false>>(BWD_LDS_BYTES, "correlation_bwd (f16 matrix cores)")) return rc;
  if (int rc = reserve_lds<corr16_bwd_kernel<true>>(BWD_LDS_BYTES, "correlation_bwd (f16 matrix cores)")) return rc;
  hipLaunchKernelGGL(corr16_bwd_kernel<false>, grid, dim3(NTHREADS), BWD_LDS_BYTES, s, static_cast<const H16 *>(topdiff),
                     static_cast<const H16 *>(in1), static_cast<H16 *>(g0), C, H, W, nbx, nrb, nch);
  hipLaunchKernelGGL(corr16_bwd_kernel<true>, grid, dim3(NTHREADS), BWD_LDS_BYTES, s, static_cast<const H16 *>(topdiff),
                     static_cast<const H16 *>(in0), static_cast<H16 *>(g1), C, H, W, nbx, nrb, nch);
  return launch_status("correlation_bwd (f16 matrix cores)");
}

}  // namespace unflow
