// FlowNetC correlation backward on the f32 matrix cores of gfx950: the shipped kernel (third generation of the row sweep,
// hot instruction stream as generated instruction blocks: corr_bwd4_asm.inc, tools/gen_corr_bwd4_asm.py).
//
// Reference semantics: ops/correlation_op.cu.cc:119-248, :366-388 (CorrelateDataBackward0/1 + CorrelationGrad) of the
// reference checkout.  Formulation and the row sweep itself: correlation_mfma.hip ("Backward"), whose first-generation
// kernel still serves fp16 storage, C % 128 != 0 and rows that are not 16-byte aligned (corr_mfma_bwd_r1).
#include "corr_mfma_common.h"
#include "corr_bwd4_asm.inc"

namespace unflow {
namespace {

// ===========================================================================
// Backward for the float32 FlowNetC shape (16-byte aligned rows, C % 128 == 0): the row sweep of
// corr_mfma_bwd_kernel in its third generation.  What the generations kept:
//   * (2nd) no B masks: the top image has an all-zero 22nd row and 22nd displacement column (the DMA
//     range check writes them) and an invalid (row, dx') simply points there;
//   * (2nd) finished tiles leave through ob while the next row computes (owners dump in front of the
//     row's barrier, everybody stores inside K-steps 3, 4 of the row after next);
//   * (2nd) persistent workgroups: one per CU walks its share of the (pair, row class, channel group,
//     strip) items; the first row of the next item is fetched during the last row of this one;
//   * (2nd) LDS images without the 4-way (top) and 2-way (A) bank conflicts of the first kernel:
//     top row stride 356, A channel groups of 8 start alternately 0 / 1 floats in;
//   * (2nd) the row barrier stands between K-steps 4 and 5: when it opens, K-step 5's MFMAs are ready and the
//     first fragments of the next row are read behind them;
//   * (3rd) the instruction stream of a row is WRITTEN, not compiled: see corr_mfma_bwd4_kernel below.
// The second generation itself (the same sweep as compiled C++: 1.05-1.07 ms per call against 0.99) is
// tools/sketches/corr_bwd2_second_generation.hip.txt.
// ===========================================================================
namespace b2 {
constexpr int TROWS2 = TROWS + 1;               // 22: + one all-zero row
constexpr int TDX2 = GW + 1;                    // 22: + one all-zero displacement column
constexpr int ZERO_ROW = TROWS;                 // 21 (the all-zero displacement column is column GW = 21 of every row)
constexpr int TRS = TDX2 * RW + 4;              // 356 floats per top-image row: 2-way instead of 4-way conflicts
constexpr int T_IMAGE = TROWS2 * TRS;           // 7832
constexpr int T_PIECES = (T_IMAGE + 255) / 256; // 31 DMA pieces of 1 KiB
constexpr int T_STAGE = T_PIECES * 256;         // 7936
constexpr int AG = 8 * ACS;                     // 480 floats: a group of 8 channel rows
constexpr int A_PIECE_F = AG / 2;               // 240 floats = 60 lanes x 4: a piece never crosses a group
constexpr int A_PIECES = CWS / 8 * 2;           // 32
constexpr int A_STAGE = (CWS * ACS + 1 + 63) / 64 * 64;   // 7744 (odd groups start one float later)
constexpr int STAGE2 = A_STAGE + T_STAGE;        // 15680 floats
constexpr int LDS_FLOATS = 2 * STAGE2 + SW_OUT;  // 39552 floats = 158208 B
constexpr int LDS_BYTES2 = 4 * LDS_FLOATS + 64;  // + the row barrier's arrival counter (one dword behind the images)
constexpr int A_PER_WAVE = A_PIECES / NWAVES;   // 4
constexpr int T_PER_WAVE = (T_PIECES + NWAVES - 1) / NWAVES;   // 4 (wave 7: 3)
static_assert(A_PIECES % NWAVES == 0 && LDS_BYTES2 <= 160 * 1024, "LDS budget");
// A fragment (ds_read_b32, lanes 0..31 = 16 channels x 2 window columns): channel n sits at
// n*60 + (n>>3): the 8 channels of a group take the 8 banks = 0 mod 4 (60 = 28 mod 32), the next
// group the banks = 1 mod 4, the second window column (+2 floats) the banks = 2, 3 mod 4.
static_assert(ACS % 32 == 28, "A image bank pattern");
// Issue priorities of the K-step blocks (s_setprio, corr_bwd4_asm.inc): falling along a row -- which begins behind
// the row barrier with K-step 5 of the previous one --, the upper four waves' one K-step later than the lower four's.
#ifdef UNFLOW_TUNE_PRIO_A                    // diagnostic: another pair of tables (tools/variants.sh)
#define UNFLOW_TUNE_PRIO_TABLE
#else
#define UNFLOW_TUNE_PRIO_A 3, 2, 2, 1, 1, 0
#define UNFLOW_TUNE_PRIO_B 3, 3, 2, 2, 1, 1
#endif
constexpr int prio_a(int q) { constexpr int p[6] = {UNFLOW_TUNE_PRIO_A}; return p[(q + 1) % 6]; }
constexpr int prio_b(int q) { constexpr int p[6] = {UNFLOW_TUNE_PRIO_B}; return p[(q + 1) % 6]; }
}  // namespace b2

// Barriers of the row sweep, written out: the compiler's own wait-count insertion
// knows nothing about the LDS side of an LDS-DMA (a plain __syncthreads() came out as vmcnt(9) in
// front of the very first row), and its fences wait for stores nobody depends on.
//   _DMA: every LDS-DMA (and everything older) of this wave has landed, LDS traffic done
//   _LDS: only this wave's LDS reads / writes are done (dump -> store hand-over through ob)
#define UNFLOW_BARRIER_DMA() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define UNFLOW_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

#ifdef UNFLOW_TUNE_NODMA
#define UNFLOW_TUNE_NODMA2 (C > 0)       // diagnostic: always true at run time, opaque to the compiler
#else
#define UNFLOW_TUNE_NODMA2 false
#endif

// ===========================================================================
// Third generation: the row sweep with its hot instruction stream written as INSTRUCTION BLOCKS
// (corr_bwd4_asm.inc, generated by tools/gen_corr_bwd4_asm.py).  Measurements of round 3
// (profiles/r03_variants.txt, profiles/ubench_mfma_lds.txt): every non-MFMA instruction of a wave costs its SIMD
// 2-3 cycles of matrix time, and the compiled second generation carried 2.4 scalar + 0.95 vector instructions per
// MFMA (a branch per 4 MFMAs, a wait count per fragment group, address arithmetic per K-step, 11-18 scalar
// instructions per DMA issue, a per-row recomputation of the sweep state with three quarter-rate multiplies, SGPR
// spills through v_readlane) -- ~7 cycles per 32-cycle MFMA, which was its 0.66 of the peak.  Here
//   * a K-step is  s_waitcnt; 3 x (s_bitcmp0, s_cbranch, 8 MFMAs); 11 ds_read_b32 with instruction offsets
//     = 7 scalar + 11 LDS instructions per 24 MFMAs, no vector ALU instruction at all;
//   * the whole of sweep row w + 1 is fetched by 8 LDS-DMA instructions between the MFMA groups of K-steps 0, 1 of
//     row w (M0 from one scalar add each, the row in the scalar offset / one v_add, one EXEC mask for the A pieces);
//   * the sweep state (per ring slot: row of the tile's life, tile, live bit, the lanes' fragment addresses) moves
//     one row further in ~27 scalar + 18 vector instructions that stand BETWEEN the MFMAs of K-step 5, the one behind
//     the row barrier (UNFLOW_BWD4_KSTEP5_ROW, round 4: as a block of its own behind the barrier it met an empty
//     matrix pipe in all eight waves at once);
//   * the ROW BARRIER is an arrival counter in LDS instead of s_barrier (round 4): a wave adds 1 behind K-step 4 and looks
//     at the counter inside K-step 5, in front of its first reads of the next row's stage; the MFMAs of K-step 5 need
//     nothing from the other waves and run while the last ones arrive (-1.6 to -1.9 %, bit-equal);
//   * the two waves of a SIMD are kept level by issue priorities that fall along the row (s_setprio at the head of
//     every K-step, prio_a / prio_b below): arbitrated by age, the older one ran ahead and waited for a quarter of
//     every row at the barrier while the younger one worked off its rest alone (cycle stamps, tools/stamps.py);
// 1.15 scalar + 0.55 vector instructions per MFMA in all (PMC), 0.99 ms per call = 0.71 of the peak, results
// bit-equal to the second generation's.  The accumulators (96) and the double-buffered fragments (22 + 2 scratch)
// live in FIXED registers v[136:255] that only the blocks touch: the blocks name them in their clobber lists, so
// the compiler counts them in the kernel's register budget, and amdgpu_num_vgpr(136) ends its own allocation
// below them (it needs ~72), so what a block leaves there is still there for the next one.  What stays C++:
// items, the tile stores (inside the blocks they measured 0.5 % slower), the dump decisions, barriers.
// ===========================================================================
typedef __attribute__((address_space(3))) const float lds_cf;
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool WRT1>
// (amdgpu_num_vgpr: the compiler's own allocation ends below the fixed registers of the instruction blocks)
__global__ __launch_bounds__(NTHREADS, 2) __attribute__((amdgpu_num_vgpr(UNFLOW_BWD4_FIRST_FIXED_VGPR))) void corr_mfma_bwd4_kernel(const float *__restrict__ topdiff,
                                                                     const float *__restrict__ fin,
                                                                     float *__restrict__ gout, int C, int H, int W,
                                                                     int nbx, int nch, int nitems) {
  using namespace b2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_assume(wave >= 0 && wave < NWAVES);
  const int cx = wave & 1, tx = (wave >> 1) & 1, ph = wave >> 2;
  const long long plane = (long long)H * W;
  const int row_bytes = S * W * 4;             // one class row further down

  // items of this workgroup: XCD x (blocks b = x mod 8) walks the contiguous range
  // [x n/8, (x+1) n/8) so that its L2 sees neighbouring strips of the same pairs
  const bool xcd_split = (nitems & 7) == 0 && (gridDim.x & 7) == 0;
  const int it_stride = xcd_split ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int it_first = xcd_split ? (int)(blockIdx.x & 7) * (nitems >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int it_end = xcd_split ? ((int)(blockIdx.x & 7) + 1) * (nitems >> 3) : nitems;
  if (it_first >= it_end) return;

  const int k = lane >> 4, n = lane & 15;
  const int oyl = n >> 2, oxl = n & 3;
  const int b_col = S * (tx * TP + oxl) + cx;  // column of this lane's output position in the region
  // A fragment of M-tile m, K-step q:  a_off + m*16*ACS + 8*q   (16 channels = two groups)
  const int a_off = n * ACS + (n >> 3) + S * (tx * TP + k) + cx;
  float *const ob = lds + 2 * STAGE2;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_cf *)lds;   // LDS byte address of the dynamic segment
  const int dump_off_ = (4 * k * TP + oyl) * RW + b_col;   // + (m*16 + j) * TP * RW
  const unsigned dump_addr = lds0 + (2 * STAGE2 + dump_off_) * 4;
  const int st_c4 = threadIdx.x & 3, st_rr = (threadIdx.x >> 2) & 3, st_cl = threadIdx.x >> 4;   // + 32 q channels
  const int st_voff = (int)((st_cl * plane + (long long)(S * st_rr) * W + st_c4 * 4) * 4);        // inside the channel group
  const float inv_c = 1.f / (float)C;
  const bool pow2_c = (C & (C - 1)) == 0;

  // ---- what the DMA of one item needs.  The last row of an item fetches nothing for it any more,
  // so its top is where the description of the NEXT item replaces this one.
  struct Fetch {
    __amdgpu_buffer_rsrc_t rtop, rin;
    unsigned a_voff[A_PER_WAVE];   // channel plane + column of the lane's A segment (or OOB); the row is the SGPR offset
    int t_base[T_PER_WAVE];        // byte offset of the lane's top segment at sweep row 0 (linear in w)
    int cy, hs;
  };
  // lane -> (row of the top image, valid) of its top pieces: the same for every item
  int t_orow[T_PER_WAVE];
#pragma unroll
  for (int j = 0; j < T_PER_WAVE; ++j) {
    const int f = ((wave + NWAVES * j) * 64 + lane) * 4;
    const int orow = f / TRS, rem = f - orow * TRS;
    const bool cell = wave + NWAVES * j < T_PIECES && orow < TROWS && rem / RW < GW;   // not the zero row / column / pad
    t_orow[j] = cell ? orow : (1 << 28);
  }
  // lanes that take part in A piece j: the 4 pad floats that end every 60-float channel row are never
  // read and must not be written either -- the last one of an odd (shifted) group is the first float
  // of the next group
  unsigned a_lanes = 0;
#pragma unroll
  for (int j = 0; j < A_PER_WAVE; ++j) {
    const int g = lane * 4 + ((wave + NWAVES * j) & 1) * A_PIECE_F;
    if (lane < A_PIECE_F / 4 && g % ACS < ROWF) a_lanes |= 1u << j;
  }
  // LDS-DMA of a row as issued by the K-step blocks: the lanes of an A piece (the same for the wave's four pieces:
  // piece p = wave + 8 j has the parity of the wave), the wave's first destinations inside a stage (consecutive
  // pieces lie A_STEP / T_STEP bytes further on), whether it has a fourth top-image piece (31 pieces: wave 7 has 3)
  const unsigned long long a_mask = __builtin_amdgcn_ballot_w64(lane < A_PIECE_F / 4 && (lane * 4 + (wave & 1) * A_PIECE_F) % ACS < ROWF);
  const unsigned a_dst0 = (unsigned)(((wave >> 1) * AG + ((wave >> 1) & 1) + (wave & 1) * A_PIECE_F) * 4);
  const unsigned t_dst0 = (unsigned)((A_STAGE + wave * 256) * 4);
  const unsigned t_has3 = wave + 3 * NWAVES < T_PIECES ? 1u : 0u;
  struct Item {
    int bx, chg, cy;
    long long pair;
  };
  auto decode = [&](int it) {
    Item d;
    d.bx = it % nbx;
    it /= nbx;
    d.chg = it % nch;
    it /= nch;
    d.cy = it & 1;
    d.pair = it >> 1;
    return d;
  };
  auto make_fetch = [&](const Item &d) {
    Fetch f;
    f.cy = d.cy;
    f.hs = (H - d.cy + S - 1) / S;
    const int xs0 = d.bx * RX, c0 = d.chg * CWS;
    f.rtop = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(topdiff + d.pair * (GW * GW) * plane), (short)0,
                                               (int)(GW * GW * plane * 4), 0x00020000);
    f.rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(fin + (d.pair * C + c0) * plane), (short)0,
                                              (int)((long long)(C - c0) * plane * 4), 0x00020000);
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
      const int p = wave + NWAVES * j;                 // piece: group p >> 1, half p & 1
      const int g = lane * 4 + (p & 1) * A_PIECE_F;    // float inside the group (lanes >= 60 take no part)
      const int r8 = g / ACS, x = g - r8 * ACS;
      const int c = (p >> 1) * 8 + r8;
      const int X = S * (xs0 - R) + x;
      f.a_voff[j] = (X >= 0 && X < W) ? (unsigned)((c * plane + X) * 4) : OOB;     // outside the image: zeros
    }
#pragma unroll
    for (int j = 0; j < T_PER_WAVE; ++j) {
      const int fl = ((wave + NWAVES * j) * 64 + lane) * 4;     // float of the top image: [orow][dx'][col]
      const int orow = min(fl / TRS, TROWS - 1), rem = fl % TRS;
      const int dxp = min(rem / RW, GW - 1), col = rem % RW;
      int dyp, Y0, Xt;
      if (WRT1) {   // position = window position: row w, column shifted by -dx
        dyp = orow;
        Y0 = d.cy;
        Xt = S * xs0 - S * (dxp - R) + col;
      } else {      // position = output position: row oy = w - R + orow
        dyp = 2 * R - orow;
        Y0 = S * (orow - R) + d.cy;
        Xt = S * xs0 + col;
      }
      // (a straddling piece only feeds masked window columns; cells that must BE zero -- the all-zero row and
      // column, the pad -- get the out-of-range base, which stays out of range whatever row offset is added)
      const bool ok = Xt + 3 >= 0 && Xt < W && t_orow[j] < TROWS;
      f.t_base[j] = ok ? (int)((((long long)(dyp * GW + dxp) * H + Y0) * W + Xt) * 4) : (int)0x80000000;
    }
    return f;
  };
  // one DMA piece of window row w of the item described by f into `stage`
  auto issue_a = [&](const Fetch &f, int j, int w, float *stage) {
    const int p = wave + NWAVES * j;
    float *dst = stage + (p >> 1) * AG + ((p >> 1) & 1) + (p & 1) * A_PIECE_F;
    if (a_lanes & (1u << j)) dma<4>(f.rin, dst, f.a_voff[j], (S * w + f.cy) * W * 4);
  };
  auto issue_t = [&](const Fetch &f, int j, int w, float *stage) {
    if (wave + NWAVES * j < T_PIECES) {
      const int oy = w - R + t_orow[j];
      const unsigned voff = (oy >= 0 && oy < f.hs && f.t_base[j] != (int)0x80000000) ? (unsigned)(f.t_base[j] + w * row_bytes) : OOB;
      dma<4>(f.rtop, stage + A_STAGE + (wave + NWAVES * j) * 256, voff, 0);
    }
  };

  // accumulators: fixed registers, touched by the instruction blocks of corr_bwd4_asm.inc only
  asm volatile(UNFLOW_BWD4_ZERO ::: UNFLOW_BWD4_CLOBBERS);
  // a finished tile of ring slot SLOT: to the epilogue image, then zeroed
#define UNFLOW_DUMP_TILE4(SLOT)                                                                              \
  do {                                                                                                       \
    const int slot_ = (SLOT);                                                                                \
    if (slot_ == 0) asm volatile(UNFLOW_BWD4_DUMP_S0 : : [addr] "v"(dump_addr) : "memory", UNFLOW_BWD4_CLOBBERS);      \
    else if (slot_ == 1) asm volatile(UNFLOW_BWD4_DUMP_S1 : : [addr] "v"(dump_addr) : "memory", UNFLOW_BWD4_CLOBBERS); \
    else asm volatile(UNFLOW_BWD4_DUMP_S2 : : [addr] "v"(dump_addr) : "memory", UNFLOW_BWD4_CLOBBERS);                 \
  } while (0)

  Item cur_item = decode(it_first);
  Fetch fc = make_fetch(cur_item);
  int rowc = 0;                                  // rows done by this workgroup: stage = rowc & 1
#pragma unroll
  for (int j = 0; j < A_PER_WAVE; ++j) issue_a(fc, j, 0, lds);
#pragma unroll
  for (int j = 0; j < T_PER_WAVE; ++j) issue_t(fc, j, 0, lds);

  // The ROW BARRIER is an arrival counter in LDS, not s_barrier: a wave adds 1 behind K-step 4 of a row and looks at the
  // counter inside K-step 5 (UNFLOW_BWD4_KSTEP5_ROW), in front of its first reads of the next row's stage -- the MFMAs of
  // K-step 5 need nothing from the other waves and run while the last ones arrive.  It never wraps (8 per row).
  const unsigned bar_addr = lds0 + LDS_FLOATS * 4;
  if (threadIdx.x == 0) lds[LDS_FLOATS] = 0.f;
  UNFLOW_BARRIER_DMA();          // row 0 of the first item has landed
#ifdef UNFLOW_TUNE_STAMPS
  int stamp_slot = -1;           // workgroups 0..3 and 128..131, waves 0 and 4 (SIMD partners), WRT1 launch, first item
  {
    const int bsel = (int)blockIdx.x < 4 ? (int)blockIdx.x : ((int)blockIdx.x >= 128 && (int)blockIdx.x < 132 ? (int)blockIdx.x - 124 : -1);
    if (WRT1 && bsel >= 0 && (wave == 0 || wave == 4)) stamp_slot = bsel * 2 + (wave >> 2);
  }
  UNFLOW_STAMP(STAMP_ROWS, 0);
#endif

  for (int it = it_first; it < it_end; it += it_stride) {
    const int cy = cur_item.cy, hs = fc.hs, ntiles = (hs + TP - 1) / TP;
    const int xs0 = cur_item.bx * RX, c0 = cur_item.chg * CWS;
    const long long pair = cur_item.pair;
    const bool last_item = it + it_stride >= it_end;
    // B fragment of K-step q, slot sl: ONE byte address per slot and row (RowState::b_addr) plus the instruction's
    // offset: dx' = (k - oxl) + 4 q (wrt input 0) or (oxl - k) + 4 (5 - q) (wrt input 1) is affine in q wherever it is
    // valid, and it is valid for every lane in K-steps 1..4.  In K-steps 0 and 5 the lanes with k < oxl / k > oxl fall
    // outside the 21 displacements and read the all-zero cell instead; for the gradient wrt input 1 a window column
    // outside the image must read zero too -- only strips within 20 columns of the image border have such lanes,
    // and a wave-uniform bit per K-step sends those through the masked addresses.
    bool v_first = k >= oxl, v_last = k <= oxl;     // lanes with a valid dx' in K-step 0 / K-step 5
    unsigned v_mid = 0;                              // per lane: bit q clear = K-step q (1..4) must read zero
    unsigned q_slow = 0;                             // wave-uniform: K-steps 1..4 that have such lanes
    if constexpr (WRT1) {
      const int xw0 = S * (xs0 + tx * TP - R) + cx;  // full-resolution column of window position (q 0, k 0)
#pragma unroll
      for (int q = 0; q < KSTEPS; ++q) {
        const int Xw = xw0 + S * (4 * q + k);
        const bool in = Xw >= 0 && Xw < W;
        if (q == 0) v_first = v_first && in;
        else if (q == KSTEPS - 1) v_last = v_last && in;
        else {
          v_mid |= in ? (1u << q) : 0u;
          if (xw0 + S * 4 * q < 0 || xw0 + S * (4 * q + 3) >= W) q_slow |= 1u << q;
        }
      }
    }
    const unsigned long long vfirst_mask = __builtin_amdgcn_ballot_w64(v_first);    // (the K-step 5 block's operand)
    const int b_lane = (((WRT1 ? oxl - k : k - oxl) * RW) + b_col + A_STAGE) * 4;     // bytes, + row, + K-step offset
    // Sweep state, kept INCREMENTALLY (block UNFLOW_BWD4_KSTEP5_ROW; the per-row recomputation of the second generation was
    // ~65 scalar + ~35 vector instructions, three of them quarter-rate multiplies).  Ring slot sl holds tile
    // t = 2u + ph, u = sl mod 3, through the 24 window rows that touch it: j = w - 4 ph + R - 8u = 0..23 is the row of
    // its life, and the slot passes on to u + 3 when j wraps.  A lane's top-image row is 20 + oyl - j: one image row up
    // per sweep row, so its LDS address moves by a wave-uniform step (-one row, or +23 rows at the wrap, plus the change
    // of stage).  All lanes are inside the 21 rows for j = 3..20; in the three rows at either end of a tile's life the
    // lanes outside read the all-zero cell.  u < 0 or a tile beyond the image is a dead slot (live bit clear): its
    // MFMAs are skipped and whatever its reads return is not used.
    int sw_j[RING], sw_u[RING];      // wave-uniform
    unsigned sw_live = 0;            // bit sl: the slot's tile exists
    unsigned sw_run[RING];           // per lane: LDS byte address of the slot's row, unmasked
    unsigned sw_b[RING];             // per lane: the address the fragments are read at (K-step offset to be added)
    unsigned sw_a, sw_zero;          // per lane: of the A fragments; wave-uniform: an all-zero cell of the stage's top image
    const int sw_tmax = (ntiles - ph + 1) / 2;       // tiles of this wave's parity
    {
      const unsigned stage = lds0 + (rowc & 1) * (STAGE2 * 4);   // row 0
      sw_a = stage + a_off * 4;
      sw_zero = stage + (ZERO_ROW * TRS + A_STAGE) * 4;
      const int v = R - 4 * ph;
      const int u_hi = v >> 3, r3 = u_hi % RING;
#pragma unroll
      for (int sl = 0; sl < RING; ++sl) {
        const int back = r3 - sl + (r3 < sl ? RING : 0);
        sw_u[sl] = u_hi - back;
        sw_j[sl] = v - 8 * sw_u[sl];
        sw_live |= (sw_u[sl] >= 0 && sw_u[sl] < sw_tmax) ? 1u << sl : 0u;
        const int orow = 2 * R + oyl - sw_j[sl];
        sw_run[sl] = stage + orow * (TRS * 4) + b_lane;
        sw_b[sl] = (orow >= 0 && orow <= 2 * R) ? sw_run[sl] : sw_zero;
      }
    }
    // the tile image leaves through a buffer descriptor of this item's gradient planes (channel
    // group c0 of the pair): one per-lane offset (channel, tile row, 4-column group), everything
    // else -- the tile, the 32-channel step of the float4 index, the strip -- in the scalar offset
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
        gout + (pair * C + c0) * plane, (short)0, (int)((long long)(C - c0) * plane * 4), 0x00020000);
    auto store_piece = [&](int t, int q) __attribute__((always_inline)) {
      const int idx = (int)threadIdx.x + NTHREADS * q;
      const int oy = 4 * t + st_rr;
      const int x = S * xs0 + st_c4 * 4;
      // Every lane of every wave issues the store: the row barrier counts on exactly four stores per wave
      // behind the last DMA piece (vmcnt(4)).  A lane outside the image gets an out-of-range VECTOR offset,
      // which the descriptor's range check drops (C % 128 == 0: every channel exists).
      f32x4 v = *reinterpret_cast<const f32x4 *>(ob + idx * 4);
      if (pow2_c) v *= inv_c;
      else v /= (float)C;
      const unsigned soff = (unsigned)(((long long)(NTHREADS / 16) * q * plane + (long long)(S * 4 * t + cy) * W + S * xs0) * 4);
      const unsigned voff = (oy < hs && x < W) ? (unsigned)st_voff : OOB;
      // The store and the two wait states gfx950 needs before anything may overwrite its data registers (a store-data
      // hazard the compiler does not cover for an SGPR soffset: tools/ubench/store_data_hazard.hip) are ONE statement,
      // so that nothing can be scheduled in between.
#ifdef UNFLOW_TUNE_NOGLOBAL     // timing only: the tile pieces are read and scaled, not stored (OOB offset: the store is issued and dropped)
      asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(v), "v"(OOB), "s"(rout), "s"(soff) : "memory");
#else
      asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(rout), "s"(soff) : "memory");
#endif
    };

    int store_tile = -1;         // tile whose image in ob is stored during this row      (workgroup-uniform)
    int dumped = -1;             // tile written to ob behind the previous row's last K-step: visible after this row's barrier
    // (row 0 of this item has landed: fetched before the loop, or during the last rows of the previous
    // item and waited for at the last row's barrier)

    // The row barrier stands between K-steps 4 and 5, not between rows: when it opens, the MFMAs of K-step 5 are
    // ready to issue (operands in registers) and the first fragments of row w + 1 -- whose image the barrier has
    // just declared complete -- are read behind its first MFMA group, so neither the barrier round trip nor the LDS
    // latency of the first fragments meets an empty matrix pipe.  Nothing reads row w's image after K-step 4, so
    // that stage takes the DMA of row w + 2 during K-steps 0, 1 of row w + 1.
    static_assert(A_PER_WAVE == 4 && T_PER_WAVE == 4, "DMA schedule of the blocks: four A and four top pieces per wave and row");
    for (int w = 0; w < hs; ++w, ++rowc) {
      UNFLOW_STAMP(rowc, 0);
      const bool more = w + 1 < hs;
      // The whole of sweep row w + 1 is fetched during K-steps 0 and 1 of row w, into the other stage (free since row
      // w - 1's barrier), three K-steps ahead of the barrier that needs it.  Behind the last row of an item that is
      // row 0 of the next item: its description replaces this one's right here.
      if (!more && !last_item) {
        cur_item = decode(it + it_stride);
        fc = make_fetch(cur_item);
      }
      const unsigned fetch = ((more || !last_item) && !UNFLOW_TUNE_NODMA2) ? 1u : 0u;
      const int wn = more ? w + 1 : 0;
      // of THIS row (the state moves on behind the barrier); bit 8: the upper four waves (the blocks' issue priorities)
      const unsigned live = (unsigned)__builtin_amdgcn_readfirstlane((int)(sw_live | ((threadIdx.x >> 8) << 8)));

      // The matrix work of K-step q and the fragment reads of K-step q + 1 (K-step 5: the first fragments of row
      // w + 1) are ONE instruction block (corr_bwd4_asm.inc): the C++ around it only supplies the addresses.
      auto kstep_block = [&](auto Q) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value, qn = (q + 1) % KSTEPS;
        constexpr int pa = prio_a(q), pb = prio_b(q);
        constexpr int ia = 4 * S * qn * 4;
        constexpr int ib = (WRT1 ? KSTEPS - 1 - qn : qn) * (4 * RW * 4);
        unsigned b0 = sw_b[0], b1 = sw_b[1], b2 = sw_b[2];
        if constexpr (qn == 0 || qn == KSTEPS - 1) {
          const bool ok = qn == 0 ? v_first : v_last;
          b0 = ok ? b0 : sw_zero;
          b1 = ok ? b1 : sw_zero;
          b2 = ok ? b2 : sw_zero;
        } else if (WRT1 && (q_slow & (1u << qn))) {   // wave-uniform, border strips only
          const bool ok = (v_mid & (1u << qn)) != 0;
          b0 = ok ? b0 : sw_zero;
          b1 = ok ? b1 : sw_zero;
          b2 = ok ? b2 : sw_zero;
        }
        const unsigned live_ = live, a_rd = sw_a;   // (asm operands must be locals of the lambda)
        if constexpr ((q & 1) == 0)
          asm volatile(UNFLOW_BWD4_KSTEP_P0 : : [live] "s"(live_), [a_rd] "v"(a_rd), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [ia] "i"(ia), [ib] "i"(ib), [pa] "i"(pa), [pb] "i"(pb)
                       : "memory", "scc", UNFLOW_BWD4_CLOBBERS);
        else
          asm volatile(UNFLOW_BWD4_KSTEP_P1 : : [live] "s"(live_), [a_rd] "v"(a_rd), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [ia] "i"(ia), [ib] "i"(ib), [pa] "i"(pa), [pb] "i"(pb)
                       : "memory", "scc", UNFLOW_BWD4_CLOBBERS);
      };
      // K-steps 0 and 1 with the LDS-DMA of row w + 1 between their MFMA groups (corr_bwd4_asm.inc)
      auto kstep_dma0 = [&]() __attribute__((always_inline)) {
        constexpr int ia = 4 * S * 1 * 4;
        constexpr int ib = (WRT1 ? KSTEPS - 2 : 1) * (4 * RW * 4);
        unsigned b0 = sw_b[0], b1 = sw_b[1], b2 = sw_b[2];
        if (WRT1 && (q_slow & 2u)) {
          const bool ok = (v_mid & 2u) != 0;
          b0 = ok ? b0 : sw_zero;
          b1 = ok ? b1 : sw_zero;
          b2 = ok ? b2 : sw_zero;
        }
        const unsigned live_ = live, a_rd = sw_a, fetch_ = (unsigned)__builtin_amdgcn_readfirstlane((int)fetch);
        const unsigned nstage = lds0 + ((rowc + 1) & 1) * (STAGE2 * 4);
        const unsigned a_m0 = nstage + a_dst0, t_m0 = nstage + t_dst0;
        const unsigned asoff = (unsigned)((S * wn + fc.cy) * W * 4), twoff = (unsigned)(wn * row_bytes);
        const unsigned long long amask = a_mask;
        const __amdgpu_buffer_rsrc_t rin = fc.rin, rtop = fc.rtop;
        const unsigned av0 = fc.a_voff[0], av1 = fc.a_voff[1], av2 = fc.a_voff[2], av3 = fc.a_voff[3];
        const unsigned tb0 = (unsigned)fc.t_base[0], tb1 = (unsigned)fc.t_base[1];
        asm volatile(UNFLOW_BWD4_KSTEP0_DMA
                     : : [live] "s"(live_), [a_rd] "v"(a_rd), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [ia] "i"(ia), [ib] "i"(ib),
                         [pa] "i"(prio_a(0)), [pb] "i"(prio_b(0)), [fetch] "s"(fetch_), [amask] "s"(amask), [a_m0] "s"(a_m0), [asoff] "s"(asoff), [rin] "s"(rin),
                         [av0] "v"(av0), [av1] "v"(av1), [av2] "v"(av2), [av3] "v"(av3),
                         [t_m0] "s"(t_m0), [twoff] "s"(twoff), [rtop] "s"(rtop), [tb0] "v"(tb0), [tb1] "v"(tb1)
                     : "memory", "scc", "vcc", "m0", UNFLOW_BWD4_CLOBBERS);   // (m0: the blocks write it in front of every LDS-DMA)
      };
      auto kstep_dma1 = [&]() __attribute__((always_inline)) {
        constexpr int ia = 4 * S * 2 * 4;
        constexpr int ib = (WRT1 ? KSTEPS - 3 : 2) * (4 * RW * 4);
        unsigned b0 = sw_b[0], b1 = sw_b[1], b2 = sw_b[2];
        if (WRT1 && (q_slow & 4u)) {
          const bool ok = (v_mid & 4u) != 0;
          b0 = ok ? b0 : sw_zero;
          b1 = ok ? b1 : sw_zero;
          b2 = ok ? b2 : sw_zero;
        }
        const unsigned live_ = live, a_rd = sw_a, fetch_ = (unsigned)__builtin_amdgcn_readfirstlane((int)fetch),
                       t3 = (unsigned)__builtin_amdgcn_readfirstlane((int)(fetch & t_has3));
        const unsigned t_m0 = lds0 + ((rowc + 1) & 1) * (STAGE2 * 4) + t_dst0, twoff = (unsigned)(wn * row_bytes);
        const __amdgpu_buffer_rsrc_t rtop = fc.rtop;
        const unsigned tb2 = (unsigned)fc.t_base[2], tb3 = (unsigned)fc.t_base[3];
        asm volatile(UNFLOW_BWD4_KSTEP1_DMA
                     : : [live] "s"(live_), [a_rd] "v"(a_rd), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [ia] "i"(ia), [ib] "i"(ib),
                         [pa] "i"(prio_a(1)), [pb] "i"(prio_b(1)), [fetch] "s"(fetch_), [t3] "s"(t3), [t_m0] "s"(t_m0), [twoff] "s"(twoff), [rtop] "s"(rtop), [tb2] "v"(tb2), [tb3] "v"(tb3)
                     : "memory", "scc", "m0", UNFLOW_BWD4_CLOBBERS);
      };
      // K-steps 3, 4: the pending tile store goes LAST among the vector-memory operations in front of the barrier
      // (behind the DMA), so that the barrier can wait for the DMA without the stores
      auto stores = [&](auto Q) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value;
        if constexpr (q == 3 || q == 4) {
          if (store_tile >= 0) {
            store_piece(store_tile, 2 * (q - 3));
            store_piece(store_tile, 2 * (q - 3) + 1);
          }
        }
      };
      if (w == 0) {
        // first row of an item: nothing ran ahead of it -- its first fragments (latency exposed once per item)
        constexpr int ib = (WRT1 ? KSTEPS - 1 : 0) * (4 * RW * 4);
        const unsigned b0 = v_first ? sw_b[0] : sw_zero, b1 = v_first ? sw_b[1] : sw_zero, b2 = v_first ? sw_b[2] : sw_zero;
        const unsigned a_rd = sw_a;
        asm volatile(UNFLOW_BWD4_FIRST_READS : : [a_rd] "v"(a_rd), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [ia] "i"(0), [ib] "i"(ib)
                     : "memory", UNFLOW_BWD4_CLOBBERS);
      }
      UNFLOW_STAMP(rowc, 1);
      kstep_dma0();
      kstep_dma1();
      kstep_block(Int<2>{});
      // (the tile stores stay C++ between the blocks: inside the blocks they measured 0.5 % slower -- their operands cost
      // more scalar instructions than the stores themselves)
      static_for<3, 5>([&](auto Q) {
        kstep_block(Q);
        stores(Q);
      });
      // Row barrier.  vmcnt counts loads, LDS-DMA and stores together, in issue order: the four tile
      // stores of this row are this wave's youngest operations, and waiting for their
      // acknowledgement (microseconds under load) is not needed for anything -- only the DMA of the
      // next row must have landed.  (Every wave issues all four: each holds lanes of tile row 0.)
      UNFLOW_STAMP(rowc, 2);
#if defined(UNFLOW_TUNE_NOROWBARRIER)        // timing only (races): the row barrier's round trip
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
#elif defined(UNFLOW_TUNE_NOROWWAIT)         // timing only (races): neither the barrier nor the wait for the DMA
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
      {
        // ARRIVAL at the row barrier's counter: this wave's DMA pieces of row w + 1 have landed (vmcnt counts loads,
        // LDS-DMA and stores together, in issue order: the four tile stores of a storing row are its youngest operations
        // and nothing waits for their acknowledgement), its last reads of row w's stage are in LDS's queue ahead of the
        // add (LDS serves a wave's operations in order).  One lane adds 1.  (Arriving earlier -- inside K-step 4, behind
        // its last fragment read -- measured the same: what is left of the barrier is the slowest wave of the row.)
        const unsigned one_ = 1u, bar_ = bar_addr;
        if (store_tile >= 0)
          asm volatile("s_waitcnt vmcnt(4)\n\ts_mov_b64 vcc, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, vcc" : : "v"(bar_), "v"(one_) : "memory", "vcc");
        else
          asm volatile("s_waitcnt vmcnt(0)\n\ts_mov_b64 vcc, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, vcc" : : "v"(bar_), "v"(one_) : "memory", "vcc");
        // the last row of an item keeps the hardware barrier: the next item's first fragments are read by a block of
        // their own (FIRST_READS), and the item-end epilogue works with s_barrier anyway
        if (!more) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
#endif
      UNFLOW_STAMP(rowc, 3);
      // every fragment of row w has been read: the sweep state moves on to row w + 1 (its stage is the other one)
      if (more) {
        const int dstage = (rowc & 1) ? -(STAGE2 * 4) : STAGE2 * 4;
        const int dstep = dstage - TRS * 4, dwrap = dstage + (WIN - 1) * (TRS * 4);
        unsigned tmp_;
        const unsigned oyl_ = (unsigned)oyl;
        // (the state is wave-uniform by construction; readfirstlane tells the compiler so)
#define UNFLOW_UNI(x) __builtin_amdgcn_readfirstlane((int)(x))
        int j0 = UNFLOW_UNI(sw_j[0]), j1 = UNFLOW_UNI(sw_j[1]), j2 = UNFLOW_UNI(sw_j[2]);
        int u0 = UNFLOW_UNI(sw_u[0]), u1 = UNFLOW_UNI(sw_u[1]), u2 = UNFLOW_UNI(sw_u[2]);
        int lv = UNFLOW_UNI(sw_live), zr = UNFLOW_UNI(sw_zero);
        const int tmax = UNFLOW_UNI(sw_tmax), ds0 = UNFLOW_UNI(dstage), ds1 = UNFLOW_UNI(dstep), ds2 = UNFLOW_UNI(dwrap);
#undef UNFLOW_UNI
        // K-step 5 of this row, the state's step and the first fragment reads of row w + 1 as ONE block (the state as a
        // block of its own behind the barrier met an empty matrix pipe in all eight waves at once: 3.5 % of the kernel)
        constexpr int ib = (WRT1 ? KSTEPS - 1 : 0) * (4 * RW * 4);
        const unsigned live_ = live;
        const unsigned long long vfirst_ = vfirst_mask;
        const unsigned bar_target = (unsigned)__builtin_amdgcn_readfirstlane(NWAVES * (rowc + 1));   // every wave has arrived for this row
        unsigned bf0_, bf1_, bf2_;
        asm volatile(UNFLOW_BWD4_KSTEP5_ROW
                     : [zero] "+s"(zr), [j0] "+s"(j0), [j1] "+s"(j1), [j2] "+s"(j2), [u0] "+s"(u0), [u1] "+s"(u1),
                       [u2] "+s"(u2), [lvn] "+s"(lv), [tmp] "=&s"(tmp_), [aaddr] "+v"(sw_a), [br0] "+v"(sw_run[0]), [br1] "+v"(sw_run[1]),
                       [br2] "+v"(sw_run[2]), [bu0] "=&v"(sw_b[0]), [bu1] "=&v"(sw_b[1]), [bu2] "=&v"(sw_b[2]),
                       [bf0] "=&v"(bf0_), [bf1] "=&v"(bf1_), [bf2] "=&v"(bf2_)
                     : [live] "s"(live_), [dstage] "s"(ds0), [dstep] "s"(ds1), [dwrap] "s"(ds2), [tmax] "s"(tmax), [oyl] "v"(oyl_),
                       [vfirst] "s"(vfirst_), [ib] "i"(ib), [pa] "i"(prio_a(KSTEPS - 1)), [pb] "i"(prio_b(KSTEPS - 1)),
                       [bar] "v"(bar_addr), [target] "s"(bar_target)
                     : "memory", "vcc", "scc", UNFLOW_BWD4_CLOBBERS);
        sw_j[0] = j0; sw_j[1] = j1; sw_j[2] = j2;
        sw_u[0] = u0; sw_u[1] = u1; sw_u[2] = u2;
        sw_live = (unsigned)lv; sw_zero = (unsigned)zr;
      } else {
        kstep_block(Int<KSTEPS - 1>{});   // last row of an item: no next row (the next item's first fragments: FIRST_READS)
      }
      // ---- a tile that ends on this row (not the last one: 4 t + 13 == w): its owners write it to ob
      // (free: its last image was stored two rows ago); the next barrier publishes it and the row after
      // that stores it inside its K-steps 3, 4.
      store_tile = dumped;
      dumped = -1;
#if defined(UNFLOW_TUNE_NOSTORE) || defined(UNFLOW_TUNE_NS_LOOP)
      if (C < 0)                               // diagnostic: never true at run time, opaque to the compiler
#endif
      if (more) {
        const int t = (w - (TP - 1 + R)) >> 2;
        if (w >= TP - 1 + R && ((w - (TP - 1 + R)) & 3) == 0 && t < ntiles) {
          if ((t & 1) == ph) UNFLOW_DUMP_TILE4((t >> 1) % RING);
          dumped = t;
        }
      }
#if defined(UNFLOW_TUNE_NOSTORE) || defined(UNFLOW_TUNE_NS_END)
      if (C > 0) continue;
#endif
      if (!more) {
        // last row: the tile published by this row's barrier is still to be stored, and every tile that
        // is still live ends here -- (store,) barrier, dump, barrier, store, ...
        if (store_tile >= 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) store_piece(store_tile, q);
          UNFLOW_BARRIER_LDS();                  // ob is rewritten below
        }
        const int t_first = max(0, (w - (TP - 1 + R) + 3) >> 2);     // first t with 4 t + 13 >= w
        for (int t = t_first; t < ntiles; ++t) {
          if ((t & 1) == ph) UNFLOW_DUMP_TILE4((t >> 1) % RING);
          UNFLOW_BARRIER_LDS();
#pragma unroll
          for (int q = 0; q < 4; ++q) store_piece(t, q);
          if (t + 1 < ntiles) UNFLOW_BARRIER_LDS();   // ob is rewritten by the next tile
        }
      }
    }
  }
#undef UNFLOW_DUMP_TILE4
#undef UNFLOW_BARRIER_DMA
#undef UNFLOW_BARRIER_LDS
}

}  // namespace
}  // namespace unflow

#ifdef UNFLOW_TUNE_STAMPS
extern "C" int unflow_debug_read_stamps(void *dst, size_t bytes) {
  const size_t n = sizeof(unflow::g_stamps) < bytes ? sizeof(unflow::g_stamps) : bytes;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(unflow::g_stamps), n, 0, hipMemcpyDeviceToHost);
}
#endif

namespace unflow {

int corr_mfma_bwd(const float *topdiff, const float *in0, const float *in1, float *g0, float *g1, int B, int C,
                  const CorrGeometry &g, hipStream_t s) {
  const int H = g.H, W = g.W;
  const int ws = (W + S - 1) / S;
  const int nbx = (ws + RX - 1) / RX, nch = (C + CWS - 1) / CWS;
  const bool wide = (W % 4) == 0 && aligned16(in0) && aligned16(in1) && aligned16(topdiff) && aligned16(g0) && aligned16(g1);
#ifndef UNFLOW_TUNE_OLD_BWD
  if (wide && (C % CWS) == 0 && H >= 4) {   // (its DMA runs two class rows ahead: a row class has >= 2 rows)
    // persistent workgroups, one per CU, walking the (pair, row class, channel group, column strip) items; one launch
    // per gradient
    const int nitems = nbx * nch * S * B;
    const int nwg = nitems < num_cus() ? nitems : num_cus();
    const size_t smem2 = b2::LDS_BYTES2;
    if (int rc = reserve_lds<corr_mfma_bwd4_kernel<false>>(smem2, "correlation_bwd (mfma)")) return rc;
    if (int rc = reserve_lds<corr_mfma_bwd4_kernel<true>>(smem2, "correlation_bwd (mfma)")) return rc;
    hipLaunchKernelGGL(corr_mfma_bwd4_kernel<false>, dim3(nwg), dim3(NTHREADS), smem2, s, topdiff, in1, g0, C, H, W, nbx, nch, nitems);
    hipLaunchKernelGGL(corr_mfma_bwd4_kernel<true>, dim3(nwg), dim3(NTHREADS), smem2, s, topdiff, in0, g1, C, H, W, nbx, nch, nitems);
    return launch_status("correlation_bwd (mfma)");
  }
#endif
  return corr_mfma_bwd_r1(topdiff, in0, in1, g0, g1, B, C, g, s);   // other channel counts / unaligned rows
}

const char *tune_flags_correlation_bwd4() {
  return ""
#ifdef UNFLOW_TUNE_STAMPS
         " UNFLOW_TUNE_STAMPS"
#endif
#ifdef UNFLOW_TUNE_NODMA
         " UNFLOW_TUNE_NODMA"
#endif
#ifdef UNFLOW_TUNE_NOSTORE
         " UNFLOW_TUNE_NOSTORE"
#endif
#ifdef UNFLOW_TUNE_OLD_BWD
         " UNFLOW_TUNE_OLD_BWD"
#endif
#ifdef UNFLOW_TUNE_NOROWBARRIER
         " UNFLOW_TUNE_NOROWBARRIER"
#endif
#ifdef UNFLOW_TUNE_NOROWWAIT
         " UNFLOW_TUNE_NOROWWAIT"
#endif
#ifdef UNFLOW_TUNE_PRIO_TABLE
         " UNFLOW_TUNE_PRIO_TABLE"
#endif
#ifdef UNFLOW_TUNE_NOGLOBAL
         " UNFLOW_TUNE_NOGLOBAL"
#endif
#ifdef UNFLOW_TUNE_NS_LOOP
         " UNFLOW_TUNE_NS_LOOP"
#endif
#ifdef UNFLOW_TUNE_NS_END
         " UNFLOW_TUNE_NS_END"
#endif
      ;
}

}  // namespace unflow
