// Shared by correlation_mfma.hip (first-generation FlowNetC kernels: fallback shapes and fp16 storage) and
// correlation_bwd4.hip (the shipped float32 backward): problem constants, the LDS-DMA helper, host-side launch helpers.
// Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "common.h"
#include "correlation.h"

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

namespace unflow {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int S = 2;            // stride_2
constexpr int R = 10;           // neighbourhood grid radius (max_displacement / stride_2)
constexpr int GW = 2 * R + 1;   // 21
constexpr int TP = 4;           // tile of class positions is TP x TP (= MFMA M 16)
constexpr int WIN = TP + 2 * R; // 24: window of one tile
constexpr int TQ = WIN / 4;     // 6 N-tiles per dimension
constexpr int NT = TQ * TQ;     // 36 N-tiles = accumulators per wave
constexpr int RY = 8, RX = 8;   // region of one class in a workgroup (2x2 tiles)
constexpr int LW = RX + 2 * R;  // 28 staged window columns per class
constexpr int LH = RY + 2 * R;  // 28 staged window rows
constexpr int ROWF = S * LW;    // 56 floats: one staged row, full resolution (both classes)
constexpr int RW = S * RX;      // 16 full-resolution columns of the output region
constexpr int NTHREADS = 512;
constexpr int NWAVES = NTHREADS / 64;
constexpr unsigned OOB = 0x80000000u;   // beyond any num_records (< 2 GiB): reads as 0
static_assert(WIN % 4 == 0, "window must tile by 4");

template <int N>
using Int = std::integral_constant<int, N>;

// f(Int<A>{}), f(Int<A+1>{}), ..., f(Int<B-1>{})
template <int A, int B, class F>
__device__ inline void static_for(F &&f) {
  if constexpr (A < B) {
    f(Int<A>{});
    static_for<A + 1, B>(f);
  }
}
// index of the nth set bit of mask (compile-time use)
constexpr int nth_live(int mask, int nth) {
  for (int b = 0; b < 31; ++b)
    if ((mask >> b) & 1) {
      if (nth == 0) return b;
      --nth;
    }
  return 0;
}

__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const float *base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), (short)0, (int)bytes, 0x00020000);
}

// One LDS-DMA wave instruction: 64 lanes x VEC floats, LDS destination linear.
template <int VEC>
__device__ inline void dma(__amdgpu_buffer_rsrc_t r, float *lds_dst, unsigned voff, int soff) {
  // (the instruction exists for 1, 2, 4, 12 and 16 bytes per lane only)
  static_assert(VEC == 4 || VEC == 1, "LDS-DMA moves 4 or 16 bytes per lane here");
  if constexpr (VEC == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_dst, 16, (int)voff, soff, 0, 0);
  else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_dst, 4, (int)voff, soff, 0, 0);
}

// Workgroup barrier that publishes an LDS-DMA stage.  Written out: the compiler's wait-count
// insertion does not know that a buffer_load ... lds has a side in LDS -- depending on where it
// stands a plain __syncthreads() came out with vmcnt(9) or with no vmcnt wait at all.
__device__ inline void dma_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// XCD-aware block order: hardware deals consecutive block ids round-robin to the
// 8 XCDs; remap so that each XCD walks a contiguous range of regions (one image
// pair's windows overlap heavily and then share that XCD's L2).
__device__ inline int xcd_remap(int b, int nblk) { return (nblk & 7) == 0 ? (b & 7) * (nblk >> 3) + (b >> 3) : b; }

// ---- row sweep of the backward kernels (both generations): see correlation_mfma.hip "Backward"
constexpr int MTS = 8;                          // M-tiles (of 16 channels) per workgroup
constexpr int CWS = MTS * 16;                   // 128 channels per workgroup
constexpr int ACS = ROWF + 4;                   // 60: channel stride of the A image (4 pad floats)
// A fragment (16 channels x 2 window columns per half): c*60 mod 32 walks the 8
// multiples of 4 twice, +2k takes the other even banks: 2-way (see forward note).
static_assert(ACS % 8 == 4, "A image channel stride must be 4*odd");
constexpr int SA_STAGE = CWS * ACS;             // 7680 floats
constexpr int TROWS = 2 * R + 1;                // 21 output rows pair with one window row
constexpr int ST_IMAGE = TROWS * GW * RW;       // 7056 floats: [row 21][dx' 21][16 columns]
constexpr int ST_STAGE = (ST_IMAGE + 255) / 256 * 256;   // 7168: whole 1-KiB DMA pieces
constexpr int SW_STAGE = SA_STAGE + ST_STAGE;   // 14848 floats = 58 KB
constexpr int SW_OUT = CWS * TP * RW;           // 8192 floats: epilogue image [channel][row 4][16 columns]
constexpr int SW_LDS = 2 * SW_STAGE + SW_OUT;   // 37888 floats = 148 KB
constexpr int KSTEPS = WIN / 4;                 // 6 K-steps per window row and tile
constexpr int RING = 3;                         // live tiles per wave

#ifdef UNFLOW_TUNE_STAMPS
// Diagnostic build only (tools/variants.sh): shader-clock stamps of a few workgroups, read back
// through unflow_debug_read_stamps().  [block 0..STAMP_BLOCKS)[wave 0 / wave 4][row][stamp]
constexpr int STAMP_BLOCKS = 8, STAMP_ROWS = 128, STAMP_PER_ROW = 4;
__device__ unsigned long long g_stamps[STAMP_BLOCKS * 2 * (STAMP_ROWS + 1) * STAMP_PER_ROW];
#define UNFLOW_STAMP(ROW, WHICH)                                                                            \
  do {                                                                                                      \
    if (stamp_slot >= 0 && (threadIdx.x & 63) == 0 && (ROW) <= STAMP_ROWS)                                  \
      g_stamps[(stamp_slot * (STAMP_ROWS + 1) + (ROW)) * STAMP_PER_ROW + (WHICH)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define UNFLOW_STAMP(ROW, WHICH) do {} while (0)
#endif

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// > 64 KiB of dynamic LDS needs an opt-in, per kernel AND per device (the attribute is set on
// the current device's copy of the function).  The kernel is a template VALUE parameter: all
// instantiations of one kernel template share a function type, and a flag keyed on the type
// would be set for the first only.  A failure is returned, never remembered.
constexpr int kMaxDevices = 64;
// compute units of the current device (the persistent kernels launch one workgroup per CU)
inline int num_cus() {
  static std::atomic<int> cached[kMaxDevices];
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev >= 0 && dev < kMaxDevices && (n = cached[dev].load(std::memory_order_relaxed)) > 0) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (dev >= 0 && dev < kMaxDevices) cached[dev].store(n, std::memory_order_relaxed);
  return n;
}
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
}  // namespace unflow
