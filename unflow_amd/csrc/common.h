// Shared host/device helpers for libunflow_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/unflow_hip.h"

namespace unflow {

constexpr int kWave = 64;  // CDNA wavefront

// Thread-local last-error string behind unflow_last_error().
char *error_buffer();
int fail(int code, const char *fmt, ...);

// The UNFLOW_TUNE_* diagnostic knobs a translation unit was compiled with ("" for a production build); api.hip
// joins them behind unflow_build_flags() so that a diagnostic build cannot pass for the real library.
const char *tune_flags_correlation_mfma();
const char *tune_flags_correlation_fwd2();
const char *tune_flags_correlation_bwd4();
const char *tune_flags_warp();

// hipGetLastError() after a launch; never synchronises.
inline int launch_status(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(UNFLOW_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return UNFLOW_OK;
}

// Final pass of every loss reduction (census.hip): *loss = sum(partial[0..n)) * inv_norm,
// one block, fixed order, double accumulation.
int reduce_partials(const float *partial, int n, double inv_norm, float *loss, hipStream_t s);
// *loss = 0 for an empty tensor (reduce_sum of nothing; TF would go on to divide 0 by 0): an empty segment when a
// reduce plan is open on this thread
int zero_loss(float *loss, const char *what, hipStream_t s);

inline hipStream_t as_stream(unflow_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline unsigned blocks_for(long long n, int block) { return (unsigned)((n + block - 1) / block); }

// Number of per-block partial sums the reductions use (fixed so the final pass
// is one 1024-thread block and the summation order never depends on the size).
constexpr int kReduceBlocks = 2048;

// Separately rounded product / sum.  TensorFlow evaluates the reference's Python
// graphs op by op, so `a*b + c` there is two roundings; hipcc contracts `x*y`
// followed by `+` into an FMA by default (and HIP's __fmul_rn is a plain `x*y`),
// so the contraction is switched off for exactly these operations.
__device__ inline float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ inline float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}

// b^a for b > 0 through the hardware log2 / exp2 (1 ulp each): relative error below
// 1e-6 for the charbonnier bases (>= eps^2 = 1e-6), a tenth of the loss tolerance;
// powf() costs ~40 instructions and made the elementwise loss kernels ALU bound.
__device__ inline float pow_pos(float b, float a) { return __builtin_amdgcn_exp2f(a * __builtin_amdgcn_logf(b)); }

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

// Sum over a block of NT threads (NT multiple of 64); result valid in thread 0.
template <int NT>
__device__ inline float block_sum(float v, float *lds /* NT/64 floats */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) lds[wid] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NT / 64; i++) r += lds[i];
  }
  return r;
}

}  // namespace unflow
