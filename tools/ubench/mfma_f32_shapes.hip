// Micro-benchmark: the two f32 MFMA shapes of gfx950 in a loop shaped like the correlation forward's channel sweep --
// four waves per SIMD (two 512-thread workgroups per CU), every operand re-read from LDS (random data) with
// ds_read_b64 whose two floats feed two MFMAs (the two column classes), one barrier per 8 channels.
//   16x16x4: per 4 channels 1 A + 9 B reads -> 18 MFMAs (18 accumulator tiles = 72 registers: corr_fwd2_kernel's loop)
//   32x32x2: per 4 channels 2 x (1 A + 2 B reads -> 4 MFMAs) (4 accumulator tiles = 64 registers)
// Prints raw TFLOP/s, the clock the chip held (s_memtime / s_memrealtime) and the USEFUL rate for the correlation:
// 441/576 of the entries of a 4x4-position tile's 24x24 window are wanted with 16x16 tiles, 441/672 of a 4x8 tile's
// 24x28 window with 32x32 tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int CHUNKS = 256;           // 8 channels each
constexpr int LDS_FLOATS = 17024;     // 2 stages of corr_fwd2_kernel

__global__ __launch_bounds__(512, 4) void k16(const float *src, float *out, unsigned long long *clk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < LDS_FLOATS; i += 512) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane >> 4, n = lane & 15;
  const float *a0 = lds + 7424 + k * 136 + ((wave >> 2) * 4 + (n >> 2)) * 16 + 2 * (((wave >> 1) & 1) * 4 + (n & 3));
  const float *b0 = lds + k * 928 + ((wave >> 2) * 4 + (n >> 2)) * 56 + 2 * (((wave >> 1) & 1) * 4 + (n & 3)) + 24 * (wave & 1);
  f32x4 acc[18];
#pragma unroll
  for (int t = 0; t < 18; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int ch = 0; ch < CHUNKS; ++ch) {
    const int st = (ch & 1) * 8512;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const f32x2 a = *reinterpret_cast<const f32x2 *>(a0 + st + kk * 4 * 136);
      f32x2 b[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) b[t] = *reinterpret_cast<const f32x2 *>(b0 + st + kk * 4 * 928 + (t / 3) * 4 * 56 + (t % 3) * 8);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        acc[2 * t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t].x, acc[2 * t], 0, 0, 0);
        acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t].y, acc[2 * t + 1], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int t = 0; t < 18; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (s == 1234.5f) out[0] = s;
  if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

__global__ __launch_bounds__(512, 4) void k32(const float *src, float *out, unsigned long long *clk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < LDS_FLOATS; i += 512) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane >> 5, n = lane & 31;
  // A: 32 positions = 8 rows x 4 columns of one tile column; B: 4 window rows x 8 columns
  const float *a0 = lds + 7424 + k * 136 + (n >> 2) * 16 + 2 * (((wave >> 1) & 1) * 4 + (n & 3));
  const float *b0 = lds + k * 928 + ((wave >> 2) * 4 + (n >> 3)) * 56 + 2 * (((wave >> 1) & 1) * 4 + (n & 7)) + 16 * (wave & 1);
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int ch = 0; ch < CHUNKS; ++ch) {
    const int st = (ch & 1) * 8512;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {          // 2 channels per step
      const f32x2 a = *reinterpret_cast<const f32x2 *>(a0 + st + kk * 2 * 136);
      f32x2 b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) b[t] = *reinterpret_cast<const f32x2 *>(b0 + st + kk * 2 * 928 + t * 4 * 56);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[2 * t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[2 * t], 0, 0, 0);
        acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[2 * t + 1], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[t][i];
  if (s == 1234.5f) out[0] = s;
  if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <class K>
int run(const char *name, K kern, double flop_per_wave_chunk, double useful, const float *src, float *out, unsigned long long *clk) {
  const int blocks = 256 * 2 * 8, smem = LDS_FLOATS * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), smem, 0, src, out, clk);      // settle the clock
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), smem, 0, src, out, clk);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  unsigned long long h[2]; CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
  const double flops = flop_per_wave_chunk * 8 * blocks * CHUNKS;
  printf("%-12s %7.3f ms  %6.1f TFLOP/s raw (%.3f of 157.3), clock %.0f MHz, useful for the correlation %6.1f TFLOP/s (x %.3f)\n", name, ms,
         flops / ms / 1e9, flops / ms / 1e9 / 157.3, (double)h[0] / (double)h[1] * 100.0, flops / ms / 1e9 * useful, useful);
  return 0;
}

int main() {
  float *src, *out; unsigned long long *clk;
  CK(hipMalloc(&src, LDS_FLOATS * 4)); CK(hipMalloc(&out, 1024)); CK(hipMalloc(&clk, 64));
  float *h = (float *)malloc(LDS_FLOATS * 4);
  srand(1);
  for (int i = 0; i < LDS_FLOATS; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  CK(hipMemcpy(src, h, LDS_FLOATS * 4, hipMemcpyHostToDevice));
  for (int r = 0; r < 2; ++r) {
    if (run("16x16x4", k16, 36.0 * 2048, 441.0 / 576, src, out, clk)) return 1;
    if (run("32x32x2", k32, 16.0 * 4096, 441.0 / 672, src, out, clk)) return 1;
  }
  return 0;
}
