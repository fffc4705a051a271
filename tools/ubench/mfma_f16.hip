// Micro-benchmark behind the f16 matrix-core correlation (BASELINE config 5):
//  (1) issue rate of the f16 MFMA shapes on gfx950,
//  (2) the lane map of ds_read_b64_tr_b16 on a [k][position] LDS image (what lets an NCHW-ordered
//      image feed the K-contiguous 16-bit MFMA operands without a transposed copy),
//  (3) the rate of a loop shaped like the forward kernel: one A fragment and NB fresh B fragments
//      (transposed reads) per NB MFMAs, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 p4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
constexpr int ITER = 2000;

__device__ inline h4 tr_read(const _Float16 *p) {
  return __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) p4 *)(p)));
}

// SHAPE 0: 16x16x16 f16, 1: 16x16x32 f16, 2: 32x32x8 f16, 3: 32x32x16 f16
template <int NACC, int SHAPE>
__global__ __launch_bounds__(512, 2) void rate_kernel(float *out, float seed) {
  h4 a4, b4;
  h8 a8, b8;
  for (int i = 0; i < 4; ++i) { a4[i] = (_Float16)(seed + i + threadIdx.x * 0.001f); b4[i] = (_Float16)(seed * 2 + i); }
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(seed + i + threadIdx.x * 0.001f); b8[i] = (_Float16)(seed * 2 + i); }
  float s = 0;
  if constexpr (SHAPE < 2) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if constexpr (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if constexpr (SHAPE == 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5] + acc[i][10] + acc[i][15];
  }
  if (s == 1234.5f) out[0] = s;
}

template <int NACC, int SHAPE>
int run_rate(const char *name, float *d, int threads, double flop_per) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 256 * 4;
  hipLaunchKernelGGL((rate_kernel<NACC, SHAPE>), dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((rate_kernel<NACC, SHAPE>), dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma = (double)blocks * (threads / 64) * ITER * NACC;
  printf("%-28s acc %2d waves/WG %d: %7.3f ms  %7.1f TFLOP/s  %5.1f cycles per MFMA per SIMD (2.4 GHz)\n", name, NACC,
         threads / 64, ms, mfma * flop_per / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / mfma);
  return 0;
}

// (2) lane map: image [16 k][64 positions] of halves with value k * 64 + pos; every lane passes the address the
// guide prescribes for a block of 4 k-rows x 16 columns: lane 4q+p of a 16-lane group -> row q, columns 4p..4p+3.
__global__ void tr_map_kernel(float *out) {
  __shared__ __attribute__((aligned(16))) _Float16 img[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) img[i] = (_Float16)(float)i;
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  // group g reads k rows 4g..4g+3, columns 16g' ... here all groups take columns 0..15
  const h4 v = tr_read(img + (4 * g + q) * 64 + 4 * p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (float)v[j];
}

// (3) forward-shaped loop.  LDS image [KCH channels][ROWS][XW halves]; per "chunk" a wave reads one A fragment and
// NB B fragments (a 4x4 tile of positions each: lane 4q+p -> channel 4(lane>>4)+q, tile row p, 4 contiguous columns)
// and issues NB MFMAs 16x16x16.
template <int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void loop_kernel(float *out, int iters) {
  constexpr int XW = 32, ROWS = 16, PL = ROWS * XW + 4;   // halves; plane stride = 8 bytes mod 256
  __shared__ __attribute__((aligned(16))) _Float16 img[16 * PL + 64];
  for (int i = threadIdx.x; i < 16 * PL + 64; i += WAVES * 64) img[i] = (_Float16)((i % 97) * 0.01f);
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const _Float16 *base = img + (4 * g + q) * PL + p * XW;
  f32x4 acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const h4 a = tr_read(base + (it & 1) * 4);
    h4 b[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) b[t] = tr_read(base + (t / 6) * 4 * XW + (t % 6) * 4);
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b[t], acc[t], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NB; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1234.5f) out[0] = s;
}

template <int NB, int WAVES>
int run_loop(float *d, int wg_per_cu) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 256 * wg_per_cu, iters = 4000;
  hipLaunchKernelGGL((loop_kernel<NB, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, d, iters);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((loop_kernel<NB, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, d, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma = (double)blocks * WAVES * iters * NB;
  printf("fwd-shaped loop: %2d B frags + 1 A per %2d MFMA(16x16x16 f16), %d waves/WG, %d WG/CU: %7.3f ms  %7.1f TFLOP/s  %5.1f cycles per MFMA per SIMD\n",
         NB, NB, WAVES, wg_per_cu, ms, mfma * 8192 / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / mfma);
  return 0;
}

int main() {
  float *d; CK(hipMalloc(&d, 4096));
  run_rate<18, 0>("16x16x16 f16", d, 512, 8192);  run_rate<18, 1>("16x16x32 f16", d, 512, 16384);
  run_rate<4, 2>("32x32x8 f16", d, 512, 16384);    run_rate<4, 3>("32x32x16 f16", d, 512, 32768);
  run_rate<18, 0>("16x16x16 f16", d, 256, 8192);  run_rate<18, 1>("16x16x32 f16", d, 256, 16384);
  run_rate<2, 0>("16x16x16 f16", d, 256, 8192);   run_rate<2, 1>("16x16x32 f16", d, 256, 16384);

  hipLaunchKernelGGL(tr_map_kernel, dim3(1), dim3(64), 0, 0, d);
  std::vector<float> h(256);
  CK(hipMemcpy(h.data(), d, 256 * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int j = 0; j < 4; ++j) {
      // expectation (guide T10): lane i of group g receives column i of the block's 4 rows, row q in element q
      const int g = lane >> 4, i = lane & 15;
      const float want = (float)((4 * g + j) * 64 + i);
      if (h[lane * 4 + j] != want) ++bad;
    }
  printf("ds_read_b64_tr_b16 lane map: %s (lane 0: %g %g %g %g; lane 17: %g %g %g %g)\n", bad ? "DIFFERS from T10" : "as T10 says",
         h[0], h[1], h[2], h[3], h[68], h[69], h[70], h[71]);

  run_loop<18, 8>(d, 2); run_loop<18, 8>(d, 1); run_loop<18, 4>(d, 2); run_loop<36, 8>(d, 1);
  return 0;
}
