// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 on gfx950 as a function of
// operand sharing, accumulator count and waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 2000;

// MODE 0: A shared, B varies (forward kernel shape); 1: B shared, A varies (backward shape); 2: both vary
template <int NACC, int MODE>
__global__ __launch_bounds__(512, 2) void k(float *out, float seed) {
  f32x4 acc[NACC];
  float av[NACC], bv[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) { acc[i] = f32x4{0, 0, 0, 0}; av[i] = seed + i + threadIdx.x; bv[i] = seed * 2 + i; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const float a = MODE == 0 ? av[0] : av[i];
      const float b = MODE == 1 ? bv[0] : bv[i];
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1234.5f) out[0] = s;
}

template <int NACC, int MODE>
int run(const char *name, float *d, int threads) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 256 * 4;
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma = (double)blocks * (threads / 64) * ITER * NACC;
  const double flops = mfma * 16 * 16 * 4 * 2;
  // cycles per MFMA per SIMD at 2.4 GHz nominal: total SIMD-cycles / MFMAs
  printf("%-34s acc %2d waves/WG %d: %7.3f ms  %6.1f TFLOP/s  %5.1f cycles per MFMA per SIMD (2.4 GHz)\n", name, NACC,
         threads / 64, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / mfma);
  return 0;
}

int main() {
  float *d; CK(hipMalloc(&d, 1024));
  run<36, 0>("A shared, B varies", d, 512); run<36, 1>("B shared, A varies", d, 512); run<36, 2>("both vary", d, 512);
  run<8, 0>("A shared, B varies", d, 512);  run<8, 1>("B shared, A varies", d, 512);  run<8, 2>("both vary", d, 512);
  run<24, 0>("A shared, B varies", d, 512); run<24, 1>("B shared, A varies", d, 512);
  run<36, 0>("A shared, B varies", d, 256); run<36, 1>("B shared, A varies", d, 256);
  run<4, 2>("both vary", d, 512); run<2, 2>("both vary", d, 512);
  return 0;
}
