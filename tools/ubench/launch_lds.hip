// How long does an (almost) empty kernel take as a function of dynamic LDS size and block count?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(512, 2) void k(float *out, int spin) {
  extern __shared__ float lds[];
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (threadIdx.x == 0) lds[0] = v;
  __syncthreads();
  if (lds[0] == 1234.5f) out[0] = v;
}
int main() {
  float *d; CK(hipMalloc(&d, 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int ldss[] = {1024, 32 * 1024, 64 * 1024, 100 * 1024, 148 * 1024, 160 * 1024};
  for (int spin : {0, 20000}) for (int lds : ldss) for (int blocks : {256, 1280, 2560}) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, 0, d, spin);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, 0, d, spin);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("spin %6d lds %6d B blocks %5d: %8.3f ms  (%.2f us per round of 256 blocks)\n", spin, lds, blocks, ms, ms * 1e3 / ((blocks + 255) / 256));
  }
  return 0;
}
