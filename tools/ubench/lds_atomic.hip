// Micro-benchmark: LDS atomic-add rates on gfx950 (decides the forward_warp design).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int N = 4096;     // LDS floats per block
constexpr int ITER = 2048;  // atomic wave-instructions per wave

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int stride, int shiftdiv) {
  __shared__ float lds[N];
  __shared__ unsigned ldu[N];
  __shared__ unsigned long long ldq[N / 2];
  for (int i = threadIdx.x; i < N; i += 256) { lds[i] = 0.f; ldu[i] = 0u; if (i < N / 2) ldq[i] = 0ull; }
  __syncthreads();
  const int lane = threadIdx.x;
  float w = 1.0f + lane * 1e-3f;
  for (int it = 0; it < ITER; ++it) {
    const int a = ((lane / shiftdiv) * stride + it * 7) & (N - 1);
    if (MODE == 0) atomicAdd(&lds[a], w);                                   // ds_add_f32
    if (MODE == 1) atomicAdd(&ldu[a], (unsigned)lane);                      // ds_add_u32
    if (MODE == 2) { float v = atomicAdd(&lds[a], w); w += v * 1e-20f; }    // ds_add_rtn_f32
    if (MODE == 3) { lds[a] += w; }                                         // plain RMW (racy, timing only)
    if (MODE == 4) atomicAdd(&ldq[a >> 1], (unsigned long long)lane);       // ds_add_u64
    if (MODE == 5) __hip_atomic_fetch_add(&lds[a], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) s += lds[i] + (float)ldu[i] + (i < N / 2 ? (float)ldq[i] : 0.f);
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
int run(const char *name, float *d, int stride, int shiftdiv) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 256 * 8;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, stride, shiftdiv);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, stride, shiftdiv);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double waveinstr = (double)blocks * 4 * ITER;
  const double cyc_per_cu = ms * 1e-3 * 2.4e9;  // nominal clock
  printf("%-34s stride %2d shiftdiv %d: %8.3f ms  %6.1f G lane-atomics/s  %6.1f cycles per wave-instr per CU\n", name, stride,
         shiftdiv, ms, waveinstr * 64 / ms / 1e6, cyc_per_cu / (waveinstr / 256));
  return 0;
}

int main() {
  float *d; CK(hipMalloc(&d, 1024));
  for (int sd = 1; sd <= 2; ++sd) {
    run<0>("ds_add_f32 (atomicAdd float)", d, 1, sd);
    run<5>("ds_add_f32 (__hip_atomic wg scope)", d, 1, sd);
    run<1>("ds_add_u32", d, 1, sd);
    run<2>("ds_add_rtn_f32", d, 1, sd);
    run<3>("plain read-add-write", d, 1, sd);
    run<4>("ds_add_u64", d, 1, sd);
  }
  run<0>("ds_add_f32", d, 32, 1);
  run<1>("ds_add_u32", d, 32, 1);
  return 0;
}
