// Does buffer_load_dwordx4 ... lds work when the LDS destination is only 4-byte aligned?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k(const float *src, float *out, int shift) {
  __shared__ __attribute__((aligned(16))) float lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), (short)0, 4096, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds + shift, 16, threadIdx.x * 16, 0, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  float *src, *out; CK(hipMalloc(&src, 4096)); CK(hipMalloc(&out, 4096));
  std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  CK(hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice));
  for (int shift = 0; shift < 4; ++shift) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, shift);
    CK(hipDeviceSynchronize());
    std::vector<float> o(1024); CK(hipMemcpy(o.data(), out, 4096, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) if (o[i + shift] != (float)i) ++bad;
    printf("shift %d floats: %d of 256 wrong; first values:", shift, bad);
    for (int i = 0; i < 10; ++i) printf(" %g", o[i]);
    printf("\n");
  }
  return 0;
}
