// Micro-benchmark behind DESIGN.md "split-precision matrix path" (VERDICT r1 item 5): what one
// 16 x 16 output tile costs per 32 channels on one SIMD of gfx950
//   (a) as 8 x v_mfma_f32_16x16x4_f32 (exact f32: what the product kernels do), and
//   (b) as 6 x v_mfma_f32_16x16x32_bf16 on operands split into three bf16 terms each
//       (hi, mid, lo: 3 x 8 = 24 mantissa bits; products hi.hi hi.mid mid.hi hi.lo lo.hi mid.mid),
// with the operands arriving as f32 from LDS exactly as the forward kernel's fragments do (one
// f32 per lane, channel and tile) and the split done in registers.  NT = output tiles that share
// one A fragment (18 in the half-window forward kernel): the A split is amortised, the B split is not.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int ITER = 400, NT = 18;

struct Split { bf16x8 hi, mid, lo; };
__device__ inline Split split8(const float (&x)[8]) {
  Split s;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)x[i];
    const float r1 = x[i] - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    s.hi[i] = h; s.mid[i] = m; s.lo[i] = (__bf16)r2;
  }
  return s;
}

template <int MODE>   // 0: f32 MFMA, 1: bf16 x3 split, 2: split only (no MFMA), 3: bf16 MFMAs only (operands pre-split once)
__global__ __launch_bounds__(512, 2) void k(float *out, const float *in) {
  __shared__ float lds[8 * 64 * (NT + 1)];
  for (int i = threadIdx.x; i < 8 * 64 * (NT + 1); i += 512) lds[i] = in[i % 4096] * (1.0f + 1e-3f * (i & 7));
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0, 0, 0, 0};
  Split pa{}, pb{};
  for (int it = 0; it < ITER; ++it) {
    float a[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = lds[(c * (NT + 1)) * 64 + ((lane + it) & 63)];
    if (MODE == 1 || MODE == 2 || (MODE == 3 && it == 0)) pa = split8(a);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float b[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) b[c] = lds[(c * (NT + 1) + t + 1) * 64 + ((lane + it) & 63)];
      if (MODE == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], b[c], acc[t], 0, 0, 0);
      } else {
        if (MODE == 1 || MODE == 2 || (MODE == 3 && it == 0)) pb = split8(b);
        if (MODE == 2) {
          acc[t][0] += (float)pb.hi[0] + (float)pb.mid[3] + (float)pb.lo[7] + (float)pa.lo[1];
        } else {
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa.hi, pb.hi, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa.hi, pb.mid, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa.mid, pb.hi, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa.hi, pb.lo, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa.lo, pb.hi, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa.mid, pb.mid, acc[t], 0, 0, 0);
        }
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
int run(const char *name, float *d, const float *in) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 256 * 2;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, in);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, in);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  // tile-steps (one 16x16 tile x 32 channels) per SIMD: blocks * 8 waves * ITER * NT / 1024 SIMDs
  const double steps = (double)blocks * 8 * ITER * NT / 1024.0;
  printf("%-46s %8.3f ms  %7.1f cycles per tile and 32 channels per SIMD (2.4 GHz)  = %6.1f exact-f32-equivalent TFLOP/s\n", name, ms,
         ms * 1e-3 * 2.4e9 / steps, (double)blocks * 8 * ITER * NT * 16 * 16 * 32 * 2 / ms / 1e9);
  return 0;
}

int main() {
  float *d, *in, h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) >> 8) / 16777216.f - 0.5f;
  CK(hipMalloc(&d, 256 * 2 * 512 * 4)); CK(hipMalloc(&in, sizeof(h))); CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  run<0>("f32 MFMA (8 x 16x16x4 f32), operands from LDS", d, in);
  run<1>("bf16 x3 split in registers + 6 x 16x16x32 bf16", d, in);
  run<2>("the split alone (no MFMA)", d, in);
  run<3>("6 x 16x16x32 bf16 alone (operands split once)", d, in);
  return 0;
}
