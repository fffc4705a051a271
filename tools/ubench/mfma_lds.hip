// Micro-benchmark: what an LDS read costs a wave that keeps the f32 matrix pipe busy.  16 waves per workgroup (4 per
// SIMD, as corr_mfma_bwd16_kernel), one workgroup per CU; a wave repeats [4 MFMAs][reads][4 MFMAs][reads][4 MFMAs][reads]
// with 12 independent accumulators; the reads' results are not used by the MFMAs (no dependency: only their issue and
// execution can cost anything) and are waited for once per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4000;

// KIND 0: ds_read_b32 conflict-free; 1: ds_read_b32 2-way; 2: ds_read_b32 4-way; 3: ds_read_b64 conflict-free;
//      4: ds_read_b128 (linear); 5: ds_read2st64_b32; 6: ds_read_b64 2-way
template <int KIND>
__device__ inline void rd(unsigned addr) {
  if constexpr (KIND <= 2) { float s; asm volatile("ds_read_b32 %0, %1" : "=v"(s) : "v"(addr)); asm volatile("" :: "v"(s)); }
  else if constexpr (KIND == 3 || KIND == 6) { f32x2 s; asm volatile("ds_read_b64 %0, %1" : "=v"(s) : "v"(addr)); asm volatile("" :: "v"(s)); }
  else if constexpr (KIND == 4) { f32x4 s; asm volatile("ds_read_b128 %0, %1" : "=v"(s) : "v"(addr)); asm volatile("" :: "v"(s)); }
  else if constexpr (KIND == 5) { f32x2 s; asm volatile("ds_read2st64_b32 %0, %1 offset1:15" : "=v"(s) : "v"(addr)); asm volatile("" :: "v"(s)); }
  else if constexpr (KIND == 7) { unsigned s; asm volatile("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(s) : "v"(addr)); asm volatile("" :: "v"(s)); }          // one VALU op
  else if constexpr (KIND == 8) { unsigned s; asm volatile("s_add_u32 %0, %1, 4" : "=s"(s) : "s"(__builtin_amdgcn_readfirstlane(addr)) : "scc"); asm volatile("" :: "s"(s)); }   // one SALU op
  else if constexpr (KIND == 9) { asm volatile("s_cmp_lt_i32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" :: "s"(__builtin_amdgcn_readfirstlane(addr)) : "scc"); }    // compare + untaken branch
  else if constexpr (KIND == 10) { asm volatile("s_waitcnt lgkmcnt(2)"); }
  else if constexpr (KIND == 11) {   // the backward kernel's B-fragment sequence: 4 address VALU ops + 3 reads
    unsigned a0, a1, a2, a3; float s0, s1, s2;
    asm volatile("v_lshl_add_u32 %0, %4, 2, %4\n\tv_lshl_add_u32 %1, %4, 2, %0\n\tv_lshl_add_u32 %2, %4, 2, %0\n\tv_lshl_add_u32 %3, %4, 2, %0"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(addr));
    asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:256\n\tds_read_b32 %2, %3 offset:512" : "=v"(s0), "=v"(s1), "=v"(s2) : "v"(addr));
    asm volatile("" :: "v"(s0), "v"(s1), "v"(s2), "v"(a1), "v"(a2), "v"(a3));
  }
}

template <int KIND, int NR>
__global__ __launch_bounds__(1024) void k(float *out, float seed) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 1024) lds[i] = seed;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)lds + (threadIdx.x >> 6) * 4096;
  const unsigned addr = base + (KIND == 0 ? lane * 4 : KIND == 1 ? lane * 8 : KIND == 2 ? lane * 16 : KIND == 3 ? lane * 8 :
                                KIND == 4 ? lane * 16 : KIND == 5 ? lane * 4 : lane * 16);
  f32x4 acc[12];
  float av[4], bv[3];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i) av[i] = seed + i + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 3; ++i) bv[i] = seed * 2 + i;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[g * 4 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[g], acc[g * 4 + m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < NR / 3; ++r) rd<KIND>(addr);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1234.5f) out[0] = s;
}

template <int KIND, int NR>
int run(const char *name, float *d, double *base_ms) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t smem = 100 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<KIND, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL((k<KIND, NR>), dim3(256), dim3(1024), smem, 0, d, 1.0f);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<KIND, NR>), dim3(256), dim3(1024), smem, 0, d, 1.0f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  if (NR == 0) *base_ms = ms;
  const double mfma_per_simd = 4.0 * ITER * 12;
  // kernel cycles per wave-read-instruction per SIMD, beyond the MFMA-only time (2.4 GHz nominal)
  const double extra = NR ? (ms - *base_ms) * 1e-3 * 2.4e9 / (4.0 * ITER * NR) : 0.0;
  printf("%-34s reads/iter %2d: %7.3f ms  %5.1f cycles per MFMA per SIMD  | extra SIMD cycles per read instruction %5.2f\n", name, NR, ms,
         ms * 1e-3 * 2.4e9 / mfma_per_simd, extra);
  return 0;
}

// Dependent mode, as in the backward kernel: the B operand of a group of 4 MFMAs is an LDS read issued DISTG groups
// earlier (1 group = 4 MFMAs of this wave); the A operands are read once per 3 groups, 1 K-step (3 groups) ahead.
template <int DISTG, int CONFLICT>
__global__ __launch_bounds__(1024) void kdep(float *out, float seed) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 1024) lds[i] = seed * 1e-3f;
  __syncthreads();
  typedef __attribute__((address_space(3))) const float lds_cf;
  const unsigned base = (unsigned)(size_t)(lds_cf *)lds + (threadIdx.x >> 6) * 2048;
  lds_cf *pb = (lds_cf *)(size_t)(base + lane * 4 * CONFLICT);
  lds_cf *pa = (lds_cf *)(size_t)(base + lane * 4);
  f32x4 acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = f32x4{0, 0, 0, 0};
  constexpr int NBUF = DISTG + 1;
  float bq[NBUF];        // ring of B operands
  float aq[2][4];
#pragma unroll
  for (int i = 0; i < NBUF; ++i) bq[i] = pb[i * 64];
#pragma unroll
  for (int i = 0; i < 4; ++i) { aq[0][i] = pa[i * 64]; aq[1][i] = pa[i * 64 + 16]; }
  for (int it = 0; it < ITER; it += NBUF * 2) {
#pragma unroll
    for (int u = 0; u < NBUF * 2; ++u) {       // unrolled so that ring indices are compile-time
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        constexpr int dummy = 0; (void)dummy;
        const int gi = (u * 3 + g);
        const int slot = gi % NBUF;
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[g * 4 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[u & 1][m], bq[slot], acc[g * 4 + m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        bq[slot] = pb[((gi * 7) & 31) * 64 / CONFLICT];          // refill: consumed DISTG groups from now... (ring of DISTG + 1)
        if (g == 0) {
#pragma unroll
          for (int m = 0; m < 4; ++m) aq[(u + 1) & 1][m] = pa[m * 64 + ((u * 5) & 15)];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1234.5f) out[0] = s;
}

template <int DISTG, int CONFLICT>
int rundep(float *d) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t smem = 100 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kdep<DISTG, CONFLICT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL((kdep<DISTG, CONFLICT>), dim3(256), dim3(1024), smem, 0, d, 1.0f);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((kdep<DISTG, CONFLICT>), dim3(256), dim3(1024), smem, 0, d, 1.0f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const int iters = (ITER + (DISTG + 1) * 2 - 1) / ((DISTG + 1) * 2) * ((DISTG + 1) * 2);
  printf("dependent B reads, distance %d group(s) of 4 MFMAs, %d-way: %7.3f ms  %5.1f cycles per MFMA per SIMD\n", DISTG, CONFLICT, ms,
         ms * 1e-3 * 2.4e9 / (4.0 * iters * 12));
  return 0;
}

int main() {
  float *d; CK(hipMalloc(&d, 1024));
  double b = 0;
  run<0, 0>("MFMAs only", d, &b);
  run<0, 3>("ds_read_b32, conflict-free", d, &b); run<0, 6>("ds_read_b32, conflict-free", d, &b); run<0, 12>("ds_read_b32, conflict-free", d, &b);
  run<1, 3>("ds_read_b32, 2-way", d, &b); run<1, 6>("ds_read_b32, 2-way", d, &b); run<1, 12>("ds_read_b32, 2-way", d, &b);
  run<2, 6>("ds_read_b32, 4-way", d, &b);
  run<3, 3>("ds_read_b64, conflict-free", d, &b); run<3, 6>("ds_read_b64, conflict-free", d, &b); run<3, 12>("ds_read_b64, conflict-free", d, &b);
  run<6, 6>("ds_read_b64, 2-way", d, &b);
  run<4, 3>("ds_read_b128, linear", d, &b); run<4, 6>("ds_read_b128, linear", d, &b);
  run<5, 3>("ds_read2st64_b32", d, &b); run<5, 6>("ds_read2st64_b32", d, &b);
  run<7, 6>("VALU op (v_lshl_add_u32)", d, &b); run<7, 12>("VALU op (v_lshl_add_u32)", d, &b); run<7, 24>("VALU op (v_lshl_add_u32)", d, &b); run<7, 48>("VALU op (v_lshl_add_u32)", d, &b);
  run<8, 12>("SALU op (s_add_u32)", d, &b); run<8, 48>("SALU op (s_add_u32)", d, &b);
  run<9, 3>("s_cmp + untaken s_cbranch", d, &b); run<9, 12>("s_cmp + untaken s_cbranch", d, &b);
  run<10, 3>("s_waitcnt lgkmcnt(2)", d, &b); run<10, 12>("s_waitcnt lgkmcnt(2)", d, &b);
  run<11, 3>("4 VALU + 3 ds_read_b32 (one unit)", d, &b); run<11, 6>("4 VALU + 3 ds_read_b32 (one unit)", d, &b);
  rundep<1, 1>(d); rundep<1, 2>(d); rundep<2, 1>(d); rundep<2, 2>(d); rundep<3, 2>(d); rundep<5, 2>(d);
  return 0;
}
