// Does an LDS-DMA (buffer_load_dwordx4 ... lds) touch VGPR v0 -- the register its unused data operand encodes?
// Observed in round 3 (tools/sketches/corr_bwd3_specialised_waves.hip.txt): a ds_read_b128 into v[0:3] issued while
// DMA of the same wave was in flight came back with v0 zeroed in the lanes whose DMA segments were out of range.
// This test pins markers in v0..v3 with inline asm, issues DMAs whose lanes are in range / out of range / masked,
// waits, and reads the markers back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(64) void k(const float *src, int nbytes, unsigned *out, int mode) {
  __shared__ __attribute__((aligned(16))) float lds[2048];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) lds[i] = -1.f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), (short)0, nbytes, 0x00020000);
  // mode 0: all lanes in range; 1: odd lanes out of range; 2: lanes (lane & 12) == 12 out of range; 3: all out of range
  unsigned voff = lane * 16;
  if (mode == 1 && (lane & 1)) voff = 0x80000000u;
  if (mode == 2 && (lane & 12) == 12) voff = 0x80000000u;
  if (mode == 3) voff = 0x80000000u;
  const unsigned m0 = (unsigned)(size_t)lds;    // LDS byte address of the destination (low 32 bits of the flat-to-local cast)
  unsigned o0, o1, o2, o3;
  asm volatile(
      "v_mov_b32 v0, 0x11110000\n\t"
      "v_mov_b32 v1, 0x22220000\n\t"
      "v_mov_b32 v2, 0x33330000\n\t"
      "v_mov_b32 v3, 0x44440000\n\t"
      "s_mov_b32 m0, %5\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %4, %6, 0 offen lds\n\t"
      "buffer_load_dwordx4 %4, %6, 0 offen offset:1024 lds\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_nop 4\n\t"
      "v_mov_b32 %0, v0\n\t"
      "v_mov_b32 %1, v1\n\t"
      "v_mov_b32 %2, v2\n\t"
      "v_mov_b32 %3, v3\n\t"
      : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
      : "v"(voff), "s"((unsigned)__builtin_amdgcn_readfirstlane((int)((__attribute__((address_space(3))) float *)lds - (__attribute__((address_space(3))) float *)0) * 4)), "s"(r)
      : "v0", "v1", "v2", "v3", "memory");
  (void)m0;
  out[lane * 4 + 0] = o0;
  out[lane * 4 + 1] = o1;
  out[lane * 4 + 2] = o2;
  out[lane * 4 + 3] = o3;
  __syncthreads();
  out[256 + lane] = __float_as_uint(lds[lane * 4]);      // what landed in LDS (first dword of each lane's 16 bytes)
}

// The round-3 sighting itself: a ds_read_b128 into v[0:3] issued while the wave's DMA is in flight (order 0: DMA
// first, order 1: read first).  The read takes a region the DMA does not write.
__global__ __launch_bounds__(64) void k2(const float *src, int nbytes, unsigned *out, int mode, int order) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) lds[i] = -1.f;
  for (int i = lane; i < 2048; i += 64) lds[2048 + i] = 5000.f + i;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), (short)0, nbytes, 0x00020000);
  unsigned voff = lane * 16;
  if (mode == 1 && (lane & 1)) voff = 0x80000000u;
  if (mode == 2 && (lane & 12) == 12) voff = 0x80000000u;
  if (mode == 3) voff = 0x80000000u;
  const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)((__attribute__((address_space(3))) float *)lds - (__attribute__((address_space(3))) float *)0) * 4);
  const unsigned raddr = base + 8192 + lane * 16;
  unsigned o0, o1, o2, o3;
  if (order == 0)
    asm volatile(
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %4, %6, 0 offen lds\n\t"
        "buffer_load_dwordx4 %4, %6, 0 offen offset:1024 lds\n\t"
        "buffer_load_dwordx4 %4, %6, 0 offen offset:2048 lds\n\t"
        "ds_read_b128 v[0:3], %7\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %0, v0\n\t"
        "v_mov_b32 %1, v1\n\t"
        "v_mov_b32 %2, v2\n\t"
        "v_mov_b32 %3, v3\n\t"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
        : "v"(voff), "s"(base), "s"(r), "v"(raddr)
        : "v0", "v1", "v2", "v3", "memory");
  else
    asm volatile(
        "s_mov_b32 m0, %5\n\t"
        "ds_read_b128 v[0:3], %7\n\t"
        "buffer_load_dwordx4 %4, %6, 0 offen lds\n\t"
        "buffer_load_dwordx4 %4, %6, 0 offen offset:1024 lds\n\t"
        "buffer_load_dwordx4 %4, %6, 0 offen offset:2048 lds\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %0, v0\n\t"
        "v_mov_b32 %1, v1\n\t"
        "v_mov_b32 %2, v2\n\t"
        "v_mov_b32 %3, v3\n\t"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
        : "v"(voff), "s"(base), "s"(r), "v"(raddr)
        : "v0", "v1", "v2", "v3", "memory");
  out[lane * 4 + 0] = o0;
  out[lane * 4 + 1] = o1;
  out[lane * 4 + 2] = o2;
  out[lane * 4 + 3] = o3;
}

int main() {
  float *src; unsigned *out;
  const int n = 4096;
  CK(hipMalloc(&src, n * 4)); CK(hipMalloc(&out, 1024 * 4));
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = 1000.f + i;
  CK(hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice));
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, n * 4, out, mode);
    std::vector<unsigned> o(1024);
    CK(hipMemcpy(o.data(), out, 1024 * 4, hipMemcpyDeviceToHost));
    int bad[4] = {0, 0, 0, 0};
    const unsigned want[4] = {0x11110000u, 0x22220000u, 0x33330000u, 0x44440000u};
    int first_bad = -1;
    for (int l = 0; l < 64; ++l)
      for (int c = 0; c < 4; ++c)
        if (o[l * 4 + c] != want[c]) { ++bad[c]; if (first_bad < 0) first_bad = l; }
    int landed = 0, zeros = 0;
    for (int l = 0; l < 64; ++l) {
      union { unsigned u; float f; } cv; cv.u = o[256 + l]; const float v = cv.f;
      if (v == 1000.f + l * 4) ++landed; else if (v == 0.f) ++zeros;
    }
    printf("mode %d: lanes with a changed marker: v0 %d  v1 %d  v2 %d  v3 %d  (first lane %d, its v0 now 0x%08x) | LDS: %d lanes got their data, %d got zeros\n",
           mode, bad[0], bad[1], bad[2], bad[3], first_bad, first_bad >= 0 ? o[first_bad * 4] : 0u, landed, zeros);
  }
  for (int order = 0; order < 2; ++order)
    for (int mode = 0; mode < 4; ++mode) {
      hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, src, n * 4, out, mode, order);
      std::vector<unsigned> o(1024);
      CK(hipMemcpy(o.data(), out, 1024 * 4, hipMemcpyDeviceToHost));
      int bad[4] = {0, 0, 0, 0}, first_bad = -1;
      for (int l = 0; l < 64; ++l)
        for (int c = 0; c < 4; ++c) {
          union { unsigned u; float f; } cv; cv.u = o[l * 4 + c];
          if (cv.f != 5000.f + l * 4 + c) { ++bad[c]; if (first_bad < 0) first_bad = l; }
        }
      printf("ds_read_b128 v[0:3] racing the DMA, %s, mode %d: lanes with a wrong dword: v0 %d  v1 %d  v2 %d  v3 %d  (first lane %d, its v0 0x%08x)\n",
             order ? "read issued first" : "DMA issued first", mode, bad[0], bad[1], bad[2], bad[3], first_bad, first_bad >= 0 ? o[first_bad * 4] : 0u);
    }
  return 0;
}
