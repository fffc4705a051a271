// Does a VALU write of a wide store's data register in the NEXT instruction reach memory?
// LLVM's hazard recognizer (GCNHazardRecognizer::createsVALUHazard) inserts a wait state between a VMEM store of more
// than 64 bits and a VALU write of its data registers -- except for MUBUF stores whose soffset is an SGPR, which it
// takes to be safe.  Round 3 saw exactly that form (buffer_store_dwordx4 v[4:7], v, s[..], s0 offen ; v_cndmask v4)
// store the overwritten dword 0 in lanes 12-15 (mod 16) a few hundred times per launch.  This test issues the pair
// back to back, with 0 / 1 / 2 s_nop in between, for an immediate-zero and an SGPR soffset, many times per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: soffset = SGPR; 1: soffset = 0 (immediate).  NOPS: s_nop count between the store and the overwrite.
template <int MODE, int NOPS>
__global__ __launch_bounds__(512) void k(unsigned *out, int nbytes, int reps, int soff_in) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, (short)0, nbytes, 0x00020000);
  const int tid = blockIdx.x * 512 + threadIdx.x;
  const int soff = __builtin_amdgcn_readfirstlane(soff_in);      // 0 at run time, an SGPR to the assembler
  for (int it = 0; it < reps; ++it) {
    const unsigned voff = (unsigned)((it * gridDim.x * 512 + tid) * 16);
#define UNFLOW_PAIR(STORE, GAP)                                                                                     \
  asm volatile("v_mov_b32 v8, 0x11111111\n\tv_mov_b32 v9, 0x22222222\n\tv_mov_b32 v10, 0x33333333\n\tv_mov_b32 v11, 0x44444444\n\t" \
               "s_nop 4\n\t" STORE "\n\t" GAP "v_mov_b32 v8, 0"                                                      \
               : : "v"(voff), "s"(r), "s"(soff) : "memory", "v8", "v9", "v10", "v11")
    if (MODE == 0) {
      if (NOPS == 0) UNFLOW_PAIR("buffer_store_dwordx4 v[8:11], %0, %1, %2 offen", "");
      else if (NOPS == 1) UNFLOW_PAIR("buffer_store_dwordx4 v[8:11], %0, %1, %2 offen", "s_nop 0\n\t");
      else UNFLOW_PAIR("buffer_store_dwordx4 v[8:11], %0, %1, %2 offen", "s_nop 1\n\t");
    } else {
      if (NOPS == 0) UNFLOW_PAIR("buffer_store_dwordx4 v[8:11], %0, %1, 0 offen", "");
      else if (NOPS == 1) UNFLOW_PAIR("buffer_store_dwordx4 v[8:11], %0, %1, 0 offen", "s_nop 0\n\t");
      else UNFLOW_PAIR("buffer_store_dwordx4 v[8:11], %0, %1, 0 offen", "s_nop 1\n\t");
    }
  }
}

template <int MODE, int NOPS>
int run(unsigned *d, size_t n_u32, std::vector<unsigned> &h, int blocks, int reps) {
  CK(hipMemset(d, 0xff, n_u32 * 4));
  hipLaunchKernelGGL((k<MODE, NOPS>), dim3(blocks), dim3(512), 0, 0, d, (int)(n_u32 * 4), reps, 0);
  CK(hipMemcpy(h.data(), d, n_u32 * 4, hipMemcpyDeviceToHost));
  long bad[4] = {0, 0, 0, 0}, lanes[4] = {0, 0, 0, 0};
  const unsigned want[4] = {0x11111111u, 0x22222222u, 0x33333333u, 0x44444444u};
  for (size_t i = 0; i < n_u32 / 4; ++i)
    for (int c = 0; c < 4; ++c)
      if (h[i * 4 + c] != want[c]) { ++bad[c]; if (c == 0) ++lanes[(i & 15) >> 2]; }
  printf("soffset %-9s s_nop between: %d  -> stores with a wrong dword 0/1/2/3: %ld %ld %ld %ld of %zu   (dword 0 by lane quad of 16: %ld %ld %ld %ld)\n",
         MODE == 0 ? "SGPR" : "immediate", NOPS, bad[0], bad[1], bad[2], bad[3], n_u32 / 4, lanes[0], lanes[1], lanes[2], lanes[3]);
  return 0;
}

int main() {
  const int blocks = 1024, reps = 16;
  const size_t n = (size_t)blocks * 512 * reps * 4;
  unsigned *d; CK(hipMalloc(&d, n * 4));
  std::vector<unsigned> h(n);
  run<0, 0>(d, n, h, blocks, reps); run<0, 1>(d, n, h, blocks, reps); run<0, 2>(d, n, h, blocks, reps);
  run<1, 0>(d, n, h, blocks, reps); run<1, 1>(d, n, h, blocks, reps); run<1, 2>(d, n, h, blocks, reps);
  return 0;
}
