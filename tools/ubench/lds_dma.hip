// Micro-benchmark: LDS-DMA (buffer_load ... lds) staging throughput on gfx950 for the
// address patterns the correlation kernels could use.  Each workgroup (8 waves)
// streams CHUNKS stages of ~54 KB into a double-buffered LDS ring, one barrier per
// stage, no compute.  12 consecutive (XCD-remapped) workgroups share their data so
// the L2 sees the same reuse as the real kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int NT = 512, NW = 8;
constexpr int STAGE_FLOATS = 13568;          // 53 KB, 212 dword pieces of 64 floats
constexpr int CHUNKS = 32;
constexpr int ROWPITCH = 160;                // floats between consecutive image rows (W8=160)

__device__ inline __amdgpu_buffer_rsrc_t rsrc(const float *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), (short)0, (int)bytes, 0x00020000);
}

template <int PAT>
__global__ __launch_bounds__(NT, 2) void k(const float *src, float *out, int nblk, int share, long long region_floats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int vb = blockIdx.x;
  if ((nblk & 7) == 0) vb = (vb & 7) * (nblk >> 3) + (vb >> 3);
  const float *base = src + (long long)(vb / share) * region_floats;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t r = rsrc(base, (unsigned)(region_floats * 4));
  for (int ch = 0; ch < CHUNKS; ++ch) {
    float *stage = lds + (ch & 1) * STAGE_FLOATS;
    const int cbase = ch * (int)(region_floats / CHUNKS) * 4;   // byte offset of this chunk's data
    if (PAT == 0) {          // dword, 256 contiguous bytes per instruction
      for (int j = 0; j < 27; ++j) { int p = wave + NW * j; if (p < 212)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, stage + p * 64, 4, cbase + p * 256 + lane * 4, 0, 0, 0); }
    } else if (PAT == 1) {   // dword, every other float of a 512-byte span (one column class)
      for (int j = 0; j < 27; ++j) { int p = wave + NW * j; if (p < 212)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, stage + p * 64, 4, cbase + (p / 2) * 512 + (p & 1) * 4 + lane * 8, 0, 0, 0); }
    } else if (PAT == 2) {   // dword, both column classes of a 224-byte row segment, rows ROWPITCH apart
      for (int j = 0; j < 27; ++j) { int p = wave + NW * j; if (p < 212) {
        int slot = p * 64 + lane; int row = slot / 56, r2 = slot % 56; int cx = r2 / 28, xp = r2 % 28;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, stage + p * 64, 4, cbase + row * ROWPITCH * 8 + (xp * 2 + cx) * 4, 0, 0, 0); } }
    } else if (PAT == 3) {   // dwordx4, 1 KiB contiguous per instruction
      for (int j = 0; j < 7; ++j) { int p = wave + NW * j; if (p < 53)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, stage + p * 256, 16, cbase + p * 1024 + lane * 16, 0, 0, 0); }
    } else if (PAT == 4) {   // dwordx4, 224-byte row segments (14 lanes per row), rows ROWPITCH apart
      for (int j = 0; j < 7; ++j) { int p = wave + NW * j; if (p < 53) {
        int seg = p * 64 + lane; int row = seg / 14, s16 = seg % 14;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, stage + p * 256, 16, cbase + row * ROWPITCH * 8 + s16 * 16, 0, 0, 0); } }
    } else if (PAT == 5) {   // register path: global_load_dwordx4 row segments + ds_write_b128
      const float4 *g = reinterpret_cast<const float4 *>(base);
      float4 v[7];
      for (int j = 0; j < 7; ++j) { int p = min(wave + NW * j, 52); int seg = p * 64 + lane; int row = seg / 14, s16 = seg % 14;
        v[j] = g[(cbase / 4 + row * ROWPITCH * 2 + s16 * 4) / 4]; }
      for (int j = 0; j < 7; ++j) { int p = wave + NW * j; if (p < 53) reinterpret_cast<float4 *>(stage)[p * 64 + lane] = v[j]; }
    }
    __syncthreads();
  }
  if (lds[threadIdx.x] == 123.456f) out[0] = 1.f;
}

template <int PAT>
int run(const char *name, const float *src, float *out, long long region_floats) {
  const int nblk = 1920, share = 12;
  const size_t smem = 2 * STAGE_FLOATS * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<PAT>, dim3(nblk), dim3(NT), smem, 0, src, out, nblk, share, region_floats);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<PAT>, dim3(nblk), dim3(NT), smem, 0, src, out, nblk, share, region_floats);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  const double bytes = (double)nblk * CHUNKS * STAGE_FLOATS * 4;
  printf("%-58s %7.3f ms  %6.2f TB/s into LDS\n", name, ms, bytes / ms / 1e9);
  return 0;
}

int main() {
  const long long region_floats = 32LL * 81920;   // 327 KB per chunk: every pattern stays inside its chunk (max 256 rows x 1280 B)
  const int nregions = 1920 / 12;
  float *src, *out;
  CK(hipMalloc(&src, (size_t)nregions * region_floats * 4)); CK(hipMalloc(&out, 1024));
  CK(hipMemset(src, 0, (size_t)nregions * region_floats * 4));
  printf("footprint %.0f MB, 1920 workgroups x 32 stages x 53 KB = 3.33 GB staged\n", nregions * region_floats * 4 / 1e6);
  run<0>("dword   256 B contiguous per instr", src, out, region_floats);
  run<1>("dword   stride-2 floats (one column class)", src, out, region_floats);
  run<2>("dword   both classes of a 224-B row segment", src, out, region_floats);
  run<3>("dwordx4 1 KiB contiguous per instr", src, out, region_floats);
  run<4>("dwordx4 224-B row segments", src, out, region_floats);
  run<5>("global_load_dwordx4 + ds_write_b128, 224-B row segments", src, out, region_floats);
  return 0;
}
