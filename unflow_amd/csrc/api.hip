// Error plumbing and version of the C ABI (include/unflow_hip.h).
#include "common.h"

#include <cstdint>

namespace unflow {

char *error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

// Measurement helper: the HBM roofline the warps are priced against is the rate of THIS
// copy (16 bytes per lane, four independent loads in flight per lane, non-temporal both
// ways) over a buffer far larger than the 256 MiB Infinity Cache.
namespace {
constexpr int kCopyBlock = 256, kCopyUnroll = 4;
typedef float copy_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kCopyBlock) void stream_copy_kernel(const copy_f32x4 *__restrict__ src,
                                                                 copy_f32x4 *__restrict__ dst, long long n16) {
  const long long base = (long long)blockIdx.x * (kCopyBlock * kCopyUnroll) + threadIdx.x;
  copy_f32x4 v[kCopyUnroll];
#pragma unroll
  for (int u = 0; u < kCopyUnroll; ++u) {
    const long long i = base + (long long)u * kCopyBlock;
    if (i < n16) v[u] = __builtin_nontemporal_load(src + i);
  }
#pragma unroll
  for (int u = 0; u < kCopyUnroll; ++u) {
    const long long i = base + (long long)u * kCopyBlock;
    if (i < n16) __builtin_nontemporal_store(v[u], dst + i);
  }
}
}  // namespace

}  // namespace unflow

extern "C" int unflow_stream_copy(const void *src, void *dst, size_t bytes, unflow_stream_t stream) {
  using namespace unflow;
  if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | bytes) & 15)
    return fail(UNFLOW_ERR_INVALID_ARGUMENT, "stream_copy: pointers and size must be multiples of 16 bytes");
  const long long n16 = (long long)(bytes / 16);
  if (n16 == 0) return UNFLOW_OK;
  const long long blocks = (n16 + kCopyBlock * kCopyUnroll - 1) / (kCopyBlock * kCopyUnroll);
  if (blocks > 0x7fffffffLL) return fail(UNFLOW_ERR_INVALID_ARGUMENT, "stream_copy: too large for one launch");
  hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(kCopyBlock), 0, as_stream(stream),
                     static_cast<const copy_f32x4 *>(src), static_cast<copy_f32x4 *>(dst), n16);
  return launch_status("stream_copy");
}

extern "C" int unflow_abi_version(void) { return UNFLOW_ABI_VERSION; }
extern "C" const char *unflow_build_flags(void) {
  static char buf[256] = {0};
  static bool done = false;
  if (!done) {      // idempotent: both parts are string literals
    snprintf(buf, sizeof(buf), "%s%s%s%s", unflow::tune_flags_correlation_mfma(), unflow::tune_flags_correlation_fwd2(),
             unflow::tune_flags_correlation_bwd4(), unflow::tune_flags_warp());
    done = true;
  }
  return buf;
}
extern "C" const char *unflow_last_error(void) { return unflow::error_buffer(); }
