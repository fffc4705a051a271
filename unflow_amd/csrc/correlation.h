// Internal interface between correlation.hip (entry points, generic kernels)
// and correlation_mfma.hip (FlowNetC-shaped fast path).
#pragma once
#include <hip/hip_runtime.h>

namespace unflow {

// CorrelationState of ops/correlation_op.h:28-68 (reference checkout).
struct CorrGeometry {
  int H, W;
  int pad, stride_1, stride_2, kernel_radius, max_displacement, kernel_size;
  int grid_radius, grid_width, padded_height, padded_width, border_size;
  int out_height, out_width, out_channels;
};

int corr_geometry_init(CorrGeometry *g, int H, int W, int kernel_size, int max_displacement, int pad,
                       int stride_1, int stride_2);

bool corr_mfma_supported(const CorrGeometry &g, int C);
bool corr_mfma_bwd_supported(const CorrGeometry &g, int C);
int corr_mfma_fwd(const float *in0, const float *in1, float *out, int B, int C, const CorrGeometry &g,
                  hipStream_t s);
// second-generation forward (correlation_fwd2.hip): the channel sweep as one hand-written instruction stream.  Needs
// C % 16 == 0, W % 4 == 0 and 16-byte aligned tensors; everything else takes corr_mfma_fwd's kernels.
bool corr_fwd2_supported(const CorrGeometry &g, int C, const void *in0, const void *in1, const void *out);
int corr_fwd2(const float *in0, const float *in1, float *out, int B, int C, const CorrGeometry &g, hipStream_t s);
// fp16 storage, f32 MFMA (extension): half in, half out, every product and the k-ordered f32 sum exact
int corr_mfma_fwd_f16(const void *in0, const void *in1, void *out, int B, int C, const CorrGeometry &g,
                      hipStream_t s);
int corr_mfma_bwd_f16(const void *topdiff, const void *in0, const void *in1, void *g0, void *g1, int B, int C,
                      const CorrGeometry &g, hipStream_t s);
// float32 backward: correlation_bwd4.hip (C % 128 == 0, 16-byte aligned rows), everything else corr_mfma_bwd_r1
int corr_mfma_bwd(const float *topdiff, const float *in0, const float *in1, float *g0, float *g1, int B, int C,
                  const CorrGeometry &g, hipStream_t s);
// first-generation row sweep (correlation_mfma.hip): any channel count, 4-byte aligned rows
int corr_mfma_bwd_r1(const float *topdiff, const float *in0, const float *in1, float *g0, float *g1, int B, int C,
                     const CorrGeometry &g, hipStream_t s);
// f16 matrix cores (correlation_f16.hip): half in, half out, v_mfma_f32_16x16x16_f16 (f32 accumulate)
int corr16_fwd(const void *in0, const void *in1, void *out, int B, int C, const CorrGeometry &g, hipStream_t s);
int corr16_bwd(const void *topdiff, const void *in0, const void *in1, void *g0, void *g1, int B, int C,
               const CorrGeometry &g, hipStream_t s);

}  // namespace unflow
