// placeholder: fast path not built yet
#include "common.h"
#include "correlation.h"
namespace unflow {
bool corr_mfma_supported(const CorrGeometry &, int) { return false; }
int corr_mfma_fwd(const float *, const float *, float *, int, int, const CorrGeometry &, hipStream_t) { return fail(UNFLOW_ERR_LAUNCH, "no mfma path"); }
int corr_mfma_bwd(const float *, const float *, const float *, float *, float *, int, int, const CorrGeometry &, hipStream_t) { return fail(UNFLOW_ERR_LAUNCH, "no mfma path"); }
}
