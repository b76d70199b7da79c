// One (tile variant, operand type, correction mode) of the implicit-GEMM kernel per translation unit: they build in parallel
// (igemm_kernel.h).
#include "igemm_kernel.h"

namespace bs {
int igemm_launch_tile2_bf16_cm2(const IgemmParams& p, bool conv, hipStream_t st) { return launch_cm<bf16, 128, 64, 2, 2, 64, 2, false, 2>(p, conv, st); }
}  // namespace bs
