// One (tile variant, operand type, correction mode) of the implicit-GEMM kernel per translation unit: they build in parallel
// (igemm_kernel.h).
#include "igemm_kernel.h"

namespace bs {
int igemm_launch_tile9_bf16_cm0(const IgemmParams& p, bool conv, hipStream_t st) { return launch_cm<bf16, 256, 256, 2, 4, 64, 2, false, 0>(p, conv, st); }
}  // namespace bs
