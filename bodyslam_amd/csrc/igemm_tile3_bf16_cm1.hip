// One (tile variant, operand type, correction mode) of the implicit-GEMM kernel per translation unit: they build in parallel
// (igemm_kernel.h).
#include "igemm_kernel.h"

namespace bs {
int igemm_launch_tile3_bf16_cm1(const IgemmParams& p, bool conv, hipStream_t st) { return launch_cm<bf16, 128, 32, 4, 1, 64, 2, false, 1>(p, conv, st); }
}  // namespace bs
