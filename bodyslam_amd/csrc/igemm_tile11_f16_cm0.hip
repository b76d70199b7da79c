// One (tile variant, operand type, correction mode) of the implicit-GEMM kernel per translation unit: they build in parallel
// (igemm_kernel.h).
#include "igemm_kernel.h"

namespace bs {
int igemm_launch_tile11_f16_cm0(const IgemmParams& p, bool conv, hipStream_t st) { return launch_cm<f16, 256, 128, 2, 2, 32, 3, false, 0>(p, conv, st); }
}  // namespace bs
