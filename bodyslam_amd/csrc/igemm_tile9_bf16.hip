// One (tile variant, operand type) of the implicit-GEMM kernel per translation unit: they build in parallel (see igemm_kernel.h).
#include "igemm_kernel.h"

namespace bs {
int igemm_launch_tile9_bf16(const IgemmParams& p, bool conv, hipStream_t st) { return launch_variant<bf16, 256, 256, 2, 4, 64, 2>(p, conv, st); }
}  // namespace bs
