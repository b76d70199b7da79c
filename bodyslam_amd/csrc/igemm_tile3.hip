// One tile variant of the implicit-GEMM kernel per translation unit (they build in parallel; see igemm_kernel.h).
#include "igemm_kernel.h"

namespace bs {
int igemm_launch_tile3(const IgemmParams& p, int dtype, bool conv, hipStream_t st) {
    if (dtype == BS_F16) return launch_variant<f16, 128, 32, 4, 1, 64, 2>(p, conv, st);
    return launch_variant<bf16, 128, 32, 4, 1, 64, 2>(p, conv, st);
}
}  // namespace bs
