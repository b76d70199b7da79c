// The persistent form of the 256 x 256 x 64 correction-mode tile (igemm_kernel.h, PERSIST): its own translation unit, like every (tile,
// operand type, correction mode).
#include "igemm_kernel.h"

namespace bs {
int igemm_launch_tile9_f16_cm1_persist(const IgemmParams& p, bool conv, hipStream_t st) { return launch_cm_persist<f16, 256, 256, 2, 4, 64, 2, 1>(p, conv, st); }
}  // namespace bs
