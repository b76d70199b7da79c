// Probe (gfx950): what v_permlane16_swap / v_permlane32_swap return through the clang builtins when both operands are the same value,
// and the DPP quad permutes used for 4-lane maxima.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
    const unsigned l = threadIdx.x;
    const auto a = __builtin_amdgcn_permlane16_swap(l, l, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(l, l, false, false);
    const unsigned m16 = max(a[0], a[1]);
    const auto c = __builtin_amdgcn_permlane32_swap(m16, m16, false, false);
    int q1 = __builtin_amdgcn_update_dpp((int)l, (int)l, 0xB1, 0xf, 0xf, false);
    int q2 = __builtin_amdgcn_update_dpp((int)l, (int)l, 0x4E, 0xf, 0xf, false);
    o[l * 8 + 0] = a[0]; o[l * 8 + 1] = a[1]; o[l * 8 + 2] = b[0]; o[l * 8 + 3] = b[1];
    o[l * 8 + 4] = max(c[0], c[1]); o[l * 8 + 5] = q1; o[l * 8 + 6] = q2; o[l * 8 + 7] = 0;
}
int main() {
    unsigned* d; unsigned h[512];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 5) printf("lane %2d: p16 (%2u, %2u)  p32 (%2u, %2u)  max over the 4 rows %2u  quad xor1 %2u xor2 %2u\n", l, h[l*8], h[l*8+1], h[l*8+2], h[l*8+3], h[l*8+4], h[l*8+5], h[l*8+6]);
    return 0;
}
