// Probe: operand lane map and scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 e4m3 operands (gfx950).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f8_scale.hip -o /tmp/mfma_f8 && /tmp/mfma_f8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// e4m3fn encode of small exact values (powers of two and small ints) for the probe
__host__ __device__ static uint8_t enc(float v) {
    if (v == 0.f) return 0;
    uint8_t s = v < 0 ? 0x80 : 0;
    v = fabsf(v);
    int e;
    float m = frexpf(v, &e);          // v = m * 2^e, m in [0.5,1)
    int E = e - 1 + 7;                // biased exponent of 1.xxx * 2^(e-1)
    int man = (int)roundf((m * 2.f - 1.f) * 8.f);
    if (man == 8) { man = 0; ++E; }
    if (E <= 0) { man = (int)roundf(v / ldexpf(1.f, -9)); return s | (uint8_t)man; }   // subnormal: man * 2^-9
    return s | (uint8_t)(E << 3) | (uint8_t)man;
}

__global__ void probe(const uint8_t* A, const uint8_t* B, float* C, int sa, int sb) {
    // A [16][128] bytes row-major, B [16][128] bytes (B^T: row = output column n), lane l: row l&15, k-group l>>4 (32 bytes)
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) {
        a[w] = *reinterpret_cast<const int*>(A + r * 128 + g * 32 + w * 4);
        b[w] = *reinterpret_cast<const int*>(B + r * 128 + g * 32 + w * 4);
    }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
    // C/D map of the 16x16 family: col = lane & 15, row = (lane >> 4) * 4 + j
    for (int j = 0; j < 4; ++j) C[(g * 4 + j) * 16 + r] = acc[j];
}

int main() {
    uint8_t hA[16 * 128], hB[16 * 128];
    float fA[16][128], fB[16][128];
    srand(1);
    const float vals[] = {0.f, 0.5f, 1.f, -1.f, 2.f, -0.25f, 1.5f, 3.f, -0.0625f, 0.001953125f};
    for (int r = 0; r < 16; ++r)
        for (int k = 0; k < 128; ++k) {
            fA[r][k] = vals[rand() % 10];
            fB[r][k] = vals[rand() % 10];
            hA[r * 128 + k] = enc(fA[r][k]);
            hB[r * 128 + k] = enc(fB[r][k]);
        }
    uint8_t *dA, *dB;
    float* dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    const int scales[][2] = {{127, 127}, {116, 127}, {127, 110}, {120, 122}};
    for (auto& sc : scales) {
        const int sa = sc[0] * 0x01010101, sb = sc[1] * 0x01010101;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, sa, sb);
        float hC[256];
        hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ref = 0;
                for (int k = 0; k < 128; ++k) ref += (double)fA[m][k] * fB[n][k];
                ref *= ldexp(1.0, sc[0] - 127 + sc[1] - 127);
                // which of C[m][n] / C[n][m] matches tells the A/B -> row/col assignment
                maxerr = fmax(maxerr, fabs(hC[m * 16 + n] - ref));
                maxref = fmax(maxref, fabs(ref));
            }
        double maxerrT = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ref = 0;
                for (int k = 0; k < 128; ++k) ref += (double)fA[m][k] * fB[n][k];
                ref *= ldexp(1.0, sc[0] - 127 + sc[1] - 127);
                maxerrT = fmax(maxerrT, fabs(hC[n * 16 + m] - ref));
            }
        printf("scale_a=%d scale_b=%d: max|C[m][n]-ref|=%.3e  max|C[n][m]-ref|=%.3e  (max ref %.3e)\n", sc[0], sc[1], maxerr, maxerrT, maxref);
    }
    return 0;
}
