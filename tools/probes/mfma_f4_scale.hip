// Probe (gfx950): v_mfma_scale_f32_16x16x128_f8f6f4 with e2m1 (FP4) operands and PER-LANE E8M0 block scales, and the packing /
// rounding of v_cvt_scalef32_pk_fp4_f32.  Answers, with exact data:
//   * does lane l's scale byte apply to ITS 32-element k block (row l & 15, block l >> 4) of its operand?
//   * nibble order: is element 2i the low nibble of byte i (as v_cvt_scalef32_pk_fp4_f32 writes it)?
//   * what does the convert do with its `scale` argument, and how does it round / saturate?
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f4_scale.hip -o /tmp/mfma_f4 && /tmp/mfma_f4
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static const float E2M1[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
static float dec4(int c) { return (c & 8 ? -1.f : 1.f) * E2M1[c & 7]; }

__global__ void probe(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* C, int opsel) {
    // A [16][64] bytes row-major (128 e2m1 values per row), B likewise (row = output column); lane l: row l & 15, block l >> 4 (16 bytes)
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int w = 0; w < 4; ++w) {
        a[w] = *reinterpret_cast<const int*>(A + r * 64 + g * 16 + w * 4);
        b[w] = *reinterpret_cast<const int*>(B + r * 64 + g * 16 + w * 4);
    }
    // the lane's own scale in byte `opsel` of the scale register, garbage in the other bytes
    const int sa = (0x5A5A5A5A & ~(0xff << (8 * opsel))) | (SA[r * 4 + g] << (8 * opsel));
    const int sb = (0x3C3C3C3C & ~(0xff << (8 * opsel))) | (SB[r * 4 + g] << (8 * opsel));
    f32x4 acc = {0, 0, 0, 0};
    if (opsel == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 4, 4, 0, sa, 0, sb);
    else if (opsel == 1) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 4, 4, 1, sa, 1, sb);
    else if (opsel == 2) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 4, 4, 2, sa, 2, sb);
    else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 4, 4, 3, sa, 3, sb);
    for (int j = 0; j < 4; ++j) C[(g * 4 + j) * 16 + r] = acc[j];
}

__global__ void cvt(const float* x, const float* scale, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    unsigned o = 0xFFFFFFFFu;
    o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(o, x[2 * i], x[2 * i + 1], scale[i], 0);
    out[i] = o;
}

int main() {
    uint8_t hA[16 * 64], hB[16 * 64], hSA[64], hSB[64];
    srand(3);
    for (int i = 0; i < 16 * 64; ++i) { hA[i] = rand() & 0xff; hB[i] = rand() & 0xff; }
    for (int i = 0; i < 64; ++i) { hSA[i] = 120 + rand() % 12; hSB[i] = 118 + rand() % 14; }
    uint8_t *dA, *dB, *dSA, *dSB;
    float* dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dSA, 64); hipMalloc(&dSB, 64); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dSA, hSA, 64, hipMemcpyHostToDevice); hipMemcpy(dSB, hSB, 64, hipMemcpyHostToDevice);
    for (int opsel = 0; opsel < 4; ++opsel) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC, opsel);
        float hC[256];
        hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        // reference: per-lane block scales, low nibble = even element
        double e_blk = 0, e_row = 0, mref = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ref = 0, ref_rowscale = 0;
                for (int g = 0; g < 4; ++g) {
                    double s = 0;
                    for (int k = 0; k < 32; ++k) {
                        const int ba = hA[m * 64 + g * 16 + k / 2], bb = hB[n * 64 + g * 16 + k / 2];
                        const int ca = (k & 1) ? ba >> 4 : ba & 15, cb = (k & 1) ? bb >> 4 : bb & 15;
                        s += (double)dec4(ca) * dec4(cb);
                    }
                    ref += s * ldexp(1.0, hSA[m * 4 + g] - 127 + hSB[n * 4 + g] - 127);
                    ref_rowscale += s * ldexp(1.0, hSA[m * 4] - 127 + hSB[n * 4] - 127);
                }
                // operands are passed (a, b): D[row = a's row][col = b's row]?  try both orientations and keep the better
                const double d1 = fabs(hC[m * 16 + n] - ref), d2 = fabs(hC[n * 16 + m] - ref);
                e_blk = fmax(e_blk, fmin(d1, d2));
                e_row = fmax(e_row, fmin(fabs(hC[m * 16 + n] - ref_rowscale), fabs(hC[n * 16 + m] - ref_rowscale)));
                mref = fmax(mref, fabs(ref));
            }
        printf("opsel %d: max err vs per-(row, 32-block) scales %.3e   vs block-0 scale for the whole row %.3e   (max |ref| %.3e)\n", opsel, e_blk, e_row, mref);
    }
    // orientation check at opsel 0
    {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC, 0);
        float hC[256];
        hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        double e1 = 0, e2 = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ref = 0;
                for (int g = 0; g < 4; ++g) {
                    double s = 0;
                    for (int k = 0; k < 32; ++k) {
                        const int ba = hA[m * 64 + g * 16 + k / 2], bb = hB[n * 64 + g * 16 + k / 2];
                        s += (double)dec4((k & 1) ? ba >> 4 : ba & 15) * dec4((k & 1) ? bb >> 4 : bb & 15);
                    }
                    ref += s * ldexp(1.0, hSA[m * 4 + g] - 127 + hSB[n * 4 + g] - 127);
                }
                e1 = fmax(e1, fabs(hC[m * 16 + n] - ref));
                e2 = fmax(e2, fabs(hC[n * 16 + m] - ref));
            }
        printf("orientation: C[(g*4+j)*16 + r] = D[a-row][b-row] err %.3e, transposed err %.3e\n", e1, e2);
    }
    // the convert: inputs x scales
    const float xs[] = {0.f, 0.2f, 0.25f, 0.3f, 0.75f, 1.25f, 1.75f, 2.5f, 3.5f, 5.0f, 5.5f, 7.0f, 100.f, -0.75f, -2.5f, -1e-9f};
    const float scs[] = {1.f, 2.f, 0.5f, 3.f, 0.25f};
    const int nx = sizeof(xs) / 4;
    float hx[64], hs[32];
    unsigned ho[32];
    float *dx, *ds;
    unsigned* dO;
    hipMalloc(&dx, sizeof hx); hipMalloc(&ds, sizeof hs); hipMalloc(&dO, sizeof ho);
    for (float sc : scs) {
        for (int i = 0; i < nx; ++i) { hx[2 * i] = xs[i]; hx[2 * i + 1] = -xs[i]; hs[i] = sc; }
        hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(ds, hs, sizeof hs, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, dx, ds, dO, nx);
        hipMemcpy(ho, dO, sizeof ho, hipMemcpyDeviceToHost);
        printf("cvt scale %.2f:", sc);
        for (int i = 0; i < nx; ++i) printf("  %g->%g|%g", xs[i], dec4(ho[i] & 15), dec4((ho[i] >> 4) & 15));
        printf("   (word0 = %08x)\n", ho[0]);
    }
    return 0;
}
