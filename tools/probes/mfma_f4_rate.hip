// Probe (gfx950): issue rate of the block-scaled MFMA with e2m1 / e4m3 operands, with per-lane VGPR scales vs literal scales, one and
// two waves per SIMD, operands in registers (no memory in the loop).  Prints shader cycles per MFMA per wave and per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(long long* cyc, float* sink, int iters, int sa_in, int sb_in) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) { a[w] = 0x11111111 * (1 + ((l + w) & 3)); b[w] = 0x22222222 ^ (l * 7 + w); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const int sa = sa_in + (MODE == 1 ? (l & 3) : 0), sb = sb_in + (MODE == 1 ? ((l >> 2) & 3) : 0);
    f16x8 ha, hb;
    for (int w = 0; w < 8; ++w) { ha[w] = (_Float16)(0.01f * (l + w)); hb[w] = (_Float16)(0.02f * (l - w)); }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 4, 4, 0, sa, 0, sb);         // fp4, uniform runtime scales
            else if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 4, 4, 0, sa, 0, sb);    // fp4, per-lane scales
            else if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 4, 4, 0, 0, 0, 0);      // fp4, literal zero scales
            else if (MODE == 3) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, sa, 0, sb);    // fp8, runtime scales
            else if (MODE == 4) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);                       // f16
            else if (MODE == 5) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 2, 2, 0, sa, 0, sb);    // fp6 e2m3
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (l == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int threads, int blocks) {
    long long* d; float* s;
    hipMalloc(&d, blocks * 8 * 8); hipMalloc(&s, blocks * 512 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, s, iters, 125, 126);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, s, iters, 125, 126);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double kper = MODE == 4 ? 32.0 : 128.0;
    const double tflops = 2.0 * 16 * 16 * kper * 8.0 * iters * (threads / 64) * blocks / (ms * 1e-3) / 1e12;
    long long h[8];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const double per = (double)h[0] / (iters * 8.0);
    const int wps = threads / 256;      // waves per SIMD
    printf("%-34s %d wave(s)/SIMD, %4d blocks: %6.2f cycles per MFMA per wave -> %6.2f per SIMD; wall %.3f ms = %7.1f TFLOP/s -> clock %.2f GHz\n", name, wps,
           blocks, per, per / wps, ms, tflops, (double)h[0] / (ms * 1e-3) / 1e9);
    hipFree(d); hipFree(s);
}

int main() {
    for (int threads : {256, 512}) {
        for (int blocks : {1, 256}) {
            run<4>("f16 16x16x32", threads, blocks);
            run<3>("fp8 scaled, runtime scales", threads, blocks);
            run<0>("fp4 scaled, uniform runtime scales", threads, blocks);
            run<1>("fp4 scaled, per-lane scales", threads, blocks);
            run<2>("fp4 scaled, literal 0 scales", threads, blocks);
            run<5>("fp6 scaled, runtime scales", threads, blocks);
        }
    }
    return 0;
}
