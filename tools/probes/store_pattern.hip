// Probe: what a GEMM tile's 16-bit output costs as a function of the store pattern, with the GEMM's occupancy (one 512-thread block per
// CU: 128 KiB of LDS) and a compute phase in front of the stores (a timed spin standing in for the main loop, so that the CUs drift apart
// as they do in bs_gemm).  Block tile 256 x 256 of a row-major [M, N] fp16 matrix, wave tile 128 x 64 (2 x 4 waves), as igemm_kernel.
//   pattern 0: igemm_kernel's register layout -- an instruction covers 16 rows x 64 contiguous bytes (lane = row frow, 16-byte piece fq)
//   pattern 1: full lines -- an instruction covers 8 rows x 128 contiguous bytes (what a transpose through LDS would give)
//   pattern 2: no stores (the spin only)
//   pattern 4: full lines in the lane order a DPP half-row exchange of the register layout gives (lane = row frow & 7, piece fq + 4 (frow >> 3))
//   pattern 3: pattern 1 with the LDS round trip that produces it (ds_write_b128 in the register layout, ds_read_b128 in the line layout)
// hipcc -O3 --offload-arch=gfx950 -o /tmp/store_pattern tools/probes/store_pattern.hip && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int PAT>
__global__ __launch_bounds__(512) void k(_Float16* out, int ntm, int ntn, int N, long long spin_ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + loc;
    const int tm = wg / ntn, tn = wg - tm * ntn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    h8 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = (_Float16)(float)(lane + i + e + tm);
    if (PAT == 2) {
        if (v[3][1] == (_Float16)12345.0f) out[0] = v[2][0];
        return;
    }
    const int frow = lane & 15, fq = lane >> 4;
    _Float16* base = out + ((long long)tm * 256 + wm * 128) * N + tn * 256 + wn * 64;
    if (PAT == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp)
                *reinterpret_cast<h8*>(base + (long long)(i * 16 + frow) * N + jp * 32 + fq * 8) = v[i * 2 + jp];
    } else if (PAT == 1) {
#pragma unroll
        for (int it = 0; it < 16; ++it)
            *reinterpret_cast<h8*>(base + (long long)(it * 8 + (lane >> 3)) * N + (lane & 7) * 8) = v[it];
    } else if (PAT == 4) {
#pragma unroll
        for (int it = 0; it < 16; ++it)
            *reinterpret_cast<h8*>(base + (long long)(it * 8 + (frow & 7)) * N + (fq + 4 * (frow >> 3)) * 8) = v[it];
    } else {
        char* sw = smem + wave * 16384;          // this wave's 128 rows x 128 bytes
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const int r = i * 16 + frow, c = jp * 4 + fq;
                *reinterpret_cast<h8*>(sw + r * 128 + ((c ^ (r & 7)) << 4)) = v[i * 2 + jp];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = it * 8 + (lane >> 3), c = lane & 7;
            const h8 x = *reinterpret_cast<const h8*>(sw + r * 128 + ((c ^ (r & 7)) << 4));
            *reinterpret_cast<h8*>(base + (long long)r * N + c * 8) = x;
        }
    }
}

int main(int argc, char** argv) {
    const int ntm = 385, N = argc > 1 ? atoi(argv[1]) : 3072, ntn = N / 256;
    const long long M = (long long)ntm * 256;
    _Float16* out;
    hipMalloc(&out, M * N * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    void (*ks[5])(_Float16*, int, int, int, long long) = {k<0>, k<1>, k<2>, k<3>, k<4>};
    for (int p = 0; p < 5; ++p) hipFuncSetAttribute((const void*)ks[p], hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    printf("N = %d: %d tiles of 256 x 256 fp16 = %.0f MB per launch; wall_clock64 at 100 MHz\n", N, ntm * ntn, M * N * 2 / 1e6);
    for (int spin_us = 0; spin_us <= 30; spin_us += 30) {
        for (int rep = 0; rep < 2; ++rep)
            for (int p = 0; p < 5; ++p) {
                for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(ks[p], dim3(ntm * ntn), dim3(512), 131072, 0, out, ntm, ntn, N, (long long)spin_us * 100);
                hipEventRecord(e0, 0);
                for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(ks[p], dim3(ntm * ntn), dim3(512), 131072, 0, out, ntm, ntn, N, (long long)spin_us * 100);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                printf("spin %2d us  pattern %d: %8.1f us per launch\n", spin_us, p, ms * 200.0f);
            }
    }
    return 0;
}
