// Probe: LDS reads and scalar loads in flight together, retired by ONE `s_waitcnt lgkmcnt(0)` -- the shape of logbinom_kernel's hidden layer in
// rounds 2-5 (four ds_read_b64 whose destination registers hold a sentinel before the read, the last one's destination being its own address
// register, interleaved with two s_load_dwordx16 from a page that changes every pass).  After the wait every lane checks what its registers hold
// against what the LDS holds: a lane that still sees the sentinel (or the address) consumed the read before it landed.
//   hipcc --offload-arch=gfx950 -O2 -o lgkm_mix_probe lgkm_mix_probe.hip ;  ./lgkm_mix_probe [seconds]      (alone and beside tools/probes/gpu_churn.py)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>

template <int V>
__global__ __launch_bounds__(256) void probe(const float* big, unsigned npages, int iters, unsigned long long* bad) {
    __shared__ __attribute__((aligned(16))) unsigned lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0x40000000u + i;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // per-lane gathers with bank conflicts, 8-byte aligned, inside [0, 32 KiB)
    const unsigned a0 = ((lane * 40 + wv * 8 + 0) & 4095) * 8, a1 = ((lane * 40 + wv * 8 + 1037) & 4095) * 8;
    const unsigned a2 = ((lane * 40 + wv * 8 + 2053) & 4095) * 8, a3 = ((lane * 40 + wv * 8 + 3079) & 4095) * 8;
    unsigned long long nbad = 0, q3 = 0;
    unsigned seed = blockIdx.x * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        seed = seed * 1664525u + 1013904223u;
        const float* p = big + (size_t)(__builtin_amdgcn_readfirstlane(seed >> 8) % npages) * 1024;       // a 4-KiB page, uniform
        unsigned o0, o1, o2, o3, o4, o5, o6, o7;
        if (V == 0)
            asm volatile("v_mov_b32 v20, -1\n v_mov_b32 v21, -1\n v_mov_b32 v22, -1\n v_mov_b32 v23, -1\n v_mov_b32 v24, -1\n v_mov_b32 v25, -1\n v_mov_b32 v26, -1\n"
                         "v_mov_b32 v27, %[a3]\n s_nop 4\n"
                         "ds_read_b64 v[20:21], %[a0]\n ds_read_b64 v[22:23], %[a1]\n s_load_dwordx16 s[36:51], %[p], 0x0\n"
                         "ds_read_b64 v[24:25], %[a2]\n ds_read_b64 v[26:27], v27\n s_load_dwordx16 s[52:67], %[p], 0x40\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         "v_mov_b32 %[o0], v20\n v_mov_b32 %[o1], v21\n v_mov_b32 %[o2], v22\n v_mov_b32 %[o3], v23\n"
                         "v_mov_b32 %[o4], v24\n v_mov_b32 %[o5], v25\n v_mov_b32 %[o6], v26\n v_mov_b32 %[o7], v27\n"
                         : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [o4] "=&v"(o4), [o5] "=&v"(o5), [o6] "=&v"(o6), [o7] "=&v"(o7)
                         : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [p] "s"(p)
                         : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47",
                           "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "memory");
        else                 // the same LDS reads with nothing scalar in flight
            asm volatile("v_mov_b32 v20, -1\n v_mov_b32 v21, -1\n v_mov_b32 v22, -1\n v_mov_b32 v23, -1\n v_mov_b32 v24, -1\n v_mov_b32 v25, -1\n v_mov_b32 v26, -1\n"
                         "v_mov_b32 v27, %[a3]\n s_nop 4\n"
                         "ds_read_b64 v[20:21], %[a0]\n ds_read_b64 v[22:23], %[a1]\n"
                         "ds_read_b64 v[24:25], %[a2]\n ds_read_b64 v[26:27], v27\n"
                         "s_waitcnt lgkmcnt(0)\n s_load_dwordx16 s[36:51], %[p], 0x0\n s_load_dwordx16 s[52:67], %[p], 0x40\n"
                         "v_mov_b32 %[o0], v20\n v_mov_b32 %[o1], v21\n v_mov_b32 %[o2], v22\n v_mov_b32 %[o3], v23\n"
                         "v_mov_b32 %[o4], v24\n v_mov_b32 %[o5], v25\n v_mov_b32 %[o6], v26\n v_mov_b32 %[o7], v27\n s_waitcnt lgkmcnt(0)\n"
                         : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [o4] "=&v"(o4), [o5] "=&v"(o5), [o6] "=&v"(o6), [o7] "=&v"(o7)
                         : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [p] "s"(p)
                         : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47",
                           "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "memory");
        const unsigned e0 = 0x40000000u + a0 / 4, e2 = 0x40000000u + a1 / 4, e4 = 0x40000000u + a2 / 4, e6 = 0x40000000u + a3 / 4;
        const int wrong = (o0 != e0) + (o1 != e0 + 1) + (o2 != e2) + (o3 != e2 + 1) + (o4 != e4) + (o5 != e4 + 1) + (o6 != e6) + (o7 != e6 + 1);
        nbad += wrong;
        if (lane >= 48) q3 += wrong;
    }
    if (nbad) {
        atomicAdd(&bad[0], nbad);
        atomicAdd(&bad[1], q3);
    }
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 20.0;
    const size_t bytes = (size_t)4 << 30;                 // 4 GiB: a new 4-KiB page per pass
    const unsigned npages = (unsigned)(bytes / 4096);
    float* big;
    unsigned long long* bad;
    if (hipMalloc(&big, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(big, 0, bytes);
    (void)hipMalloc(&bad, 4 * sizeof(unsigned long long));
    (void)hipMemset(bad, 0, 4 * sizeof(unsigned long long));
    long launches = 0;
    const int blocks = 256 * 3, iters = 2048;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, big, npages, iters, bad);
        hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, big, npages, iters, bad + 2);
        (void)hipDeviceSynchronize();
        ++launches;
    }
    unsigned long long h[4];
    (void)hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    printf("%ld launches of each form, %d waves x %d passes x 8 registers per launch\n", launches, blocks * 4, iters);
    printf("  LDS reads with scalar loads in flight, one wait:   wrong registers %llu  (of them in lanes 48-63: %llu)\n", h[0], h[1]);
    printf("  LDS reads alone, scalar loads behind their wait:    wrong registers %llu  (of them in lanes 48-63: %llu)\n", h[2], h[3]);
    return 0;
}
