// Probe: per-lane LDS reads of one kernel beside MFMA instructions of ANOTHER kernel on the same CUs (two streams of one process).
// profiles/r06_reproducibility.txt (8): the four-byte-gather form of bs_logbinom_depth_ex returns wrong values in the last 16 lanes of a wave when
// bs_rank1_bias (v_mfma_f32_16x16x32_bf16, 17 KiB of LDS: it fits beside the victim's blocks) runs on a second stream -- and not when that kernel's
// MFMA is taken out.  Here: stream A = a kernel that does nothing but ds_read_b32 / b64 / b128 gathers into registers holding a sentinel and checks
// them; stream B = a kernel that does nothing but MFMA (bf16 16x16x32, f16 16x16x32, f16 32x32x16) or plain FMAs.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_lds_probe mfma_lds_probe.hip ;  ./mfma_lds_probe [seconds per combination]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int W, int VALU>      // bytes per lane and read: 4, 8, 16; packed FMAs per pass
__global__ __launch_bounds__(256) void victim(int iters, unsigned long long* bad) {
    f32x2 p0 = {1.0f + threadIdx.x * 1e-6f, 0.5f}, p1 = {0.25f, 0.125f};
    const f32x2 pk = {0.999f, 1.001f};
    __shared__ __attribute__((aligned(16))) unsigned lds[10240];          // 40 KiB: three blocks per CU, like the kernel it stands for
    for (int i = threadIdx.x; i < 10240; i += 256) lds[i] = 0x40000000u + i;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // four cells of 160 bytes per lane, rows of the tile share cells (broadcast + bank conflicts), the last 16 lanes read other cells
    const unsigned cell = (lane & 15) / 2 + (lane >> 4 == 3 ? 9 : 0) + wv * 18;
    const unsigned a0 = cell * 160, a1 = (cell + 1) * 160, a2 = (cell + 9) * 160, a3 = (cell + 10) * 160;
    unsigned long long nbad = 0, q3 = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned off = (it % 10) * 16;
        unsigned o[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[j][e] = 0xffffffffu;
        const unsigned b0 = a0 + off, b1 = a1 + off, b2 = a2 + off, b3 = a3 + off;
        if (W == 4)
            asm volatile("s_nop 4\n ds_read_b32 %0, %4\n ds_read_b32 %1, %5\n ds_read_b32 %2, %6\n ds_read_b32 %3, %7\n s_waitcnt lgkmcnt(0)\n"
                         : "+v"(o[0][0]), "+v"(o[1][0]), "+v"(o[2][0]), "+v"(o[3][0]) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "memory");
        else if (W == 8) {
            unsigned long long r0 = ~0ull, r1 = ~0ull, r2 = ~0ull, r3 = ~0ull;
            asm volatile("s_nop 4\n ds_read_b64 %0, %4\n ds_read_b64 %1, %5\n ds_read_b64 %2, %6\n ds_read_b64 %3, %7\n s_waitcnt lgkmcnt(0)\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "memory");
            o[0][0] = (unsigned)r0; o[0][1] = (unsigned)(r0 >> 32); o[1][0] = (unsigned)r1; o[1][1] = (unsigned)(r1 >> 32);
            o[2][0] = (unsigned)r2; o[2][1] = (unsigned)(r2 >> 32); o[3][0] = (unsigned)r3; o[3][1] = (unsigned)(r3 >> 32);
        } else {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 r0 = {~0u, ~0u, ~0u, ~0u}, r1 = r0, r2 = r0, r3 = r0;
            asm volatile("s_nop 4\n ds_read_b128 %0, %4\n ds_read_b128 %1, %5\n ds_read_b128 %2, %6\n ds_read_b128 %3, %7\n s_waitcnt lgkmcnt(0)\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "memory");
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[0][e] = r0[e]; o[1][e] = r1[e]; o[2][e] = r2[e]; o[3][e] = r3[e]; }
        }
        const unsigned bb[4] = {b0, b1, b2, b3};
        int wrong = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < W / 4; ++e) wrong += o[j][e] != 0x40000000u + bb[j] / 4 + e;
        nbad += wrong;
        if (lane >= 48) q3 += wrong;
        if (VALU) {          // the victim's own arithmetic: packed-fp32 FMAs (other waves of the SIMD run these while this one reads LDS) and a transcendental
#pragma unroll
            for (int r = 0; r < VALU; ++r) {
                p0 = __builtin_elementwise_fma(p0, pk, p1);
                p1 = __builtin_elementwise_fma(p1, pk, p0);
            }
            p0[0] += __builtin_amdgcn_rcpf(p1[1] + 2.0f);
        }
    }
    if (p0[0] == 123.456f) nbad += 1;                      // (keeps the arithmetic)
    if (nbad) { atomicAdd(&bad[0], nbad); atomicAdd(&bad[1], q3); }
}

template <int KIND>   // 0: v_mfma_f32_16x16x32_bf16, 1: v_mfma_f32_16x16x32_f16, 2: v_mfma_f32_32x32x16_f16, 3: plain FMAs
__global__ __launch_bounds__(256) void neighbour(int iters, float* sink) {
    __shared__ float pad[4352];                                            // 17 KiB, as bs_rank1_bias
    pad[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc16 = {};
    bf16x8 ab, bb;
    f16x8 ah, bh;
    for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)(0.001f * (threadIdx.x + e)); bb[e] = (__bf16)1.0f; ah[e] = (_Float16)(0.001f * (threadIdx.x + e)); bh[e] = (_Float16)1.0f; }
    float f = pad[(threadIdx.x * 7) & 255];
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc, 0, 0, 0);
        else if (KIND == 1) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
        else if (KIND == 2) acc16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc16, 0, 0, 0);
        else f = __builtin_fmaf(f, 1.0000001f, 0.5f);
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc16[0] + f;
}

// the victim as the real kernel lives: MANY SHORT blocks, each stages its 40 KiB from global memory (16-byte LDS writes: other blocks of the CU are staging
// while this one gathers), one barrier, 40 passes of four per-lane reads, done
template <int W>
__global__ __launch_bounds__(256) void victim_staged(const unsigned* src, unsigned long long* bad) {
    extern __shared__ __attribute__((aligned(16))) unsigned dl[];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned* g = src + (size_t)(blockIdx.x & 255) * 10240;
    for (int i = threadIdx.x; i < 2560; i += 256) *reinterpret_cast<u32x4*>(dl + 4 * i) = *reinterpret_cast<const u32x4*>(g + 4 * i);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, tag = (blockIdx.x & 255) << 16;
    const unsigned cell = (lane & 15) / 2 + (lane >> 4 == 3 ? 9 : 0) + wv * 18;
    const unsigned a0 = cell * 160, a1 = (cell + 1) * 160, a2 = (cell + 9) * 160, a3 = (cell + 10) * 160;
    unsigned long long nbad = 0, q3 = 0;
    for (int it = 0; it < 40; ++it) {
        const unsigned off = (it % 10) * 16 + (it / 10) * 4;
        const unsigned b0 = a0 + off, b1 = a1 + off, b2 = a2 + off, b3 = a3 + off;
        unsigned o0 = ~0u, o1 = ~0u, o2 = ~0u, o3 = ~0u;
        asm volatile("s_nop 4\n ds_read_b32 %0, %4\n ds_read_b32 %1, %5\n ds_read_b32 %2, %6\n ds_read_b32 %3, %7\n s_waitcnt lgkmcnt(0)\n"
                     : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "memory");
        const int wrong = (o0 != tag + b0 / 4) + (o1 != tag + b1 / 4) + (o2 != tag + b2 / 4) + (o3 != tag + b3 / 4);
        nbad += wrong;
        if (lane >= 48) q3 += wrong;
    }
    if (nbad) { atomicAdd(&bad[0], nbad); atomicAdd(&bad[1], q3); }
}

template <int KIND>
static void combo_staged(double secs, unsigned long long* bad, float* sink, const unsigned* src, hipStream_t sA, hipStream_t sB, const char* nname) {
    (void)hipMemset(bad, 0, 4 * sizeof(unsigned long long));
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        if (KIND >= 0)
            for (int r = 0; r < 24; ++r) hipLaunchKernelGGL(neighbour<(KIND < 0 ? 3 : KIND)>, dim3(192), dim3(256), 0, sB, 512, sink);
        for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(victim_staged<4>, dim3(4608), dim3(256), 41856, sA, src, bad);
        (void)hipDeviceSynchronize();
        launches += 8;
    }
    unsigned long long h[2];
    (void)hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    printf("  staged ds_read_b32 beside %-28s %7ld launches of 4 608 short blocks   wrong registers %10llu   (in lanes 48-63: %llu)\n", nname, launches, h[0], h[1]);
    fflush(stdout);
}

template <int W, int KIND, int VALU = 0>
static void combo(double secs, unsigned long long* bad, float* sink, hipStream_t sA, hipStream_t sB, const char* vname, const char* nname) {
    (void)hipMemset(bad, 0, 4 * sizeof(unsigned long long));
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        if (KIND >= 0)
            for (int r = 0; r < 24; ++r) hipLaunchKernelGGL(neighbour<(KIND < 0 ? 3 : KIND)>, dim3(192), dim3(256), 0, sB, 512, sink);
        for (int r = 0; r < 16; ++r) hipLaunchKernelGGL((victim<W, VALU>), dim3(288), dim3(256), 0, sA, 512, bad);
        (void)hipDeviceSynchronize();
        launches += 16;
    }
    unsigned long long h[2];
    (void)hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    printf("  %-14s beside %-28s %7ld launches   wrong registers %10llu   (in lanes 48-63: %llu)\n", vname, nname, launches, h[0], h[1]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    unsigned long long* bad;
    float* sink;
    (void)hipMalloc(&bad, 4 * sizeof(unsigned long long));
    (void)hipMalloc(&sink, 192 * 256 * 4);
    hipStream_t sA, sB;
    (void)hipStreamCreate(&sA);
    (void)hipStreamCreate(&sB);
    printf("victim: 288 blocks x 4 waves x 512 passes x four per-lane LDS reads per launch; neighbour: 192 blocks x 4 waves x 2048 instructions, four launches per burst\n");
    combo<4, -1>(secs, bad, sink, sA, sB, "ds_read_b32", "nothing");
    combo<4, 3>(secs, bad, sink, sA, sB, "ds_read_b32", "plain FMAs");
    combo<4, 0>(secs, bad, sink, sA, sB, "ds_read_b32", "mfma_f32_16x16x32_bf16");
    combo<4, 1>(secs, bad, sink, sA, sB, "ds_read_b32", "mfma_f32_16x16x32_f16");
    combo<4, 2>(secs, bad, sink, sA, sB, "ds_read_b32", "mfma_f32_32x32x16_f16");
    combo<8, 0>(secs, bad, sink, sA, sB, "ds_read_b64", "mfma_f32_16x16x32_bf16");
    combo<8, 2>(secs, bad, sink, sA, sB, "ds_read_b64", "mfma_f32_32x32x16_f16");
    combo<16, 0>(secs, bad, sink, sA, sB, "ds_read_b128", "mfma_f32_16x16x32_bf16");
    combo<16, 2>(secs, bad, sink, sA, sB, "ds_read_b128", "mfma_f32_32x32x16_f16");
    printf("the victim with 32 + 32 packed FMAs and a v_rcp after every pass:\n");
    combo<4, -1, 32>(secs, bad, sink, sA, sB, "ds_read_b32", "nothing");
    combo<4, 0, 32>(secs, bad, sink, sA, sB, "ds_read_b32", "mfma_f32_16x16x32_bf16");
    combo<4, 1, 32>(secs, bad, sink, sA, sB, "ds_read_b32", "mfma_f32_16x16x32_f16");
    combo<8, 1, 32>(secs, bad, sink, sA, sB, "ds_read_b64", "mfma_f32_16x16x32_f16");
    combo<16, 1, 32>(secs, bad, sink, sA, sB, "ds_read_b128", "mfma_f32_16x16x32_f16");
    unsigned* src;
    (void)hipMalloc(&src, (size_t)256 * 10240 * 4);
    {
        unsigned* hsrc = (unsigned*)malloc((size_t)256 * 10240 * 4);
        for (unsigned b = 0; b < 256; ++b)
            for (unsigned i = 0; i < 10240; ++i) hsrc[(size_t)b * 10240 + i] = (b << 16) + i;
        (void)hipMemcpy(src, hsrc, (size_t)256 * 10240 * 4, hipMemcpyHostToDevice);
        free(hsrc);
    }
    printf("the victim as many short blocks that stage their LDS content from global memory first (dynamic LDS, 41 856 bytes):\n");
    combo_staged<-1>(secs, bad, sink, src, sA, sB, "nothing");
    combo_staged<3>(secs, bad, sink, src, sA, sB, "plain FMAs");
    combo_staged<0>(secs, bad, sink, src, sA, sB, "mfma_f32_16x16x32_bf16");
    combo_staged<1>(secs, bad, sink, src, sA, sB, "mfma_f32_16x16x32_f16");
    return 0;
}
