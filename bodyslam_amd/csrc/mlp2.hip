// Two 1x1 convolutions back to back in one kernel: out = act2(relu(x W1^T + b1) W2^T + b2), the attractor MLP of the metric-bins head
// (HF modeling_zoedepth.py:665-700: Conv2d(128 -> 256, 1) + ReLU + Conv2d(256 -> 2 n_attractors, 1) + softplus, both heads' MLPs stacked).
// As two bs_gemm launches the 256-channel hidden map of the finest level (6.3 M pixels at the bench batch) is written and read back: 3.2 GB
// each way, 1.6 + 0.9 ms per step, for a product whose inputs and outputs are 1.6 + 0.2 GB.  Here the hidden tile never leaves the CU:
//   block = 256 pixels, 512 threads.  LDS (128 KiB): x tile [2 segments][256 rows][128 B] + W1 [2][256][128 B], both by LDS-DMA with the
//   16-byte chunks of a row XOR-swizzled by (row & 7) -- the implicit-GEMM kernel's image, conflict-free ds_read_b128 for the 16x16x32 MFMA.
//   product 1: waves 4 x 2, each 64 rows x 128 hidden units (128 accumulators); + b1, ReLU, rounded to the 16-bit type -- the value the
//   two-launch path stores -- and written back over the SAME LDS as the hidden tile [4 segments][256 rows][128 B].
//   product 2: each wave 32 rows x (<= 32) outputs over K = 256, W2 fragments straight from global (<= 16 KiB, L2-resident); + b2,
//   softplus, fp32 rows out.
// Same MFMA, same K order, same roundings as the two launches: bit-identical results (tests/test_ops_gpu.py::test_mlp2).
#include "common.h"

namespace bs {

constexpr int MLP_BM = 256, MLP_K1 = 128, MLP_N1 = 256, MLP_LDS = 131072;

template <typename T>
__global__ __launch_bounds__(512) void mlp2_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ W1, const float* __restrict__ b1,
                                                   const T* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ out, int M, int N2,
                                                   int act2) {
    typedef typename T16<T>::v8 v8;
    typedef typename T16<T>::v4 v4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * MLP_BM;
    const int frow = lane & 15, fq = lane >> 4;

    // ---- stage x and W1: a wave instruction fills 8 rows x 128 B (lane -> row l / 8, chunk position l % 8 <- source chunk (l % 8) ^ (row & 7))
    {
        const int r8 = lane >> 3, chunk = (lane & 7) ^ r8;      // (row & 7 = r8: the row groups start at multiples of 8)
        const bool whole = m0 + MLP_BM <= M;                     // (block-uniform) the last block clamps its rows
        const char* xl = reinterpret_cast<const char*>(x + (int64_t)(m0 + r8) * ldx + chunk * 8);
        const char* wl = reinterpret_cast<const char*>(W1 + (int64_t)r8 * MLP_K1 + chunk * 8);
        const int64_t xstep = (int64_t)8 * ldx * 2;
#pragma unroll
        for (int it = 0; it < 4; ++it) {                          // this wave's row groups: wave, wave + 8, wave + 16, wave + 24
            const int rg = wave + it * 8;
            const char* xs = xl + rg * xstep;
            if (!whole) {
                int m = m0 + rg * 8 + r8;
                m = m < M ? m : M - 1;
                xs = reinterpret_cast<const char*>(x + (int64_t)m * ldx + chunk * 8);
            }
#pragma unroll
            for (int seg = 0; seg < 2; ++seg) {
                glds16(xs + seg * 128, smem + seg * 32768 + rg * 1024);
                glds16(wl + (int64_t)rg * 8 * MLP_K1 * 2 + seg * 128, smem + 65536 + seg * 32768 + rg * 1024);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- product 1: acc1[i][j] rows 64 wr + 16 i + frow, hidden units 128 wc + 16 j + 4 fq + e
    const int wr = wave >> 1, wc = wave & 1;
    f32x4 acc1[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sw = frow & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int seg = ks >> 1, kc = (ks & 1) * 4;
        const int coff = ((kc + fq) ^ sw) << 4;
        v8 xf[4], wf[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const v8*>(smem + seg * 32768 + (wr * 64 + i * 16 + frow) * 128 + coff);
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[j] = *reinterpret_cast<const v8*>(smem + 65536 + seg * 32768 + (wc * 128 + j * 16 + frow) * 128 + coff);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc1[i][j] = T16<T>::mfma16(wf[j], xf[i], acc1[i][j]);
    }
    // W2 fragments of product 2 (rows n = 16 j + frow of W2, k = 32 ks + 8 fq ..): issued now, used after the hidden tile is in LDS
    const int NF2 = (N2 + 15) >> 4;
    v8 w2f[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int n = j * 16 + frow;
            v8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = T16<T>::from_f32(0.f);
            w2f[j][ks] = (j < NF2 && n < N2) ? *reinterpret_cast<const v8*>(W2 + (int64_t)n * MLP_N1 + ks * 32 + fq * 8) : z;
        }
    __syncthreads();            // every wave is done reading x / W1: the hidden tile takes their place
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = wc * 128 + j * 16 + fq * 4;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b1 + n);
        const int seg = n >> 6, chunk = (n & 63) >> 3, sub = (n & 7) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = wr * 64 + i * 16 + frow;
            v4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = T16<T>::from_f32(fmaxf(acc1[i][j][e] + bb[e], 0.0f));
            *reinterpret_cast<v4*>(smem + seg * 32768 + m * 128 + ((chunk ^ (m & 7)) << 4) + sub) = h;
        }
    }
    __syncthreads();

    // ---- product 2: rows 32 wave + 16 i + frow, outputs 16 j + 4 fq + e
    f32x4 acc2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int seg = ks >> 1, kc = (ks & 1) * 4;
        const int coff = ((kc + fq) ^ sw) << 4;
        v8 af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const v8*>(smem + seg * 32768 + (wave * 32 + i * 16 + frow) * 128 + coff);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (j < NF2) acc2[i][j] = T16<T>::mfma16(w2f[j][ks], af[i], acc2[i][j]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = j * 16 + fq * 4;
        if (j >= NF2 || n >= N2) continue;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + n);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wave * 32 + i * 16 + frow;
            if (m >= M) continue;
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = apply_act(acc2[i][j][e] + bb[e], act2);
            *reinterpret_cast<f32x4*>(out + (int64_t)m * N2 + n) = y;
        }
    }
}

template <typename T>
static int launch_mlp2(const void* x, int ldx, const void* W1, const float* b1, const void* W2, const float* b2, float* out, int M, int N2, int act2,
                       hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        BS_CHECK_HIP(hipFuncSetAttribute((const void*)mlp2_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS));
        attr_done = true;
    }
    hipLaunchKernelGGL((mlp2_kernel<T>), dim3(cdiv(M, MLP_BM)), dim3(512), MLP_LDS, st, (const T*)x, ldx, (const T*)W1, b1, (const T*)W2, b2, out, M, N2,
                       act2);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

}  // namespace bs

extern "C" int bs_mlp2(const void* x, int32_t ldx, const void* W1, const float* b1, const void* W2, const float* b2, float* out, int32_t M,
                       int32_t K1, int32_t N1, int32_t N2, int32_t act2, int32_t dtype, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("bs_mlp2: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(x && W1 && b1 && W2 && b2 && out, "bs_mlp2: null operand");
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "bs_mlp2: dtype must be f16 or bf16");
    BS_REQUIRE(K1 == MLP_K1 && N1 == MLP_N1, "bs_mlp2: built for K1 = %d, N1 = %d (got %d, %d)", MLP_K1, MLP_N1, K1, N1);
    BS_REQUIRE(N2 > 0 && N2 <= 32 && N2 % 4 == 0, "bs_mlp2: N2 = %d must be a multiple of 4 in 4 .. 32", N2);
    BS_REQUIRE(M > 0 && ldx >= K1 && ldx % 8 == 0, "bs_mlp2: M = %d, ldx = %d (rows of >= K1 16-bit values, 16-byte aligned)", M, ldx);
    BS_REQUIRE(act2 == BS_ACT_NONE || act2 == BS_ACT_RELU || act2 == BS_ACT_SOFTPLUS || act2 == BS_ACT_GELU, "bs_mlp2: unknown activation %d", act2);
    hipStream_t st = (hipStream_t)stream;
    return dtype == BS_F16 ? launch_mlp2<f16>(x, ldx, W1, b1, W2, b2, out, M, N2, act2, st) : launch_mlp2<bf16>(x, ldx, W1, b1, W2, b2, out, M, N2, act2, st);
}
