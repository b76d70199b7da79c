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

// FUSE (bs_mlp2_add): the input row is not read but formed, x[m] = emb[m] + bilinear_align_corners(prev)[m] rounded to the 16-bit type -- the
// hi half of what bs_add_resized would store (HF modeling_zoedepth.py:726-730; same operations in the same order, same bits), SPLIT: emb and
// prev hold (hi | lo) pairs.  The attractor level's sum then exists only as this kernel's LDS tile.
struct MlpAddGeom {
    const void* prev;
    int Hp, Wp, H, W;
    float sy, sx;
};

template <typename T, int FUSE, int SPLIT>
__global__ __launch_bounds__(512) void mlp2_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ W1, const float* __restrict__ b1,
                                                   const T* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ out, int M, int N2,
                                                   int act2, MlpAddGeom ag) {
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
                if (!FUSE) glds16(xs + seg * 128, smem + seg * 32768 + rg * 1024);
                glds16(wl + (int64_t)rg * 8 * MLP_K1 * 2 + seg * 128, smem + 65536 + seg * 32768 + rg * 1024);
            }
        }
    }
    if constexpr (FUSE) {
        // item = (row, 8-channel group): consecutive threads take the 16 groups of a row (256 contiguous bytes of each operand half)
        const T* prev = reinterpret_cast<const T*>(ag.prev);
        constexpr int C = MLP_K1, PS = SPLIT ? 2 : 1;
        auto ld8 = [&](const T* ptr, float (&dst)[8]) {
            const v8 h = *reinterpret_cast<const v8*>(ptr);
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e] = (float)h[e];
            if (SPLIT) {
                const v8 l = *reinterpret_cast<const v8*>(ptr + C);
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[e] += (float)l[e];
            }
        };
        // per-row table behind the tiles (8 KiB): the four corner offsets into prev and the interpolation weights of the block's 256 pixels,
        // formed once per row instead of once per (row, channel group) -- two integer divisions and the coordinate arithmetic, x 16
        typedef int i32x4_ __attribute__((ext_vector_type(4)));
        i32x4_* rtab = reinterpret_cast<i32x4_*>(smem + MLP_LDS);
        if (tid < MLP_BM) {
            int m = m0 + tid;
            m = m < M ? m : M - 1;
            const int hw = ag.H * ag.W;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / ag.W, ox = rem - oy * ag.W;
            const float fy = ag.sy * (float)oy, fx = ag.sx * (float)ox;
            int y0 = (int)fy, x0 = (int)fx;
            y0 = y0 > ag.Hp - 1 ? ag.Hp - 1 : y0;
            x0 = x0 > ag.Wp - 1 ? ag.Wp - 1 : x0;
            const int y1 = y0 + (y0 < ag.Hp - 1 ? 1 : 0), x1 = x0 + (x0 < ag.Wp - 1 ? 1 : 0);
            const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
            const int pb0 = b * ag.Hp * ag.Wp;
            rtab[2 * tid] = i32x4_{(pb0 + y0 * ag.Wp + x0) * C * PS, (pb0 + y0 * ag.Wp + x1) * C * PS, (pb0 + y1 * ag.Wp + x0) * C * PS,
                                   (pb0 + y1 * ag.Wp + x1) * C * PS};
            rtab[2 * tid + 1] = i32x4_{__float_as_int(hx), __float_as_int(lx), __float_as_int(hy), __float_as_int(ly)};
        }
        __syncthreads();
#pragma unroll 2
        for (int k = 0; k < MLP_BM * 16 / 512; ++k) {       // (unrolled by four: slower, 2.07 vs 1.89 ms at the finest level)
            const int idx = tid + k * 512, r = idx >> 4, c8 = idx & 15;
            int m = m0 + r;
            m = m < M ? m : M - 1;
            const i32x4_ po = rtab[2 * r], pw = rtab[2 * r + 1];
            const float hx = __int_as_float(pw[0]), lx = __int_as_float(pw[1]), hy = __int_as_float(pw[2]), ly = __int_as_float(pw[3]);
            const T* pb = prev + c8 * 8;
            float q00[8], q01[8], q10[8], q11[8], av[8];
            ld8(pb + po[0], q00);
            ld8(pb + po[1], q01);
            ld8(pb + po[2], q10);
            ld8(pb + po[3], q11);
            ld8(x + (int64_t)m * ldx + c8 * 8, av);
            const f32x2_ hx2 = {hx, hx}, lx2 = {lx, lx}, hy2 = {hy, hy}, ly2 = {ly, ly};
            v8 o;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                f32x2_ v = bilerp2(f32x2_{q00[e], q00[e + 1]}, f32x2_{q01[e], q01[e + 1]}, f32x2_{q10[e], q10[e + 1]}, f32x2_{q11[e], q11[e + 1]}, hx2, lx2,
                                   hy2, ly2);
                v += f32x2_{av[e], av[e + 1]};
                o[e] = T16<T>::from_f32(v[0]);
                o[e + 1] = T16<T>::from_f32(v[1]);
            }
            *reinterpret_cast<v8*>(smem + (c8 >> 3) * 32768 + r * 128 + (((c8 & 7) ^ (r & 7)) << 4)) = o;
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

template <typename T, int FUSE, int SPLIT>
static int launch_mlp2(const void* x, int ldx, const void* W1, const float* b1, const void* W2, const float* b2, float* out, int M, int N2, int act2,
                       const MlpAddGeom& ag, hipStream_t st) {
    BS_MAX_DYNAMIC_LDS(((const void*)mlp2_kernel<T, FUSE, SPLIT>), MLP_LDS + (FUSE ? MLP_BM * 32 : 0));
    hipLaunchKernelGGL((mlp2_kernel<T, FUSE, SPLIT>), dim3(cdiv(M, MLP_BM)), dim3(512), MLP_LDS + (FUSE ? MLP_BM * 32 : 0), st, (const T*)x, ldx, (const T*)W1, b1, (const T*)W2, b2,
                       out, M, N2, act2, ag);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

static int mlp2_check(const char* who, const void* x, const void* W1, const float* b1, const void* W2, const float* b2, float* out, int M, int K1,
                      int N1, int N2, int act2, int dtype) {
    if (!initialized()) { set_error("%s: call bs_init first", who); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(x && W1 && b1 && W2 && b2 && out, "%s: null operand", who);
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "%s: dtype must be f16 or bf16", who);
    BS_REQUIRE(K1 == MLP_K1 && N1 == MLP_N1, "%s: built for K1 = %d, N1 = %d (got %d, %d)", who, MLP_K1, MLP_N1, K1, N1);
    BS_REQUIRE(N2 > 0 && N2 <= 32 && N2 % 4 == 0, "%s: N2 = %d must be a multiple of 4 in 4 .. 32", who, N2);
    BS_REQUIRE(M > 0, "%s: M = %d", who, M);
    BS_REQUIRE(act2 == BS_ACT_NONE || act2 == BS_ACT_RELU || act2 == BS_ACT_SOFTPLUS || act2 == BS_ACT_SOFTPLUS_FAST || act2 == BS_ACT_GELU, "%s: unknown activation %d", who, act2);
    return BS_OK;
}

}  // namespace bs

extern "C" int bs_mlp2(const void* x, int32_t ldx, const void* W1, const float* b1, const void* W2, const float* b2, float* out, int32_t M,
                       int32_t K1, int32_t N1, int32_t N2, int32_t act2, int32_t dtype, void* stream) {
    using namespace bs;
    const int rc = mlp2_check("bs_mlp2", x, W1, b1, W2, b2, out, M, K1, N1, N2, act2, dtype);
    if (rc != BS_OK) return rc;
    BS_REQUIRE(ldx >= K1 && ldx % 8 == 0, "bs_mlp2: ldx = %d (rows of >= K1 16-bit values, 16-byte aligned)", ldx);
    hipStream_t st = (hipStream_t)stream;
    const MlpAddGeom ag{};
    return dtype == BS_F16 ? launch_mlp2<f16, 0, 0>(x, ldx, W1, b1, W2, b2, out, M, N2, act2, ag, st)
                           : launch_mlp2<bf16, 0, 0>(x, ldx, W1, b1, W2, b2, out, M, N2, act2, ag, st);
}

extern "C" int bs_mlp2_add(const void* emb, const void* prev, const void* W1, const float* b1, const void* W2, const float* b2, float* out,
                           int32_t B, int32_t Hp, int32_t Wp, int32_t H, int32_t W, int32_t K1, int32_t N1, int32_t N2, int32_t act2, int32_t dtype,
                           void* stream) {
    using namespace bs;
    const int split = (dtype & 16) ? 1 : 0;           // bit 4: emb and prev hold (hi | lo) pairs of K1 channels each
    dtype &= 15;
    BS_REQUIRE(B > 0 && Hp > 0 && Wp > 0 && H > 0 && W > 0 && (int64_t)B * H * W < 0x7FFFFFFFll, "bs_mlp2_add: bad geometry");
    const int M = B * H * W;
    const int rc = mlp2_check("bs_mlp2_add", emb, W1, b1, W2, b2, out, M, K1, N1, N2, act2, dtype);
    if (rc != BS_OK) return rc;
    BS_REQUIRE(prev, "bs_mlp2_add: null operand");
    BS_REQUIRE((int64_t)B * Hp * Wp * K1 * (split ? 2 : 1) < 0x7FFFFFFFll, "bs_mlp2_add: prev too large for 32-bit element offsets");
    MlpAddGeom ag;
    ag.prev = prev; ag.Hp = Hp; ag.Wp = Wp; ag.H = H; ag.W = W;
    ag.sy = H > 1 ? (float)(Hp - 1) / (float)(H - 1) : 0.f;      // align_corners = True, as bs_add_resized
    ag.sx = W > 1 ? (float)(Wp - 1) / (float)(W - 1) : 0.f;
    const int ldx = K1 * (split ? 2 : 1);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BS_F16)
        return split ? launch_mlp2<f16, 1, 1>(emb, ldx, W1, b1, W2, b2, out, M, N2, act2, ag, st)
                     : launch_mlp2<f16, 1, 0>(emb, ldx, W1, b1, W2, b2, out, M, N2, act2, ag, st);
    return split ? launch_mlp2<bf16, 1, 1>(emb, ldx, W1, b1, W2, b2, out, M, N2, act2, ag, st)
                 : launch_mlp2<bf16, 1, 0>(emb, ldx, W1, b1, W2, b2, out, M, N2, act2, ag, st);
}
