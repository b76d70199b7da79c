// One level of the metric-bins head's projector path in ONE launch (HF modeling_zoedepth.py:749-772 Projector, :726-730 the attractor level's
// sum, and -- at the last level -- the embedding half of the conditional log-binomial MLP's first layer, :376-491):
//
//   e1  = relu(bilinear_align_corners(z) + b_c1)            z: the projector's first 1x1 convolution taken at the LOW resolution (the plan moves
//                                                           it in front of the upsampling; rounds 5: bs_resize_bias_relu_nhwc wrote e1, 256 B / pixel)
//   emb = W_c2 e1 + b_c2                                    rounds 2-5: a 3-pass bs_gemm on (hi | lo) pairs, 256 B in + 512 B out per pixel
//   x   = round16(emb + bilinear_align_corners(emb_prev))   rounds 4-5: formed inside bs_mlp2_add from emb (512 B) + the corners of emb_prev
//   Eh  = W_e e1 + b_e  (last level only)                   rounds 2-5: another 3-pass bs_gemm over e1, 256 B in + 320 B out per pixel
//
// z and emb_prev live on the SAME low-resolution grid (every level doubles the previous one), so one set of corners and weights serves both
// gathers.  Here e1 exists only as a wave's 8 KiB LDS tile and emb only in accumulators: per pixel of the finest level the kernel reads the
// corners of z and emb_prev (768 B per low-resolution pixel, shared by four outputs) and writes x (256 B), Eh (320 B) and -- at the levels that
// hand their embedding on -- emb as (hi | lo) pairs (512 B).  Rounds 2-5 moved 2 368 B per pixel through HBM for the same result.
//
// Block = 512 threads, persistent; the four weight planes (W_c2 hi / lo, W_e hi / lo: 52 KiB) are staged ONCE per block by LDS-DMA under the
// implicit-GEMM kernel's XOR swizzle.  After that there is no block-wide barrier: a wave owns 32 consecutive pixels of one output row (W is a
// multiple of 32), its own 8 KiB LDS slice, and runs gather -> products -> epilogue on its own, so the eight waves of a CU drift apart and one
// wave's gathers run under another's MFMAs.  Products: v_mfma_f32_16x16x32 in the K order of the 3-pass pair GEMM they replace
// (e1_hi W_hi, e1_lo W_hi, e1_hi W_lo): the accumulators hold the bits bs_gemm's held.
#include "common.h"

namespace bs {

constexpr int PL_PM = 64, PL_E = 128, PL_NE_MAX = 80;
constexpr int PL_WE_OFF = 2 * PL_E * 128;                          // W_c2 hi | lo planes first (rows of 128 B)
constexpr int PL_W_BYTES = PL_WE_OFF + 2 * PL_NE_MAX * 128;        // 53 248
constexpr int PL_SLICE = 8192, PL_LDS = PL_W_BYTES + 8 * PL_SLICE;

struct PlArgs {
    const void *z, *prev, *Wc2, *We;
    const float *b1, *bc2, *be;
    void *x_out, *emb_out;
    float* eh_out;
    int B, Hl, Wl, H, W, NE;
    float sy, sx;
    int nslices;
};

// (hipcc, ROCm 7.2, has interleaved the vector reads of a product's accumulators with MFMAs still in flight -- a whole row fragment wrong now and
// then, profiles/r05_gemm_experiments.txt (3): all MFMAs of a product first, this fence, then the reads)
#define PL_MFMA_FENCE()                                    \
    do {                                                   \
        __builtin_amdgcn_sched_barrier(0);                 \
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)

template <typename T>
__global__ __launch_bounds__(512) void projector_level_kernel(PlArgs a) {
    typedef typename T16<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4, sw = frow & 7;
    const int c8 = lane & 7, pq = lane >> 3;
    const int NE = a.NE, nje = NE >> 4;

    // ---- the weight planes, once: a wave instruction fills 8 rows x 128 B (lane -> row l / 8, chunk position l % 8 <- source chunk (l % 8) ^ (row & 7)).
    // Source rows are [W_hi (PM) | W_hi (PM) | W_lo (PM)] (the 3-pass pair GEMM's packing): plane 0 = columns 0 .., plane 1 = columns 2 PM ..
    {
        const int r8 = lane >> 3, chunk = (lane & 7) ^ r8;
        const int ngc = PL_E / 8, nge = NE / 8, total = 2 * ngc + (a.We ? 2 * nge : 0);
        for (int g = wave; g < total; g += 8) {
            const T* src;
            char* dst;
            if (g < 2 * ngc) {
                const int part = g / ngc, rg = g - part * ngc;
                src = reinterpret_cast<const T*>(a.Wc2) + (int64_t)(rg * 8 + r8) * (3 * PL_PM) + part * (2 * PL_PM) + chunk * 8;
                dst = smem + part * (PL_E * 128) + rg * 1024;
            } else {
                const int g2 = g - 2 * ngc, part = g2 / nge, rg = g2 - part * nge;
                src = reinterpret_cast<const T*>(a.We) + (int64_t)(rg * 8 + r8) * (3 * PL_PM) + part * (2 * PL_PM) + chunk * 8;
                dst = smem + PL_WE_OFF + part * (NE * 128) + rg * 1024;
            }
            glds16(src, dst);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    char* my = smem + PL_W_BYTES + wave * PL_SLICE;
    const T* zp = reinterpret_cast<const T*>(a.z);
    const T* pp = reinterpret_cast<const T*>(a.prev);
    T* xo = reinterpret_cast<T*>(a.x_out);
    T* eo = reinterpret_cast<T*>(a.emb_out);
    float bz[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bz[e] = a.b1[c8 * 8 + e];

    auto ld8 = [&](const T* ptr, int lo_off, float (&dst)[8]) {          // hi + lo of 8 channels
        const v8 h = *reinterpret_cast<const v8*>(ptr);
        const v8 l = *reinterpret_cast<const v8*>(ptr + lo_off);
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = (float)h[e];
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] += (float)l[e];
    };

    for (int s = blockIdx.x * 8 + wave; s < a.nslices; s += gridDim.x * 8) {
        // the slice: 32 consecutive pixels of output row (b, oy), columns ox0 .. ox0 + 31 -- all wave-uniform
        const int m0 = s * 32;
        const int row = m0 / a.W, ox0 = m0 - row * a.W;
        const int b = row / a.H, oy = row - b * a.H;
        const float fy = a.sy * (float)oy;
        int y0 = (int)fy;
        y0 = y0 > a.Hl - 1 ? a.Hl - 1 : y0;
        const int y1 = y0 + (y0 < a.Hl - 1 ? 1 : 0);
        const float ly = fy - (float)y0, hy = 1.0f - ly;
        const f32x2_ hy2 = {hy, hy}, ly2 = {ly, ly};
        const int64_t img = (int64_t)b * a.Hl * a.Wl;
        const int r0 = y0 * a.Wl, r1 = y1 * a.Wl;

        // ---- e1 = relu(bilinear(z) + b_c1) as (hi | lo) pairs into the wave's slice: [hi: 32 rows x 128 B][lo: 32 rows x 128 B], chunk c of row p at c ^ (p & 7)
        const T* zb = zp + img * (2 * PL_PM) + c8 * 8;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int p = it * 8 + pq, ox = ox0 + p;
            const float fx = a.sx * (float)ox;
            int x0 = (int)fx;
            x0 = x0 > a.Wl - 1 ? a.Wl - 1 : x0;
            const int x1 = x0 + (x0 < a.Wl - 1 ? 1 : 0);
            const float lx = fx - (float)x0, hx = 1.0f - lx;
            float q00[8], q01[8], q10[8], q11[8];
            ld8(zb + (int64_t)(r0 + x0) * (2 * PL_PM), PL_PM, q00);
            ld8(zb + (int64_t)(r0 + x1) * (2 * PL_PM), PL_PM, q01);
            ld8(zb + (int64_t)(r1 + x0) * (2 * PL_PM), PL_PM, q10);
            ld8(zb + (int64_t)(r1 + x1) * (2 * PL_PM), PL_PM, q11);
            const f32x2_ hx2 = {hx, hx}, lx2 = {lx, lx};
            v8 o, ol;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                f32x2_ v = bilerp2(f32x2_{q00[e], q00[e + 1]}, f32x2_{q01[e], q01[e + 1]}, f32x2_{q10[e], q10[e + 1]}, f32x2_{q11[e], q11[e + 1]}, hx2, lx2,
                                   hy2, ly2);
                v += f32x2_{bz[e], bz[e + 1]};
                v = f32x2_{fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f)};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    o[e + u] = T16<T>::from_f32(v[u]);
                    ol[e + u] = T16<T>::from_f32(v[u] - (float)o[e + u]);
                }
            }
            const int pos = p * 128 + ((c8 ^ (p & 7)) << 4);
            *reinterpret_cast<v8*>(my + pos) = o;
            *reinterpret_cast<v8*>(my + 4096 + pos) = ol;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (the wave's own LDS traffic is in order; this keeps the compiler's order too)
        __builtin_amdgcn_wave_barrier();

        // ---- Eh = W_e e1 + b_e (last level): rows m = 16 i + frow, outputs n = 16 j + 4 fq + e
        if (a.eh_out) {
            f32x4 acc[2][PL_NE_MAX / 16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < PL_NE_MAX / 16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pass = 0; pass < 3; ++pass) {                    // e1_hi W_hi, e1_lo W_hi, e1_hi W_lo
                const int aseg = pass == 1 ? 4096 : 0, wpart = pass == 2 ? NE * 128 : 0;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int coff = ((ks * 4 + fq) ^ sw) << 4;
                    v8 xf[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) xf[i] = *reinterpret_cast<const v8*>(my + aseg + (i * 16 + frow) * 128 + coff);
#pragma unroll
                    for (int j = 0; j < PL_NE_MAX / 16; ++j) {
                        if (j < nje) {
                            const v8 wf = *reinterpret_cast<const v8*>(smem + PL_WE_OFF + wpart + (j * 16 + frow) * 128 + coff);
#pragma unroll
                            for (int i = 0; i < 2; ++i) acc[i][j] = T16<T>::mfma16(wf, xf[i], acc[i][j]);
                        }
                    }
                }
            }
            PL_MFMA_FENCE();
#pragma unroll
            for (int j = 0; j < PL_NE_MAX / 16; ++j) {
                if (j < nje) {
                    const int n = j * 16 + fq * 4;
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(a.be + n);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const f32x4 y = acc[i][j] + bb;
                        *reinterpret_cast<f32x4*>(a.eh_out + (int64_t)(m0 + i * 16 + frow) * NE + n) = y;
                    }
                }
            }
        }

        // ---- emb = W_c2 e1 + b_c2
        f32x4 acc[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            const int aseg = pass == 1 ? 4096 : 0, wpart = pass == 2 ? PL_E * 128 : 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int coff = ((ks * 4 + fq) ^ sw) << 4;
                v8 xf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) xf[i] = *reinterpret_cast<const v8*>(my + aseg + (i * 16 + frow) * 128 + coff);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const v8 wf = *reinterpret_cast<const v8*>(smem + wpart + (j * 16 + frow) * 128 + coff);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = T16<T>::mfma16(wf, xf[i], acc[i][j]);
                }
            }
        }
        PL_MFMA_FENCE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // every fragment of e1 has been read: the slice becomes the staging tile
        __builtin_amdgcn_wave_barrier();

        // ---- x = round16(emb + bilinear(emb_prev)), 64 output channels at a time through the slice as fp32 [32 rows][64], 16-byte chunk c of
        // row m at c ^ (m & 15): the accumulator layout (a lane: 4 channels of one row) is turned into the gather's (a lane: 8 channels of a pixel)
        const T* pb = pp + img * (2 * PL_E) + c8 * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bc2 + h * 64 + jj * 16 + fq * 4);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = i * 16 + frow;
                    *reinterpret_cast<f32x4*>(my + m * 256 + (((jj * 4 + fq) ^ frow) << 4)) = acc[i][h * 4 + jj] + bb;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int p = it * 8 + pq, ox = ox0 + p;
                const float fx = a.sx * (float)ox;
                int x0 = (int)fx;
                x0 = x0 > a.Wl - 1 ? a.Wl - 1 : x0;
                const int x1 = x0 + (x0 < a.Wl - 1 ? 1 : 0);
                const float lx = fx - (float)x0, hx = 1.0f - lx;
                float q00[8], q01[8], q10[8], q11[8];
                const T* pc = pb + h * 64;
                ld8(pc + (int64_t)(r0 + x0) * (2 * PL_E), PL_E, q00);
                ld8(pc + (int64_t)(r0 + x1) * (2 * PL_E), PL_E, q01);
                ld8(pc + (int64_t)(r1 + x0) * (2 * PL_E), PL_E, q10);
                ld8(pc + (int64_t)(r1 + x1) * (2 * PL_E), PL_E, q11);
                const f32x4 e0 = *reinterpret_cast<const f32x4*>(my + p * 256 + (((2 * c8) ^ (p & 15)) << 4));
                const f32x4 e1v = *reinterpret_cast<const f32x4*>(my + p * 256 + (((2 * c8 + 1) ^ (p & 15)) << 4));
                const float em[8] = {e0[0], e0[1], e0[2], e0[3], e1v[0], e1v[1], e1v[2], e1v[3]};
                const f32x2_ hx2 = {hx, hx}, lx2 = {lx, lx};
                v8 o;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    f32x2_ v = bilerp2(f32x2_{q00[e], q00[e + 1]}, f32x2_{q01[e], q01[e + 1]}, f32x2_{q10[e], q10[e + 1]}, f32x2_{q11[e], q11[e + 1]}, hx2,
                                       lx2, hy2, ly2);
                    v += f32x2_{em[e], em[e + 1]};
                    o[e] = T16<T>::from_f32(v[0]);
                    o[e + 1] = T16<T>::from_f32(v[1]);
                }
                *reinterpret_cast<v8*>(xo + (int64_t)(m0 + p) * PL_E + h * 64 + c8 * 8) = o;
                if (eo) {                    // the level hands its embedding on: emb itself as a (hi | lo) pair
                    v8 eh, el;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        eh[e] = T16<T>::from_f32(em[e]);
                        el[e] = T16<T>::from_f32(em[e] - (float)eh[e]);
                    }
                    T* ep = eo + (int64_t)(m0 + p) * (2 * PL_E) + h * 64 + c8 * 8;
                    *reinterpret_cast<v8*>(ep) = eh;
                    *reinterpret_cast<v8*>(ep + PL_E) = el;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the staging tile has been read: the next half (or the next slice's e1) may overwrite it
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <typename T>
static int launch_projector_level(const PlArgs& a, hipStream_t st) {
    BS_MAX_DYNAMIC_LDS(((const void*)projector_level_kernel<T>), PL_LDS);
    const int nblk = cdiv(a.nslices, 8);
    const int grid = nblk < cu_count() ? nblk : cu_count();          // one 512-thread block per CU (LDS), persistent over the slices
    hipLaunchKernelGGL((projector_level_kernel<T>), dim3(grid), dim3(512), PL_LDS, st, a);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

}  // namespace bs

extern "C" int bs_projector_level(const void* z, const float* b_c1, const void* emb_prev, const void* Wc2, const float* b_c2, const void* We,
                                  const float* b_e, void* x_out, void* emb_out, float* eh_out, int32_t B, int32_t Hl, int32_t Wl, int32_t H,
                                  int32_t W, int32_t PM, int32_t E, int32_t NE, int32_t dtype, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("bs_projector_level: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(z && b_c1 && emb_prev && Wc2 && b_c2 && x_out, "bs_projector_level: null operand");
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "bs_projector_level: dtype must be f16 or bf16");
    BS_REQUIRE(PM == PL_PM && E == PL_E, "bs_projector_level: built for a %d-channel hidden map and a %d-channel embedding (got %d, %d)", PL_PM, PL_E, PM, E);
    BS_REQUIRE((We == nullptr) == (eh_out == nullptr) && (We == nullptr || b_e != nullptr), "bs_projector_level: We, b_e and eh_out go together");
    BS_REQUIRE(We == nullptr || (NE > 0 && NE <= PL_NE_MAX && NE % 16 == 0), "bs_projector_level: NE = %d must be a multiple of 16 up to %d", NE, PL_NE_MAX);
    BS_REQUIRE(B > 0 && Hl > 0 && Wl > 0 && H > 0 && W > 0 && W % 32 == 0, "bs_projector_level: bad geometry (W must be a multiple of 32: a wave takes 32 pixels of one row)");
    BS_REQUIRE((int64_t)B * H * W < 0x7FFFFFFFll && (int64_t)Hl * Wl * 2 * PL_E < 0x7FFFFFFFll, "bs_projector_level: too large for 32-bit pixel indices");
    PlArgs a;
    a.z = z; a.prev = emb_prev; a.Wc2 = Wc2; a.We = We; a.b1 = b_c1; a.bc2 = b_c2; a.be = b_e;
    a.x_out = x_out; a.emb_out = emb_out; a.eh_out = eh_out;
    a.B = B; a.Hl = Hl; a.Wl = Wl; a.H = H; a.W = W; a.NE = We ? NE : 16;
    a.sy = H > 1 ? (float)(Hl - 1) / (float)(H - 1) : 0.f;          // align_corners = True, as bs_resize_bias_relu_nhwc / bs_add_resized
    a.sx = W > 1 ? (float)(Wl - 1) / (float)(W - 1) : 0.f;
    a.nslices = (int)((int64_t)B * H * W / 32);
    hipStream_t st = (hipStream_t)stream;
    return dtype == BS_F16 ? launch_projector_level<f16>(a, st) : launch_projector_level<bf16>(a, st);
}
