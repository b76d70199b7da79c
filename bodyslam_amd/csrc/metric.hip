// Metric-bins head (ZoeD_NK) per-pixel kernels and the domain router's small pieces.
// The [B,64,384,512] bin/probability tensors of the reference are never materialised: the final
// kernel interpolates bin centres, evaluates the per-pixel MLP tail, the log-binomial softmax and
// the expectation in registers.
//   HF modeling_zoedepth.py:376-491 (log-binomial), :665-746 (attractor, unnormed), :885-962 (router),
//   :965-1103 (multi-head forward)
#include "common.h"

namespace bs {

// bilinear, align_corners=True source coordinates
struct Lerp2 {
    int y0, y1, x0, x1;
    float ly, lx, hy, hx;
};
__device__ __forceinline__ Lerp2 lerp_ac(int oy, int ox, int Hin, int Win, float sy, float sx) {
    Lerp2 l;
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    l.y0 = (int)fy;
    l.x0 = (int)fx;
    l.y0 = l.y0 > Hin - 1 ? Hin - 1 : l.y0;
    l.x0 = l.x0 > Win - 1 ? Win - 1 : l.x0;
    l.y1 = l.y0 + (l.y0 < Hin - 1 ? 1 : 0);
    l.x1 = l.x0 + (l.x0 < Win - 1 ? 1 : 0);
    l.ly = fy - (float)l.y0;
    l.lx = fx - (float)l.x0;
    l.hy = 1.0f - l.ly;
    l.hx = 1.0f - l.lx;
    return l;
}

// ---------------------------------------------------------------------------------------------
// attractor step.  bins are NHWC fp32 with G groups (heads) of nb bins; A has G groups of na
// attractors.  thread = (pixel, group, 4 consecutive bins).  With `route`, only the group that an
// image is routed to is computed (the other group's output is left untouched).
// ---------------------------------------------------------------------------------------------
// NQ: bin quads per thread (2 where n_bins % 8 == 0: twice the bytes in flight per thread, half the threads -- the kernel is bound by the
// latency of its few loads, not by arithmetic or bytes)
template <int NQ>
__global__ __launch_bounds__(256) void attractor_kernel(const float* A, const float* bins_prev, float* bins_out, const int32_t* route,
                                                         int B, int Hp, int Wp, int H, int W, int G, int nb, int na, float sy, float sx) {
    // grid.y = (image, output row); grid.x covers (column, group, bin quad group) of that row: no 64-bit div/mod per thread
    // with `route` the grid covers the routed group only (no idle lanes for the head that is not computed)
    const int q4 = nb / (4 * NQ);
    const int GG = route ? 1 : G;
    const unsigned per_row = (unsigned)W * GG * q4;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per_row) return;
    const int b = blockIdx.y / H, oy = blockIdx.y - b * H;
    const int q = idx % q4;
    const int g = route ? route[b] : (int)((idx / q4) % G);
    const int ox = idx / (q4 * GG);
    const int64_t pix = ((int64_t)b * H + oy) * W + ox;
    const Lerp2 l = lerp_ac(oy, ox, Hp, Wp, sy, sx);
    const int CB = G * nb, CA = G * na;
    const float* pb = bins_prev + (int64_t)b * Hp * Wp * CB + g * nb + q * 4 * NQ;
    f32x4 p00[NQ], p01[NQ], p10[NQ], p11[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        p00[u] = *reinterpret_cast<const f32x4*>(pb + ((int64_t)l.y0 * Wp + l.x0) * CB + 4 * u);
        p01[u] = *reinterpret_cast<const f32x4*>(pb + ((int64_t)l.y0 * Wp + l.x1) * CB + 4 * u);
        p10[u] = *reinterpret_cast<const f32x4*>(pb + ((int64_t)l.y1 * Wp + l.x0) * CB + 4 * u);
        p11[u] = *reinterpret_cast<const f32x4*>(pb + ((int64_t)l.y1 * Wp + l.x1) * CB + 4 * u);
    }
    float c[4 * NQ], dsum[4 * NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            c[4 * u + e] = l.hy * (l.hx * p00[u][e] + l.lx * p01[u][e]) + l.ly * (l.hx * p10[u][e] + l.lx * p11[u][e]);
            dsum[4 * u + e] = 0.f;
        }
    const float* pa = A + pix * CA + g * na;
    for (int a4 = 0; a4 < na; a4 += 4) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(pa + a4);
        // inv_attractor defaults alpha=300, gamma=2: dx / (1 + 300 dx^2) with v_rcp (1 ulp) instead of the IEEE division sequence; two bins
        // per instruction (v_pk_*_f32), the products fused into their adds -- 6 issue slots per term instead of 9
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int h = 0; h < 2 * NQ; ++h) {
                const f32x2_ dx = f32x2_{av[k], av[k]} - f32x2_{c[2 * h], c[2 * h + 1]};
                const f32x2_ den = __builtin_elementwise_fma(dx * f32x2_{300.0f, 300.0f}, dx, f32x2_{1.0f, 1.0f});
                const f32x2_ r = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
                const f32x2_ d2 = __builtin_elementwise_fma(dx, r, f32x2_{dsum[2 * h], dsum[2 * h + 1]});
                dsum[2 * h] = d2[0];
                dsum[2 * h + 1] = d2[1];
            }
    }
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = c[4 * u + e] + dsum[4 * u + e] / (float)na;
        *reinterpret_cast<f32x4*>(bins_out + pix * CB + g * nb + q * 4 * NQ + 4 * u) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// conditional log-binomial + expectation.  thread = one output pixel.
//   hidden = gelu( interp(Eh)[40] + W0_last[40x32] . last[32] )     (Eh already holds W0_emb.emb + b0)
//   pt = softplus(W2[4x40] . hidden + b2); p = (pt0+eps)/(pt0+pt1+2eps); T = (max-min)*t + min
//   y_k = logC(n-1,k) + k log p + (n-1-k) log(1-p); depth = sum_k softmax(y/T)_k * interp(bins)_k
// ---------------------------------------------------------------------------------------------
constexpr int LB_IN = 32, LB_BINS = 64;      // hidden width LB_HID is a template parameter: 40 (NK head) or 80 (single head)
constexpr int LB_T = 16;                 // output tile edge (256 threads = 16 x 16 pixels)
constexpr int LB_MAXSRC = 12;            // most low-res rows / columns a 16-pixel span may touch (>= 1.7x upsampling); the launch sizes LDS for the actual ratio

// everything the wave has in flight on the LDS / scalar-memory counter has arrived (a workgroup-scope fence on the LDS address space is
// `s_waitcnt lgkmcnt(0)`, dropped by the compiler where nothing is outstanding), and the scheduler moves nothing across.  A fence and not
// __builtin_amdgcn_s_waitcnt or inline assembly: behind either of those the compiler no longer treats the MLP weights as unmodified and
// reads them with vector loads (92 global_load per kernel instead of s_load: measured in the ISA).
#define LB_LGKM_FENCE()                              \
    do {                                             \
        __builtin_amdgcn_sched_barrier(0);           \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local"); \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)

#ifdef BS_DIAG
// diagnostics build only: per-pixel intermediates (pt[0..3], sum of the inputs, max logit, den, num) for tools/probes/rerun_determinism.py
__device__ float* lb_dbg = nullptr;
#endif

template <typename T, int LSPLIT, int LB_HID>
__global__ __launch_bounds__(256) void logbinom_kernel(const T* last, const float* Eh, const float* bins, const float* w0_last,
                                                        const float* w2, const float* b2, const float* rel_w, const int32_t* route,
                                                        float* depth, int B, int H, int W, int He, int We, float sy, float sx,
                                                        float min_temp, float max_temp, int ncell_max, int interleaved) {
    // LDS: the block's low-res patch of bin centres [rows][cols][64] and of Eh [rows][cols][40] for the routed head only,
    // loaded once (coalesced) instead of 4 x (256 + 160) bytes per output pixel; plus the small MLP weights.
    // dynamic LDS sized by the launcher for the window the upsampling ratio really needs (2x: 10 x 10 cells = 41.6 KB, 3 blocks/CU)
    extern __shared__ __attribute__((aligned(16))) float lb_smem[];
    float* s_lb = lb_smem;                                   // [64]
    float* s_bins = lb_smem + LB_BINS;                       // [cells][64]
    float* s_eh = s_bins + ncell_max * LB_BINS;              // [cells][LB_HID]
    const int b = blockIdx.z;
    const int g = route[b];
    const int ty0 = blockIdx.y * LB_T, tx0 = blockIdx.x * LB_T;
    // low-res window covered by this tile (align_corners=True: src = scale * dst)
    const int ty1 = min(ty0 + LB_T - 1, H - 1), tx1 = min(tx0 + LB_T - 1, W - 1);
    const int sy0 = min((int)(sy * (float)ty0), He - 1), sx0 = min((int)(sx * (float)tx0), We - 1);
    const int sy1 = min((int)(sy * (float)ty1) + 1, He - 1), sx1 = min((int)(sx * (float)tx1) + 1, We - 1);
    const int nr = sy1 - sy0 + 1, nc = sx1 - sx0 + 1;        // <= LB_MAXSRC (checked by the launcher's scale test)
    for (int i = threadIdx.x; i < nr * nc * (LB_BINS / 4); i += 256) {
        const int k4 = i % (LB_BINS / 4), cell = i / (LB_BINS / 4);
        const int rr = cell / nc, cc = cell - rr * nc;
        const float* src = bins + (((int64_t)b * He + sy0 + rr) * We + sx0 + cc) * (2 * LB_BINS) + g * LB_BINS + k4 * 4;
        *reinterpret_cast<f32x4*>(s_bins + cell * LB_BINS + k4 * 4) = *reinterpret_cast<const f32x4*>(src);
    }
    for (int i = threadIdx.x; i < nr * nc * (LB_HID / 4); i += 256) {
        const int k4 = i % (LB_HID / 4), cell = i / (LB_HID / 4);
        const int rr = cell / nc, cc = cell - rr * nc;
        const float* src = Eh + (((int64_t)b * He + sy0 + rr) * We + sx0 + cc) * (2 * LB_HID) + g * LB_HID + k4 * 4;
        *reinterpret_cast<f32x4*>(s_eh + cell * LB_HID + k4 * 4) = *reinterpret_cast<const f32x4*>(src);
    }
    if (threadIdx.x < LB_BINS) {
        // log_binom(n = 63, k) with the reference's eps placement (modeling_zoedepth.py:376-381)
        const float e = 1e-7f;
        const float n = (float)(LB_BINS - 1) + e, k = (float)threadIdx.x + e;
        s_lb[threadIdx.x] = n * logf(n) - k * logf(k) - (n - k) * logf(n - k + e);
    }
    __syncthreads();
    const int oy = ty0 + (threadIdx.x >> 4), ox = tx0 + (threadIdx.x & 15);
    if (oy >= H || ox >= W) return;
    const int64_t gid = ((int64_t)b * H + oy) * W + ox;
    const Lerp2 l = lerp_ac(oy, ox, He, We, sy, sx);
    const int c00 = (l.y0 - sy0) * nc + (l.x0 - sx0), c01 = (l.y0 - sy0) * nc + (l.x1 - sx0);
    const int c10 = (l.y1 - sy0) * nc + (l.x0 - sx0), c11 = (l.y1 - sy0) * nc + (l.x1 - sx0);

    // the 32 `last` features of this pixel
    float xin[LB_IN];
#ifdef BS_DIAG
    if ((interleaved >> 4) & 16) {
#pragma unroll
        for (int c = 0; c < LB_IN; ++c) xin[c] = 0.0f;
    } else
#endif
    {
        typedef typename T16<T>::v8 v8;
        const T* lp = last + gid * LB_IN * (LSPLIT ? 2 : 1);     // LSPLIT: (hi | lo) pairs, 32 + 32 values per pixel
#pragma unroll
        for (int v = 0; v < LB_IN / 8; ++v) {
            const v8 t = *reinterpret_cast<const v8*>(lp + v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) xin[v * 8 + e] = (float)t[e];
            if (LSPLIT == 1) {
                const v8 tl = *reinterpret_cast<const v8*>(lp + LB_IN + v * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) xin[v * 8 + e] += (float)tl[e];
            } else if (LSPLIT == 2) {     // (hi16 | hi8 | lo8): the lo8 plane starts 1.5 * 32 elements into the pixel
                const char* l8 = reinterpret_cast<const char*>(lp + LB_IN + LB_IN / 2) + v * 8;
                float l0[4], l1[4];
                f8_unpack4(*reinterpret_cast<const int*>(l8), l0);
                f8_unpack4(*reinterpret_cast<const int*>(l8 + 4), l1);
                const float sc = __builtin_ldexpf(1.0f, -F8_ACT_LO_EXP);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xin[v * 8 + e] += l0[e] * sc;
                    xin[v * 8 + 4 + e] += l1[e] * sc;
                }
            }
        }
    }
    // The MLP weights are the same for every thread of the block (g is block-uniform): they are read straight from global
    // memory with uniform addresses, i.e. through the scalar cache into SGPRs (s_load + v_fmac with a scalar operand), not
    // through 1280 LDS broadcast reads per pixel.
    const float* __restrict__ gw0 = w0_last + g * LB_HID * LB_IN;
    const float* __restrict__ gw2 = w2 + g * 4 * LB_HID;
    float pt[4] = {b2[g * 4 + 0], b2[g * 4 + 1], b2[g * 4 + 2], b2[g * 4 + 3]};
    // single-head models feed the relative depth relu(conv3(last)) as a 33rd input (HF modeling_zoedepth.py:1186-1191, :367-371):
    // rel_w[g] = [W0 column of that input (LB_HID) | conv3 weight (32) | conv3 bias]
    const float* __restrict__ grel = rel_w ? rel_w + g * (LB_HID + LB_IN + 1) : nullptr;
    float rd = 0.0f;
    if (grel) {
        rd = grel[LB_HID + LB_IN];
#pragma unroll
        for (int c = 0; c < LB_IN; ++c) rd += grel[LB_HID + c] * xin[c];
        rd = fmaxf(rd, 0.0f);
    }
    // two hidden units per pass: the 32-term dot products run on v_pk_fma_f32 (both halves read the same input, the weights of units h and
    // h + 1 sit in scalar registers) -- the build has -ffp-contract=off, so without the explicit fused form every term was a multiply
    // AND an add (2 560 issue slots per pixel for this loop; now 640)
    //
    // Round 6: the interpolated Eh of EIGHT units is read from LDS in one go (eight ds_read_b128), between two `s_waitcnt lgkmcnt(0)` fences.
    // Rounds 2-5 read the four corners of two units inside the loop with per-lane 8-byte (in other builds 4-byte) gathers.  Those forms compute with
    // wrong LDS read results in the LAST SIXTEEN LANES of a wave when a kernel of ANOTHER stream (or process) shares the CU and issues MFMA --
    // bs_rank1_bias is the plan's one MFMA kernel small enough to do so: 11-14 % of the launches beside it, 0 with its MFMA removed, 0 alone
    // (profiles/r06_reproducibility.txt (5)-(8); tools/probes/gather_beside_stream.py).  Fencing the small reads off from the scalar loads does not
    // help; reading 16 bytes per lane does: this form has not differed in 57 600 reruns with the other forms failing in the same calls, and it is
    // 6 % faster.  Why the hardware does this is not known.  BS_LOGBINOM_INTERLEAVED=1 / 4 / 5 keep the small-read forms in the diagnostics build.
    static_assert(LB_HID % 8 == 0, "hidden width");
#ifdef BS_DIAG
    // diagnostics build: BS_LOGBINOM_SKIP (bits 4.. of the switch): 1 = no dot products, 2 = no GELU, 4 = no softmax phase (the sum of the MLP's outputs is
    // written instead), 8 = no output-layer weights (no scalar load in the loop), 16 = the pixel's inputs not loaded -- wrong results; which part of this kernel has to be there for another stream's MFMA kernel to disturb its small LDS reads
    const int skip = interleaved >> 4;
    interleaved &= 15;
    float dg_interp = 0.f, dg_pre = 0.f, dg_act = 0.f;       // sums over the hidden units of: interpolated Eh, pre-activation, activation
#endif
    for (int h0 = 0; h0 < LB_HID; h0 += 8) {
        float ehv[8];
#ifdef BS_DIAG
        if (interleaved == 5) {       // the block position of the shipped form, but as 32 four-byte per-lane reads (volatile: not merged) between the fences
            LB_LGKM_FENCE();
            float g00[8], g01[8], g10[8], g11[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                g00[k] = *reinterpret_cast<volatile const float*>(s_eh + c00 * LB_HID + h0 + k);
                g01[k] = *reinterpret_cast<volatile const float*>(s_eh + c01 * LB_HID + h0 + k);
                g10[k] = *reinterpret_cast<volatile const float*>(s_eh + c10 * LB_HID + h0 + k);
                g11[k] = *reinterpret_cast<volatile const float*>(s_eh + c11 * LB_HID + h0 + k);
            }
            LB_LGKM_FENCE();
#pragma unroll
            for (int k = 0; k < 8; ++k) ehv[k] = l.hy * (l.hx * g00[k] + l.lx * g01[k]) + l.ly * (l.hx * g10[k] + l.lx * g11[k]);
        } else if (!interleaved)
#endif
        {
            LB_LGKM_FENCE();
            f32x4 e00[2], e01[2], e10[2], e11[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                e00[q] = *reinterpret_cast<const f32x4*>(s_eh + c00 * LB_HID + h0 + 4 * q);
                e01[q] = *reinterpret_cast<const f32x4*>(s_eh + c01 * LB_HID + h0 + 4 * q);
                e10[q] = *reinterpret_cast<const f32x4*>(s_eh + c10 * LB_HID + h0 + 4 * q);
                e11[q] = *reinterpret_cast<const f32x4*>(s_eh + c11 * LB_HID + h0 + 4 * q);
            }
            LB_LGKM_FENCE();
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)      // same bilinear operation order as torch: hy*(hx*p00 + lx*p01) + ly*(hx*p10 + lx*p11)
                    ehv[4 * q + e] = l.hy * (l.hx * e00[q][e] + l.lx * e01[q][e]) + l.ly * (l.hx * e10[q][e] + l.lx * e11[q][e]);
        }
#pragma unroll
        for (int hh = 0; hh < 8; hh += 2) {
            const int h = h0 + hh;
            // each unit's dot product as an (even inputs, odd inputs) pair: the weight pairs are adjacent in memory -- they arrive as aligned
            // scalar register pairs, no s_mov to pair up weights of two rows -- and so are the input pairs in the vector registers
            f32x2_ sv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                sv[u][0] = ehv[hh + u];
#ifdef BS_DIAG
                if (interleaved == 1)      // the rounds 2-5 form: the compiler places these reads among the scalar loads of the units' weights
                    sv[u][0] = l.hy * (l.hx * s_eh[c00 * LB_HID + h + u] + l.lx * s_eh[c01 * LB_HID + h + u]) +
                               l.ly * (l.hx * s_eh[c10 * LB_HID + h + u] + l.lx * s_eh[c11 * LB_HID + h + u]);
                else if (interleaved == 4) {      // the same reads, fenced off from the scalar loads on both sides
                    LB_LGKM_FENCE();
                    const float g00 = s_eh[c00 * LB_HID + h + u], g01 = s_eh[c01 * LB_HID + h + u];
                    const float g10 = s_eh[c10 * LB_HID + h + u], g11 = s_eh[c11 * LB_HID + h + u];
                    LB_LGKM_FENCE();
                    sv[u][0] = l.hy * (l.hx * g00 + l.lx * g01) + l.ly * (l.hx * g10 + l.lx * g11);
                }
                dg_interp += sv[u][0];
#endif
                sv[u][1] = 0.0f;
            }
#ifdef BS_DIAG
            if (!(skip & 1))
#endif
#pragma unroll
            for (int c = 0; c < LB_IN; c += 2) {
                const f32x2_ x2 = {xin[c], xin[c + 1]};
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    sv[u] = __builtin_elementwise_fma(*reinterpret_cast<const f32x2_*>(gw0 + (h + u) * LB_IN + c), x2, sv[u]);
            }
            f32x2_ a2 = {sv[0][0] + sv[0][1], sv[1][0] + sv[1][1]};
            if (grel) a2 = __builtin_elementwise_fma(f32x2_{grel[h], grel[h + 1]}, f32x2_{rd, rd}, a2);
#ifdef BS_DIAG
            const f32x2_ ga = (skip & 2) ? a2 : gelu_erf_as2(a2);
#else
            const f32x2_ ga = gelu_erf_as2(a2);
#endif
#ifdef BS_DIAG
            dg_pre += a2[0] + a2[1];
            dg_act += ga[0] + ga[1];
#endif
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int o = 0; o < 4; ++o) {
#ifdef BS_DIAG
                    if (skip & 8) pt[o] += ga[u];
                    else
#endif
                    pt[o] = fmaf(gw2[o * LB_HID + h + u], ga[u], pt[o]);
                }
        }
    }
    LB_LGKM_FENCE();       // (the softmax loop below reads the bin centres from LDS: nothing scalar in flight there either)
#ifdef BS_DIAG
    if (skip & 4) {
        depth[gid] = (pt[0] + pt[1]) + (pt[2] + pt[3]);
        return;
    }
    const float dg_pt[4] = {pt[0], pt[1], pt[2], pt[3]};
    float dg_x = 0.f;
#pragma unroll
    for (int c = 0; c < LB_IN; ++c) dg_x += xin[c];
#endif
#pragma unroll
    for (int o = 0; o < 4; ++o) pt[o] = softplus20(pt[o]);
    const float eps = 1e-4f;
    float p = (pt[0] + eps) / ((pt[0] + eps) + (pt[1] + eps));
    float t = (pt[2] + eps) / ((pt[2] + eps) + (pt[3] + eps));
    t = (max_temp - min_temp) * t + min_temp;
    float omp = 1.0f - p;
    omp = fminf(fmaxf(omp, eps), 1.0f);
    p = fminf(fmaxf(p, eps), 1.0f);
    const float lp_ = logf(p), lomp = logf(omp);
    // y_k / T = (lb_k + k log p + (63 - k) log(1 - p)) / T.  The 64 logits are not kept in registers (that cost 64 VGPRs and
    // halved the occupancy): they are three FMAs each and are evaluated twice, once for the max and once for the exponentials;
    // the division by T is one reciprocal (T > 0), applied after the max.
    const float rt = 1.0f / t;
    // two bins per instruction (v_pk_fma_f32 / v_pk_mul_f32): (lb_k + k log p) + (63 - k) log(1 - p), each product fused into its add
    const f32x2_ lp2 = {lp_, lp_}, lo2 = {lomp, lomp};
    auto logit2 = [&](int k) {      // bins k, k + 1 (k even)
        const f32x2_ kk = {(float)k, (float)(k + 1)}, nk = {(float)(LB_BINS - 1 - k), (float)(LB_BINS - 2 - k)};
        const f32x2_ lb = *reinterpret_cast<const f32x2_*>(s_lb + k);
        return __builtin_elementwise_fma(nk, lo2, __builtin_elementwise_fma(kk, lp2, lb));
    };
    float vmx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < LB_BINS; k += 2) {
        const f32x2_ v = logit2(k);
        vmx = fmaxf(fmaxf(vmx, v[0]), v[1]);
    }
    const f32x2_ vm2 = {vmx, vmx}, rt2 = {rt, rt};
    const f32x2_ hx2 = {l.hx, l.hx}, lx2 = {l.lx, l.lx}, hy2 = {l.hy, l.hy}, ly2 = {l.ly, l.ly};
    f32x2_ den2 = {0.f, 0.f}, num2 = {0.f, 0.f};
#pragma unroll
    for (int k4 = 0; k4 < LB_BINS / 4; ++k4) {
        const f32x4 b00 = *reinterpret_cast<const f32x4*>(s_bins + c00 * LB_BINS + k4 * 4);
        const f32x4 b01 = *reinterpret_cast<const f32x4*>(s_bins + c01 * LB_BINS + k4 * 4);
        const f32x4 b10 = *reinterpret_cast<const f32x4*>(s_bins + c10 * LB_BINS + k4 * 4);
        const f32x4 b11 = *reinterpret_cast<const f32x4*>(s_bins + c11 * LB_BINS + k4 * 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2_ a = (logit2(k4 * 4 + 2 * h) - vm2) * rt2;
            const f32x2_ w = {__expf(a[0]), __expf(a[1])};
            const f32x2_ p00 = {b00[2 * h], b00[2 * h + 1]}, p01 = {b01[2 * h], b01[2 * h + 1]};
            const f32x2_ p10 = {b10[2 * h], b10[2 * h + 1]}, p11 = {b11[2 * h], b11[2 * h + 1]};
            // torch's bilinear order hy*(hx*p00 + lx*p01) + ly*(hx*p10 + lx*p11), products fused into the adds
            const f32x2_ top = __builtin_elementwise_fma(lx2, p01, hx2 * p00), bot = __builtin_elementwise_fma(lx2, p11, hx2 * p10);
            const f32x2_ c = __builtin_elementwise_fma(ly2, bot, hy2 * top);
            den2 += w;
            num2 = __builtin_elementwise_fma(w, c, num2);
        }
    }
    const float den = den2[0] + den2[1], num = num2[0] + num2[1];
    depth[gid] = num / den;
#ifdef BS_DIAG
    if (lb_dbg) {
        float* d = lb_dbg + gid * 8;
        d[0] = dg_x; d[1] = dg_interp; d[2] = dg_pre; d[3] = dg_act;
        d[4] = dg_pt[0]; d[5] = dg_pt[1]; d[6] = dg_pt[2]; d[7] = dg_pt[3];
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// router pieces: small multi-head attention (S <= 256, head_dim 32, no mask), argmax
// qkv fp32 [B*S, 3*D] (q | k | v), out 16-bit [B*S, D]
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void small_attention_kernel(const float* qkv, T* out, int S, int nheads, float scale) {
    constexpr int HD = 32;
    const int b = blockIdx.x / nheads, h = blockIdx.x % nheads;
    const int D = nheads * HD;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // K and V rows of this (image, head) in LDS, 32 floats = 128 bytes each: every thread of the block reads the SAME row at the same
    // time (a broadcast: no bank conflicts whatever the stride), so the rows are left unpadded and go out as eight ds_read_b128
    // instead of 32 ds_read_b32 -- the kernel was bound by LDS instruction issue, and by a multiply AND an add per term (the build
    // has -ffp-contract=off): both loops now run fused multiply-adds on whole float4s.
    float* ks = reinterpret_cast<float*>(smem_raw);  // [S][HD]
    float* vs = ks + S * HD;
    for (int i = threadIdx.x; i < S * HD; i += blockDim.x) {
        const int t = i / HD, d = i % HD;
        const float* row = qkv + ((int64_t)b * S + t) * 3 * D + h * HD + d;
        ks[i] = row[D];
        vs[i] = row[2 * D];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < S; t += blockDim.x) {
        f32x4 q[HD / 4], o[HD / 4];
        const float* qr = qkv + ((int64_t)b * S + t) * 3 * D + h * HD;
#pragma unroll
        for (int d = 0; d < HD / 4; ++d) {
            q[d] = *reinterpret_cast<const f32x4*>(qr + 4 * d);
            o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        auto score = [&](int j) {
            f32x4 acc = q[0] * *reinterpret_cast<const f32x4*>(ks + j * HD);
#pragma unroll
            for (int d = 1; d < HD / 4; ++d) acc = __builtin_elementwise_fma(q[d], *reinterpret_cast<const f32x4*>(ks + j * HD + 4 * d), acc);
            return ((acc[0] + acc[1]) + (acc[2] + acc[3])) * scale;
        };
        float mx = -3.0e38f;
        for (int j = 0; j < S; ++j) mx = fmaxf(mx, score(j));
        float den = 0.f;
        for (int j = 0; j < S; ++j) {
            const float w = expf(score(j) - mx);
            den += w;
            const f32x4 w4 = {w, w, w, w};
#pragma unroll
            for (int d = 0; d < HD / 4; ++d) o[d] = __builtin_elementwise_fma(w4, *reinterpret_cast<const f32x4*>(vs + j * HD + 4 * d), o[d]);
        }
        T* orow = out + ((int64_t)b * S + t) * D + h * HD;
        const float inv = 1.0f / den;
#pragma unroll
        for (int d = 0; d < HD / 4; ++d)
#pragma unroll
            for (int e = 0; e < 4; ++e) orow[4 * d + e] = T16<T>::from_f32(o[d][e] * inv);
    }
}

__global__ void route_argmax_kernel(const float* logits, int ld, int32_t* route, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    // torch.argmax returns the first maximal index
    route[b] = logits[(int64_t)b * ld + 1] > logits[(int64_t)b * ld] ? 1 : 0;
}

}  // namespace bs

using namespace bs;
#define BS_ENTRY(name) \
    if (!initialized()) { set_error(name ": call bs_init first"); return BS_ERR_NOT_INIT; }

extern "C" int bs_attractor_step(const float* A, const float* bins_prev, float* bins_out, const int32_t* route, int32_t B, int32_t Hp,
                                 int32_t Wp, int32_t H, int32_t W, int32_t groups, int32_t n_bins, int32_t n_attr, void* stream) {
    BS_ENTRY("bs_attractor_step");
    BS_REQUIRE(A && bins_prev && bins_out && B >= 0 && Hp > 0 && Wp > 0 && H > 0 && W > 0 && groups > 0 && n_bins % 4 == 0 && n_attr > 0,
               "bs_attractor_step: bad argument");
    if (B == 0) return BS_OK;
    const float sy = H > 1 ? (float)(Hp - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(Wp - 1) / (float)(W - 1) : 0.f;
    BS_REQUIRE(n_attr % 4 == 0, "bs_attractor_step: n_attr must be a multiple of 4");
    static const bool one_quad = diag_env("BS_ATTRACTOR_NQ1") != nullptr;        // diagnostics (A / B)
    if (n_bins % 8 == 0 && !one_quad) {      // (four quads per thread: 1.65 ms against 1.59 for the bench's four levels; one: 1.72)
        const unsigned per_row = (unsigned)W * (route ? 1 : groups) * (n_bins / 8);
        hipLaunchKernelGGL(attractor_kernel<2>, dim3(cdiv((int)per_row, 256), B * H), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), A,
                           bins_prev, bins_out, route, B, Hp, Wp, H, W, groups, n_bins, n_attr, sy, sx);
    } else {
        const unsigned per_row = (unsigned)W * (route ? 1 : groups) * (n_bins / 4);
        hipLaunchKernelGGL(attractor_kernel<1>, dim3(cdiv((int)per_row, 256), B * H), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), A,
                           bins_prev, bins_out, route, B, Hp, Wp, H, W, groups, n_bins, n_attr, sy, sx);
    }
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_logbinom_depth_ex(const void* last, const float* Eh, const float* bins, const float* w0_last, const float* w2,
                                    const float* b2, const float* rel_w, int32_t hid, const int32_t* route, float* depth, int32_t B,
                                    int32_t H, int32_t W, int32_t He, int32_t We, float min_temp, float max_temp, int32_t dtype,
                                    void* stream) {
    BS_ENTRY("bs_logbinom_depth");
    BS_REQUIRE(last && Eh && bins && w0_last && w2 && b2 && route && depth && B >= 0 && H > 0 && W > 0 && He > 0 && We > 0,
               "bs_logbinom_depth: bad argument");
    BS_REQUIRE((dtype & 15) == BS_F16 || (dtype & 15) == BS_BF16, "bs_logbinom_depth: dtype");
    BS_REQUIRE(hid == 40 || hid == 80, "bs_logbinom_depth: hidden width %d (built: 40, 80)", hid);
    if (B == 0) return BS_OK;
    const float sy = H > 1 ? (float)(He - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(We - 1) / (float)(W - 1) : 0.f;
    // a 16-pixel output span must fit the LDS patch: span * scale + 2 <= LB_MAXSRC
    BS_REQUIRE(sy * (LB_T - 1) + 3.0f <= (float)LB_MAXSRC && sx * (LB_T - 1) + 3.0f <= (float)LB_MAXSRC,
               "bs_logbinom_depth: the bins map must be upsampled by at least ~1.7x (He,We=%d,%d -> H,W=%d,%d)", He, We, H, W);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(cdiv(W, LB_T), cdiv(H, LB_T), B);
    // rows / columns of the low-res window of a 16-pixel span: floor(scale * 15 + frac) + 2 <= floor(scale * 15) + 3
    const int nr_max = (int)(sy * (LB_T - 1)) + 3, nc_max = (int)(sx * (LB_T - 1)) + 3;
    const int ncell_max = nr_max * nc_max;
    const size_t lds = sizeof(float) * (size_t)(LB_BINS + ncell_max * (LB_BINS + hid));
    static const int interleaved = (diag_env("BS_LOGBINOM_INTERLEAVED") ? atoi(diag_env("BS_LOGBINOM_INTERLEAVED")) : 0) +
                                   16 * (diag_env("BS_LOGBINOM_SKIP") ? atoi(diag_env("BS_LOGBINOM_SKIP")) : 0);      // diagnostics build only: 1 = the rounds 2-5 form of the hidden layer, 4 = that form's reads between fences, 5 = the shipped position with four-byte reads
    const int lsplit = (dtype & 32) ? 2 : ((dtype & 16) ? 1 : 0);   // bit 4: `last` holds (hi | lo) 16-bit pairs; bit 5: (hi16 | hi8 | lo8)
    dtype &= 15;
#define BS_LB_LAUNCH(TT, LS, HD)                                                                                                 \
    do {                                                                                                                         \
        BS_MAX_DYNAMIC_LDS(reinterpret_cast<const void*>(&logbinom_kernel<TT, LS, HD>), 96 * 1024); \
        hipLaunchKernelGGL((logbinom_kernel<TT, LS, HD>), grid, dim3(256), lds, st, (const TT*)last, Eh, bins, w0_last, w2, b2,   \
                           rel_w, route, depth, B, H, W, He, We, sy, sx, min_temp, max_temp, ncell_max, interleaved);            \
    } while (0)
#define BS_LB_HID(TT, LS)             \
    do {                              \
        if (hid == 40) BS_LB_LAUNCH(TT, LS, 40); \
        else BS_LB_LAUNCH(TT, LS, 80);           \
    } while (0)
    if (dtype == BS_F16) {
        if (lsplit == 2) BS_LB_HID(f16, 2);
        else if (lsplit == 1) BS_LB_HID(f16, 1);
        else BS_LB_HID(f16, 0);
    } else {
        if (lsplit == 2) BS_LB_HID(bf16, 2);
        else if (lsplit == 1) BS_LB_HID(bf16, 1);
        else BS_LB_HID(bf16, 0);
    }
#undef BS_LB_HID
#undef BS_LB_LAUNCH
    BS_CHECK_LAUNCH();
    return BS_OK;
}

#ifdef BS_DIAG
extern "C" int bs_diag_logbinom_buffer(float* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(bs::lb_dbg), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int bs_logbinom_depth(const void* last, const float* Eh, const float* bins, const float* w0_last, const float* w2,
                                 const float* b2, const int32_t* route, float* depth, int32_t B, int32_t H, int32_t W, int32_t He,
                                 int32_t We, float min_temp, float max_temp, int32_t dtype, void* stream) {
    return bs_logbinom_depth_ex(last, Eh, bins, w0_last, w2, b2, nullptr, 40, route, depth, B, H, W, He, We, min_temp, max_temp, dtype, stream);
}

extern "C" int bs_small_attention(const float* qkv, void* out, int32_t B, int32_t S, int32_t nheads, int32_t dtype, void* stream) {
    BS_ENTRY("bs_small_attention");
    BS_REQUIRE(qkv && out && B >= 0 && S > 0 && S <= 512 && nheads > 0, "bs_small_attention: bad argument");
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "bs_small_attention: dtype");
    if (B == 0) return BS_OK;
    const size_t smem = (size_t)2 * S * 33 * sizeof(float);
    BS_REQUIRE(smem <= 64 * 1024, "bs_small_attention: S too large for LDS");
    const float scale = 1.0f / sqrtf(32.0f);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == BS_F16)
        hipLaunchKernelGGL(small_attention_kernel<f16>, dim3(B * nheads), dim3(256), smem, st, qkv, (f16*)out, S, nheads, scale);
    else
        hipLaunchKernelGGL(small_attention_kernel<bf16>, dim3(B * nheads), dim3(256), smem, st, qkv, (bf16*)out, S, nheads, scale);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_route_argmax(const float* logits, int32_t ld, int32_t* route, int32_t B, void* stream) {
    BS_ENTRY("bs_route_argmax");
    BS_REQUIRE(logits && route && B >= 0 && ld >= 2, "bs_route_argmax: bad argument");
    if (B == 0) return BS_OK;
    hipLaunchKernelGGL(route_argmax_kernel, dim3(cdiv(B, 64)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), logits, ld, route, B);
    BS_CHECK_LAUNCH();
    return BS_OK;
}
