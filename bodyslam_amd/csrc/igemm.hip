// Implicit-GEMM on MFMA for gfx950: out[m, n] = epilogue(sum_k A(m, k) * W[n, k]).
//
// One kernel family serves every Linear / Conv2d / ConvTranspose2d(k == stride) on the hot path
// (include/bodyslam_hip.h: bs_gemm lists the reference call sites).  Design:
//   * NHWC activations, weights [N][KH][KW][Cin]: a BK = 64 slice of K is ONE filter tap and 64
//     contiguous channels, i.e. one 128-byte line per output pixel.  The A tile is therefore a row
//     gather: each lane of a `global_load_lds_dwordx4` supplies the address of 16 bytes of its pixel
//     (or of a zero page when the tap falls into the padding) and the data lands in LDS without
//     touching VGPRs.  No im2col buffer exists anywhere.
//   * LDS image per operand: [rows][64] 16-bit, 128-byte rows, 16-byte chunk c of row r stored at
//     chunk position c ^ (r & 7).  The DMA destination is lane-linear, so the XOR is applied to the
//     SOURCE chunk each lane fetches and again on the ds_read_b128 address (conflict-free for the
//     16x16x32 operand pattern: 16 distinct rows x one chunk per lane group).
//   * v_mfma_f32_16x16x32_{f16,bf16}; operands swapped (W fragment as "A", activation fragment as
//     "B") so a lane ends with 4 consecutive n for one m: 8/16-byte epilogue stores.
//   * double-buffered LDS, one barrier per K tile: the DMA of tile t+1 is in flight while tile t
//     is multiplied.
//   * 1-D grid with the bijective XCD remap: the N-tiles of one M-tile (which share the gathered
//     activations) are consecutive work ids and land on one XCD's L2.
// Epilogue fuses bias (optionally per row group), ReLU/GELU/softplus, per-channel scale
// (BEiT layer-scale), residual add (fp32 or 16-bit), and three store layouts (plain with row
// regrouping, ConvTranspose pixel shuffle, Q/K/V^T head scatter).
#include "common.h"

namespace bs {

struct IgemmParams {
    const void* A;
    const void* W;
    const void* zero;
    long long a_bytes;   // extent of A in bytes (bounds of the buffer descriptor)
    int M, N, K, lda;
    int Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad_h, pad_w;
    int tiles_per_tap;
    int relu_a;
    const float* bias;
    int bias_group_rows;
    int act;
    const float* scale;
    const void* res;
    const void* res2;
    int res_dtype, ldr;
    void* out;
    void* out2;
    void* out3;
    int out_dtype, ldo, out_mode;
    int out_group_rows, out_group_stride, out_row_offset;
    int shuffle_s, shuffle_cout;
    int qkv_hidden, qkv_tokens, qkv_sp;
    float q_scale;
    int ntm, ntn;
};

template <typename T>
__device__ __forceinline__ typename T16<T>::v8 relu8(typename T16<T>::v8 x) {
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 b = __builtin_bit_cast(s16x8, x);
    s16x8 neg = b >> 15;  // 0xFFFF where the sign bit is set
    b = b & ~neg;
    return __builtin_bit_cast(typename T16<T>::v8, b);
}

template <typename T>
__device__ __forceinline__ void store4(void* base, int64_t off, int out_dtype, const float (&y)[4]) {
    if (out_dtype == BS_F32) {
        f32x4 v = {y[0], y[1], y[2], y[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off) = v;
    } else {
        typename T16<T>::v4 v;
        v[0] = T16<T>::from_f32(y[0]);
        v[1] = T16<T>::from_f32(y[1]);
        v[2] = T16<T>::from_f32(y[2]);
        v[3] = T16<T>::from_f32(y[3]);
        *reinterpret_cast<typename T16<T>::v4*>(reinterpret_cast<T*>(base) + off) = v;
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Tile variants.  BK is the K slice per stage (one 64- or 128-byte LDS row per tile row), STAGES the depth
// of the LDS ring: STAGES-1 tiles are in flight (global_load_lds) while one is multiplied.
// MODE: 0 plain GEMM rows, 1 implicit conv, 2 implicit conv with ReLU applied to A on load
template <typename T, int BM, int BN, int WM, int WN, int BK, int STAGES, int MODE>
__global__ __launch_bounds__(WM* WN * 64) void igemm_kernel(const IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-descriptor builtins exist in the device pass only; the host pass needs just the stub
    constexpr bool CONV = MODE != 0;
    constexpr bool RELU_A = MODE == 2;
    constexpr int NT = WM * WN * 64;
    constexpr int ROWB = BK * 2;        // bytes per LDS row
    constexpr int LPR = ROWB / 16;      // lanes (16-byte chunks) per row: 8 (BK 64) or 4 (BK 32)
    constexpr int RPR = NT / LPR;       // rows staged per DMA round
    constexpr int RPW = 64 / LPR;       // rows per wave-instruction (1 KiB)
    constexpr int RA = BM / RPR, RB = BN / RPR;
    static_assert(BK == 64 || BK == 32, "BK");
    static_assert(BM % RPR == 0 && BN % RPR == 0, "tile rows must be a multiple of the DMA round");
    constexpr int GL = RA + RB;         // LDS-DMA instructions per wave per stage
    static_assert(GL * (STAGES - 1) <= 63, "vmcnt range");
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile must be a multiple of 16");
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int LDS_BYTES = STAGES * STAGE;
    constexpr unsigned OOB = 0x80000000u;   // voffset sentinel: beyond every descriptor (num_records < 2^31) -> the DMA writes zeros
    typedef typename T16<T>::v8 v8;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- work id -> tile, XCD-aware (blocks b and b+8 share an XCD; give each XCD a contiguous
    // run of work ids so that the N-tiles of one M-tile hit the same L2)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + loc;
    const int tm = wg / p.ntn, tn = wg - tm * p.ntn;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int srow = tid / LPR;
    // LDS image: 16-byte chunk c of row r sits at chunk position c ^ swz(r); swz(r) = r & 7 (128-byte rows) or
    // (-(r >> 2)) & 3 (64-byte rows): conflict-free ds_read_b128 for the 16x16x32 operand pattern in both cases
    // (the b128 lane groups pair rows {0-3,12-15} at chunk c with rows {4-11} at chunk c^1).
    const int sswz = (BK == 64) ? (srow & 7) : ((0 - (srow >> 2)) & 3);
    const int cs16 = ((tid & (LPR - 1)) ^ sswz) * 16;  // byte offset of the SOURCE chunk this lane fetches

    // ---- buffer descriptors (wave-uniform): A window starting at this tile's first image / row, W whole.
    // Out-of-range lanes of a `buffer_load ... lds` write ZEROS to LDS (probed: tools/probes/lds_dma_oob.hip):
    // that is the convolution's zero padding -- no zero page, no per-lane pointer select, 32-bit offsets only.
    const int m0 = tm * BM;
    long long a_base_el;
    int img0 = 0;
    if (CONV) {
        img0 = m0 / (p.Hout * p.Wout);
        a_base_el = (long long)img0 * p.Hin * p.Win * p.lda;
    } else {
        a_base_el = (long long)m0 * p.lda;
    }
    long long a_left = p.a_bytes - a_base_el * 2;
    a_left = a_left > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : a_left;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(reinterpret_cast<const T*>(p.A)) + a_base_el, 0, (int)a_left, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.W), 0, (int)((long long)p.N * p.K * 2), 0x00020000);

    // ---- per-lane row bookkeeping for the DMA rounds: a 32-bit byte offset and (conv) a tap-validity bitmask
    unsigned a_off[RA];
    unsigned a_mask[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        int m = m0 + j * RPR + srow;
        m = m < p.M ? m : p.M - 1;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            const int iy0 = oy * p.stride - p.pad_h, ix0 = ox * p.stride - p.pad_w;
            // offset of tap (0,0); may be "negative" (wraps): it is only used where the tap is valid, where the sum is in range
            a_off[j] = (unsigned)((((b - img0) * p.Hin + iy0) * p.Win + ix0) * p.lda * 2 + cs16);
            unsigned mk = 0;
            for (int ky = 0; ky < p.KH; ++ky)
                for (int kx = 0; kx < p.KW; ++kx) {
                    const bool ok = (unsigned)(iy0 + ky) < (unsigned)p.Hin && (unsigned)(ix0 + kx) < (unsigned)p.Win;
                    mk |= (ok ? 1u : 0u) << (ky * p.KW + kx);
                }
            a_mask[j] = mk;
        } else {
            a_off[j] = (unsigned)((m - m0) * p.lda * 2 + cs16);
            a_mask[j] = 1u;
        }
    }
    unsigned w_off[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        int n = tn * BN + j * RPR + srow;
        n = n < p.N ? n : p.N - 1;
        w_off[j] = (unsigned)(n * p.K * 2 + cs16);
    }

    // running state of the NEXT tile to stage: tap index / tap byte offset / channel byte offset (conv), k byte offset
    int s_tap = 0, s_kx = 0, s_tapoff = 0, s_c0 = 0, s_k = 0;

    auto stage = [&](int buf) {
        char* sa = smem + buf * STAGE;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            unsigned vo;
            if (CONV) {
                vo = ((a_mask[j] >> s_tap) & 1u) ? a_off[j] + (unsigned)s_tapoff : OOB;
            } else {
                vo = a_off[j];
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(sa + (j * RPR + wave * RPW) * ROWB), 16, vo,
                                                     CONV ? s_c0 : s_k, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < RB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(sb + (j * RPR + wave * RPW) * ROWB), 16, w_off[j],
                                                     s_k, 0, 0);
        s_k += BK * 2;
        if (CONV) {
            s_c0 += BK * 2;
            if (s_c0 >= p.Cin * 2) {
                s_c0 = 0;
                ++s_tap;
                s_tapoff += p.lda * 2;
                if (++s_kx >= p.KW) {
                    s_kx = 0;
                    s_tapoff += (p.Win - p.KW) * p.lda * 2;
                }
            }
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row = 16-row fragment base + (lane & 15)
    const int frow = lane & 15, fq = lane >> 4;
    const int fswz = (BK == 64) ? (lane & 7) : ((0 - ((lane & 15) >> 2)) & 3);
    const int koff0 = ((0 + fq) ^ fswz) << 4, koff1 = (BK == 64) ? (((4 + fq) ^ fswz) << 4) : 0;
    const int a_base = (wm * TM + frow) * ROWB, b_base = (wn * TN + frow) * ROWB;

    const int nt = p.K / BK;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nt) stage(s);
    int cbuf = 0, sbuf = STAGES - 1;   // buffer multiplied this iteration / buffer staged this iteration
    for (int t = 0; t < nt; ++t) {
        // my own DMA for tile t has landed once at most (tiles issued after t) x GL operations are outstanding
        const int younger = nt - 1 - t;
        if (STAGES >= 4 && younger >= 2) wait_vmcnt<(STAGES >= 4 ? 2 : 0) * GL>();
        else if (STAGES >= 3 && younger >= 1) wait_vmcnt<(STAGES >= 3 ? 1 : 0) * GL>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // everyone's tile t has landed; everyone is done reading buffer sbuf (tile t-1)
        asm volatile("" ::: "memory");
        if (t + STAGES - 1 < nt) stage(sbuf);
        const char* sa = smem + cbuf * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            const int ko = kk ? koff1 : koff0;
            v8 af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                af[i] = *reinterpret_cast<const v8*>(sa + a_base + i * 16 * ROWB + ko);
                if (RELU_A) af[i] = relu8<T>(af[i]);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = *reinterpret_cast<const v8*>(sb + b_base + j * 16 * ROWB + ko);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = T16<T>::mfma16(bf[j], af[i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        }
        cbuf = cbuf + 1 == STAGES ? 0 : cbuf + 1;
        sbuf = sbuf + 1 == STAGES ? 0 : sbuf + 1;
    }

    // ---- epilogue.  Each wave stages ITS OWN accumulator sub-tile through a private LDS region (XOR-swizzled
    // float4 slots, no padding) and streams it out row by row: one block barrier in total (the main loop must be
    // done with the LDS), no barrier between passes, the fused epilogue code exists once (a runtime loop), and a
    // wave-instruction stores whole rows (TN*2 or TN*4 contiguous bytes per row).
    // lane holds acc[m = ..+(lane&15)][n = ..+(lane>>4)*4 + 0..3]
    constexpr int NW = WM * WN;
    constexpr int S4 = TN / 4;                                     // float4 slots per staged row
    constexpr int MAXR = LDS_BYTES / (NW * TN * 4);
    constexpr int PASS_R = MAXR >= TM ? TM : (MAXR >= TM / 2 ? TM / 2 : (MAXR >= TM / 4 ? TM / 4 : TM / 8));
    static_assert(PASS_R >= 16 && PASS_R % 16 == 0 && NW * PASS_R * TN * 4 <= LDS_BYTES, "epilogue staging does not fit the main-loop LDS");
    static_assert(64 % S4 == 0, "rows per sweep");
    constexpr int PASSES = TM / PASS_R;
    constexpr int RPS = 64 / S4;                                   // rows per 64-lane sweep
    float* sc = reinterpret_cast<float*>(smem) + wave * (PASS_R * TN);
    const int n_wave = tn * BN + wn * TN;
    const bool v_tile = (p.out_mode == BS_OUT_QKV) && (tn * BN >= 2 * p.qkv_hidden);
    __syncthreads();  // every wave is done reading the main-loop LDS
    for (int ps = 0; ps < PASSES; ++ps) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if ((i * 16) / PASS_R == ps) {
                const int r = i * 16 - ps * PASS_R + frow;
#pragma unroll
                for (int j = 0; j < FN; ++j) *reinterpret_cast<f32x4*>(sc + r * TN + (((j * 4 + fq) ^ (r & (S4 - 1))) << 2)) = acc[i][j];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int m_base = tm * BM + wm * TM + ps * PASS_R;
        if (!v_tile) {
            for (int it = 0; it < PASS_R / RPS; ++it) {
                const int r = it * RPS + lane / S4, c4 = lane % S4;
                const int m = m_base + r, n0 = n_wave + c4 * 4;
                if (m >= p.M || n0 >= p.N) continue;
                const f32x4 a = *reinterpret_cast<const f32x4*>(sc + r * TN + ((c4 ^ (r & (S4 - 1))) << 2));
                float y[4] = {a[0], a[1], a[2], a[3]};
                if (p.bias) {
                    const float* bias_row = p.bias + (p.bias_group_rows ? (int64_t)(m / p.bias_group_rows) * p.N : 0);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(bias_row + n0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] += b[e];
                }
                if (p.act != BS_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = apply_act(y[e], p.act);
                }
                if (p.scale) {
                    const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.scale + n0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] *= s4[e];
                }
                int64_t orow = m;
                if (p.out_mode == BS_OUT_PLAIN && p.out_group_rows) {
                    const int g = m / p.out_group_rows;
                    orow = (int64_t)g * p.out_group_stride + (m - g * p.out_group_rows) + p.out_row_offset;
                }
                if (p.res) {
                    const int64_t ro = orow * p.ldr + n0;  // the residual lives in the OUTPUT row geometry
                    if (p.res_dtype == BS_F32) {
                        const f32x4 rr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + ro);
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] += rr[e];
                    } else {
                        const typename T16<T>::v4 rr = *reinterpret_cast<const typename T16<T>::v4*>(reinterpret_cast<const T*>(p.res) + ro);
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] += T16<T>::to_f32(rr[e]);
                    }
                }
                if (p.res2) {  // second residual, 16-bit, same row geometry (fusion: fused + residual_unit(skip))
                    const typename T16<T>::v4 rr = *reinterpret_cast<const typename T16<T>::v4*>(reinterpret_cast<const T*>(p.res2) + orow * p.ldr + n0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] += T16<T>::to_f32(rr[e]);
                }
                if (p.out_mode == BS_OUT_PLAIN) {
                    store4<T>(p.out, orow * p.ldo + n0, p.out_dtype, y);
                } else if (p.out_mode == BS_OUT_SHUFFLE) {
                    const int hw = p.Hout * p.Wout;
                    const int ob = m / hw, rem = m - ob * hw;
                    const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                    const int s = p.shuffle_s;
                    const int tap = n0 / p.shuffle_cout, co = n0 - tap * p.shuffle_cout;
                    const int ky = tap / s, kx = tap - ky * s;
                    const int64_t pix = ((int64_t)(ob * p.Hout + oy) * s + ky) * (p.Wout * s) + ox * s + kx;
                    store4<T>(p.out, pix * p.ldo + co, p.out_dtype, y);
                } else {  // Q or K part of the fused QKV projection
                    const int ob = m / p.qkv_tokens, otok = m - ob * p.qkv_tokens;
                    const int which = n0 / p.qkv_hidden, rem = n0 - which * p.qkv_hidden;
                    const int hh = rem >> 6, d = rem & 63, nh = p.qkv_hidden >> 6;
                    const int64_t off = (((int64_t)ob * nh + hh) * p.qkv_sp + otok) * 64 + d;
                    if (which == 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] *= p.q_scale;
                        store4<T>(p.out, off, p.out_dtype, y);
                    } else {
                        store4<T>(p.out2, off, p.out_dtype, y);
                    }
                }
            }
        } else {
            // V part: written transposed (V^T [B,nh,64,Sp]); consecutive lanes take consecutive tokens
            T* vt = reinterpret_cast<T*>(p.out3);
            const int nh = p.qkv_hidden >> 6;
            for (int idx = lane; idx < PASS_R * TN; idx += 64) {
                const int c = idx / PASS_R, r = idx - c * PASS_R;
                const int m = m_base + r, n = n_wave + c;
                if (m >= p.M || n >= p.N) continue;
                float y = sc[r * TN + ((((c >> 2) ^ (r & (S4 - 1))) << 2) | (c & 3))];
                if (p.bias) y += p.bias[n];
                const int ob = m / p.qkv_tokens, otok = m - ob * p.qkv_tokens;
                const int rem = n - 2 * p.qkv_hidden;
                const int hh = rem >> 6, d = rem & 63;
                vt[(((int64_t)ob * nh + hh) * 64 + d) * p.qkv_sp + otok] = T16<T>::from_f32(y);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the next pass overwrites this wave's region
    }
#endif
}

template <typename T, int BM, int BN, int WM, int WN, int BK, int STAGES, int MODE>
static int launch_mode(const IgemmParams& p, hipStream_t st) {
    constexpr int smem = STAGES * (BM + BN) * BK * 2;
    dim3 grid(p.ntm * p.ntn), block(WM * WN * 64);
    auto k = igemm_kernel<T, BM, BN, WM, WN, BK, STAGES, MODE>;
    static bool attr = false;
    if (!attr) {
        BS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr = true;
    }
    hipLaunchKernelGGL(k, grid, block, smem, st, p);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

template <typename T, int BM, int BN, int WM, int WN, int BK, int STAGES>
static int launch_variant(const IgemmParams& p, bool conv, hipStream_t st) {
    if (!conv) return launch_mode<T, BM, BN, WM, WN, BK, STAGES, 0>(p, st);
    if (p.relu_a) return launch_mode<T, BM, BN, WM, WN, BK, STAGES, 2>(p, st);
    return launch_mode<T, BM, BN, WM, WN, BK, STAGES, 1>(p, st);
}

// tile ids: 1 128x128x64 2-stage (2 blocks/CU) | 2 128x64 | 3 128x32 | 4 256x128x64 2-stage
//           5 256x128x64 3-stage (144 KiB) | 6 256x256x32 4-stage (128 KiB) | 7 128x128x64 3-stage | 8 256x128x32 4-stage
//           9 256x256x64 2-stage (128 KiB)
static int auto_tile(int M, int N, int tile, bool conv) {
    if (tile != 0) return tile;
    if (N <= 32) return 3;
    if (N <= 64) return 2;
    // the 256x256x64 tile (1 block/CU, 128 KiB LDS) wins once the grid covers the 256 CUs often enough to amortise its
    // coarse tail; measured with tools/bench_kernels.py (plain K=1024..4096 GEMMs: from ~1.5 rounds; 3x3 convs: from 2)
    const long long blocks = (long long)cdiv(M, 256) * (N / 256);
    if (N % 256 == 0 && blocks >= (conv ? 512 : 384)) return 9;
    return 1;
}

static void tile_dims(int tile, int& BM, int& BN) {
    switch (tile) {
        case 2: BM = 128; BN = 64; break;
        case 3: BM = 128; BN = 32; break;
        case 4: case 5: case 8: BM = 256; BN = 128; break;
        case 6: case 9: BM = 256; BN = 256; break;
        default: BM = 128; BN = 128; break;
    }
}

template <typename T>
static int dispatch(IgemmParams& p, bool conv, int tile, hipStream_t st) {
    tile = auto_tile(p.M, p.N, tile, conv);
    if (tile < 1 || tile > 9) { set_error("bs_gemm: unknown tile %d", tile); return BS_ERR_INVALID; }
    int BM, BN;
    tile_dims(tile, BM, BN);
    p.ntm = cdiv(p.M, BM);
    p.ntn = cdiv(p.N, BN);
    switch (tile) {
        case 1: return launch_variant<T, 128, 128, 2, 2, 64, 2>(p, conv, st);
        case 2: return launch_variant<T, 128, 64, 2, 2, 64, 2>(p, conv, st);
        case 3: return launch_variant<T, 128, 32, 4, 1, 64, 2>(p, conv, st);
        case 4: return launch_variant<T, 256, 128, 4, 2, 64, 2>(p, conv, st);
        case 5: return launch_variant<T, 256, 128, 4, 2, 64, 3>(p, conv, st);
        case 6: return launch_variant<T, 256, 256, 2, 4, 32, 4>(p, conv, st);
        case 7: return launch_variant<T, 128, 128, 2, 2, 64, 3>(p, conv, st);
        case 8: return launch_variant<T, 256, 128, 4, 2, 32, 4>(p, conv, st);
        default: return launch_variant<T, 256, 256, 2, 4, 64, 2>(p, conv, st);
    }
}

}  // namespace bs

extern "C" int bs_gemm_tile(const bs_gemm_desc* d) { return d ? bs::auto_tile(d->M, d->N, d->tile, d->conv != 0) : BS_ERR_INVALID; }

extern "C" int bs_gemm(const bs_gemm_desc* d, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("bs_gemm: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(d && d->A && d->W && d->out, "bs_gemm: null operand");
    BS_REQUIRE(d->dtype == BS_F16 || d->dtype == BS_BF16, "bs_gemm: dtype must be f16 or bf16");
    BS_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "bs_gemm: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
    BS_REQUIRE(d->N % 4 == 0, "bs_gemm: N=%d must be a multiple of 4", d->N);
    BS_REQUIRE(d->K % 64 == 0, "bs_gemm: K=%d must be a multiple of 64", d->K);
    BS_REQUIRE(d->lda % 8 == 0, "bs_gemm: lda=%d must be a multiple of 8 (16-byte rows)", d->lda);
    BS_REQUIRE(d->out_dtype == BS_F32 || d->out_dtype == d->dtype, "bs_gemm: out_dtype must be f32 or the operand dtype");
    BS_REQUIRE(!d->res || d->res_dtype == BS_F32 || d->res_dtype == d->dtype, "bs_gemm: res_dtype must be f32 or the operand dtype");
    IgemmParams p{};
    p.A = d->A; p.W = d->W; p.zero = zero_page();
    p.a_bytes = d->conv ? (long long)(d->M / (d->Hout > 0 && d->Wout > 0 ? d->Hout * d->Wout : 1)) * d->Hin * d->Win * d->lda * 2
                        : ((long long)(d->M - 1) * d->lda + d->K) * 2;
    BS_REQUIRE((long long)d->N * d->K * 2 < 0x7FFFFFF0ll, "bs_gemm: weight matrix too large for one descriptor");
    p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda;
    p.Hin = d->Hin; p.Win = d->Win; p.Cin = d->Cin; p.Hout = d->Hout; p.Wout = d->Wout;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
    if (d->conv) {
        BS_REQUIRE(d->Cin > 0 && d->Cin % 64 == 0, "bs_gemm: conv Cin=%d must be a multiple of 64", d->Cin);
        BS_REQUIRE(d->K == d->KH * d->KW * d->Cin, "bs_gemm: conv K=%d != KH*KW*Cin", d->K);
        BS_REQUIRE(d->Hout > 0 && d->Wout > 0 && d->M % (d->Hout * d->Wout) == 0, "bs_gemm: conv M=%d not a multiple of Hout*Wout", d->M);
        BS_REQUIRE(d->stride > 0 && d->lda >= d->Cin, "bs_gemm: bad conv stride/lda");
        BS_REQUIRE(d->KH > 0 && d->KW > 0 && d->KH * d->KW <= 32, "bs_gemm: conv window %dx%d: at most 32 taps", d->KH, d->KW);
        BS_REQUIRE((long long)d->Hin * d->Win * d->lda * 2 * 4 < 0x7FFFFFF0ll, "bs_gemm: image too large for the 2 GiB tile window");
        p.tiles_per_tap = d->Cin / 64;
    }
    p.relu_a = d->relu_a;
    BS_REQUIRE(!d->relu_a || d->conv, "bs_gemm: relu_a is only built for conv mode");
    p.bias = d->bias; p.bias_group_rows = d->bias_group_rows; p.act = d->act; p.scale = d->scale;
    p.res = d->res; p.res2 = d->res2; p.res_dtype = d->res_dtype; p.ldr = d->ldr;
    BS_REQUIRE(!d->res2 || d->res, "bs_gemm: res2 needs res (they share ldr)");
    p.out = d->out; p.out2 = d->out2; p.out3 = d->out3; p.out_dtype = d->out_dtype; p.ldo = d->ldo; p.out_mode = d->out_mode;
    p.out_group_rows = d->out_group_rows; p.out_group_stride = d->out_group_stride; p.out_row_offset = d->out_row_offset;
    p.shuffle_s = d->shuffle_s; p.shuffle_cout = d->shuffle_cout;
    p.qkv_hidden = d->qkv_hidden; p.qkv_tokens = d->qkv_tokens; p.qkv_sp = d->qkv_sp; p.q_scale = d->q_scale;
    if (d->out_mode == BS_OUT_SHUFFLE) {
        BS_REQUIRE(!d->conv && d->shuffle_s > 0 && d->shuffle_cout % 4 == 0 && d->N == d->shuffle_s * d->shuffle_s * d->shuffle_cout,
                   "bs_gemm: bad shuffle geometry");
        BS_REQUIRE(d->Hout > 0 && d->Wout > 0 && d->M % (d->Hout * d->Wout) == 0, "bs_gemm: shuffle needs the input grid in Hout/Wout");
    } else if (d->out_mode == BS_OUT_QKV) {
        BS_REQUIRE(d->out2 && d->out3 && d->qkv_hidden % 64 == 0 && d->N == 3 * d->qkv_hidden && d->qkv_tokens > 0 &&
                       d->M % d->qkv_tokens == 0 && d->qkv_sp >= d->qkv_tokens && d->out_dtype == d->dtype,
                   "bs_gemm: bad qkv geometry");
    } else {
        BS_REQUIRE(d->out_mode == BS_OUT_PLAIN && d->ldo >= d->N, "bs_gemm: bad out_mode/ldo");
        BS_REQUIRE(d->ldo % 4 == 0, "bs_gemm: ldo must be a multiple of 4");
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == BS_F16) return dispatch<f16>(p, d->conv != 0, d->tile, st);
    return dispatch<bf16>(p, d->conv != 0, d->tile, st);
}
