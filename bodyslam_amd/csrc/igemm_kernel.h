// Implicit-GEMM on MFMA for gfx950: out[m, n] = epilogue(sum_k A(m, k) * W[n, k]).
//
// One kernel family serves every Linear / Conv2d / ConvTranspose2d(k == stride) on the hot path
// (include/bodyslam_hip.h: bs_gemm lists the reference call sites).  Design:
//   * NHWC activations, weights [N][Cin/64][KH][KW][64] (chunk > tap > channel, see the K walk below): a
//     BK = 64 slice of K is ONE filter tap and 64 contiguous channels, i.e. one 128-byte line per output
//     pixel.  The A tile is therefore a row gather: each lane of a `buffer_load_dwordx4 ... lds` supplies
//     the 32-bit offset of 16 bytes of its pixel (an out-of-range offset when the tap falls into the
//     padding: the DMA then writes zeros) and the data lands in LDS without touching VGPRs.  No im2col
//     buffer exists anywhere.
//   * LDS image per operand: [rows][64] 16-bit, 128-byte rows, 16-byte chunk c of row r stored at
//     chunk position c ^ (r & 7).  The DMA destination is lane-linear, so the XOR is applied to the
//     SOURCE chunk each lane fetches and again on the ds_read_b128 address (conflict-free for the
//     16x16x32 operand pattern: 16 distinct rows x one chunk per lane group).
//   * v_mfma_f32_16x16x32_{f16,bf16}; operands swapped (W fragment as "A", activation fragment as
//     "B") so a lane ends with 4 consecutive n for one m, and the W rows of a tile are permuted on load so
//     that a fragment pair gives a lane 8 consecutive n: 16-byte epilogue stores.
//   * accurate mode: K segments / FP8 correction stages evaluate a split-precision product in one launch
//     (IgemmParams::cin1, f8_stages; the F8 template flag keeps that code out of the fast-mode kernels).
//   * double-buffered LDS, one barrier per K tile: the DMA of tile t+1 is in flight while tile t
//     is multiplied.
//   * 1-D grid with the bijective XCD remap: the N-tiles of one M-tile (which share the gathered
//     activations) are consecutive work ids and land on one XCD's L2.
// Epilogue fuses bias (optionally per row group), ReLU/GELU/softplus, per-channel scale
// (BEiT layer-scale), residual add (fp32 or 16-bit), and three store layouts (plain with row
// regrouping, ConvTranspose pixel shuffle, Q/K/V^T head scatter).
#pragma once
#include <type_traits>

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
    int cin1;            // second K segment: channels per tap (conv) / K extent (plain); 0 = none.  Segment 0 is Cin (conv) / K - cin1 (plain)
    int split_off;       // > 0: 16-bit output is written as a (hi, lo) pair, lo at +split_off elements (PLAIN mode)
    int res_split_off;   // > 0: the 16-bit residual(s) are (hi, lo) pairs, lo at +res_split_off elements
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
    int qkv_cls_last;    // Q / K / V^T hold an image's tokens patches first, cls (token 0) last
    int qkv_cls_rows;    // > 0: grouped rows -- the first qkv_cls_rows rows are the images' cls tokens, patches from qkv_patch_row0 on
    int qkv_patch_row0;
    int qkv_lo_off;      // > 0 (BS_OUT_QKV): also store y - round16(y) of Q / K / V^T, this many elements behind the value
    int f8_wonly_from;   // tiles starting at a row >= this (> 0; -1: every tile) skip the second FP8 half (activation-rounding correction)
    float q_scale;
    int ntm, ntn;
    int m_begin;        // first output row this launch covers (a GEMM may be issued as a main launch + a tail launch)
    int f8_seg;          // > 0: the last f8_seg K-ELEMENTS of every A / W row are FP8 (e4m3) bytes, multiplied on the block-scaled
                         //      FP8 MFMA (2x the 16-bit rate): the correction products of a split-precision GEMM.  Two halves of
                         //      f8_seg / 2 elements each with their own power-of-two scales (E8M0 exponents, 127 = 1.0)
    int f8_sa0, f8_sb0, f8_sa1, f8_sb1;
    int f8_stages;       // number of trailing 128-byte K stages that are FP8 (host: f8_seg / 128, times the taps for a conv)
    int res_f8;          // != 0: the 16-bit residual(s) are (hi16 | hi8 | lo8) rows of N channels; value = hi16 + lo8 * 2^-BS_F8_ACT_LO_EXP
    int out_f8;          // > 0 with split_off: the output pair is (hi16 | hi8 | lo8), see store8_f8
    int out2_relu;       // != 0 (lean (hi16 | hi8 | lo8) epilogue only): out2 receives relu(y) in the same format and geometry
    int out_lo8_rows;    // > 0: only output rows below this index need their lo8 plane (the consumer drops the activation-rounding
                         //      correction on the others, f8_wonly_from): tiles past it skip that plane
    int f8_skip_from;    // > 0: tiles that start at a row >= this run no FP8 stage at all; -1: no tile does (a product calibrated down to one 16-bit pass)
    int out_planes_rows; // > 0: tiles that start at a row >= this store hi16 only (no FP8 plane; fc1 -> fc2 with f8_skip_from)
    const float* bias2;  // optional fp32 [groups, N] added to rows >= bias2_row0, group = (m - bias2_row0) / bias2_group_rows
    int bias2_row0, bias2_group_rows;
    int strip;    // work id -> tile order: 0 = row-major (N fastest over the whole width), w > 0 = strips of w N-tiles
    int ablate;   // diagnostics only (tools/bench_kernels.py): 1 = no DMA after the prologue, 2 = no LDS fragment reads after tile 0, 4 = no epilogue, 8 = no tail split (host side), 128 = the Q / K epilogue without its stores, 256 = no fast Q / K patch-tile path, 512 = no fast V^T patch-tile path, 1024 = no LDS-staged full-line stores (Q / K tiles, fc1's hi16 tiles), 2048 = no lean (hi16 | hi8 | lo8) epilogue (neck),
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
__device__ __forceinline__ void store4(void* base, int64_t off, int out_dtype, const float (&y)[4], int split_off = 0) {
    if (split_off > 0) {   // (hi, lo) pair: y = hi + lo to ~22 bits
        typename T16<T>::v4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = T16<T>::from_f32(y[e]);
            l[e] = T16<T>::from_f32(y[e] - T16<T>::to_f32(h[e]));
        }
        *reinterpret_cast<typename T16<T>::v4*>(reinterpret_cast<T*>(base) + off) = h;
        *reinterpret_cast<typename T16<T>::v4*>(reinterpret_cast<T*>(base) + off + split_off) = l;
        return;
    }
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

// 8 consecutive outputs from one lane (two fragments whose columns interleave, see the W-row permutation in igemm_kernel):
// one 16-byte store for 16-bit outputs, two for fp32 or (hi, lo) pairs
template <typename T>
__device__ __forceinline__ void store8(void* base, int64_t off, int out_dtype, const float (&y)[8], int split_off = 0) {
    typedef typename T16<T>::v8 v8;
    if (split_off > 0) {
        v8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            h[e] = T16<T>::from_f32(y[e]);
            l[e] = T16<T>::from_f32(y[e] - T16<T>::to_f32(h[e]));
        }
        *reinterpret_cast<v8*>(reinterpret_cast<T*>(base) + off) = h;
        *reinterpret_cast<v8*>(reinterpret_cast<T*>(base) + off + split_off) = l;
        return;
    }
    if (out_dtype == BS_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off) = f32x4{y[0], y[1], y[2], y[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off + 4) = f32x4{y[4], y[5], y[6], y[7]};
    } else {
        v8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = T16<T>::from_f32(y[e]);
        *reinterpret_cast<v8*>(reinterpret_cast<T*>(base) + off) = v;
    }
}

// (hi16 | hi8 | lo8) output pair of 8 consecutive columns: hi = round16(y) at off; e4m3(y * 2^ea) at byte n of the hi8 plane,
// e4m3((y - hi) * 2^el) in the lo8 plane.  Planes (in 16-bit element units from the row start): hi16 [0, N), hi8 [N, N + N/2),
// lo8 [N + N/2, 2N).  `off` = row start + n, plane_off = N (= split_off).
template <typename T>
__device__ __forceinline__ void store8_f8(void* base, int64_t row_off, int n, int plane_off, const float (&y)[8], int ea, int el, bool lo = true,
                                          bool hi8 = true) {
    typedef typename T16<T>::v8 v8;
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    // (two values per instruction for the scalings, one v_med3_f32 for each clamp: the epilogues that emit planes are VALU-bound)
    v8 h;
    f32x2_ yh[4];
    const float sa = __builtin_ldexpf(1.0f, ea), sl = __builtin_ldexpf(1.0f, el);
    const f32x2_ sa2 = {sa, sa}, sl2 = {sl, sl};
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        h[e] = T16<T>::from_f32(y[e]);
        h[e + 1] = T16<T>::from_f32(y[e + 1]);
        const f32x2_ v = f32x2_{y[e], y[e + 1]} * sa2;
        yh[e >> 1] = f32x2_{__builtin_amdgcn_fmed3f(v[0], -448.0f, 448.0f), __builtin_amdgcn_fmed3f(v[1], -448.0f, 448.0f)};
    }
    T* rowp = reinterpret_cast<T*>(base) + row_off;
    *reinterpret_cast<v8*>(rowp + n) = h;
    if (!hi8) return;   // (wave-uniform) a map whose every consumer runs one 16-bit pass: no plane is read
    int ph0 = 0, ph1 = 0;
    ph0 = __builtin_amdgcn_cvt_pk_fp8_f32(yh[0][0], yh[0][1], ph0, false);
    ph0 = __builtin_amdgcn_cvt_pk_fp8_f32(yh[1][0], yh[1][1], ph0, true);
    ph1 = __builtin_amdgcn_cvt_pk_fp8_f32(yh[2][0], yh[2][1], ph1, false);
    ph1 = __builtin_amdgcn_cvt_pk_fp8_f32(yh[3][0], yh[3][1], ph1, true);
    char* bytes = reinterpret_cast<char*>(rowp + plane_off);
    *reinterpret_cast<i32x2*>(bytes + n) = i32x2{ph0, ph1};
    if (lo) {       // (wave-uniform) the plane of the rounding residuals: only where the consumer evaluates A_lo8 W_hi8
        f32x2_ yl[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2_ r = (f32x2_{y[e], y[e + 1]} + f32x2_{-T16<T>::to_f32(h[e]), -T16<T>::to_f32(h[e + 1])}) * sl2;
            yl[e >> 1] = f32x2_{__builtin_amdgcn_fmed3f(r[0], -448.0f, 448.0f), __builtin_amdgcn_fmed3f(r[1], -448.0f, 448.0f)};
        }
        int pl0 = 0, pl1 = 0;
        pl0 = __builtin_amdgcn_cvt_pk_fp8_f32(yl[0][0], yl[0][1], pl0, false);
        pl0 = __builtin_amdgcn_cvt_pk_fp8_f32(yl[1][0], yl[1][1], pl0, true);
        pl1 = __builtin_amdgcn_cvt_pk_fp8_f32(yl[2][0], yl[2][1], pl1, false);
        pl1 = __builtin_amdgcn_cvt_pk_fp8_f32(yl[3][0], yl[3][1], pl1, true);
        *reinterpret_cast<i32x2*>(bytes + plane_off + n) = i32x2{pl0, pl1};
    }
}

// y[0..3] += the lo8 plane of a (hi16 | hi8 | lo8) row of C channels at columns n..n+3
template <typename T>
__device__ __forceinline__ void add_lo8(float* y, const void* base, int64_t row_off, int C, int n) {
    const char* lp = reinterpret_cast<const char*>(reinterpret_cast<const T*>(base) + row_off + C + (C >> 1)) + n;
    const int packed = *reinterpret_cast<const int*>(lp);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(packed, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(packed, true);
    const float sc = __builtin_ldexpf(1.0f, -BS_F8_ACT_LO_EXP);
    y[0] += a[0] * sc; y[1] += a[1] * sc; y[2] += b[0] * sc; y[3] += b[1] * sc;
}

constexpr int NW_CHECK(int a, int b) { return a * b; }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Tile variants.  BK is the K slice per stage (one 64- or 128-byte LDS row per tile row), STAGES the depth
// of the LDS ring: STAGES-1 tiles are in flight (global_load_lds) while one is multiplied.
// MODE: 0 plain GEMM rows, 1 implicit conv, 2 implicit conv with ReLU applied to A on load
// CM (correction mode): 1 = the instantiation that knows the FP8 correction stages and the (hi16 | hi8 | lo8) epilogue formats
// (accurate mode); 0 = the plain instantiation, which carries none of that code, so fast-mode launches are not affected by its
// register pressure.  (CM = 2, FP4 correction stages with per-block scales, existed in round 3: +1.4 % frames/s for 1.5x the depth error --
// removed in round 4, profiles/r03_fp4_corrections.txt keeps the measurements.)
template <typename T, int BM, int BN, int WM, int WN, int BK, int STAGES, int MODE, bool PP = false, int CM = 0>
__global__ __launch_bounds__(WM* WN * 64, 2) void igemm_kernel(const IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-descriptor builtins exist in the device pass only; the host pass needs just the stub
    constexpr bool CONV = MODE != 0;
    constexpr bool RELU_A = MODE == 2;
    constexpr bool F8 = CM != 0;
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
    int tm, tn;
    if (p.strip > 0) {       // column strips of p.strip N-tiles, M fastest across strips: a strip's W tiles stay in the XCD's L2 while A streams through
        const int per = p.ntm * p.strip, s = wg / per, r = wg - s * per;
        const int wl = p.ntn - s * p.strip < p.strip ? p.ntn - s * p.strip : p.strip;
        tm = r / wl;
        tn = s * p.strip + (r - tm * wl);
    } else {
        tm = wg / p.ntn;
        tn = wg - tm * p.ntn;
    }

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
    const int m0 = p.m_begin + tm * BM;
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
    const int w_pitch = p.K * 2;     // bytes per W row
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.W), 0, (int)((long long)p.N * w_pitch), 0x00020000);

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
    // LDS row r of the W tile holds output column perm(r): inside every group of 32 rows the two 16-row fragments take
    // INTERLEAVED groups of 4 columns (fragment e, MFMA column c -> column 8*(c>>2) + 4*e + (c&3)), so that a lane's 4+4
    // accumulator values of a fragment pair are 8 CONSECUTIVE output columns: one 16-byte store, 64 contiguous bytes per row.
    static_assert(BN % 32 == 0 && TN % 32 == 0, "column permutation works on 32-column groups");
    unsigned w_off[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int r = j * RPR + srow;
        int n = tn * BN + ((r & ~31) | ((r & 12) << 1) | ((r & 16) >> 2) | (r & 3));
        n = n < p.N ? n : p.N - 1;
        w_off[j] = (unsigned)(n * w_pitch + cs16);
    }
    // running state of the NEXT tile to stage.  W side: s_k (byte offset along K, runs straight through).  A side: the
    // K axis is up to two SEGMENTS that both walk the same rows of A -- segment 0 with seg0 bytes per tap, then segment 1
    // with seg1 bytes per tap (taps restart).  This is how split-precision products are expressed without a second
    // kernel: A = [hi | lo] channels, W' = [W_hi | W_hi] then [W_lo] against the hi channels again (Numerics).
    //
    // Order of K inside a segment: 64-channel CHUNK outermost, filter tap inside it, the chunk's 64 channels innermost
    // (W is laid out to match: [N][segment][chunk][tap][64]).  Tiles that run side by side on an XCD walk K in step, so
    // the three kernel rows a tap triple (ky = 0,1,2) needs from one input row are requested within 3 K-steps of each other
    // and by neighbouring tiles within a few more: the re-reads hit the XCD's 4 MiB L2.  (Tap-outermost order spreads them a
    // whole Cin apart: 9x the L2 miss traffic on the 256-channel, 192x256 feature maps.)
    const int seg0 = (CONV ? p.Cin : p.K - p.cin1) * 2, seg1 = p.cin1 * 2;
    const int ntaps = CONV ? p.KH * p.KW : 1;
    int s_tap = 0, s_kx = 0, s_tapoff = 0, s_cb = 0, s_sub = 0, s_k = 0, s_seg = seg0;

    auto stage = [&](int buf) {
        char* sa = smem + buf * STAGE;
        char* sb = sa + A_BYTES;
        const int s_c0 = s_cb + s_sub;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            unsigned vo;
            if (CONV) {
                vo = ((a_mask[j] >> s_tap) & 1u) ? a_off[j] + (unsigned)s_tapoff : OOB;
            } else {
                vo = a_off[j];
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(sa + (j * RPR + wave * RPW) * ROWB), 16, vo,
                                                     s_c0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < RB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(sb + (j * RPR + wave * RPW) * ROWB), 16, w_off[j],
                                                     s_k, 0, 0);
        s_k += BK * 2;
        if (BK == 32) {
            s_sub ^= 64;                       // second half of the 64-channel chunk, same tap
            if (s_sub) return;
        }
        if (CONV) {                            // next tap of this chunk
            ++s_tap;
            s_tapoff += p.lda * 2;
            if (++s_kx >= p.KW) {
                s_kx = 0;
                s_tapoff += (p.Win - p.KW) * p.lda * 2;
            }
        }
        if (!CONV || s_tap >= ntaps) {         // chunk finished: next chunk, taps restart
            s_tap = 0;
            s_kx = 0;
            s_tapoff = 0;
            s_cb += 128;
            if (s_cb >= s_seg) {               // segment 0 finished: segment 1 walks the same rows again
                s_cb = 0;
                s_seg = seg1;
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

    const int nt_all = p.K / BK;
    const int ncorr = p.f8_stages;
    const int nt16 = F8 ? nt_all - ncorr : nt_all;   // stages multiplied as 16-bit data; the rest are FP8 / FP4 correction stages (BK = 64 tiles)
    // the second half of the FP8 stages (A_lo8 W_hi8) is dropped for tiles past f8_wonly_from: they stop after the first half
    const bool wonly = F8 && p.f8_wonly_from != 0 && (p.f8_wonly_from < 0 || m0 >= p.f8_wonly_from);
    const bool skip8 = F8 && p.f8_skip_from != 0 && (p.f8_skip_from < 0 || m0 >= p.f8_skip_from);
    const int nt = skip8 ? nt16 : (wonly ? nt_all - (ncorr >> 1) : nt_all);
    if constexpr (!PP) {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nt) stage(s);
        int cbuf = 0, sbuf = STAGES - 1;   // buffer multiplied this iteration / buffer staged this iteration
        v8 af[FM], bf[FN];
        // one iteration of the ring; F8 selects the multiply of the stage (two plain loops, not a branch inside one: a branch
        // made the register allocator spill the accumulators)
        auto iteration = [&](int t, auto f8_tag) {
            constexpr bool F8S = decltype(f8_tag)::value == 1;
            // my own DMA for tile t has landed once at most (tiles issued after t) x GL operations are outstanding
            const int younger = nt - 1 - t;
            if (STAGES >= 4 && younger >= 2) wait_vmcnt<(STAGES >= 4 ? 2 : 0) * GL>();
            else if (STAGES >= 3 && younger >= 1) wait_vmcnt<(STAGES >= 3 ? 1 : 0) * GL>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();   // everyone's tile t has landed; everyone is done reading buffer sbuf (tile t-1)
            asm volatile("" ::: "memory");
            if (t + STAGES - 1 < nt && !(p.ablate & 1)) stage(sbuf);
            const char* sa = smem + cbuf * STAGE;
            const char* sb = sa + A_BYTES;
            if constexpr (F8S) {
                // ---- FP8 correction stage: the 128-byte LDS rows hold 128 e4m3 values.  A lane supplies 32 of them per
                // row: the same two 16-byte chunks (fq and 4 + fq) the 16-bit path reads -- A and W use the same
                // permutation of k, so the product is unchanged -- as one 8-register operand of the 16x16x128 MFMA.
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                typedef int i32x8 __attribute__((ext_vector_type(8)));
                const bool second = t >= nt16 + ((nt_all - nt16) >> 1);
                const int sca = (second ? p.f8_sa1 : p.f8_sa0) * 0x01010101, scb = (second ? p.f8_sb1 : p.f8_sb0) * 0x01010101;
                auto ld8 = [&](const char* base) {
                    const i32x4 lo = *reinterpret_cast<const i32x4*>(base + koff0);
                    const i32x4 hi = *reinterpret_cast<const i32x4*>(base + koff1);
                    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                };
                i32x8 b8[FN];
#pragma unroll
                for (int j = 0; j < FN; ++j) b8[j] = ld8(sb + b_base + j * 16 * ROWB);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < FM; ++i) {     // one A fragment (8 registers) at a time: the kernel is at the 256-VGPR line
                    const i32x8 a8 = ld8(sa + a_base + i * 16 * ROWB);
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8[j], a8, acc[i][j], 0, 0, 0, scb, 0, sca);
                }
                __builtin_amdgcn_s_setprio(0);
            } else {
#pragma unroll
                for (int kk = 0; kk < BK / 32; ++kk) {
                    const int ko = kk ? koff1 : koff0;
                    if (!(p.ablate & 2) || t == 0) {
#pragma unroll
                        for (int i = 0; i < FM; ++i) {
                            af[i] = *reinterpret_cast<const v8*>(sa + a_base + i * 16 * ROWB + ko);
                            if (RELU_A) af[i] = relu8<T>(af[i]);
                        }
#pragma unroll
                        for (int j = 0; j < FN; ++j) bf[j] = *reinterpret_cast<const v8*>(sb + b_base + j * 16 * ROWB + ko);
                    }
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) acc[i][j] = T16<T>::mfma16(bf[j], af[i], acc[i][j]);
                    __builtin_amdgcn_s_setprio(0);
                }
            }
            cbuf = cbuf + 1 == STAGES ? 0 : cbuf + 1;
            sbuf = sbuf + 1 == STAGES ? 0 : sbuf + 1;
        };
        int t = 0;
        // (kernel-argument loads still outstanding from the prologue are retired here, before the first LDS fragment read: no scalar load
        // shares a wait with an LDS read -- tools/probes/lgkm_mix_audit.py; a precaution from round 6, neutral in time)
        __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0) alone: a fence would also wait for the LDS-DMA of the staged tiles (vmcnt)
        for (; t < nt16; ++t) iteration(t, std::integral_constant<int, 0>{});
        if constexpr (F8 && BK == 64 && !RELU_A) {
            for (; t < nt; ++t) iteration(t, std::integral_constant<int, 1>{});
        }
    } else {
        // ---- ping-pong schedule (BK = 32, 4-stage ring, 8 waves = two groups of four, one wave of each group per
        // SIMD).  A wave alternates a LOAD phase (issue its 4 LDS-DMA pieces of tile t+3, read ALL 12 fragments of
        // its tile into registers) with a COMPUTE phase (32 MFMAs from registers, nothing else); the groups run half a
        // tile apart, so on every SIMD one wave feeds the matrix pipe while its partner moves data.
        //   phase 2t   : group 0 computes tile t          | group 1 issues DMA(t+3), loads tile t
        //   phase 2t+1 : group 0 issues DMA(t+3), loads t+1 | group 1 computes tile t
        // One barrier per phase.  Before the barrier that opens an odd phase every wave waits (counted vmcnt) for its
        // own pieces of the tile that group 0 is about to read; the ring slot of tile t+3 was last read two phases ago.
        static_assert(BK == 32 && STAGES == 4 && WM == 2 && NW_CHECK(WM, WN) == 8, "ping-pong variant: 256x256x32, 8 waves");
        const int grp = wm;
        v8 af[FM], bf[FN];
        auto load_frags = [&](int buf) {
            const char* sa = smem + buf * STAGE;
            const char* sb = sa + A_BYTES;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                af[i] = *reinterpret_cast<const v8*>(sa + a_base + i * 16 * ROWB + koff0);
                if (RELU_A) af[i] = relu8<T>(af[i]);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = *reinterpret_cast<const v8*>(sb + b_base + j * 16 * ROWB + koff0);
        };
        auto compute = [&]() {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = T16<T>::mfma16(bf[j], af[i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        };
        auto wait_tiles = [&](int younger) {   // leave `younger` of my tiles (GL pieces each) in flight
            if (younger >= 2) wait_vmcnt<2 * GL>();
            else if (younger == 1) wait_vmcnt<GL>();
            else wait_vmcnt<0>();
        };
        auto phase_barrier = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (s < nt) stage(s);
        wait_tiles((nt - 1 < 2 ? nt - 1 : 2));
        phase_barrier();                       // tile 0 has landed everywhere
        // Both groups run the SAME loop; group 1 runs it one phase later (it sits out phase -1 and issues tile 3 on
        // entering phase 0), group 0 sits out the last phase.
        int issued = nt - 1 < 2 ? nt - 1 : 2;  // highest tile this wave has issued
        if (grp == 1) {
            phase_barrier();
            if (3 < nt) { stage(3); issued = 3; }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // (as above)
        load_frags(0);
        int buf = 0;                           // ring slot of tile t
        const int last = nt - 1;
        for (int t = 0; t < nt; ++t) {
            const int nbuf = buf == 3 ? 0 : buf + 1;
            // group 1: this barrier opens the odd phase in which group 0 reads tile t+1 -> my pieces of t+1 must have landed
            if (grp == 1 && t + 1 < nt) wait_tiles(issued - (t + 1));
            phase_barrier();
            compute();
            if (grp == 0 && t + 1 < nt) wait_tiles(issued - (t + 1));
            phase_barrier();
            if (t + 1 < nt) {
                if (issued < last) { ++issued; stage(issued & 3); }
                load_frags(nbuf);
            }
            buf = nbuf;
        }
        if (grp == 0) phase_barrier();
    }

    // ---- epilogue.  A lane holds, per 16x16 fragment, 4 CONSECUTIVE n of one row m (operands were swapped), so it
    // stores 8 bytes (16-bit out) or 16 bytes (fp32 out) straight from registers: no LDS round trip, no barrier.
    // Everything that depends only on the column (bias, layer-scale, head / tap decomposition) is hoisted per fragment
    // column j, everything that depends only on the row (regrouped row, image / token / pixel decomposition) per fragment
    // row i; the activation is selected once, outside the fragment loops.  Only the V third of the fused QKV projection
    // still goes through LDS: it is written TRANSPOSED (V^T [B,nh,64,Sp]), consecutive lanes taking consecutive tokens.
    //   lane: m = m0 + wm*TM + i*16 + (lane & 15),  n = tn*BN + wn*TN + (j>>1)*32 + (lane >> 4)*8 + (j&1)*4 + 0..3
    // (The lane id afresh: the thread id and what derives from it need not live through the main loop, which runs at 245-252 of 256
    // registers, for the epilogue's sake.)
    // Each epilogue form reads it again at its top (BS_FRESH_LANE, two instructions the compiler may not merge): merged into one value
    // it is spilled to scratch right behind the main loop and reloaded per form.
#define BS_FRESH_LANE                                                                                                       \
    int lane;                                                                                                               \
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));                             \
    const int frow = lane & 15, fq = lane >> 4;                                                                             \
    (void)frow; (void)fq;
    {
    const int n_wave = tn * BN + wn * TN;
    const bool v_tile = (p.out_mode == BS_OUT_QKV) && (tn * BN >= 2 * p.qkv_hidden);
    if (p.ablate & 4) {
        if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.out)[0] = 1.f;   // keep the accumulators live
        return;
    }
    // bias2 (per-group second bias): g2 >= 0 when every row of this tile belongs to one group (folded into the column biases),
    // b2_rows when the tile straddles groups or the first covered row (looked up per row)
    int g2 = -1;
    bool b2_rows = false;
    if (p.bias2) {
        const int lo = m0 - p.bias2_row0, hi = (m0 + BM < p.M ? m0 + BM : p.M) - 1 - p.bias2_row0;
        if (hi >= 0) {
            if (lo >= 0 && lo / p.bias2_group_rows == hi / p.bias2_group_rows) g2 = lo / p.bias2_group_rows;
            else b2_rows = true;
        }
    }
    if (!v_tile) {
        auto epi = [&](auto act_tag) {
            constexpr int ACT = decltype(act_tag)::value;
            BS_FRESH_LANE
            f32x4 bj[FN], sj[FN];
            int n0j[FN];
            int64_t coff[FN];        // column part of the store offset (elements)
            int which[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n0 = n_wave + (j >> 1) * 32 + fq * 8 + (j & 1) * 4;
                n0j[j] = n0;
                const bool ok = n0 < p.N;
                bj[j] = (p.bias && !p.bias_group_rows && ok) ? *reinterpret_cast<const f32x4*>(p.bias + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (g2 >= 0 && ok) bj[j] += *reinterpret_cast<const f32x4*>(p.bias2 + (int64_t)g2 * p.N + n0);
                sj[j] = (p.scale && ok) ? *reinterpret_cast<const f32x4*>(p.scale + n0) : f32x4{1.f, 1.f, 1.f, 1.f};
                which[j] = 0;
                if (p.out_mode == BS_OUT_PLAIN) {
                    coff[j] = n0;
                } else if (p.out_mode == BS_OUT_SHUFFLE) {
                    const int s = p.shuffle_s;
                    const int tap = n0 / p.shuffle_cout, co = n0 - tap * p.shuffle_cout;
                    const int ky = tap / s, kx = tap - ky * s;
                    coff[j] = ((int64_t)ky * (p.Wout * s) + kx) * p.ldo + co;
                } else {
                    const int w_ = n0 / p.qkv_hidden, rem = n0 - w_ * p.qkv_hidden;
                    which[j] = w_;
                    coff[j] = (int64_t)(rem >> 6) * p.qkv_sp * 64 + (rem & 63);
                    if (w_ == 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { sj[j][e] *= p.q_scale; bj[j][e] *= 1.0f; }
                    }
                }
            }
            // 16-byte stores need every row start and the split offset on an 8-element boundary (always true for the plans in
            // zoedepth.py; the 4-wide path stays for odd leading dimensions)
            const bool wide_ok = (p.out_mode == BS_OUT_QKV) || (p.ldo % 8 == 0 && p.split_off % 8 == 0);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm * TM + i * 16 + frow;
                if (m >= p.M) continue;
                int64_t orow = m;        // row in the output / residual geometry
                int64_t roff;            // row part of the store offset (elements)
                const float* brow = nullptr;
                if (p.bias && p.bias_group_rows) brow = p.bias + (int64_t)(m / p.bias_group_rows) * p.N;
                const float* brow2 = (b2_rows && m >= p.bias2_row0) ? p.bias2 + (int64_t)((m - p.bias2_row0) / p.bias2_group_rows) * p.N : nullptr;
                if (p.out_mode == BS_OUT_PLAIN) {
                    if (p.out_group_rows) {
                        const int g = m / p.out_group_rows;
                        orow = (int64_t)g * p.out_group_stride + (m - g * p.out_group_rows) + p.out_row_offset;
                    }
                    roff = orow * p.ldo;
                } else if (p.out_mode == BS_OUT_SHUFFLE) {
                    const int hw = p.Hout * p.Wout;
                    const int ob = m / hw, rem = m - ob * hw;
                    const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                    const int s = p.shuffle_s;
                    roff = (((int64_t)(ob * p.Hout + oy) * s) * (p.Wout * s) + ox * s) * p.ldo;
                } else {
                    int ob, otok;
                    if (p.qkv_cls_rows > 0) {      // grouped rows: cls tokens first, then the patches image by image (positions: cls last)
                        if (m >= p.qkv_cls_rows && m < p.qkv_patch_row0) continue;      // padding rows between the two groups
                        const int mp = m - p.qkv_patch_row0;
                        ob = mp < 0 ? m : mp / (p.qkv_tokens - 1);
                        otok = mp < 0 ? p.qkv_tokens - 1 : mp - ob * (p.qkv_tokens - 1);
                    } else {
                        ob = m / p.qkv_tokens;
                        otok = m - ob * p.qkv_tokens;
                        if (p.qkv_cls_last) otok = otok == 0 ? p.qkv_tokens - 1 : otok - 1;
                    }
                    roff = ((int64_t)ob * (p.qkv_hidden >> 6) * p.qkv_sp + otok) * 64;
                }
                float yall[FN / 2][8];
#pragma unroll
                for (int jp = 0; jp < FN / 2; ++jp) {
                    float (&y8)[8] = yall[jp];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int j = jp * 2 + h;
                        float* y = y8 + h * 4;
                        if (n0j[j] >= p.N) continue;
                        // (the same association as a tile that lies in one group, where bias2 is folded into bj: acc + (bias + bias2) --
                        // a row's result must not depend on which rows share its tile, i.e. on the batch)
                        f32x4 bsum = brow ? *reinterpret_cast<const f32x4*>(brow + n0j[j]) : bj[j];
                        if (brow2) bsum += *reinterpret_cast<const f32x4*>(brow2 + n0j[j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = acc[i][j][e] + bsum[e];
                        const bool res32 = p.res && p.res_dtype == BS_F32;
                        if (ACT == BS_ACT_GELU) {                 // two values per instruction (bit-identical to gelu_erf per element)
                            const f32x2_ g0 = gelu_erf2(f32x2_{y[0], y[1]}), g1 = gelu_erf2(f32x2_{y[2], y[3]});
                            y[0] = g0[0]; y[1] = g0[1]; y[2] = g1[0]; y[3] = g1[1];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (ACT == BS_ACT_RELU) y[e] = fmaxf(y[e], 0.0f);
                            else if (ACT == BS_ACT_SOFTPLUS) y[e] = softplus20(y[e]);
                            else if (ACT == BS_ACT_SOFTPLUS_FAST) y[e] = softplus_fast(y[e]);
                            if (!res32) y[e] *= sj[j][e];
                        }
                        if (p.res) {
                            const int64_t ro = orow * p.ldr + n0j[j];   // residuals live in the OUTPUT row geometry
                            if (p.res_dtype == BS_F32) {
                                const f32x4 rr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + ro);
#pragma unroll
                                for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf(y[e], sj[j][e], rr[e]);     // (one fused step, as the o_proj / fc2 form below)
                            } else {
                                const typename T16<T>::v4 rr = *reinterpret_cast<const typename T16<T>::v4*>(reinterpret_cast<const T*>(p.res) + ro);
#pragma unroll
                                for (int e = 0; e < 4; ++e) y[e] += T16<T>::to_f32(rr[e]);
                                if (p.res_split_off > 0) {
                                    const typename T16<T>::v4 rl = *reinterpret_cast<const typename T16<T>::v4*>(reinterpret_cast<const T*>(p.res) + ro + p.res_split_off);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) y[e] += T16<T>::to_f32(rl[e]);
                                } else if (F8 && p.res_f8) {
                                    add_lo8<T>(y, p.res, orow * p.ldr, p.N, n0j[j]);
                                }
                            }
                            if (p.res2) {   // second residual, 16-bit (fusion: fused + residual_unit(skip))
                                const typename T16<T>::v4 rr = *reinterpret_cast<const typename T16<T>::v4*>(reinterpret_cast<const T*>(p.res2) + ro);
#pragma unroll
                                for (int e = 0; e < 4; ++e) y[e] += T16<T>::to_f32(rr[e]);
                                if (p.res_split_off > 0) {
                                    const typename T16<T>::v4 rl = *reinterpret_cast<const typename T16<T>::v4*>(reinterpret_cast<const T*>(p.res2) + ro + p.res_split_off);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) y[e] += T16<T>::to_f32(rl[e]);
                                } else if (F8 && p.res_f8) {
                                    add_lo8<T>(y, p.res2, orow * p.ldr, p.N, n0j[j]);
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int jp = 0; jp < FN / 2; ++jp) {
                    float (&y8)[8] = yall[jp];
                    const int j0 = jp * 2, j1 = j0 + 1;
                    if (n0j[j0] >= p.N) continue;
                    void* dst = (p.out_mode == BS_OUT_QKV && which[j0] == 1) ? p.out2 : p.out;
                    const int so = p.out_mode != BS_OUT_QKV ? p.split_off : p.qkv_lo_off;   // (QKV: the residual tensors of the split-precision attention)
                    // the pair is one 8-wide store when both halves exist, are adjacent in the output and 16-byte aligned
                    if (F8 && p.out_f8) {     // PLAIN / SHUFFLE, channels % 8 == 0 (checked on the host): (hi16 | hi8 | lo8) planes per row / pixel
                        const int nloc = p.out_mode == BS_OUT_SHUFFLE ? n0j[j0] % p.shuffle_cout : n0j[j0];
                        store8_f8<T>(dst, roff + coff[j0] - nloc, nloc, so, y8, p.out_f8 & 0xff, (p.out_f8 >> 8) & 0xff);
                    } else if (wide_ok && n0j[j1] < p.N && coff[j1] == coff[j0] + 4 && which[j1] == which[j0]) {
                        store8<T>(dst, roff + coff[j0], p.out_dtype, y8, so);
                    } else {
                        const float(&ya)[4] = *reinterpret_cast<const float(*)[4]>(y8);
                        store4<T>(dst, roff + coff[j0], p.out_dtype, ya, so);
                        if (n0j[j1] < p.N) {
                            const float(&yb)[4] = *reinterpret_cast<const float(*)[4]>(y8 + 4);
                            void* dst1 = (p.out_mode == BS_OUT_QKV && which[j1] == 1) ? p.out2 : p.out;
                            store4<T>(dst1, roff + coff[j1], p.out_dtype, yb, so);
                        }
                    }
                }
            }
        };
        // ---- two specialised forms of the same arithmetic for the backbone's hot shapes (wave-uniform selection) -----------------
        const bool plain_full = p.out_mode == BS_OUT_PLAIN && !p.out_group_rows && p.bias && !p.bias_group_rows && p.N % BN == 0 &&
                                p.ldo % 8 == 0 && !b2_rows && !(p.ablate & 16);
        if (plain_full && p.act == BS_ACT_NONE && p.res && p.res_dtype == BS_F32 && !p.res2 && p.out_dtype == BS_F32 && p.split_off == 0 &&
            !(F8 && p.out_f8) && p.scale && p.ldr % 4 == 0) {
            // o_proj / fc2: x += scale * (acc + bias), fp32 in place.  The generic loop reads the residual where it needs it and the
            // kernel has no register to spare for the compiler to hoist those loads: every fragment pair waited out a full memory
            // latency (220 us of a 500 us o_proj launch).  Here the loads run DEPTH pairs ahead of their use.
            BS_FRESH_LANE
            constexpr int NP = FM * (FN / 2), DEPTH = 8;
            const float* resp = reinterpret_cast<const float*>(p.res);
            float* outp = reinterpret_cast<float*>(p.out);
            f32x4 bj[FN], sj[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n0 = n_wave + (j >> 1) * 32 + fq * 8 + (j & 1) * 4;
                bj[j] = *reinterpret_cast<const f32x4*>(p.bias + n0);
                if (g2 >= 0) bj[j] += *reinterpret_cast<const f32x4*>(p.bias2 + (int64_t)g2 * p.N + n0);
                sj[j] = *reinterpret_cast<const f32x4*>(p.scale + n0);
            }
            const int mrow = m0 + wm * TM + frow;
            f32x4 rq[DEPTH][2];
            auto issue = [&](int it, int slot) {
                const int i = it / (FN / 2), jp = it - i * (FN / 2);
                int m = mrow + i * 16;
                m = m < p.M ? m : p.M - 1;
                const float* rp = resp + (int64_t)m * p.ldr + n_wave + jp * 32 + fq * 8;
                rq[slot][0] = *reinterpret_cast<const f32x4*>(rp);
                rq[slot][1] = *reinterpret_cast<const f32x4*>(rp + 4);
            };
#pragma unroll
            for (int it = 0; it < DEPTH; ++it) issue(it, it);
#pragma unroll
            for (int it = 0; it < NP; ++it) {
                const int i = it / (FN / 2), jp = it - i * (FN / 2);
                const int m = mrow + i * 16;
                f32x4 y0, y1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y0[e] = __builtin_fmaf(acc[i][2 * jp][e] + bj[2 * jp][e], sj[2 * jp][e], rq[it % DEPTH][0][e]);
                    y1[e] = __builtin_fmaf(acc[i][2 * jp + 1][e] + bj[2 * jp + 1][e], sj[2 * jp + 1][e], rq[it % DEPTH][1][e]);
                }
                if (it + DEPTH < NP) issue(it + DEPTH, it % DEPTH);
                if (m < p.M) {
                    float* op = outp + (int64_t)m * p.ldo + n_wave + jp * 32 + fq * 8;
                    *reinterpret_cast<f32x4*>(op) = y0;
                    *reinterpret_cast<f32x4*>(op + 4) = y1;
                }
            }
            return;
        }
        if constexpr (F8) {
            // The neck's convolutions and 1x1 projections (round 4): (acc + bias) [ReLU] [+ residual(s)] as (hi16 | hi8 | lo8) rows.  The
            // generic loop above decides mode, activation and residual format per fragment and loads every residual piece (8 + 4 bytes,
            // two tensors in the fusion blocks' second conv) right where it adds it: fu3.r1.c2's epilogue was 56 us of a tile against
            // 18 us for the same conv without residuals (tools/probes/plan_epilogue_share.py).  Here the residual pieces of a fragment
            // pair (16 + 8 bytes per tensor) run RDEPTH pairs ahead of their use; same operations in the same order, same bits.
            if (p.out_mode == BS_OUT_PLAIN && !p.out_group_rows && !p.bias_group_rows && p.N % BN == 0 && p.ldo % 8 == 0 && !b2_rows &&
                !(p.ablate & (16 | 2048)) && p.out_f8 && !p.scale && (p.act == BS_ACT_NONE || p.act == BS_ACT_RELU) && p.split_off == p.N &&
                p.N % 8 == 0 && p.out_dtype != BS_F32 &&
                (!p.res || (p.res_f8 && p.res_dtype != BS_F32 && p.res_split_off == 0 && p.ldr % 8 == 0)) && (!p.res2 || p.res)) {
                BS_FRESH_LANE
                typedef typename T16<T>::v8 v8;
                typedef int i32x2 __attribute__((ext_vector_type(2)));
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                constexpr int NP = FM * (FN / 2), RDEPTH = 4;
                f32x4 bj[FN];
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int n0 = n_wave + (j >> 1) * 32 + fq * 8 + (j & 1) * 4;
                    bj[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
                    if (g2 >= 0) bj[j] += *reinterpret_cast<const f32x4*>(p.bias2 + (int64_t)g2 * p.N + n0);
                }
                const bool relu = p.act == BS_ACT_RELU;
                const int mrow = m0 + wm * TM + frow;
                const int ea = p.out_f8 & 0xff, el = (p.out_f8 >> 8) & 0xff;
                // (tile-uniform) no lo8 plane past out_lo8_rows: every consumer of this map drops the activation-rounding correction
                const bool lo_plane = !(p.out_lo8_rows > 0 && m0 >= p.out_lo8_rows);
                // ... and no plane at all past out_planes_rows: every consumer runs one 16-bit pass (the calibration's "plain" sites)
                const bool hi_plane = !(p.out_planes_rows > 0 && m0 >= p.out_planes_rows);
                const float lsc = __builtin_ldexpf(1.0f, -BS_F8_ACT_LO_EXP);
                auto body = [&](auto nres_tag) {
                    constexpr int NRES = decltype(nres_tag)::value;
                    v8 rh[NRES ? RDEPTH : 1][NRES ? NRES : 1];
                    i32x2 rl[NRES ? RDEPTH : 1][NRES ? NRES : 1];
                    auto issue = [&](int it, int slot) {
                        const int i = it / (FN / 2), jp = it - i * (FN / 2);
                        int m = mrow + i * 16;
                        m = m < p.M ? m : p.M - 1;
                        const int n = n_wave + jp * 32 + fq * 8;
#pragma unroll
                        for (int q = 0; q < NRES; ++q) {
                            const T* rp = reinterpret_cast<const T*>(q ? p.res2 : p.res) + (int64_t)m * p.ldr;
                            rh[slot][q] = *reinterpret_cast<const v8*>(rp + n);
                            rl[slot][q] = *reinterpret_cast<const i32x2*>(reinterpret_cast<const char*>(rp + p.N + (p.N >> 1)) + n);
                        }
                    };
                    if constexpr (NRES > 0) {
#pragma unroll
                        for (int it = 0; it < RDEPTH; ++it) issue(it, it);
                    }
#pragma unroll
                    for (int it = 0; it < NP; ++it) {
                        const int i = it / (FN / 2), jp = it - i * (FN / 2);
                        const int m = mrow + i * 16;
                        float y8[8];
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float y = acc[i][2 * jp + h][e] + bj[2 * jp + h][e];
                                if (relu) y = fmaxf(y, 0.0f);
                                y8[4 * h + e] = y;
                            }
                        if constexpr (NRES > 0) {
#pragma unroll
                            for (int q = 0; q < NRES; ++q) {
                                const v8 hv = rh[it % RDEPTH][q];
                                const i32x2 lv = rl[it % RDEPTH][q];
#pragma unroll
                                for (int h = 0; h < 2; ++h) {      // (the generic loop's order: a fragment's four hi16 values, then its four lo8 values)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) y8[4 * h + e] += T16<T>::to_f32(hv[4 * h + e]);
                                    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(lv[h], false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lv[h], true);
                                    y8[4 * h] += a[0] * lsc; y8[4 * h + 1] += a[1] * lsc; y8[4 * h + 2] += b[0] * lsc; y8[4 * h + 3] += b[1] * lsc;
                                }
                            }
                            if (it + RDEPTH < NP) issue(it + RDEPTH, it % RDEPTH);
                        }
                        if (m < p.M) store8_f8<T>(p.out, (int64_t)m * p.ldo, n_wave + jp * 32 + fq * 8, p.split_off, y8, ea, el, lo_plane && hi_plane, hi_plane);
                        if (p.out2_relu) {      // (wave-uniform) the ReLU'd copy the next residual unit's first convolution reads
#pragma unroll
                            for (int e = 0; e < 8; ++e) y8[e] = fmaxf(y8[e], 0.0f);
                            if (m < p.M) store8_f8<T>(p.out2, (int64_t)m * p.ldo, n_wave + jp * 32 + fq * 8, p.split_off, y8, ea, el);
                        }
                    }
                };
                if (p.res2) body(std::integral_constant<int, 2>{});
                else if (p.res) body(std::integral_constant<int, 1>{});
                else body(std::integral_constant<int, 0>{});
                return;
            }
            if (plain_full && p.act == BS_ACT_GELU && !p.res && p.out_f8 && !p.scale && p.split_off == p.N && p.N % 8 == 0) {
                // fc1: gelu(acc + bias) as (hi16 | hi8 | lo8) rows; VALU-bound (a third of the launch): no scale multiply, and past
                // out_lo8_rows no lo8 plane
                BS_FRESH_LANE
                const bool lo = !(p.out_lo8_rows > 0 && m0 >= p.out_lo8_rows);
                const bool planes = !(p.out_planes_rows > 0 && m0 >= p.out_planes_rows);
                f32x4 bj[FN];
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int n0 = n_wave + (j >> 1) * 32 + fq * 8 + (j & 1) * 4;
                    bj[j] = *reinterpret_cast<const f32x4*>(p.bias + n0);
                    if (g2 >= 0) bj[j] += *reinterpret_cast<const f32x4*>(p.bias2 + (int64_t)g2 * p.N + n0);
                }
                // hi16-only tiles (the patch tiles when fc2 runs "wmean"): full-line stores through the wave's LDS region as for the Q / K
                // tiles below, but 32 rows at a time, BETWEEN the GELU arithmetic of consecutive row groups (staging the whole tile and
                // storing it in one burst behind the arithmetic loses: 930 vs 920 us -- the arithmetic is what hides the stores).
                if constexpr (TM == 128 && TN == 64 && LDS_BYTES >= WM * WN * 16384) {
                    if (!planes && m0 + BM <= p.M && !(p.ablate & 1024)) {
                        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                        __syncthreads();          // every wave is done reading the main-loop LDS
                        char* sw = smem + wave * 16384;
                        const int s7 = frow & 7;
                        const int wb0 = frow * 128 + ((fq ^ s7) << 4), wb1 = frow * 128 + (((4 + fq) ^ s7) << 4);
                        const int rr = lane >> 3, pc = lane & 7;
                        const int rb = rr * 128 + ((pc ^ rr) << 4);
                        T* op = reinterpret_cast<T*>(p.out) + (int64_t)(m0 + wm * TM + rr) * p.ldo + n_wave + pc * 8;
                        const int64_t step = (int64_t)8 * p.ldo;
#pragma unroll
                        for (int ip = 0; ip < FM; ip += 2) {
                            char* sr = sw + ((ip >> 1) & 1) * 4096;       // two 32-row regions in turn: round k + 1 writes while round k's reads drain
#pragma unroll
                            for (int ii = 0; ii < 2; ++ii) {
                                const int i = ip + ii;
#pragma unroll
                                for (int jp = 0; jp < FN / 2; ++jp) {
                                    typename T16<T>::v8 h8v;
#pragma unroll
                                    for (int h = 0; h < 2; ++h) {
                                        const f32x4 v = acc[i][2 * jp + h] + bj[2 * jp + h];
                                        const f32x2_ g0 = gelu_erf2(f32x2_{v[0], v[1]}), g1 = gelu_erf2(f32x2_{v[2], v[3]});
                                        h8v[4 * h] = T16<T>::from_f32(g0[0]); h8v[4 * h + 1] = T16<T>::from_f32(g0[1]);
                                        h8v[4 * h + 2] = T16<T>::from_f32(g1[0]); h8v[4 * h + 3] = T16<T>::from_f32(g1[1]);
                                    }
                                    *reinterpret_cast<typename T16<T>::v8*>(sr + (jp ? wb1 : wb0) + ii * 2048) = h8v;
                                }
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int it = 0; it < 4; ++it) {
                                *reinterpret_cast<u32x4_*>(op) = *reinterpret_cast<const u32x4_*>(sr + rb + it * 1024);
                                op += step;
                            }
                        }
                        return;
                    }
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int m = m0 + wm * TM + i * 16 + frow;
                    if (m >= p.M) continue;
#pragma unroll
                    for (int jp = 0; jp < FN / 2; ++jp) {
                        float y8[8];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {             // (two values per instruction: gelu_erf2)
                            const f32x4 v = acc[i][2 * jp + h] + bj[2 * jp + h];
                            const f32x2_ g0 = gelu_erf2(f32x2_{v[0], v[1]}), g1 = gelu_erf2(f32x2_{v[2], v[3]});
                            y8[4 * h] = g0[0]; y8[4 * h + 1] = g0[1]; y8[4 * h + 2] = g1[0]; y8[4 * h + 3] = g1[1];
                        }
                        if (planes) {
                            store8_f8<T>(p.out, (int64_t)m * p.ldo, n_wave + jp * 32 + fq * 8, p.split_off, y8, p.out_f8 & 0xff, (p.out_f8 >> 8) & 0xff, lo);
                        } else {
                            typename T16<T>::v8 h;
#pragma unroll
                            for (int e = 0; e < 8; ++e) h[e] = T16<T>::from_f32(y8[e]);
                            *reinterpret_cast<typename T16<T>::v8*>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.ldo + n_wave + jp * 32 + fq * 8) = h;
                        }
                    }
                }
                return;
            }
        }
        // (F8 instantiations only: in the plain ones the generic loop is as fast, measured 705 vs 718 us)
        if (F8 && p.out_mode == BS_OUT_QKV && p.qkv_hidden % BN == 0 && p.act == BS_ACT_NONE && !p.res && p.bias && !p.bias_group_rows && !b2_rows &&
            p.out_dtype != BS_F32 && !(p.ablate & 16)) {
            // Q / K tile of the QKV product (a 256-column tile lies inside one of the three): (acc + bias) * scale as 16-bit rows of
            // [image, head, position, 64].  The generic loop re-derives the part, the column offset and the store shape per fragment
            // pair; at K = 1024 that epilogue was 47 % of the launch (920 -> 816 us).
            BS_FRESH_LANE
            const int part = (tn * BN) / p.qkv_hidden;                      // 0 = Q, 1 = K (V tiles take the transposing path below)
            T* dst = reinterpret_cast<T*>(part == 1 ? p.out2 : p.out);
            const float qs = part == 0 ? p.q_scale : 1.0f;
            f32x4 bj[FN], sj[FN];
            int64_t coff[FN / 2];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n0 = n_wave + (j >> 1) * 32 + fq * 8 + (j & 1) * 4;
                bj[j] = *reinterpret_cast<const f32x4*>(p.bias + n0);
                if (g2 >= 0) bj[j] += *reinterpret_cast<const f32x4*>(p.bias2 + (int64_t)g2 * p.N + n0);
                sj[j] = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n0) : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) sj[j][e] *= qs;
                if (!(j & 1)) {
                    const int rem = n0 - part * p.qkv_hidden;
                    coff[j >> 1] = (int64_t)(rem >> 6) * p.qkv_sp * 64 + (rem & 63);
                }
            }
            const int nhead = p.qkv_hidden >> 6;
            // Patch tiles of the grouped layout (all but the first M-tile of a backbone launch): the rows of a lane are 16 apart, so
            // (image, token) is decoded by ONE division and then advanced; the stores go through a buffer descriptor with 32-bit
            // offsets.  The loop below re-derives both per row with 64-bit addresses: 17 integer divisions and ~1 700 instructions per
            // wave, 6 us of a 46 us tile at two waves per SIMD (profiles/r04_gemm_experiments.txt).  Same values, same bits.
            const long long q_bytes = (long long)p.qkv_cls_rows * nhead * p.qkv_sp * 128;
            if (p.qkv_cls_rows > 0 && m0 >= p.qkv_patch_row0 && p.qkv_lo_off == 0 && !(p.ablate & (128 | 256)) && q_bytes < 0x7FFFFFF0ll) {
                typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)q_bytes, 0x00020000);
                const int Tk = p.qkv_tokens - 1;
                int m = m0 + wm * TM + frow;
                const int mp0 = m - p.qkv_patch_row0;
                const int ob0 = mp0 / Tk;
                int otok = mp0 - ob0 * Tk;
                const unsigned img_bytes = (unsigned)(nhead * p.qkv_sp) * 128u;
                unsigned roff = (unsigned)ob0 * img_bytes + (unsigned)otok * 128u;
                if constexpr (TM == 128 && TN == 64 && LDS_BYTES >= WM * WN * 16384) {
                    // Whole tiles (round 4): the 16-bit rows go through the wave's 16 KiB of the (now idle) main-loop LDS so that an
                    // instruction stores 8 FULL 128-byte lines, consecutive lanes consecutive 16-byte pieces.  In the register layout an
                    // instruction covers 16 rows x 64 bytes with lane = row: 64 separate requests, 3x the time per tile when the stores sit
                    // between compute phases (tools/probes/store_pattern.hip: +4.8 us against +1.6 us per 256 x 256 tile).
                    if (m0 + BM <= p.M && Tk % 8 == 0 && (m0 - p.qkv_patch_row0) % 8 == 0 && !(p.ablate & 1024)) {
                        __syncthreads();          // every wave is done reading the main-loop LDS
                        char* sw = smem + wave * 16384;
                        const int s7 = frow & 7;
                        const int wb0 = frow * 128 + ((fq ^ s7) << 4), wb1 = frow * 128 + (((4 + fq) ^ s7) << 4);
#pragma unroll
                        for (int i = 0; i < FM; ++i) {
#pragma unroll
                            for (int jp = 0; jp < FN / 2; ++jp) {
                                typename T16<T>::v8 v;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    v[e] = T16<T>::from_f32((acc[i][2 * jp][e] + bj[2 * jp][e]) * sj[2 * jp][e]);
                                    v[4 + e] = T16<T>::from_f32((acc[i][2 * jp + 1][e] + bj[2 * jp + 1][e]) * sj[2 * jp + 1][e]);
                                }
                                *reinterpret_cast<typename T16<T>::v8*>(sw + (jp ? wb1 : wb0) + i * 2048) = v;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const int rr = lane >> 3, pc = lane & 7;
                        const int mpl = m0 + wm * TM + rr - p.qkv_patch_row0;
                        const int obl = mpl / Tk;
                        int tokl = mpl - obl * Tk;
                        const int rem0 = n_wave - part * p.qkv_hidden;           // the wave's 64 columns are one head
                        unsigned ro = (unsigned)obl * img_bytes + (unsigned)tokl * 128u + (unsigned)(rem0 >> 6) * (unsigned)p.qkv_sp * 128u + pc * 16u;
                        const int rb = rr * 128 + ((pc ^ rr) << 4);
#pragma unroll
                        for (int it = 0; it < 16; ++it) {
                            const u32x4_ x = *reinterpret_cast<const u32x4_*>(sw + rb + it * 1024);
                            __builtin_amdgcn_raw_buffer_store_b128(x, orsrc, ro, 0, 0);
                            tokl += 8;
                            ro += 8u * 128u;
                            if (tokl >= Tk) {
                                tokl -= Tk;
                                ro += img_bytes - (unsigned)Tk * 128u;
                            }
                        }
                        return;
                    }
                }
                unsigned cb[FN / 2];
#pragma unroll
                for (int jp = 0; jp < FN / 2; ++jp) cb[jp] = (unsigned)coff[jp] * 2u;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    if (m < p.M) {
#pragma unroll
                        for (int jp = 0; jp < FN / 2; ++jp) {
                            typename T16<T>::v8 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[e] = T16<T>::from_f32((acc[i][2 * jp][e] + bj[2 * jp][e]) * sj[2 * jp][e]);
                                v[4 + e] = T16<T>::from_f32((acc[i][2 * jp + 1][e] + bj[2 * jp + 1][e]) * sj[2 * jp + 1][e]);
                            }
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), orsrc, roff + cb[jp], 0, 0);
                        }
                    }
                    m += 16;
                    otok += 16;
                    roff += 16u * 128u;
                    if (otok >= Tk) {          // next image: its block of [head, position, 64] starts img_bytes further on
                        otok -= Tk;
                        roff += img_bytes - (unsigned)Tk * 128u;
                    }
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm * TM + i * 16 + frow;
                if (m >= p.M) continue;
                int ob, otok;
                if (p.qkv_cls_rows > 0) {
                    if (m >= p.qkv_cls_rows && m < p.qkv_patch_row0) continue;
                    const int mp = m - p.qkv_patch_row0;
                    ob = mp < 0 ? m : mp / (p.qkv_tokens - 1);
                    otok = mp < 0 ? p.qkv_tokens - 1 : mp - ob * (p.qkv_tokens - 1);
                } else {
                    ob = m / p.qkv_tokens;
                    otok = m - ob * p.qkv_tokens;
                    if (p.qkv_cls_last) otok = otok == 0 ? p.qkv_tokens - 1 : otok - 1;
                }
                T* rowp = dst + ((int64_t)ob * nhead * p.qkv_sp + otok) * 64;
#pragma unroll
                for (int jp = 0; jp < FN / 2; ++jp) {
                    typename T16<T>::v8 v;
                    float yq[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        yq[e] = (acc[i][2 * jp][e] + bj[2 * jp][e]) * sj[2 * jp][e];
                        yq[4 + e] = (acc[i][2 * jp + 1][e] + bj[2 * jp + 1][e]) * sj[2 * jp + 1][e];
                        v[e] = T16<T>::from_f32(yq[e]);
                        v[4 + e] = T16<T>::from_f32(yq[4 + e]);
                    }
                    if (p.ablate & 128) {                 // diagnostics: everything but the store
                        asm volatile("" ::"v"(v));
                        continue;
                    }
                    *reinterpret_cast<typename T16<T>::v8*>(rowp + coff[jp]) = v;
                    if (p.qkv_lo_off) {       // (wave-uniform) the rounding residuals, for bs_attention_table_corr
                        typename T16<T>::v8 vl;
#pragma unroll
                        for (int e = 0; e < 8; ++e) vl[e] = T16<T>::from_f32(yq[e] - T16<T>::to_f32(v[e]));
                        *reinterpret_cast<typename T16<T>::v8*>(rowp + coff[jp] + p.qkv_lo_off) = vl;
                    }
                }
            }
            return;
        }
        if (p.act == BS_ACT_GELU) epi(std::integral_constant<int, BS_ACT_GELU>{});
        else if (p.act == BS_ACT_RELU) epi(std::integral_constant<int, BS_ACT_RELU>{});
        else if (p.act == BS_ACT_SOFTPLUS) epi(std::integral_constant<int, BS_ACT_SOFTPLUS>{});
        else if (p.act == BS_ACT_SOFTPLUS_FAST) epi(std::integral_constant<int, BS_ACT_SOFTPLUS_FAST>{});
        else epi(std::integral_constant<int, BS_ACT_NONE>{});
    } else {
        // V part: per-wave LDS transpose, then consecutive lanes store consecutive tokens of one (head, d) row
        constexpr int NW = WM * WN;
        constexpr int S4 = TN / 4;
        constexpr int MAXR = LDS_BYTES / (NW * TN * 4);
        constexpr int PASS_R = MAXR >= TM ? TM : (MAXR >= TM / 2 ? TM / 2 : (MAXR >= TM / 4 ? TM / 4 : TM / 8));
        static_assert(PASS_R >= 16 && PASS_R % 16 == 0 && NW * PASS_R * TN * 4 <= LDS_BYTES, "epilogue staging does not fit the main-loop LDS");
        constexpr int PASSES = TM / PASS_R;
        float* sc = reinterpret_cast<float*>(smem) + wave * (PASS_R * TN);
        T* vt = reinterpret_cast<T*>(p.out3);
        const int nh = p.qkv_hidden >> 6;
        __syncthreads();  // every wave is done reading the main-loop LDS
        if constexpr (PASS_R == 64 && TN == 64) {
            // Patch tiles of the grouped layout (round 4): a 64-row pass is 64 consecutive tokens of ONE image and the wave's 64 columns
            // are ONE head, so V^T gets, per d, 128 contiguous bytes.  fp32 rows to LDS, then a lane takes 8 consecutive tokens of one d: eight ds_read_b32, one 16-byte store -- an
            // instruction writes 8 full 128-byte lines.  (The general form below loads its bias per column inside the loop and stores
            // 2 bytes per lane: 128 serialised load -> read -> store round trips per wave, 3x the time of a Q / K tile; PMC study,
            // profiles/r04_gemm_experiments.txt (6).)  LDS swizzle f(r): 16-byte group g of row r sits at g ^ f(r); conflict-free for the
            // b128 writes (16 rows x one group) and for the b32 reads (8 token octets x 8 d).
            const int tpi = p.qkv_tokens - 1;
            const bool fastv = p.qkv_cls_rows > 0 && m0 >= p.qkv_patch_row0 && m0 + BM <= p.M && !b2_rows && !(p.qkv_sp & 7) && tpi % 64 == 0 &&
                               (m0 - p.qkv_patch_row0) % 64 == 0 && !(p.ablate & (128 | 512));
            if (fastv) {
                BS_FRESH_LANE
                const int tg = lane & 7, dl = lane >> 3;
                float bd[8];        // bias of this lane's eight d (d = 8 it + dl)
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int n = n_wave + it * 8 + dl;
                    bd[it] = p.bias ? p.bias[n] : 0.f;
                    if (g2 >= 0) bd[it] += p.bias2[(int64_t)g2 * p.N + n];
                }
                // Byte address of (row r, 16-byte group g) in the wave's region: r * 256 + ((g ^ f(r)) << 4), f(r) = 2 (r >> 3) ^ h(r & 7),
                // h(k) = {0, 1, 4, 5, 8, 9, 12, 13}[k].  Row and group bits do not overlap, so the swizzle is ONE xor of a lane base with a
                // compile-time constant (the row's multiple of 16 or 8 and the fragment / column indices are unrolled).
                auto hk = [](int k) { return ((k & 6) << 1) | (k & 1); };
                char* scb = reinterpret_cast<char*>(sc);
                // write side: r = 16 i' + frow, g = 8 (j >> 1) + 2 fq + (j & 1)
                const int wbase = frow * 256 + (((fq << 1) ^ ((frow >> 3) << 1) ^ hk(frow & 7)) << 4);
                // read side: r = 8 tg + k, g = 2 it + (dl >> 2), word dl & 3
                const int rbase = tg * 2048 + ((((dl >> 2) ^ (tg << 1)) << 4) | ((dl & 3) << 2));
                const int head = (n_wave - 2 * p.qkv_hidden) >> 6;
                for (int ps = 0; ps < PASSES; ++ps) {
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        if ((i * 16) / PASS_R == ps) {
                            const int i4 = i & (PASS_R / 16 - 1);
#pragma unroll
                            for (int j = 0; j < FN; ++j) {
                                const int c = ((j >> 1) * 8 + (j & 1)) ^ (((i4 * 2) & 7) << 1);
                                *reinterpret_cast<f32x4*>(scb + (wbase ^ (c << 4)) + i4 * 16 * 256) = acc[i][j];
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int mp = m0 + wm * TM + ps * PASS_R - p.qkv_patch_row0;
                    const int ob = mp / tpi, otok = mp - ob * tpi + tg * 8;
                    T* vrow = vt + (((int64_t)ob * nh + head) * 64) * p.qkv_sp + otok;
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int d = it * 8 + dl;
                        float y[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            y[k] = *reinterpret_cast<const float*>(scb + (rbase ^ (((it * 2) ^ hk(k)) << 4)) + k * 256) + bd[it];
                        typename T16<T>::v8 vh;
#pragma unroll
                        for (int k = 0; k < 8; ++k) vh[k] = T16<T>::from_f32(y[k]);
                        *reinterpret_cast<typename T16<T>::v8*>(vrow + (int64_t)d * p.qkv_sp) = vh;
                        if (p.qkv_lo_off) {
                            typename T16<T>::v8 vl;
#pragma unroll
                            for (int k = 0; k < 8; ++k) vl[k] = T16<T>::from_f32(y[k] - T16<T>::to_f32(vh[k]));
                            *reinterpret_cast<typename T16<T>::v8*>(vrow + (int64_t)d * p.qkv_sp + p.qkv_lo_off) = vl;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();   // the next pass overwrites this wave's region
                }
                return;
            }
        }
        BS_FRESH_LANE
        for (int ps = 0; ps < PASSES; ++ps) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                if ((i * 16) / PASS_R == ps) {
                    const int r = i * 16 - ps * PASS_R + frow;
#pragma unroll
                    for (int j = 0; j < FN; ++j)   // group-of-4 index of this lane's columns in the wave tile (permuted columns, see w_off)
                        *reinterpret_cast<f32x4*>(sc + r * TN + ((((j >> 1) * 8 + fq * 2 + (j & 1)) ^ (r & (S4 - 1))) << 2)) = acc[i][j];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int m_base = m0 + wm * TM + ps * PASS_R;
            // lane -> row r (token), loop over the TN columns; PASS_R is 32 or 64 rows: 1 or 2 columns per sweep
            constexpr int CPS = 64 / PASS_R >= 1 ? 64 / PASS_R : 1;     // columns per 64-lane sweep
            const int r = lane % PASS_R, csub = lane / PASS_R;
            const int m = m_base + r;
            int ob, otok;
            bool pad_row = false;
            if (p.qkv_cls_rows > 0) {
                pad_row = m >= p.qkv_cls_rows && m < p.qkv_patch_row0;
                const int mp = m - p.qkv_patch_row0;
                ob = mp < 0 ? m : mp / (p.qkv_tokens - 1);
                otok = mp < 0 ? p.qkv_tokens - 1 : mp - ob * (p.qkv_tokens - 1);
            } else {
                ob = m / p.qkv_tokens;
                otok = m - ob * p.qkv_tokens;
                if (p.qkv_cls_last) otok = otok == 0 ? p.qkv_tokens - 1 : otok - 1;
            }
            const float* b2 = (p.bias2 && m < p.M && m >= p.bias2_row0) ? p.bias2 + (int64_t)((m - p.bias2_row0) / p.bias2_group_rows) * p.N : nullptr;
            if (m < p.M && !pad_row) {
                for (int c0 = 0; c0 < TN; c0 += CPS) {
                    const int c = c0 + csub;
                    const int n = n_wave + c;
                    if (n >= p.N) continue;
                    float y = sc[r * TN + ((((c >> 2) ^ (r & (S4 - 1))) << 2) | (c & 3))];
                    // acc + (bias + bias2), the association of every other form: a row's bits must not depend on which form its tile takes
                    float bsum = p.bias ? p.bias[n] : 0.0f;
                    if (b2) bsum += b2[n];
                    y += bsum;
                    const int rem = n - 2 * p.qkv_hidden;
                    const int64_t vo = (((int64_t)ob * nh + (rem >> 6)) * 64 + (rem & 63)) * p.qkv_sp + otok;
                    const T vh = T16<T>::from_f32(y);
                    if (p.ablate & 128) {                 // diagnostics: everything but the store
                        asm volatile("" ::"v"(y));
                        continue;
                    }
                    vt[vo] = vh;
                    if (p.qkv_lo_off) vt[vo + p.qkv_lo_off] = T16<T>::from_f32(y - T16<T>::to_f32(vh));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // the next pass overwrites this wave's region
        }
    }
    }
#undef BS_FRESH_LANE
#endif
}

template <typename T, int BM, int BN, int WM, int WN, int BK, int STAGES, int MODE, bool PP = false, int CM = 0>
inline int launch_mode(const IgemmParams& p, hipStream_t st) {
    constexpr int smem = STAGES * (BM + BN) * BK * 2;
    dim3 grid(p.ntm * p.ntn), block(WM * WN * 64);
    auto k = igemm_kernel<T, BM, BN, WM, WN, BK, STAGES, MODE, PP, CM>;
    BS_MAX_DYNAMIC_LDS(reinterpret_cast<const void*>(k), smem);
    hipLaunchKernelGGL(k, grid, block, smem, st, p);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

// One (tile, operand type, correction mode) per translation unit (igemm_tile*_cm*.hip): they build in parallel.
template <typename T, int BM, int BN, int WM, int WN, int BK, int STAGES, bool PP, int CM>
inline int launch_cm(const IgemmParams& p, bool conv, hipStream_t st) {
    if constexpr (CM == 0) {
        if (!conv) return launch_mode<T, BM, BN, WM, WN, BK, STAGES, 0, PP, 0>(p, st);
        if (p.relu_a) return launch_mode<T, BM, BN, WM, WN, BK, STAGES, 2, PP, 0>(p, st);
        return launch_mode<T, BM, BN, WM, WN, BK, STAGES, 1, PP, 0>(p, st);
    } else {
        static_assert(BK == 64 && !PP, "the correction instantiations are built for the BK = 64 ring tiles");
        if (!conv) return launch_mode<T, BM, BN, WM, WN, BK, STAGES, 0, PP, CM>(p, st);
        return launch_mode<T, BM, BN, WM, WN, BK, STAGES, 1, PP, CM>(p, st);
    }
}

}  // namespace bs
