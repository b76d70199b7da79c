// BEiT self-attention with additive relative-position bias, flash-style on MFMA (head_dim 64).
//   HF modeling_beit.py:268-341 (eager_attention_forward) + :179-265 (relative position bias)
//
// Layouts (written by the QKV projection's BS_OUT_QKV epilogue): Q [B,nh,Sp,64] pre-scaled by
// log2(e)/sqrt(64), K [B,nh,Sp,64], V^T [B,nh,64,Sp]; rows / columns >= S are zero.  bias fp32
// [nh,Sp,Sp], pre-multiplied by log2(e), with -1e30 in key columns >= S (no in-kernel masking).
//
// One wave owns 32 queries; the scores are computed TRANSPOSED, S^T = K Q^T, with
// v_mfma_f32_32x32x16: the query sits on the lane (column), the 32 keys of a sub-tile sit in the 16
// accumulator registers of the two lane halves.  Row statistics are then lane-local (one
// cross-half exchange per sub-tile for the max), and the exponentiated accumulator registers ARE
// the B operand of the second product O^T += V^T P^T -- no LDS round trip for P.  The MFMA row ->
// key assignment is permuted (bits 2 and 3 of the row swapped) so that each lane's 8 P values of a
// k-step are 8 CONSECUTIVE keys, i.e. one 16-byte read of the V^T image.
// K and V^T tiles (64 keys) are staged by global_load_lds into double-buffered, XOR-swizzled LDS
// images shared by the QW waves of a block.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace bs {
// Maximum of a float over the two lane halves (lane l and lane l ^ 32), the same value in both: one v_permlane32_swap on the
// order-preserving integer image of the float.  (An fmaxf of the two halves of the swap is FOLDED AWAY by hipcc / ROCm 7.2 -- the
// ISA compared sw[0] alone, so every lane took the lower half's value: consistent between the halves, which kept the softmax
// right, but a large score among the upper half's keys never moved the running max.  The integer maximum is emitted as written.)
__device__ __forceinline__ float half_swap_max(float v) {
    int k = __builtin_bit_cast(int, v);
    k ^= (k >> 31) & 0x7fffffff;
    const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)k, (unsigned)k, false, false);
    const int a = (int)sw[0], b = (int)sw[1];
    int m = a > b ? a : b;
    m ^= (m >> 31) & 0x7fffffff;
    return __builtin_bit_cast(float, m);
}

// Barrier of the K / V^T ring.  A tile is written to LDS by `global_load_lds` pieces issued by ALL waves of the block and read by
// all of them, so the order is: every wave waits for ITS OWN pieces (vmcnt counts LDS-DMA), then the workgroup barrier, then the
// reads.  `__syncthreads()` alone is NOT that: inside the tile loops hipcc (ROCm 7.2) compiles it to `s_waitcnt lgkmcnt(0);
// s_barrier` -- no vmcnt -- although an LDS-DMA is in flight (ISA of attention_tab2_kernel, round 4: the only vmcnt(0) of the loop
// belonged to a spill reload behind the stage; the register-leaner rewrites of round 3 lost that accident and read tiles that
// had not landed -- the "nondeterminism at NB = 128").  The wait is written out, as the LDS-DMA rules demand (counted vmcnt, then
// the barrier, then the ds_read).
__device__ __forceinline__ void dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}


template <typename T, int QW>
__global__ __launch_bounds__(QW * 64) void attention_kernel(const T* __restrict__ Q, const T* __restrict__ K, const T* __restrict__ Vt,
                                                             const float* __restrict__ bias, T* __restrict__ out, int split, int B, int nh, int S,
                                                             int Sp, int nqt, int nqb, int ablate) {
    typedef typename T16<T>::v8 v8;
    constexpr int STAGE = 16 * 1024;  // K tile 8 KiB + V^T tile 8 KiB
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // work id -> (head, q-block, image): all images of one (head, q-block) are consecutive ids and the
    // chunked XCD remap keeps them on one XCD, so its L2 serves the shared bias rows
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + loc;
    const int b = wg % B;
    const int qblk = (wg / B) % nqb;
    const int head = wg / (B * nqb);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h2 = lane >> 5;
    int qt = qblk * QW + wave;
    const bool active = qt < nqt;
    qt = active ? qt : nqt - 1;
    const int q0 = qt * 32;
    const int64_t bh = (int64_t)b * nh + head;

    const T* Qg = Q + bh * Sp * 64;
    const T* Kg = K + bh * Sp * 64;
    const T* Vg = Vt + bh * 64 * Sp;

    v8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const v8*>(Qg + (int64_t)(q0 + r) * 64 + ks * 16 + h2 * 8);

    // LDS images: [64 rows][64 x 16-bit], 16-byte chunk c of row r at chunk position c ^ ((r >> 1) & 7).  The 32x32x16
    // operand read has every ds_read_b128 lane group on 16 rows {0-3,12-15,20-27} / {4-11,16-19,28-31} at ONE chunk: with
    // the pair (row parity, (r >> 1) & 7) those 16 rows land on 16 different 16-byte slots (r & 7 alone collides 2-way).
    const int srow = lane >> 3;                       // row inside one 1-KiB DMA piece (8 rows x 128 bytes)
    auto stage = [&](int kt, int buf) {
        char* sb = smem + buf * STAGE;
        for (int i = wave; i < 16; i += QW) {
            const int row = (i & 7) * 8 + srow;       // row of the K (i < 8) or V^T (i >= 8) image
            const int cs8 = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
            const T* src;
            if (i < 8) {
                src = Kg + (int64_t)(kt * 64 + row) * 64 + cs8;
            } else {
                src = Vg + (int64_t)row * Sp + kt * 64 + cs8;
            }
            glds16(src, sb + i * 1024);
        }
    };

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        oacc[0][i] = 0.f;
        oacc[1][i] = 0.f;
    }
    float m_run = -1.0e30f, l_run = 0.f;

    // MFMA row rho -> key kappa(rho): swap bits 2 and 3
    const int kap = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
    const float* bias_row = bias + ((int64_t)head * Sp + q0 + r) * Sp + 8 * h2;

    const int nkt = (S + 63) >> 6;
    // Scores live in the log2 domain (Q and the bias carry log2(e)): the exponential is one v_exp_f32.
    // VALU diet (the loop is VALU-bound, not MFMA-bound): the accumulator of S^T = K Q^T is INITIALISED with
    // (bias - m_run), so the MFMA chain delivers s + bias - m_run and p = exp2(acc) needs no add and no subtract; the
    // running max is only moved when a sub-tile exceeds it by more than 2^THR (deferred max: p <= 2^THR stays exact in
    // fp32 and well inside fp16 for the P operand).  The bias rows of the NEXT sub-tile are fetched one sub-tile ahead.
    constexpr float THR = 6.0f;
    f32x4 bnext[4];
    auto load_bias = [&](int key0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bnext[2 * s] = *reinterpret_cast<const f32x4*>(bias_row + key0 + 16 * s);
            bnext[2 * s + 1] = *reinterpret_cast<const f32x4*>(bias_row + key0 + 16 * s + 4);
        }
    };
    m_run = 0.f;
    bool first = true;
    load_bias(0);
    stage(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        dma_barrier();
        if (kt + 1 < nkt && !(ablate & 2)) stage(kt + 1, (kt + 1) & 1);
        const char* sk = smem + (kt & 1) * STAGE;
        const char* sv = sk + 8 * 1024;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int key0 = kt * 64 + sub * 32;
            if (key0 >= S) break;
            // registers 8s..8s+7 of lane half h2 are keys key0 + 16s + 8*h2 + 0..7
            f32x16 sacc;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sacc[8 * s + e] = bnext[2 * s][e] - m_run;
                    sacc[8 * s + 4 + e] = bnext[2 * s + 1][e] - m_run;
                }
            if (key0 + 32 < Sp && !(ablate & 1)) load_bias(key0 + 32);
            // ---- S^T = K Q^T (+ bias - m_run)
            const int krow = sub * 32 + kap;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int chunk = 2 * ks + h2;
                const v8 kf = *reinterpret_cast<const v8*>(sk + krow * 128 + ((chunk ^ ((krow >> 1) & 7)) << 4));
                sacc = T16<T>::mfma32(kf, qf[ks], sacc);
            }
            float mloc = fmaxf(fmaxf(sacc[0], sacc[1]), sacc[2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) mloc = fmaxf(fmaxf(mloc, sacc[i]), sacc[i + 1]);
            mloc = fmaxf(mloc, sacc[15]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            if (first || __any(mloc > THR)) {   // wave-uniform: move the running max (always on the first sub-tile)
                // the branch is wave-uniform but the shift is the lane's own: a query whose tile maximum lies BELOW its running max keeps it
                // (shift 0) -- exp2(-mloc) of a maximum 128 below the running one overflowed to inf (a key that out-scores the rest by
                // 2^128 earlier in the sequence; found by the spiked-key test).  On the first sub-tile the accumulators are zero: no scaling.
                const float sh = first ? mloc : fmaxf(mloc, 0.0f);
                const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-sh);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    oacc[0][i] *= alpha;
                    oacc[1][i] *= alpha;
                    sacc[i] -= sh;
                }
                m_run += sh;
                first = false;
            }
            float p[16];
            float psum = 0.f;
            if (ablate & 4) {
#pragma unroll
                for (int i = 0; i < 16; ++i) p[i] = sacc[i];
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    p[i] = __builtin_amdgcn_exp2f(sacc[i]);
                    psum += p[i];
                }
            }
            l_run += psum;
            // ---- O^T += V^T P^T
            v8 pf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[s][e] = T16<T>::from_f32(p[8 * s + e]);
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {
                const int drow = dh * 32 + r;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int chunk = sub * 4 + 2 * s + h2;
                    const v8 vf = *reinterpret_cast<const v8*>(sv + drow * 128 + ((chunk ^ ((drow >> 1) & 7)) << 4));
                    oacc[dh] = T16<T>::mfma32(vf, pf[s], oacc[dh]);
                }
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + r;
    if (active && q < S) {
        // split: rows of (hi | lo) pairs [., 2*nh*64], o = hi + lo to ~22 bits (operand of a split-precision o_proj)
        T* orow = out + ((int64_t)b * S + q) * (nh * 64) * (split ? 2 : 1) + head * 64;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                typename T16<T>::v4 o, ol;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = oacc[dh][gg * 4 + e] * inv;
                    o[e] = T16<T>::from_f32(y);
                    ol[e] = T16<T>::from_f32(y - T16<T>::to_f32(o[e]));
                }
                *reinterpret_cast<typename T16<T>::v4*>(orow + dh * 32 + 8 * gg + 4 * h2) = o;
                if (split == 2) {      // (hi16 | hi8 | lo8) planes of a 4*nh*64-byte row
                    const int col = head * 64 + dh * 32 + 8 * gg + 4 * h2;
                    char* planes = reinterpret_cast<char*>(orow - head * 64 + nh * 64);
                    const float sh = __builtin_ldexpf(1.0f, F8_ACT_HI_EXP), sl = __builtin_ldexpf(1.0f, F8_ACT_LO_EXP);
                    float yv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) yv[e] = oacc[dh][gg * 4 + e] * inv;
                    *reinterpret_cast<int*>(planes + col) = f8_pack4(yv[0] * sh, yv[1] * sh, yv[2] * sh, yv[3] * sh);
                    *reinterpret_cast<int*>(planes + nh * 64 + col) =
                        f8_pack4((yv[0] - T16<T>::to_f32(o[0])) * sl, (yv[1] - T16<T>::to_f32(o[1])) * sl, (yv[2] - T16<T>::to_f32(o[2])) * sl,
                                 (yv[3] - T16<T>::to_f32(o[3])) * sl);
                } else if (split) {
                    *reinterpret_cast<typename T16<T>::v4*>(orow + nh * 64 + dh * 32 + 8 * gg + 4 * h2) = ol;
                }
            }
    }
}

// Epilogue shared by the table kernels: a wave's 32 x 64 output tile (lane = query r + 32*h2, 8 groups of 4 consecutive d) goes
// through an 8 KiB region of LDS so that every global store instruction covers whole rows -- the row-per-lane stores it
// replaces touched 64 different 128-byte lines per instruction (8 + 4 + 4 bytes per lane and line) and cost the accurate
// mode's (hi16 | hi8 | lo8) output 200 us per launch at NB = 128, all of it address-coalescer time.
//   LDS image per wave: hi16 [32][64] 16-bit (128-byte rows, 16-byte chunk c of row r at c ^ (r & 7)), then the two FP8
//   planes [32][64] bytes (64-byte rows, chunk c at c ^ ((r >> 1) & 3)).
template <typename T>
__device__ __forceinline__ void store_out_tile(char* wlds, const f32x16 (&oacc)[2], float inv, T* __restrict__ out, int split, int nh, int head,
                                               int64_t row0, int64_t row_cls, int S, int q0, bool active, int lane, bool cls_planes = false) {
    typedef typename T16<T>::v8 v8;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const int r = lane & 31, h2 = lane >> 5;
    char* l16 = wlds;
    char* lh8 = wlds + 4096;
    char* ll8 = wlds + 6144;
    char* llo = wlds + 4096;     // split == 1: the 16-bit lo plane, same image as hi16
    // cls_planes: only the cls query's row needs its FP8 planes (the consumer runs no FP8 stage on patch rows): tiles without
    // that row skip the packing and the plane stores (wave-uniform)
    const bool planes = split == 2 && (!cls_planes || (q0 <= S - 1 && S - 1 < q0 + 32));
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            typename T16<T>::v4 o, ol;
            float yv[4], rl[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                yv[e] = oacc[dh][gg * 4 + e] * inv;
                o[e] = T16<T>::from_f32(yv[e]);
                rl[e] = yv[e] - T16<T>::to_f32(o[e]);
                ol[e] = T16<T>::from_f32(rl[e]);
            }
            const int d0 = dh * 32 + 8 * gg + 4 * h2;                 // first of this lane's 4 consecutive d
            const int c16 = d0 >> 3, w16 = (d0 & 7) * 2;              // 16-byte chunk of the 128-byte hi16 row, byte inside it
            *reinterpret_cast<typename T16<T>::v4*>(l16 + r * 128 + ((c16 ^ (r & 7)) << 4) + w16) = o;
            if (planes) {
                const float sh = __builtin_ldexpf(1.0f, F8_ACT_HI_EXP), sl = __builtin_ldexpf(1.0f, F8_ACT_LO_EXP);
                const int c8 = d0 >> 4, w8 = d0 & 15;                  // 16-byte chunk of the 64-byte plane row
                const int pos = r * 64 + ((c8 ^ ((r >> 1) & 3)) << 4) + w8;
                *reinterpret_cast<int*>(lh8 + pos) = f8_pack4(yv[0] * sh, yv[1] * sh, yv[2] * sh, yv[3] * sh);
                *reinterpret_cast<int*>(ll8 + pos) = f8_pack4(rl[0] * sl, rl[1] * sl, rl[2] * sl, rl[3] * sl);
            } else if (split == 1) {
                *reinterpret_cast<typename T16<T>::v4*>(llo + r * 128 + ((c16 ^ (r & 7)) << 4) + w16) = ol;
            }
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (!active) return;
    const int W = nh * 64 * (split ? 2 : 1);                          // row pitch of `out` in 16-bit elements
    auto out_row = [&](int rr) -> T* {                                // query position -> row of `out` (nullptr: padding query)
        const int qpos = q0 + rr;
        if (qpos >= S) return nullptr;
        const int64_t row = qpos == S - 1 ? row_cls : row0 + qpos;
        return out + row * W;
    };
    // hi16 (and the 16-bit lo plane): 8 lanes cover one 128-byte row, a wave-instruction 8 rows
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int rr = ps * 8 + (lane >> 3), c = lane & 7;
        T* orow = out_row(rr);
        if (orow) {
            *reinterpret_cast<v8*>(orow + head * 64 + c * 8) = *reinterpret_cast<const v8*>(l16 + rr * 128 + ((c ^ (rr & 7)) << 4));
            if (split == 1) *reinterpret_cast<v8*>(orow + nh * 64 + head * 64 + c * 8) = *reinterpret_cast<const v8*>(llo + rr * 128 + ((c ^ (rr & 7)) << 4));
        }
    }
    if (planes) {   // FP8 planes: 4 lanes cover one 64-byte row, a wave-instruction 16 rows
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int rr = ps * 16 + (lane >> 2), c = lane & 3;
            T* orow = out_row(rr);
            if (orow && (!cls_planes || q0 + rr == S - 1)) {
                char* planes = reinterpret_cast<char*>(orow + nh * 64);
                const int pos = rr * 64 + ((c ^ ((rr >> 1) & 3)) << 4);
                *reinterpret_cast<i32x4*>(planes + head * 64 + c * 16) = *reinterpret_cast<const i32x4*>(lh8 + pos);
                *reinterpret_cast<i32x4*>(planes + nh * 64 + head * 64 + c * 16) = *reinterpret_cast<const i32x4*>(ll8 + pos);
            }
        }
    }
}

// the head's bias table -> LDS: by LDS-DMA when its byte size and offset keep 16-byte alignment (even hp), else through registers.
// The last 1-KiB piece may run past the table: its lanes re-read the table's last 16 bytes (the words they write lie past ntab).
template <int NT_>
__device__ __forceinline__ void load_table(float* tab, const float* __restrict__ table, int head, int ntab, int tid, int wave, int lane) {
    const float* src = table + (int64_t)head * ntab;
    if ((ntab & 3) == 0) {
        const int bytes = ntab * 4, pieces = (bytes + 1023) >> 10;
        for (int pc = wave; pc < pieces; pc += NT_ / 64) {
            int off = pc * 1024 + lane * 16;
            off = off > bytes - 16 ? bytes - 16 : off;
            glds16(reinterpret_cast<const char*>(src) + off, reinterpret_cast<char*>(tab) + pc * 1024);
        }
    } else {
        for (int i = tid; i < ntab; i += NT_) tab[i] = src[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Table variant (the full-size networks: window hp x 32 patches).  The additive bias of (query, key) depends only on the
// offset (qy - ky, qx - kx) of the two patches -- HF gathers a [S, S] matrix per head from a ((2hp-1)(2wp-1) + 3)-entry
// table (modeling_beit.py:194-265).  Here the per-head table (12 KiB fp32, pre-multiplied by log2 e) sits in LDS and a lane
// reads its 16 values of a sub-tile straight from it: no [nh, Sp, Sp] tensor exists and nothing but the K / V^T tiles is
// fetched inside the loop (the materialised-bias kernel above streamed 4 KiB of bias per wave and 32-key sub-tile -- 2.5x the
// K / V^T bytes -- through VGPR-destination loads whose waits also drained the tile prefetch).
// Token order inside Q / K / V^T is patches first, cls LAST (written so by the QKV epilogue, bs_gemm_desc.qkv_cls_last):
// a 32-query tile is then one patch row (qy = tile index, qx = lane) and a 32-key sub-tile one key row (ky, kx = 0..31), and
//   bias = body[(qy - ky + hp - 1) * 63 + (qx - kx + 31)]
// is a wave-uniform row base plus a per-lane offset plus an immediate: conflict-free ds_read_b32 (32 consecutive words per
// lane half).  The table operand holds the body REVERSED (see bs_attention_table), so consecutive keys are ascending words.  The cls key is the single valid key of the last sub-tile (tab[nrd-2]; tab[nrd-1] for the cls query), the cls
// query tile takes tab[nrd-3] for every patch key.  Output rows are written back in the residual stream's order (cls first).
template <typename T, int QW>
__global__ __launch_bounds__(QW * 64) void attention_tab_kernel(const T* __restrict__ Q, const T* __restrict__ K, const T* __restrict__ Vt,
                                                                 const float* __restrict__ table, T* __restrict__ out, int split, int B, int nh,
                                                                 int hp, int Sp, int nqb, int ntab, int grouped, int ablate) {
    BS_ARG_NOW(out);
    const bool cls_planes = (split & 4) != 0;   // FP8 planes for the cls query's row only (bs_attention_table, dtype bit 6)
    split &= 3;
    typedef typename T16<T>::v8 v8;
    constexpr int STAGE = 16 * 1024;  // K tile 8 KiB + V^T tile 8 KiB
    constexpr int WP = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem + 2 * STAGE);
    const int S = hp * WP + 1, nqt = hp + 1;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + loc;
    // the q-blocks of one (image, head) are consecutive work ids: they stream the same K / V^T tiles at about the same time,
    // from the same XCD's L2
    const int qblk = wg % nqb;
    const int b = (wg / nqb) % B;
    const int head = wg / (B * nqb);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h2 = lane >> 5;
    int qt = qblk * QW + wave;
    const bool active = qt < nqt;
    qt = active ? qt : nqt - 1;
    const int q0 = qt * 32;
    const bool cls_tile = qt == hp;           // wave-uniform: the tile whose only valid query is the cls token
    const int64_t bh = (int64_t)b * nh + head;

    const T* Qg = Q + bh * Sp * 64;
    const T* Kg = K + bh * Sp * 64;
    const T* Vg = Vt + bh * 64 * Sp;

    // the head's table -> LDS (visible after the first barrier of the loop)
    if (!(ablate & 64)) load_table<QW * 64>(tab, table, head, ntab, tid, wave, lane);

    v8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const v8*>(Qg + (int64_t)(q0 + r) * 64 + ks * 16 + h2 * 8);

    const int srow = lane >> 3;
    auto stage = [&](int kt, int buf) {
        char* sb = smem + buf * STAGE;
        for (int i = wave; i < 16; i += QW) {
            const int row = (i & 7) * 8 + srow;
            const int cs8 = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
            const T* src;
            if (i < 8) {
                src = Kg + (int64_t)(kt * 64 + row) * 64 + cs8;
            } else {
                src = Vg + (int64_t)row * Sp + kt * 64 + cs8;
            }
            glds16(src, sb + i * 1024);
        }
    };

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        oacc[0][i] = 0.f;
        oacc[1][i] = 0.f;
    }
    float m_run = 0.f, l_run = 0.f;
    const int kap = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
    const int nkt = (S + 63) >> 6;
    constexpr float THR = 6.0f;

    // registers 8s..8s+7 of lane half h2 are keys key0 + 16s + 8*h2 + e -> kx = 16s + 8*h2 + e.  The table body arrives REVERSED
    // (entry i at word nbody - 1 - i), so the 8 keys of a register group are 8 ascending words:
    //   word = [nbody - 63 - (qt + hp - 1) * 63] + ky * 63 + (31 - r + 8*h2) + 16s + e
    const int nbody = ntab - 3;
    const float* lane_tab = tab + (nbody - (2 * WP - 1) - (qt + hp - 1) * (2 * WP - 1)) + (31 - r + 8 * h2);
    const float NEG = -1.0e30f;
    float bnext[16];
    auto load_bias = [&](int ky) {      // the 16 bias values of the sub-tile holding key row ky (ky == hp: the cls key + padding)
        if (ky < hp) {
            if (cls_tile) {
                const float c3 = tab[ntab - 3];
#pragma unroll
                for (int i = 0; i < 16; ++i) bnext[i] = c3;
            } else {
                const float* rowp = lane_tab + ky * (2 * WP - 1);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bnext[8 * s2 + e] = rowp[16 * s2 + e];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) bnext[i] = NEG;
            if (h2 == 0) bnext[0] = cls_tile ? tab[ntab - 1] : tab[ntab - 2];
        }
    };

    bool first = true;
    stage(0, 0);
    dma_barrier();                    // tile 0 and the table have landed
    if (1 < nkt) stage(1, 1);
    load_bias(0);
    for (int kt = 0; kt < nkt; ++kt) {
        if (ablate & 32) break;
        if (kt > 0) {
            dma_barrier();
            if (kt + 1 < nkt && !(ablate & 16)) stage(kt + 1, (kt + 1) & 1);
        }
        const char* sk = smem + (kt & 1) * STAGE;
        const char* sv = sk + 8 * 1024;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int ky = kt * 2 + sub;
            if (ky > hp) break;
            f32x16 sacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[i] = bnext[i] - m_run;
            if (ky + 1 <= hp && !(ablate & 8)) load_bias(ky + 1);
            const int krow = sub * 32 + kap;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ablate & 4) break;
                const int chunk = 2 * ks + h2;
                const v8 kf = *reinterpret_cast<const v8*>(sk + krow * 128 + ((chunk ^ ((krow >> 1) & 7)) << 4));
                sacc = T16<T>::mfma32(kf, qf[ks], sacc);
            }
            float mloc = fmaxf(fmaxf(sacc[0], sacc[1]), sacc[2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) mloc = fmaxf(fmaxf(mloc, sacc[i]), sacc[i + 1]);
            mloc = fmaxf(mloc, sacc[15]);
            mloc = half_swap_max(mloc);      // the other lane half holds the other 16 keys of this query
            if (first || __any(mloc > THR)) {
                // the branch is wave-uniform but the shift is the lane's own: a query whose tile maximum lies BELOW its running max keeps it
                // (shift 0) -- exp2(-mloc) of a maximum 128 below the running one overflowed to inf (a key that out-scores the rest by
                // 2^128 earlier in the sequence; found by the spiked-key test).  On the first sub-tile the accumulators are zero: no scaling.
                const float sh = first ? mloc : fmaxf(mloc, 0.0f);
                const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-sh);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    oacc[0][i] *= alpha;
                    oacc[1][i] *= alpha;
                    sacc[i] -= sh;
                }
                m_run += sh;
                first = false;
            }
            float p[16];
            float psum = 0.f;
            if (ablate & 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) p[i] = sacc[i];
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    p[i] = __builtin_amdgcn_exp2f(sacc[i]);
                    psum += p[i];
                }
            }
            l_run += psum;
            v8 pf[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[s2][e] = T16<T>::from_f32(p[8 * s2 + e]);
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {
                if (ablate & 2) break;
                const int drow = dh * 32 + r;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int chunk = sub * 4 + 2 * s2 + h2;
                    const v8 vf = *reinterpret_cast<const v8*>(sv + drow * 128 + ((chunk ^ ((drow >> 1) & 7)) << 4));
                    oacc[dh] = T16<T>::mfma32(vf, pf[s2], oacc[dh]);
                }
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    __syncthreads();                     // every wave is done with the K / V^T ring and the table: the LDS becomes the store staging
    store_out_tile<T>(smem + wave * 8192, oacc, inv, out, split, nh, head, grouped ? (int64_t)grouped + (int64_t)b * (S - 1) : (int64_t)b * S + 1,
                      grouped ? (int64_t)b : (int64_t)b * S, S, q0, active, lane, cls_planes);
}

// The cls query on its own (round 5, CLS2 builds of attention_tab2_kernel).  As a 25th query tile the cls query costs a whole wave for one
// valid row, and 25 tiles do not split into blocks that fill a CU's wave slots (13 + 12 waves: one block per CU, 13 of 16 slots, one SIMD
// with four waves; 4 x 7: 14 of 16).  The 24 patch tiles do -- three blocks of 8, two blocks per CU -- and block 0 of every (image, head) runs
// this pass after its tiles: the hp + 1 key rows (32-key sub-tiles; the last holds the cls key alone) are dealt over the block's waves, a wave reads
// its K / V^T fragments straight from global memory (no ring: a sub-tile is read once, by one wave) and keeps a running max / sum / O^T of its
// own; the partial results meet in LDS and wave 0 merges them (sum_w 2^(m_w - M) (l_w, O_w)) and stores the row through the usual epilogue.
// Same operands and the same fp32 accumulation as a tile of the main loop; the order of the key rows within a sum differs from it (and is fixed:
// by the wave count, not by batch or launch).  `area`: 8 KiB of LDS behind the table, the partials and then the staging of the store.
template <typename T, int QW, bool CORR>
__device__ __forceinline__ void cls_query_pass(const T* __restrict__ Qg, const T* __restrict__ Kg, const T* __restrict__ Vg, const T* __restrict__ Qlg,
                                               const T* __restrict__ Klg, const T* __restrict__ Vlg, int Sp, int hp, const float* __restrict__ tail3,
                                               char* area, int wave, int lane, T* __restrict__ out, int split, int nh, int head, int64_t row0,
                                               int64_t row_cls, int S, bool cls_planes) {
    typedef typename T16<T>::v8 v8;
    const int r = lane & 31, h2 = lane >> 5;
    const int kap = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
    const int q0 = hp * 32;
    constexpr float THR = 6.0f;
    const float b_cp = tail3[0], b_cc = tail3[2];          // cls -> patch, cls -> cls
    v8 qf[4], qlf[CORR ? 4 : 1];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = *reinterpret_cast<const v8*>(Qg + (int64_t)(q0 + r) * 64 + ks * 16 + h2 * 8);
        if constexpr (CORR) qlf[ks] = *reinterpret_cast<const v8*>(Qlg + (int64_t)(q0 + r) * 64 + ks * 16 + h2 * 8);
    }
    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        oacc[0][i] = 0.f;
        oacc[1][i] = 0.f;
    }
    float m_run = 0.f, l_run = 0.f;
    bool first = true;
    // K fragments of the wave's next key row are fetched under the arithmetic of the current one, V^T fragments at the top of the row's own step
    // (they are not needed before the probabilities exist)
    auto load_k = [&](int ky, v8 (&kf)[4], v8 (&kfl)[CORR ? 4 : 1]) {
        const int64_t krow = (int64_t)ky * 32 + kap;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = *reinterpret_cast<const v8*>(Kg + krow * 64 + (2 * ks + h2) * 8);
            if constexpr (CORR) kfl[ks] = *reinterpret_cast<const v8*>(Klg + krow * 64 + (2 * ks + h2) * 8);
        }
    };
    v8 kf[4], kfl[CORR ? 4 : 1];
    load_k(wave, kf, kfl);             // (hp + 1 >= QW: every wave has a key row)
#pragma unroll 1
    for (int ky = wave; ky <= hp; ky += QW) {
        v8 vf[4], vfl[CORR ? 4 : 1];
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int64_t voff = (int64_t)(dh * 32 + r) * Sp + ky * 32 + (2 * s2 + h2) * 8;
                vf[dh * 2 + s2] = *reinterpret_cast<const v8*>(Vg + voff);
                if constexpr (CORR) vfl[dh * 2 + s2] = *reinterpret_cast<const v8*>(Vlg + voff);
            }
        v8 kn[4], knl[CORR ? 4 : 1];
        const int kyn = ky + QW <= hp ? ky + QW : ky;        // (the last step re-reads its own row: no branch around the loads)
        load_k(kyn, kn, knl);
        f32x16 sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = (ky < hp ? b_cp : -1.0e30f) - m_run;
        if (ky == hp && h2 == 0) sacc[0] = b_cc - m_run;     // the cls key: key 0 of the last sub-tile, the rest is padding
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            sacc = T16<T>::mfma32(kf[ks], qf[ks], sacc);
            if constexpr (CORR) {
                sacc = T16<T>::mfma32(kf[ks], qlf[ks], sacc);
                sacc = T16<T>::mfma32(kfl[ks], qf[ks], sacc);
            }
        }
        float mloc = sacc[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mloc = fmaxf(mloc, sacc[i]);
        mloc = half_swap_max(mloc);
        if (first || __any(mloc > THR)) {                    // as in the main loop: the shift is the lane's own, never negative after the first
            const float sh = first ? mloc : fmaxf(mloc, 0.0f);
            const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-sh);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                oacc[0][i] *= alpha;
                oacc[1][i] *= alpha;
                sacc[i] -= sh;
            }
            m_run += sh;
            first = false;
        }
        float psum = 0.f;
        v8 pf[2], pfl[CORR ? 2 : 1];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pv = __builtin_amdgcn_exp2f(sacc[8 * s2 + e]);
                psum += pv;
                pf[s2][e] = T16<T>::from_f32(pv);
                if constexpr (CORR) pfl[s2][e] = T16<T>::from_f32(pv - T16<T>::to_f32(pf[s2][e]));
            }
        l_run += psum;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                oacc[dh] = T16<T>::mfma32(vf[dh * 2 + s2], pf[s2], oacc[dh]);
                if constexpr (CORR) {
                    oacc[dh] = T16<T>::mfma32(vf[dh * 2 + s2], pfl[s2], oacc[dh]);
                    oacc[dh] = T16<T>::mfma32(vfl[dh * 2 + s2], pf[s2], oacc[dh]);
                }
            }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = kn[ks];
            if constexpr (CORR) kfl[ks] = knl[ks];
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);   // the other lane half holds the other 16 keys of every sub-tile
    // partials of query column 0 (the cls query: lanes 0 and 32), per wave and lane half: [m, l, 32 x O^T]
    float* part = reinterpret_cast<float*>(area);
    if (r == 0) {
        float* pp = part + (wave * 2 + h2) * 36;
        pp[0] = m_run;
        pp[1] = l_tot;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            pp[2 + i] = oacc[0][i];
            pp[18 + i] = oacc[1][i];
        }
    }
    __syncthreads();
    if (wave != 0) return;
    float M = part[h2 * 36];
#pragma unroll 1
    for (int w = 1; w < QW; ++w) M = fmaxf(M, part[(w * 2 + h2) * 36]);
    float L = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        oacc[0][i] = 0.f;
        oacc[1][i] = 0.f;
    }
#pragma unroll 1
    for (int w = 0; w < QW; ++w) {
        const float* pp = part + (w * 2 + h2) * 36;
        const float f = __builtin_amdgcn_exp2f(pp[0] - M);
        L = fmaf(pp[1], f, L);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            oacc[0][i] = fmaf(pp[2 + i], f, oacc[0][i]);
            oacc[1][i] = fmaf(pp[18 + i], f, oacc[1][i]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the partials are read; the region becomes this wave's store staging
    __builtin_amdgcn_wave_barrier();
    store_out_tile<T>(area, oacc, 1.0f / L, out, split, nh, head, row0, row_cls, S, q0, true, lane, cls_planes);
}

// ---------------------------------------------------------------------------------------------------------------------
// Pipelined form of the table variant (hp even: the 384x512 and 416x512 network inputs).  The loop above runs a wave through
// QK^T -> softmax -> PV strictly in turn and branches on the kind of key row / query tile inside the loop (the merges cost
// dozens of register copies per sub-tile: 153 VALU + 63 SALU instructions per 32x32 sub-tile by PMC, twice the arithmetic).  Here
//   * one loop iteration = one 64-key tile = two full key rows, no special cases: the cls key (the single valid key of the
//     last tile) is a peeled step after the loop, the cls-query tile reads its constant bias through the same addressing
//     (row stride 0 into a constant region of LDS);
//   * the bias table sits in LDS REVERSED, so a lane's 8 consecutive keys are 8 ascending words (no re-ordering moves), and it is
//     read straight into the score accumulator, which the MFMA chain then continues from (bias - running max);
//   * both sub-tiles' QK^T MFMAs are issued before the first softmax: the matrix pipe works under the exp / convert / sum
//     VALU stream of the same wave, and PV of the first half overlaps the softmax of the second.  A rescale of the running max
//     also shifts the accumulator still in flight.
// Table operand: `table` holds per head [(2hp-1)*63 body entries in REVERSED order | cls->patch, patch->cls, cls->cls].
//
// CORR (bs_attention_table_corr): the split-precision product.  Q, K, V^T each come with a second 16-bit tensor holding the rounding
// residual (x - round16(x), unscaled; written by the QKV epilogue, bs_gemm_desc.qkv_lo_off), and the probabilities are split the
// same way in registers:  S = Q_hi K_hi + Q_lo K_hi + Q_hi K_lo,  O = V_hi P_hi + V_hi P_lo + V_lo P_hi  -- three MFMA passes
// accumulating into the same fp32 registers, ~22 significant bits per operand.  With outlier channels after the LayerNorms (trained
// BEiT checkpoints) the single 16-bit Q / K / V / P of the plain kernel cost 0.9-2.4e-4 m of depth EACH
// (tools/probes/outlier_rounding_study.py); all four corrected: 1.7e-6 m.  The K / V^T ring holds the lo tiles behind the hi tiles
// (32 KiB per stage, two blocks per CU).
template <typename T, int QW, int WPE, bool CORR, bool PK = false, bool CLS2 = false>
__global__ __launch_bounds__(QW * 64, WPE) void attention_tab2_kernel(const T* __restrict__ Q, const T* __restrict__ K, const T* __restrict__ Vt,
                                                                  const T* __restrict__ Ql, const T* __restrict__ Kl, const T* __restrict__ Vtl,
                                                                  const float* __restrict__ table, T* __restrict__ out, int split, int B, int nh,
                                                                  int hp, int Sp, int nqb, int ntab, int grouped) {
    BS_ARG_NOW(out);
    const bool cls_planes = (split & 4) != 0;   // FP8 planes for the cls query's row only (bs_attention_table, dtype bit 6)
    const int abl = split >> 3;                 // diagnostics (BS_ATTN_ABL): 1 = no ring barrier, 2 = no DMA wait, 4 = no DMA after tile 1 -- wrong results, timing only
    split &= 3;
    typedef typename T16<T>::v8 v8;
    constexpr int HALF = 16 * 1024;             // K tile 8 KiB + V^T tile 8 KiB
    constexpr int STAGE = CORR ? 2 * HALF : HALF;   // CORR: [K_hi | V^T_hi | K_lo | V^T_lo]
    constexpr int WP = 32, RW = 2 * WP - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem + 2 * STAGE);
    const int ntab_pad = (ntab + 255) & ~255; // whole 1-KiB DMA pieces
    float* creg = tab + ntab_pad;             // 64 words of the cls->patch entry: the cls query's bias towards every patch key
    const int S = hp * WP + 1, nqt = CLS2 ? hp : hp + 1;       // CLS2: patch tiles only, the cls query is cls_query_pass

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + loc;
    // the q-blocks of one (image, head) are consecutive work ids: they stream the same K / V^T tiles at about the same time,
    // from the same XCD's L2
    const int qblk = wg % nqb;
    const int b = (wg / nqb) % B;
    const int head = wg / (B * nqb);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h2 = lane >> 5;
    int qt = qblk * QW + wave;
    const bool active = qt < nqt;
    qt = active ? qt : nqt - 1;
    const int q0 = qt * 32;
    const bool cls_tile = !CLS2 && qt == hp;
    const int64_t bh = (int64_t)b * nh + head;

    const T* Qg = Q + bh * Sp * 64;
    const T* Kg = K + bh * Sp * 64;
    const T* Vg = Vt + bh * 64 * Sp;
    const T* Klg = CORR ? Kl + bh * Sp * 64 : nullptr;
    const T* Vlg = CORR ? Vtl + bh * 64 * Sp : nullptr;

    load_table<QW * 64>(tab, table, head, ntab, tid, wave, lane);
    if (tid < 64) creg[tid] = table[(int64_t)head * ntab + ntab - 3];

    v8 qf[4], qlf[CORR ? 4 : 1];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const v8*>(Qg + (int64_t)(q0 + r) * 64 + ks * 16 + h2 * 8);
    if constexpr (CORR) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qlf[ks] = *reinterpret_cast<const v8*>(Ql + bh * Sp * 64 + (int64_t)(q0 + r) * 64 + ks * 16 + h2 * 8);
    }

    const int srow = lane >> 3;
    auto stage = [&](int kt, int buf) {
        char* sb = smem + buf * STAGE;
        for (int i = wave; i < (CORR ? 32 : 16); i += QW) {
            const int j = i & 15;                         // piece of the hi (i < 16) or lo half of the stage
            const int row = (j & 7) * 8 + srow;
            const int cs8 = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
            const T* kb = (CORR && i >= 16) ? Klg : Kg;
            const T* vb = (CORR && i >= 16) ? Vlg : Vg;
            const T* src = j < 8 ? kb + (int64_t)(kt * 64 + row) * 64 + cs8 : vb + (int64_t)row * Sp + kt * 64 + cs8;
            glds16(src, sb + i * 1024);
        }
    };

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        oacc[0][i] = 0.f;
        oacc[1][i] = 0.f;
    }
    float m_run = 0.f, l_run = 0.f;
    bool first = true;
    const int kap = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
    constexpr float THR = 6.0f;

    // Bias of (query (qt, r), key (ky, kx)), kx = 16s + 8*h2 + e for accumulator register 8s + e: body entry
    // (qt - ky + hp - 1) * 63 + (r - kx + 31), i.e. word (nbody - 1) - that of the reversed table
    //   = [nbody - 63 - (qt + hp - 1) * 63] + ky * 63 + (31 - r + 8*h2) + 16s + e        (ascending in e).
    // The cls-query tile reads the constant region with row stride 0.
    const int nbody = ntab - 3;
    const float* lt = cls_tile ? creg : tab + (nbody - RW - (qt + hp - 1) * RW) + (31 - r + 8 * h2);
    const int rstep = cls_tile ? 0 : RW;
    auto load_bias = [&](f32x16& sacc, int ky) {
        const float* rowp = lt + ky * rstep;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int e = 0; e < 8; ++e) sacc[8 * s2 + e] = rowp[16 * s2 + e];
    };
    // scores of one sub-tile: (bias - running max) + K Q^T, four MFMAs left in flight
    auto qk = [&](f32x16& sacc, const char* sk, int sub) {
        // The pre-shift by the running max.  (Round 3 found packed / dropped forms of it "nondeterministic at NB = 128" and concluded the
        // MFMAs needed these sixteen VALU reads of the freshly loaded bias.  They do not: those forms had lost a spill reload whose
        // `s_waitcnt vmcnt(0)` was the only thing ordering the tile DMA before the barrier -- dma_barrier() above.  PK: the same
        // subtraction as eight v_pk_add_f32 of the negated max, bit-identical.)
        if constexpr (PK) {
            const f32x2_ nm = {-m_run, -m_run};
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2_ v = f32x2_{sacc[i], sacc[i + 1]} + nm;
                sacc[i] = v[0];
                sacc[i + 1] = v[1];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[i] -= m_run;
        }
        const int krow = sub * 32 + kap;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int chunk = 2 * ks + h2;
            const int off = krow * 128 + ((chunk ^ ((krow >> 1) & 7)) << 4);
            const v8 kf = *reinterpret_cast<const v8*>(sk + off);
            sacc = T16<T>::mfma32(kf, qf[ks], sacc);
            if constexpr (CORR) {
                sacc = T16<T>::mfma32(kf, qlf[ks], sacc);
                const v8 kfl = *reinterpret_cast<const v8*>(sk + HALF + off);
                sacc = T16<T>::mfma32(kfl, qf[ks], sacc);
            }
        }
    };
    // softmax of one sub-tile's scores + O^T += V^T P^T.  OTHER: `other` is the accumulator of the next sub-tile, already shifted
    // by the current running max and in flight on the matrix pipe: a rescale shifts it too.  NEXT: once the probabilities are
    // packed the score registers are free and take the bias of key row next_ky (for the tile after this one).
    auto softmax_pv = [&](f32x16& sacc, f32x16& other, const char* sv, int sub, int next_ky, auto other_tag, auto next_tag) {
        constexpr bool OTHER = decltype(other_tag)::value;
        constexpr bool NEXT = decltype(next_tag)::value;
        float mloc = fmaxf(fmaxf(sacc[0], sacc[1]), sacc[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mloc = fmaxf(fmaxf(mloc, sacc[i]), sacc[i + 1]);
        mloc = fmaxf(mloc, sacc[15]);
        mloc = half_swap_max(mloc);          // the other lane half holds the other 16 keys of this query
        if (first || __any(mloc > THR)) {
            // the branch is wave-uniform but the shift is the lane's own: a query whose tile maximum lies BELOW its running max keeps it
                // (shift 0) -- exp2(-mloc) of a maximum 128 below the running one overflowed to inf (a key that out-scores the rest by
                // 2^128 earlier in the sequence; found by the spiked-key test).  On the first sub-tile the accumulators are zero: no scaling.
                const float sh = first ? mloc : fmaxf(mloc, 0.0f);
                const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-sh);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                oacc[0][i] *= alpha;
                oacc[1][i] *= alpha;
                sacc[i] -= sh;
                if (OTHER) other[i] -= sh;
            }
            m_run += sh;
            first = false;
        }
        float psum = 0.f;
        v8 pf[2], pfl[CORR ? 2 : 1];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pv = __builtin_amdgcn_exp2f(sacc[8 * s2 + e]);
                psum += pv;
                pf[s2][e] = T16<T>::from_f32(pv);
                if constexpr (CORR) pfl[s2][e] = T16<T>::from_f32(pv - T16<T>::to_f32(pf[s2][e]));
            }
        l_run += psum;
        if (NEXT) load_bias(sacc, next_ky);
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
            const int drow = dh * 32 + r;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int chunk = sub * 4 + 2 * s2 + h2;
                const int off = drow * 128 + ((chunk ^ ((drow >> 1) & 7)) << 4);
                const v8 vf = *reinterpret_cast<const v8*>(sv + off);
                oacc[dh] = T16<T>::mfma32(vf, pf[s2], oacc[dh]);
                if constexpr (CORR) {
                    oacc[dh] = T16<T>::mfma32(vf, pfl[s2], oacc[dh]);
                    const v8 vfl = *reinterpret_cast<const v8*>(sv + HALF + off);
                    oacc[dh] = T16<T>::mfma32(vfl, pf[s2], oacc[dh]);
                }
            }
        }
    };

    const int NT = hp >> 1;               // full 64-key tiles (two key rows each); tile NT holds the cls key
    f32x16 sa, sb2;
    typedef std::true_type Yes;
    typedef std::false_type No;
    stage(0, 0);
    dma_barrier();                      // tile 0 and the table have landed
    load_bias(sa, 0);
    load_bias(sb2, 1);
    for (int kt = 0; kt < NT - 1; ++kt) {
        if (kt > 0) {                   // tile kt has landed; everyone is done with tile kt - 1
            if (abl == 0) dma_barrier();
            else {
                if (!(abl & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(abl & 1)) __syncthreads();
            }
        }
        if (!(abl & 4) || kt == 0) stage(kt + 1, (kt + 1) & 1);
        const char* sk = smem + (kt & 1) * STAGE;
        qk(sa, sk, 0);
        qk(sb2, sk, 1);
        softmax_pv(sa, sb2, sk + 8 * 1024, 0, 2 * kt + 2, Yes{}, Yes{});
        softmax_pv(sb2, sa, sk + 8 * 1024, 1, 2 * kt + 3, No{}, Yes{});
    }
    {   // last full tile; then the cls key (sub-tile 0 of tile NT: one valid key, the rest padding)
        if (NT > 1) dma_barrier();
        stage(NT, NT & 1);
        const char* sk = smem + ((NT - 1) & 1) * STAGE;
        qk(sa, sk, 0);
        qk(sb2, sk, 1);
        softmax_pv(sa, sb2, sk + 8 * 1024, 0, 0, Yes{}, No{});
#pragma unroll
        for (int i = 0; i < 16; ++i) sa[i] = -1.0e30f;
        if (h2 == 0) sa[0] = cls_tile ? tab[ntab - 1] : tab[ntab - 2];
        softmax_pv(sb2, sa, sk + 8 * 1024, 1, 0, No{}, No{});
        dma_barrier();
        const char* sk2 = smem + (NT & 1) * STAGE;
        qk(sa, sk2, 0);
        softmax_pv(sa, sb2, sk2 + 8 * 1024, 0, 0, No{}, No{});
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    __syncthreads();                     // every wave is done with the K / V^T ring and the table: the LDS becomes the store staging
    store_out_tile<T>(smem + wave * 8192, oacc, inv, out, split, nh, head, grouped ? (int64_t)grouped + (int64_t)b * (S - 1) : (int64_t)b * S + 1,
                      grouped ? (int64_t)b : (int64_t)b * S, S, q0, active, lane, cls_planes);
    if constexpr (CLS2) {
        if (qblk == 0) {    // block-uniform: after their own tiles (K / V^T of this (image, head) are warm in L2) the block's waves share the cls query's key rows
            __syncthreads();                 // every wave's store staging is read out
            cls_query_pass<T, QW, CORR>(Qg, Kg, Vg, CORR ? Ql + bh * Sp * 64 : nullptr, Klg, Vlg, Sp, hp, table + (int64_t)head * ntab + ntab - 3,
                                        smem + 2 * STAGE + (ntab_pad + 64) * 4, wave, lane, out, split, nh, head,
                                        grouped ? (int64_t)grouped + (int64_t)b * (S - 1) : (int64_t)b * S + 1, grouped ? (int64_t)b : (int64_t)b * S, S,
                                        cls_planes);
        }
    }
}

template <typename T>
static int launch_attn_tab(const void* q, const void* k, const void* vt, const void* ql, const void* kl, const void* vtl, const float* table, void* out,
                           int split, int B, int nh, int hp, int Sp, int grouped, hipStream_t st) {
    constexpr int QW = 5;
    const int nqt = hp + 1, nqb = cdiv(nqt, QW), ntab = (2 * hp - 1) * 63 + 3;
    const int tab_bytes = (((ntab + 255) & ~255) + 64) * 4;
    if (ql) {     // split-precision operands (bs_attention_table_corr): 64 KiB ring, the 256-register budget = two waves per SIMD, 8 per CU
        // Round 5: 7- or 8-wave blocks.  A 5-wave block leaves 3 of the CU's 8 wave slots empty (a second block of 5 does not fit them): blocks of
        // ceil(nqt / ceil(nqt / 8)) waves fill them (25 query tiles: 4 blocks of 7) and stage a tile once per 7-8 query tiles.  Same bits.
        static const bool cls2_ok = diag_env("BS_ATTN_NO_CLS2") == nullptr;     // diagnostics: the round-5 shape before the cls query left the tiles
        if (cls2_ok && hp % 8 == 0) {
            // the cls query as cls_query_pass: 24 patch tiles = 3 blocks of 8 waves, every wave slot of the CU (2 per SIMD at this register budget).
            // (Three waves per SIMD -- 168 registers, 6-wave blocks, two per CU -- spill 29 registers and run 1 886 us against 1 313.)
            auto kb = attention_tab2_kernel<T, 8, 2, true, false, true>;
            BS_MAX_DYNAMIC_LDS(reinterpret_cast<const void*>(kb), 96 * 1024);
            hipLaunchKernelGGL(kb, dim3(B * nh * (hp / 8)), dim3(8 * 64), 64 * 1024 + tab_bytes + 8192, st, (const T*)q, (const T*)k, (const T*)vt,
                               (const T*)ql, (const T*)kl, (const T*)vtl, table, (T*)out, split, B, nh, hp, Sp, hp / 8, ntab, grouped);
            BS_CHECK_LAUNCH();
            return BS_OK;
        }
        {
            const int nb8 = cdiv(nqt, 8), qw8 = cdiv(nqt, nb8);
#define BS_ATTN_CORR_BIG(QWB)                                                                                                                \
    do {                                                                                                                                     \
        auto kb = attention_tab2_kernel<T, QWB, 2, true>;                                                                                    \
        BS_MAX_DYNAMIC_LDS(reinterpret_cast<const void*>(kb), 96 * 1024); \
        hipLaunchKernelGGL(kb, dim3(B * nh * nb8), dim3(QWB * 64), 64 * 1024 + tab_bytes, st, (const T*)q, (const T*)k, (const T*)vt,        \
                           (const T*)ql, (const T*)kl, (const T*)vtl, table, (T*)out, split, B, nh, hp, Sp, nb8, ntab, grouped);             \
        BS_CHECK_LAUNCH();                                                                                                                   \
        return BS_OK;                                                                                                                        \
    } while (0)
            if (qw8 == 7) BS_ATTN_CORR_BIG(7);
            if (qw8 == 8) BS_ATTN_CORR_BIG(8);
#undef BS_ATTN_CORR_BIG
        }
        auto kc = attention_tab2_kernel<T, QW, 2, true>;
        // the cap covers the largest table the entry admits (hp = 40: 64 KiB + 20.3 KiB); up to hp = 30 the ring + table stay within
            // 80 KiB and two blocks share a CU, beyond that one block per CU (round-4 advisor: the cap was 80 KiB and hp 32 ... 40 failed)
        BS_MAX_DYNAMIC_LDS(reinterpret_cast<const void*>(kc), 88 * 1024);
        hipLaunchKernelGGL(kc, dim3(B * nh * nqb), dim3(QW * 64), 64 * 1024 + tab_bytes, st, (const T*)q, (const T*)k, (const T*)vt, (const T*)ql,
                           (const T*)kl, (const T*)vtl, table, (T*)out, split, B, nh, hp, Sp, nqb, ntab, grouped);
        BS_CHECK_LAUNCH();
        return BS_OK;
    }
    int smem = 32 * 1024 + ((ntab * 4 + 1023) & ~1023);
    smem = smem < QW * 8192 ? QW * 8192 : smem;                    // the epilogue stages 8 KiB per wave
    auto kern = attention_tab_kernel<T, QW>;
    BS_MAX_DYNAMIC_LDS(reinterpret_cast<const void*>(kern), 64 * 1024);
    static const bool pipelined_ok = diag_env("BS_ATTN_NO_PIPE") == nullptr;       // diagnostics: the unpipelined loop
    if (hp % 2 == 0 && pipelined_ok) {
        // 4 waves per SIMD (128 registers, three blocks per CU) with the packed pre-shift: one spilled register, 628 us per NB = 128 launch.
        // With the scalar pre-shift (BS_ATTN_NO_PK, the round-3 kernel) ten registers spill and their reloads put an `s_waitcnt vmcnt(0)`
        // right behind the stage -- the next tile's DMA is waited for at once instead of under the tile's arithmetic: 700 us.  The spill-free
        // 3-waves-per-SIMD build (BS_ATTN_WPE3) overlaps the DMA too and is still slower (811 us): occupancy hides more than the
        // prefetch does (profiles/r04_attention_variants.txt).
        static const bool wpe3 = diag_env("BS_ATTN_WPE3") != nullptr, pk = diag_env("BS_ATTN_NO_PK") == nullptr;
        auto kern2 = wpe3 ? (pk ? attention_tab2_kernel<T, QW, 3, false, true> : attention_tab2_kernel<T, QW, 3, false>)
                          : (pk ? attention_tab2_kernel<T, QW, 4, false, true> : attention_tab2_kernel<T, QW, 4, false>);
        BS_MAX_DYNAMIC_LDS(reinterpret_cast<const void*>(kern2), 64 * 1024);
        static const int abl = diag_env("BS_ATTN_ABL") ? atoi(diag_env("BS_ATTN_ABL")) : 0;      // diagnostics (timing only)
        // Round 5: large blocks.  With 5 waves per block an (image, head)'s 25 query tiles are 5 blocks, each staging every K / V^T tile for itself
        // (the staging is 9 % of the launch by ablation, profiles/r05_attention_ablations.txt); with 11-14 waves per block -- 2 blocks for 25 or
        // 27 query tiles, 3 for 33 or 41 -- a tile is staged once per 11-14 query tiles: 717 -> 669 us at NB = 128 (one block per CU: the epilogue
        // stages 8 KiB per wave; same bits -- a wave's arithmetic does not depend on its block).  Query-tile counts that do not split into
        // blocks of 11-14 keep the 5-wave blocks.
        static const bool cls2_ok = diag_env("BS_ATTN_NO_CLS2") == nullptr;     // diagnostics: the large blocks below
        if (!wpe3 && pk && abl == 0 && cls2_ok && hp % 8 == 0) {
            // Round 5, second step: the cls query leaves the tiles (cls_query_pass).  24 (32, 40) patch tiles are 3 (4, 5) blocks of 8 waves, 64 KiB
            // of LDS each: two blocks per CU fill its 16 wave slots, four waves on every SIMD (13 + 12 waves left 3 slots empty and one SIMD with
            // four waves against three).  profiles/r05_attention_blocks.txt.
            auto kb = attention_tab2_kernel<T, 8, 4, false, true, true>;
            BS_MAX_DYNAMIC_LDS(reinterpret_cast<const void*>(kb), 128 * 1024);
            const int ldsb = 8 * 8192 > 32 * 1024 + tab_bytes + 8192 ? 8 * 8192 : 32 * 1024 + tab_bytes + 8192;
            hipLaunchKernelGGL(kb, dim3(B * nh * (hp / 8)), dim3(8 * 64), ldsb, st, (const T*)q, (const T*)k, (const T*)vt, (const T*)nullptr,
                               (const T*)nullptr, (const T*)nullptr, table, (T*)out, split, B, nh, hp, Sp, hp / 8, ntab, grouped);
            BS_CHECK_LAUNCH();
            return BS_OK;
        }
        if (!wpe3 && pk && abl == 0) {
            const int nb_big = cdiv(nqt, 14), qw_big = cdiv(nqt, nb_big);
#define BS_ATTN_BIG(QWB)                                                                                                                     \
    do {                                                                                                                                     \
        auto kb = attention_tab2_kernel<T, QWB, 4, false, true>;                                                                             \
        BS_MAX_DYNAMIC_LDS(reinterpret_cast<const void*>(kb), 128 * 1024); \
        const int ldsb = QWB * 8192 > 32 * 1024 + tab_bytes ? QWB * 8192 : 32 * 1024 + tab_bytes;                                            \
        hipLaunchKernelGGL(kb, dim3(B * nh * nb_big), dim3(QWB * 64), ldsb, st, (const T*)q, (const T*)k, (const T*)vt, (const T*)nullptr,   \
                           (const T*)nullptr, (const T*)nullptr, table, (T*)out, split, B, nh, hp, Sp, nb_big, ntab, grouped);               \
        BS_CHECK_LAUNCH();                                                                                                                   \
        return BS_OK;                                                                                                                        \
    } while (0)
            if (qw_big == 11) BS_ATTN_BIG(11);
            if (qw_big == 12) BS_ATTN_BIG(12);
            if (qw_big == 13) BS_ATTN_BIG(13);
            if (qw_big == 14) BS_ATTN_BIG(14);
#undef BS_ATTN_BIG
        }
        hipLaunchKernelGGL(kern2, dim3(B * nh * nqb), dim3(QW * 64), 32 * 1024 + tab_bytes, st, (const T*)q, (const T*)k, (const T*)vt, (const T*)nullptr,
                           (const T*)nullptr, (const T*)nullptr, table, (T*)out, split | (abl << 3), B, nh, hp, Sp, nqb, ntab, grouped);
        BS_CHECK_LAUNCH();
        return BS_OK;
    }
    hipLaunchKernelGGL(kern, dim3(B * nh * nqb), dim3(QW * 64), smem, st, (const T*)q, (const T*)k, (const T*)vt, table, (T*)out, split, B,
                       nh, hp, Sp, nqb, ntab, grouped, diag_env("BS_ATTN_ABLATE") ? atoi(diag_env("BS_ATTN_ABLATE")) : 0);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

template <typename T, int QW>
static int launch_attn(const void* q, const void* k, const void* vt, const float* bias, void* out, int split, int B, int nh, int S, int Sp,
                       hipStream_t st) {
    const int nqt = cdiv(S, 32), nqb = cdiv(nqt, QW);
    hipLaunchKernelGGL((attention_kernel<T, QW>), dim3(B * nh * nqb), dim3(QW * 64), 32 * 1024, st, (const T*)q, (const T*)k,
                       (const T*)vt, bias, (T*)out, split, B, nh, S, Sp, nqt, nqb, diag_env("BS_ATTN_ABLATE") ? atoi(diag_env("BS_ATTN_ABLATE")) : 0);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

template <typename T>
static int dispatch_attn(const void* q, const void* k, const void* vt, const float* bias, void* out, int split, int B, int nh, int S, int Sp,
                         hipStream_t st) {
    const int nqt = cdiv(S, 32);
    // waves (32-query tiles) per block: 5 shares a staged K / V^T tile among the most queries and measured fastest (769 tokens:
    // 498 us vs 572 / 584 us for 4 / 3 waves), so it is used whenever there are at least 5 query tiles even if the last block
    // is partly idle (833 tokens = 27 tiles: 6 blocks); tiny sequences take the size that fits
    if (nqt >= 5) return launch_attn<T, 5>(q, k, vt, bias, out, split, B, nh, S, Sp, st);
    if (nqt % 3 == 0) return launch_attn<T, 3>(q, k, vt, bias, out, split, B, nh, S, Sp, st);
    return launch_attn<T, 4>(q, k, vt, bias, out, split, B, nh, S, Sp, st);
}

}  // namespace bs

extern "C" int bs_attention(const void* q, const void* k, const void* vt, const float* bias, void* out, int32_t B, int32_t nh, int32_t S,
                            int32_t Sp, int32_t dtype, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("bs_attention: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(q && k && vt && bias && out && B >= 0 && nh > 0 && S > 0, "bs_attention: bad argument");
    BS_REQUIRE(Sp % 64 == 0 && Sp >= S, "bs_attention: Sp=%d must be a multiple of 64 and >= S=%d", Sp, S);
    const int split = (dtype & 32) ? 2 : ((dtype & 16) ? 1 : 0);   // bit 4: (hi | lo) 16-bit pairs; bit 5: (hi16 | hi8 | lo8); [B*S, 2*nh*64]
    dtype &= 15;
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "bs_attention: dtype");
    if (B == 0) return BS_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dtype == BS_F16 ? dispatch_attn<f16>(q, k, vt, bias, out, split, B, nh, S, Sp, st)
                           : dispatch_attn<bf16>(q, k, vt, bias, out, split, B, nh, S, Sp, st);
}

static int attention_table_entry(const char* who, const void* q, const void* k, const void* vt, const void* ql, const void* kl, const void* vtl,
                                 const float* table, void* out, int32_t B, int32_t nh, int32_t hp, int32_t wp, int32_t Sp, int32_t grouped,
                                 int32_t dtype, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("%s: call bs_init first", who); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(q && k && vt && table && out && B >= 0 && nh > 0 && hp > 0, "%s: bad argument", who);
    BS_REQUIRE(wp == 32, "%s: built for windows of 32 patches per row (wp=%d): use bs_attention", who, wp);
    BS_REQUIRE(hp <= 40, "%s: hp=%d: the table must fit 32 KiB of LDS", who, hp);
    BS_REQUIRE(grouped == 0 || grouped >= B, "%s: grouped=%d is the first patch row, >= B", who, grouped);
    BS_REQUIRE(!ql || hp % 2 == 0, "%s: the split-precision kernel is built for an even number of patch rows (hp=%d)", who, hp);
    const int S = hp * wp + 1;
    BS_REQUIRE(Sp % 64 == 0 && Sp >= S, "%s: Sp=%d must be a multiple of 64 and >= S=%d", who, Sp, S);
    int split = (dtype & 32) ? 2 : ((dtype & 16) ? 1 : 0);
    if ((dtype & 64) && split == 2) split |= 4;     // bit 6: only the cls rows of `out` need their FP8 planes
    dtype &= 15;
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "%s: dtype", who);
    if (B == 0) return BS_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dtype == BS_F16 ? launch_attn_tab<f16>(q, k, vt, ql, kl, vtl, table, out, split, B, nh, hp, Sp, grouped, st)
                           : launch_attn_tab<bf16>(q, k, vt, ql, kl, vtl, table, out, split, B, nh, hp, Sp, grouped, st);
}

extern "C" int bs_attention_table(const void* q, const void* k, const void* vt, const float* table, void* out, int32_t B, int32_t nh,
                                  int32_t hp, int32_t wp, int32_t Sp, int32_t grouped, int32_t dtype, void* stream) {
    return attention_table_entry("bs_attention_table", q, k, vt, nullptr, nullptr, nullptr, table, out, B, nh, hp, wp, Sp, grouped, dtype, stream);
}

extern "C" int bs_attention_table_corr(const void* q, const void* k, const void* vt, const void* q_lo, const void* k_lo, const void* vt_lo,
                                       const float* table, void* out, int32_t B, int32_t nh, int32_t hp, int32_t wp, int32_t Sp, int32_t grouped,
                                       int32_t dtype, void* stream) {
    if (!q_lo || !k_lo || !vt_lo) { bs::set_error("bs_attention_table_corr: null residual tensor"); return BS_ERR_INVALID; }
    return attention_table_entry("bs_attention_table_corr", q, k, vt, q_lo, k_lo, vt_lo, table, out, B, nh, hp, wp, Sp, grouped, dtype, stream);
}
