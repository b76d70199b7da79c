// Dispatcher + C entry point of the implicit GEMM (the kernel template lives in igemm_kernel.h, the tile variants in
// igemm_tile*.hip).
#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>

#include "igemm_kernel.h"

namespace bs {

#define BS_DECL_TILE(t, cm)                                                  \
    int igemm_launch_tile##t##_f16_cm##cm(const IgemmParams&, bool, hipStream_t); \
    int igemm_launch_tile##t##_bf16_cm##cm(const IgemmParams&, bool, hipStream_t);
BS_DECL_TILE(1, 0) BS_DECL_TILE(1, 1)
BS_DECL_TILE(2, 0) BS_DECL_TILE(2, 1)
BS_DECL_TILE(3, 0) BS_DECL_TILE(3, 1)
BS_DECL_TILE(9, 0) BS_DECL_TILE(9, 1)
BS_DECL_TILE(10, 0) BS_DECL_TILE(11, 0)
#undef BS_DECL_TILE

// tile ids (BMxBNxBK, LDS stages): 1 128x128x64 s2 (2 blocks/CU) | 2 128x64x64 s2 | 3 128x32x64 s2 | 9 256x256x64 s2 (128 KiB)
//   10 256x256x32 s4 ping-pong (two wave groups alternate MFMA / load phases) | 11 256x128x32 s3, 4 waves (2 blocks/CU)
static int auto_tile(int M, int N, int K, int tile, bool conv) {
    if (tile != 0) return tile;
    if (N <= 32) return 3;
    if (N <= 64) return 2;
    // the 256x256x64 tile (1 block/CU, 128 KiB LDS) wins once the grid covers the 256 CUs often enough to amortise its
    // coarse tail; measured with tools/bench_kernels.py (plain K=1024..4096 GEMMs: from ~1.5 rounds; 3x3 convs: from 2)
    const long long blocks = (long long)cdiv(M, 256) * (N / 256);
    if (N % 256 == 0 && blocks >= (conv ? 512 : 384)) return (!conv && N <= 1024 && K <= 1024) ? 10 : 9;
    return 1;
}

static void tile_dims(int tile, int& BM, int& BN) {
    switch (tile) {
        case 2: BM = 128; BN = 64; break;
        case 3: BM = 128; BN = 32; break;
        case 11: BM = 256; BN = 128; break;
        case 9: case 10: BM = 256; BN = 256; break;
        default: BM = 128; BN = 128; break;
    }
}

static int launch_tile(IgemmParams& p, int dtype, bool conv, int tile, hipStream_t st) {
    int BM, BN;
    tile_dims(tile, BM, BN);
    p.ntm = cdiv(p.M - p.m_begin, BM);
    p.ntn = cdiv(p.N, BN);
    // Tile order.  Wide plain products on the 256 x 256 tile (fc1, the reassemble ConvT's: >= 8 N-tiles) walk column strips of four N-tiles,
    // M fastest: the strip's W tiles stay in the XCD's L2 while the A panels stream through once per strip -- 37-40 % fewer L2 refills at the
    // same launch time (profiles/r04_gemm_experiments.txt (7)).  Not the QKV product: its V^T tiles would bunch up at the end (+45 %).
    // BS_GEMM_STRIP = w forces a width (0 = row-major) for every non-convolution launch: the one diagnostic switch of this path (read once;
    // tools/probes/gemm_strip.sh).  (The thirds-interleaved strip walk for the QKV product, round 4's experiment, is gone: -50 % L2 refills, +5.6 % time.)
    static const int strip_env = diag_env("BS_GEMM_STRIP") ? atoi(diag_env("BS_GEMM_STRIP")) : -1;
    const bool qkv = p.out_mode == BS_OUT_QKV;
    p.strip = conv ? 0 : (strip_env >= 0 ? strip_env : ((BM == 256 && BN == 256 && !qkv && p.ntn >= 8) ? 4 : 0));
    // correction mode of the instantiation: 1 = FP8 stages / (hi16 | hi8 | lo8) formats, 0 = plain
    const int cm = (p.f8_stages > 0 || p.out_f8 || p.res_f8) ? 1 : 0;
    const bool h = dtype == BS_F16;
#define BS_TILE(t, c) (h ? igemm_launch_tile##t##_f16_cm##c(p, conv, st) : igemm_launch_tile##t##_bf16_cm##c(p, conv, st))
    switch (tile * 10 + cm) {
        case 10: return BS_TILE(1, 0);
        case 11: return BS_TILE(1, 1);
        case 20: return BS_TILE(2, 0);
        case 21: return BS_TILE(2, 1);
        case 30: return BS_TILE(3, 0);
        case 31: return BS_TILE(3, 1);
        case 90: return BS_TILE(9, 0);
        case 91: return BS_TILE(9, 1);
        case 100: return BS_TILE(10, 0);
        case 110: return BS_TILE(11, 0);
#undef BS_TILE
        default: set_error("bs_gemm: tile %d is not built for correction mode %d (tiles 1, 2, 3, 9, 10, 11)", tile, cm); return BS_ERR_INVALID;
    }
}

// Side stream + fork / join events of the tail split, one set per (device, caller stream): two plan lanes (or two host
// threads) issuing split GEMMs on different streams never share events, and a set created on one device is never used on
// another.  Created on first use under a mutex; they live for the life of the process.
struct TailLane {
    hipStream_t side;
    hipEvent_t fork, join;
};
static int tail_lane(hipStream_t caller, hipStream_t& side, hipEvent_t& ev_fork, hipEvent_t& ev_join) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, TailLane> lanes;
    int dev = 0;
    BS_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = lanes.find({dev, caller});
    if (it == lanes.end()) {
        TailLane l{};
        BS_CHECK_HIP(hipStreamCreateWithFlags(&l.side, hipStreamNonBlocking));
        BS_CHECK_HIP(hipEventCreateWithFlags(&l.fork, hipEventDisableTiming));
        BS_CHECK_HIP(hipEventCreateWithFlags(&l.join, hipEventDisableTiming));
        it = lanes.emplace(std::make_pair(dev, caller), l).first;
    }
    side = it->second.side; ev_fork = it->second.fork; ev_join = it->second.join;
    return BS_OK;
}

// Tail split.  A 256-row tile grid over M = 64 x 769 token rows is 192.25 tiles tall: the quarter-full last tile row costs a
// whole extra round of the chip (772 blocks on 256 CUs = 4 rounds for 3.02 rounds of work in o_proj / fc2).  When dropping the
// ragged rows saves a round, the full tiles go out as one launch and the <= 128 ragged rows as a second, small-tile launch
// on the same stream (same kernel family; IgemmParams::m_begin offsets its rows).
static int dispatch(IgemmParams& p, int dtype, bool conv, int tile, hipStream_t st) {
    tile = auto_tile(p.M, p.N, p.K, tile, conv);
    if ((p.f8_seg > 0 || p.out_f8 || p.res_f8) && (tile == 10 || tile == 11)) tile = 9;    // the FP8 path is built for the BK = 64 ring tiles
    int BM, BN;
    tile_dims(tile, BM, BN);
    const int rem = p.M % BM, full = p.M / BM, ntn = cdiv(p.N, BN), cus = cu_count();
    // (measured, tools/bench_kernels.py tiles 9 vs 809: -8 % on fc2 (K = 4096); neutral to slightly negative for K = 1024, where a
    // block is short and the second launch costs as much as the saved blocks)
    static const bool no_split = diag_env("BS_NO_TAIL_SPLIT") != nullptr;   // diagnostics
    if (!no_split && !(p.ablate & 8) && !conv && BM == 256 && p.K >= 2048 && full > 0 && rem > 0 && rem <= 128 && p.N % 128 == 0 &&
        cdiv(full * ntn, cus) < cdiv((full + 1) * ntn, cus)) {
        IgemmParams main = p;
        main.M = full * BM;                    // rows [0, full*BM): the kernel clamps and masks against M
        main.m_begin = 0;
        p.m_begin = full * BM;                 // rows [full*BM, M)
        // The two launches touch disjoint rows, so the short, latency-bound tail (8-32 small blocks, ~60 us on its own) runs on
        // a side stream beside the main launch and fills CUs the main grid leaves idle in its last round: fork / join by events.
        static const bool side_ok = diag_env("BS_NO_TAIL_STREAM") == nullptr;
        hipStream_t side = nullptr;
        hipEvent_t ev_fork = nullptr, ev_join = nullptr;
        if (side_ok) {
            const int rc = tail_lane(st, side, ev_fork, ev_join);
            if (rc != BS_OK) return rc;
        }
        if (!side_ok) {
            const int rc = launch_tile(main, dtype, conv, tile, st);
            if (rc != BS_OK) return rc;
            return launch_tile(p, dtype, conv, 1, st);
        }
        BS_CHECK_HIP(hipEventRecord(ev_fork, st));
        BS_CHECK_HIP(hipStreamWaitEvent(side, ev_fork, 0));
        int rc = launch_tile(main, dtype, conv, tile, st);
        if (rc != BS_OK) return rc;
        rc = launch_tile(p, dtype, conv, 1, side);
        if (rc != BS_OK) return rc;
        BS_CHECK_HIP(hipEventRecord(ev_join, side));
        BS_CHECK_HIP(hipStreamWaitEvent(st, ev_join, 0));
        return BS_OK;
    }
    p.m_begin = 0;
    return launch_tile(p, dtype, conv, tile, st);
}

}  // namespace bs

extern "C" int bs_gemm_tile(const bs_gemm_desc* d) {
    if (!d) return BS_ERR_INVALID;
    const int tile = bs::auto_tile(d->M, d->N, d->K + d->f8_seg / 2, d->tile % 100, d->conv != 0);
    return ((d->f8_seg > 0 || d->out_f8 || d->res_f8) && (tile == 10 || tile == 11)) ? 9 : tile;
}

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
                        : ((long long)(d->M - 1) * d->lda + (d->K + d->f8_seg / 2 - d->seg1)) * 2;
    BS_REQUIRE((long long)d->N * (d->K + d->f8_seg / 2) * 2 < 0x7FFFFFF0ll, "bs_gemm: weight matrix too large for one descriptor");
    p.M = d->M; p.N = d->N; p.K = d->K + d->f8_seg / 2; p.lda = d->lda;     // kernel K: 128-byte stages x 64
    p.f8_seg = d->f8_seg;
    p.f8_sa0 = d->f8_scales & 0xff; p.f8_sb0 = (d->f8_scales >> 8) & 0xff; p.f8_sa1 = (d->f8_scales >> 16) & 0xff; p.f8_sb1 = d->f8_scales >> 24;
    p.out_f8 = d->out_f8;
    p.res_f8 = d->res_f8;
    BS_REQUIRE(!d->res_f8 || (d->res && d->res_dtype == d->dtype && d->res_split_off == 0 && d->ldr >= 2 * d->N && d->N % 8 == 0), "bs_gemm: res_f8 needs a 16-bit residual in (hi16 | hi8 | lo8) rows, ldr >= 2N");
    BS_REQUIRE(d->f8_seg >= 0 && d->f8_seg % 256 == 0, "bs_gemm: f8_seg=%d must be a multiple of 256 (two halves of whole 128-byte stages)", d->f8_seg);
    BS_REQUIRE(d->f8_seg == 0 || (d->seg1 == 0 && !d->relu_a), "bs_gemm: the FP8 segment excludes seg1 and relu_a");
    p.f8_stages = d->f8_seg / 128;
    BS_REQUIRE(d->out_f8 == 0 || (d->out_dtype == d->dtype && d->ldo % 8 == 0 &&
                                  ((d->out_mode == BS_OUT_PLAIN && d->out_split_off == d->N && d->N % 8 == 0) ||
                                   (d->out_mode == BS_OUT_SHUFFLE && d->out_split_off == d->shuffle_cout && d->shuffle_cout % 8 == 0))),
               "bs_gemm: out_f8 needs a plain / shuffle 16-bit output with out_split_off = channels, channels %% 8 == 0");
    p.Hin = d->Hin; p.Win = d->Win; p.Cin = d->Cin; p.Hout = d->Hout; p.Wout = d->Wout;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
    if (d->conv) {
        BS_REQUIRE(d->Cin > 0 && d->Cin % 64 == 0, "bs_gemm: conv Cin=%d must be a multiple of 64", d->Cin);
        BS_REQUIRE(d->K == d->KH * d->KW * (d->Cin + d->seg1), "bs_gemm: conv K=%d != KH*KW*(Cin+seg1)", d->K);
        BS_REQUIRE(d->Hout > 0 && d->Wout > 0 && d->M % (d->Hout * d->Wout) == 0, "bs_gemm: conv M=%d not a multiple of Hout*Wout", d->M);
        BS_REQUIRE(d->stride > 0 && d->lda >= d->Cin, "bs_gemm: bad conv stride/lda");
        BS_REQUIRE(d->KH > 0 && d->KW > 0 && d->KH * d->KW <= 32, "bs_gemm: conv window %dx%d: at most 32 taps", d->KH, d->KW);
        BS_REQUIRE((long long)d->Hin * d->Win * d->lda * 2 * 4 < 0x7FFFFFF0ll, "bs_gemm: image too large for the 2 GiB tile window");
        p.tiles_per_tap = d->Cin / 64;
        if (d->f8_seg > 0) {
            // pixel vector = [hi16 x Cin | hi8 x Cin | lo8 x Cin]: the chunk walk simply continues through the two FP8 planes
            BS_REQUIRE(d->f8_seg == 2 * d->Cin && d->Cin % 128 == 0 && d->lda >= 2 * d->Cin, "bs_gemm: conv FP8 segment needs f8_seg = 2*Cin, Cin %% 128 == 0, lda >= 2*Cin");
            p.Cin = 2 * d->Cin;
            p.K = d->KH * d->KW * p.Cin;
            p.f8_stages = d->KH * d->KW * (d->f8_seg / 128);
            BS_REQUIRE((long long)d->N * p.K * 2 < 0x7FFFFFF0ll, "bs_gemm: weight matrix too large for one descriptor");
        }
    }
    p.relu_a = d->relu_a;
    p.cin1 = d->seg1;
    p.split_off = d->out_split_off;
    p.res_split_off = d->res_split_off;
    BS_REQUIRE(d->seg1 >= 0 && d->seg1 % 64 == 0, "bs_gemm: seg1=%d must be a multiple of 64", d->seg1);
    BS_REQUIRE(d->conv || d->seg1 < d->K, "bs_gemm: seg1 must be smaller than K");
    BS_REQUIRE(d->out_split_off == 0 || (d->out_mode != BS_OUT_QKV && d->out_dtype == d->dtype), "bs_gemm: split output needs a plain / shuffle 16-bit output");
    BS_REQUIRE(d->res_split_off == 0 || (d->res && d->res_dtype == d->dtype), "bs_gemm: split residual must be 16-bit");
    p.ablate = d->tile >= 100 ? d->tile / 100 : 0;

    BS_REQUIRE(!d->relu_a || d->conv, "bs_gemm: relu_a is only built for conv mode");
    BS_REQUIRE(d->act >= BS_ACT_NONE && d->act <= BS_ACT_SOFTPLUS_FAST, "bs_gemm: unknown activation %d", d->act);
    p.bias = d->bias; p.bias_group_rows = d->bias_group_rows; p.act = d->act; p.scale = d->scale;
    p.res = d->res; p.res2 = d->res2; p.res_dtype = d->res_dtype; p.ldr = d->ldr;
    BS_REQUIRE(!d->res2 || d->res, "bs_gemm: res2 needs res (they share ldr)");
    p.out = d->out; p.out2 = d->out2; p.out3 = d->out3; p.out_dtype = d->out_dtype; p.ldo = d->ldo; p.out_mode = d->out_mode;
    p.out_group_rows = d->out_group_rows; p.out_group_stride = d->out_group_stride; p.out_row_offset = d->out_row_offset;
    p.shuffle_s = d->shuffle_s; p.shuffle_cout = d->shuffle_cout;
    p.qkv_hidden = d->qkv_hidden; p.qkv_tokens = d->qkv_tokens; p.qkv_sp = d->qkv_sp; p.q_scale = d->q_scale;
    p.qkv_cls_last = d->qkv_cls_last;
    p.qkv_cls_rows = d->qkv_cls_rows;
    p.qkv_patch_row0 = d->qkv_patch_row0;
    p.qkv_lo_off = d->qkv_lo_off;
    p.out2_relu = d->out2_relu;
    // (the second output exists in the lean (hi16 | hi8 | lo8) epilogue only: the conditions below are that form's, igemm_kernel.h)
    BS_REQUIRE(!d->out2_relu || (d->out2 && d->out_mode == BS_OUT_PLAIN && d->out_f8 && !d->scale && !d->out_group_rows && !d->bias_group_rows &&
                                 !d->bias2 && (d->act == BS_ACT_NONE || d->act == BS_ACT_RELU) && d->out_split_off == d->N && d->N % 256 == 0 &&
                                 d->ldo % 8 == 0 && d->out_dtype == d->dtype && (!d->res || d->res_f8)),
               "bs_gemm: out2_relu needs the plain (hi16 | hi8 | lo8) output form (out_f8, N %% 256 == 0, no scale / row groups / bias2)");
    BS_REQUIRE(d->qkv_lo_off == 0 || (d->out_mode == BS_OUT_QKV && d->qkv_lo_off > 0 && d->qkv_lo_off % 8 == 0),
               "bs_gemm: qkv_lo_off needs BS_OUT_QKV and a multiple of 8 elements");
    p.f8_wonly_from = d->f8_wonly_from;
    p.out_lo8_rows = d->out_lo8_rows;
    p.f8_skip_from = d->f8_skip_from;
    p.out_planes_rows = d->out_planes_rows;
    BS_REQUIRE(d->out_planes_rows == 0 || (d->out_f8 && d->out_planes_rows > 0 && d->out_planes_rows % 256 == 0),
               "bs_gemm: out_planes_rows needs out_f8 and a multiple of 256 rows");
    p.bias2 = d->bias2;
    p.bias2_row0 = d->bias2_row0;
    p.bias2_group_rows = d->bias2_group_rows;
    BS_REQUIRE(d->f8_skip_from == 0 || (d->f8_seg > 0 && (d->f8_skip_from == -1 || (d->f8_skip_from > 0 && d->f8_skip_from % 256 == 0 && !d->conv))),
               "bs_gemm: f8_skip_from needs the FP8 correction segment and is -1 (every tile) or, for a plain GEMM, a multiple of 256 rows");
    BS_REQUIRE(!d->bias2 || (d->bias2_group_rows > 0 && d->bias2_row0 >= 0 && d->bias_group_rows == 0 && !d->conv && d->out_mode != BS_OUT_SHUFFLE &&
                             d->N % 4 == 0),
               "bs_gemm: bias2 needs bias2_group_rows > 0, a plain GEMM (PLAIN or QKV output) and no bias_group_rows");
    BS_REQUIRE(d->out_lo8_rows == 0 || (d->out_f8 && d->out_lo8_rows % 256 == 0), "bs_gemm: out_lo8_rows needs out_f8 and a multiple of 256 rows");
    BS_REQUIRE(d->f8_wonly_from == 0 || d->f8_seg > 0, "bs_gemm: f8_wonly_from needs the FP8 correction segment");
    BS_REQUIRE(d->qkv_cls_rows == 0 || (d->out_mode == BS_OUT_QKV && d->qkv_cls_last && d->qkv_tokens > 1 && d->qkv_patch_row0 >= d->qkv_cls_rows &&
                                        d->M == d->qkv_patch_row0 + d->qkv_cls_rows * (d->qkv_tokens - 1)),
               "bs_gemm: qkv_cls_rows needs BS_OUT_QKV, qkv_cls_last and M = qkv_patch_row0 + qkv_cls_rows * (tokens - 1)");
    if (d->out_mode == BS_OUT_SHUFFLE) {
        BS_REQUIRE(!d->conv && d->shuffle_s > 0 && d->shuffle_cout % 4 == 0 && d->N == d->shuffle_s * d->shuffle_s * d->shuffle_cout,
                   "bs_gemm: bad shuffle geometry");
        BS_REQUIRE(d->Hout > 0 && d->Wout > 0 && d->M % (d->Hout * d->Wout) == 0, "bs_gemm: shuffle needs the input grid in Hout/Wout");
    } else if (d->out_mode == BS_OUT_QKV) {
        BS_REQUIRE(d->out2 && d->out3 && d->qkv_hidden % 64 == 0 && d->N == 3 * d->qkv_hidden && d->qkv_tokens > 0 &&
                       (d->qkv_cls_rows > 0 || d->M % d->qkv_tokens == 0) && d->qkv_sp >= d->qkv_tokens && d->out_dtype == d->dtype,
                   "bs_gemm: bad qkv geometry");
    } else {
        BS_REQUIRE(d->out_mode == BS_OUT_PLAIN && d->ldo >= d->N, "bs_gemm: bad out_mode/ldo");
        BS_REQUIRE(d->ldo % 4 == 0, "bs_gemm: ldo must be a multiple of 4");
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int tile = d->tile % 100;   // tile ids >= 100 carry ablation bits in the hundreds (diagnostics)
    return dispatch(p, d->dtype, d->conv != 0, tile, st);
}
