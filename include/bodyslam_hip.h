/* bodyslam_hip.h -- C ABI of libbodyslam_hip.so (gfx950 / MI355X only).
 *
 * The reference (GuidoManni/BodySLAM) is pure Python with no FFI of its own: every op on its hot
 * path is a PyTorch-CUDA dispatch.  Each entry point below names the reference call it replaces
 * (paths relative to the reference repo; "HF" = transformers 5.15.0, the installed restatement of
 * the un-vendored isl-org/ZoeDepth network the reference pulls through torch.hub at
 * BodySLAM_Refactored/src/depth_estimation/interface.py:46).
 *
 * Conventions: plain pointers are DEVICE pointers unless marked host; `stream` is a hipStream_t
 * passed as void*; every function returns 0 or a negative bs_status and never throws; the library
 * allocates nothing per call (callers own every buffer; bs_init allocates one 4 KiB zero page).
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 */
#ifndef BODYSLAM_HIP_H
#define BODYSLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    BS_OK = 0,
    BS_ERR_INVALID = -1,   /* bad argument / unsupported shape */
    BS_ERR_HIP = -2,       /* a HIP runtime call failed; see bs_last_error() */
    BS_ERR_NOT_INIT = -3
} bs_status;

enum { BS_F32 = 0, BS_F16 = 1, BS_BF16 = 2 };
/* BS_ACT_SOFTPLUS: torch.nn.Softplus(beta = 1, threshold = 20) through libm (log1pf(expf(x))), what the oracle computes.
 * BS_ACT_SOFTPLUS_FAST: the same function on v_exp / v_log (relative error <= 4e-6): the attractor MLPs only, whose epilogue was bound by
 * libm's arithmetic and whose inputs are 16-bit hidden units anyway; the seed regressors (bin start values) and the reference
 * precision use the exact form. */
enum { BS_ACT_NONE = 0, BS_ACT_RELU = 1, BS_ACT_GELU = 2, BS_ACT_SOFTPLUS = 3, BS_ACT_SOFTPLUS_FAST = 4 };
enum { BS_OUT_PLAIN = 0, BS_OUT_SHUFFLE = 1, BS_OUT_QKV = 2 };

/* library ------------------------------------------------------------------------------------ */
int bs_init(int device);                 /* replaces .to("cuda") at interface.py:49-51, mpem_interface.py:34,60 */
const char* bs_last_error(void);
int bs_version(void);

/* implicit GEMM on MFMA ---------------------------------------------------------------------- *
 * out[m, n] = epilogue( sum_k A(m, k) * W[n, k] ),  A/W fp16 or bf16, fp32 accumulate.
 * Replaces every nn.Linear / nn.Conv2d / nn.ConvTranspose2d(k == stride) on the path:
 *   BEiT q/k/v/o_proj, fc1, fc2           HF modeling_beit.py:296-357
 *   patch embedding Conv16x16 s16         HF modeling_beit.py:63-90
 *   DPT readout / 1x1 / ConvT / 3x3 convs HF modeling_zoedepth.py:55-149,153-329,332-373
 *   metric-head 1x1 convs                 HF modeling_zoedepth.py:494-547,665-772
 *   CyclePose Conv7x7 / Conv3x3 s2        MPEM/architecture_v3.py:120-147
 * conv == 0: A is [M, K] with row stride lda.   conv == 1: A is an NHWC batch [B, Hin, Win, lda>=Cin],
 * K = KH*KW*Cin, M = B*Hout*Wout, zero padding pad_h/pad_w (may be negative: a crop).  K order, conv mode:
 * 64-channel chunk outermost, then the filter tap, then the chunk's channels -- W is [N][Cin/64][KH][KW][64]
 * (bodyslam_amd/_lib.py conv_weight()); this keeps the taps that re-read one input row a few K-steps apart, inside
 * the XCD's L2.  Cin (conv) / K (plain) must be a multiple of 64.
 * epilogue: y = acc + bias[(m / bias_group_rows) * N + n]  (bias_group_rows == 0: bias[n]);
 *           y = act(y); y *= scale[n]; y += res[orow * ldr + n] (orow = m unless regrouped, below); store.
 * out_mode BS_OUT_PLAIN  : out[orow * ldo + n], orow = (m / out_group_rows) * out_group_stride
 *                          + m % out_group_rows + out_row_offset (out_group_rows == 0: orow = m)
 *          BS_OUT_SHUFFLE: ConvTranspose2d(kernel == stride == shuffle_s): n = (ky*s + kx)*Cout + co,
 *                          out is NHWC [B, Hout*s, Wout*s, shuffle_cout]
 *          BS_OUT_QKV    : n = which*hidden + head*64 + d; tokens_per_image rows per image;
 *                          out -> Q [B,nh,Sp,64] (scaled by q_scale), out2 -> K [B,nh,Sp,64],
 *                          out3 -> V^T [B,nh,64,Sp]
 */
/* The (hi16 | hi8 | lo8) operand format of accurate mode's FP8 correction passes: a row of C features is
 * [round16(y) x C | e4m3(y * 2^BS_F8_ACT_HI_EXP) x C | e4m3((y - round16(y)) * 2^BS_F8_ACT_LO_EXP) x C] = 4C bytes.
 * Producers take the flag `| 32` on their dtype argument (bs_layernorm, bs_attention, bs_preprocess_patches; bs_layernorm also
 * `| (rows << 8)`: only rows below `rows` write the planes; bs_attention_table `| 64`: only the cls rows do) or
 * bs_gemm_desc.out_f8; the consumer is bs_gemm with f8_seg = 2C. */
#define BS_F8_ACT_HI_EXP 0
#define BS_F8_ACT_LO_EXP 11
typedef struct bs_gemm_desc {
    const void* A;
    const void* W;
    int32_t dtype;                 /* BS_F16 | BS_BF16 */
    int32_t M, N, K;
    int32_t lda;
    int32_t conv;
    int32_t Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad_h, pad_w;
    int32_t relu_a;                /* ReLU on A while loading (pre-activation residual unit) */
    const float* bias;
    int32_t bias_group_rows;
    int32_t act;
    const float* scale;
    const void* res;
    int32_t res_dtype;             /* BS_F32 | BS_F16 | BS_BF16 */
    int32_t ldr;
    const void* res2;              /* optional second residual, operand dtype, same ldr */
    void* out;
    void* out2;
    void* out3;
    int32_t out_dtype;
    int32_t ldo;
    int32_t out_mode;
    int32_t out_group_rows, out_group_stride, out_row_offset;
    int32_t shuffle_s, shuffle_cout;
    int32_t qkv_hidden, qkv_tokens, qkv_sp;
    float q_scale;
    int32_t tile;                  /* 0 = auto; else forces a tile variant (tests / tuning) */
    /* split-precision support (Numerics).  The K axis may consist of two segments that walk the SAME rows of A:
     * segment 0 (Cin channels per tap for conv, K - seg1 for plain) then segment 1 (seg1 channels per tap / K columns);
     * W holds both segments back to back along K.  With A = [hi | lo] and W = [W_hi | W_hi | W_lo] one launch evaluates
     * A_hi W_hi + A_lo W_hi + A_hi W_lo.  out_split_off > 0 stores the 16-bit output as a (hi, lo) pair, lo at that
     * element offset; res_split_off > 0 reads the 16-bit residual(s) as such pairs. */
    int32_t seg1;
    int32_t out_split_off;
    int32_t res_split_off;
    /* FP8 correction segment (plain GEMMs).  f8_seg > 0: every A row and every W row continues, after its K 16-bit values, with
     * f8_seg bytes of OCP e4m3 values that are multiplied on the block-scaled FP8 MFMA (2x the 16-bit rate).  They come in
     * two halves of f8_seg / 2 (a multiple of 128) each; half h contributes 2^(sa_h - 127 + sb_h - 127) * sum_k A8[k] W8[k].
     * f8_scales packs the four E8M0 exponents: sa0 | sb0 << 8 | sa1 << 16 | sb1 << 24.  With A = [hi16 | hi8 | lo8] and
     * W = [W_hi16 | W_lo8 | W_hi8] this is A_hi W_hi + A_hi W_lo + A_lo W_hi at 2 pass-equivalents instead of 3.
     * Conv mode: the pixel vector is [hi16 x Cin | hi8 x Cin | lo8 x Cin] (f8_seg = 2*Cin, Cin % 128 == 0) and W is
     * [W_hi16: chunk64, tap, 64][W_lo8: chunk128, tap, 128][W_hi8: ...] (bodyslam_amd/_lib.py f8_conv_weight()).
     * out_f8 != 0 (with out_split_off = channels, ldo = 2*channels, channels % 8 == 0): the output row / pixel is written in that A format,
     * [hi16 x N | e4m3(y * 2^ea) x N | e4m3((y - hi) * 2^el) x N], out_f8 = ea | el << 8. */
    int32_t f8_seg;
    uint32_t f8_scales;
    int32_t out_f8;
    int32_t res_f8;                /* != 0: res / res2 are rows in that format too (N channels, ldr >= 2N) */
    int32_t qkv_cls_last;          /* BS_OUT_QKV: token 0 of an image (cls) is stored at position tokens-1 of Q / K / V^T and token t
                                    * at position t-1 (patches first: the layout bs_attention_table reads) */
    int32_t qkv_cls_rows;          /* BS_OUT_QKV, > 0: the rows are GROUPED -- rows [0, qkv_cls_rows) are the cls tokens of the images,
                                    * row qkv_patch_row0 + b*(tokens-1) + t is patch t of image b (needs qkv_cls_last); the rows
                                    * in between are padding and are not stored */
    int32_t qkv_patch_row0;        /* >= qkv_cls_rows; M = qkv_patch_row0 + qkv_cls_rows * (tokens-1) */
    int32_t f8_wonly_from;         /* with f8_seg: 0 = every row gets both correction products; k > 0 = the 256-row tiles that start at
                                    * a row >= k evaluate only the first FP8 half (A_hi8 W_lo8, the weight-rounding correction) and
                                    * stop before the second (A_lo8 W_hi8, the activation-rounding correction); -1 = all rows.
                                    * Activation rounding is per-row noise that stays incoherent in the depth map except on the
                                    * cls-token rows, whose error shifts the whole map (Numerics): grouped cls rows + this
                                    * switch run the backbone at 1.5 instead of 2 pass-equivalents. */
    int32_t out_lo8_rows;          /* with out_f8, > 0: only output rows below this index store their lo8 plane (their consumer is a
                                    * GEMM with f8_wonly_from = this value, which never reads it on the other rows) */
    int32_t f8_skip_from;          /* with f8_seg, k > 0: tiles that start at a row >= k run NO FP8 stage (one 16-bit pass); their
                                    * weight-rounding error is corrected by its token-independent part instead, see bias2.  -1 = no tile
                                    * runs one (a neck product the calibration took down to one pass; also with conv) */
    int32_t bias2_row0;            /* bias2 applies to rows m >= bias2_row0 ... */
    int32_t bias2_group_rows;      /* ... with group (m - bias2_row0) / bias2_group_rows (> 0 when bias2 is given) */
    int32_t out_planes_rows;       /* with out_f8, > 0: 256-row tiles that start at a row >= this store the hi16 values only (no FP8
                                    * plane: their consumer sets f8_skip_from to this value, or -1 with 256 here) */
    const float* bias2;            /* optional fp32 [groups, N], added like the bias: y = acc + bias[n] + bias2[group, n].  The backbone
                                    * uses it for the rank-1 part of the weight-rounding error of a single-pass product:
                                    * A dW^T ~ 1 (mean_tokens(A) dW^T) per image (Numerics); excludes bias_group_rows */
    int32_t qkv_lo_off;            /* BS_OUT_QKV, > 0: the rounding residuals of Q, K and V^T (y - round16(y) as a second 16-bit value, unscaled)
                                    * are stored too, this many ELEMENTS behind the respective value in out / out2 / out3 (each tensor
                                    * allocated twice over): the operands of bs_attention_table_corr */
    int32_t out2_relu;             /* BS_OUT_PLAIN with out_f8, != 0: out2 receives relu(y) in the same (hi16 | hi8 | lo8) row format and geometry as
                                    * out (a second output of the epilogue).  The fusion stage's pre-activation residual units read both x (the skip)
                                    * and relu(x) (their first convolution's input): the producing convolution writes the two instead of a
                                    * bs_relu_split launch re-reading x (HF modeling_zoedepth.py:262-297) */
} bs_gemm_desc;
int bs_gemm(const bs_gemm_desc* d, void* stream);
/* the tile variant bs_gemm will pick for this descriptor (1: 128x128, 2: 128x64, 3: 128x32, 4: 256x128) */
int bs_gemm_tile(const bs_gemm_desc* d);

/* column means over a sample of the rows of each group: out[g, k] = mean_{j < rows_per_group, j % row_step == 0} A[row0 + g*rows_per_group + j, k]
 * for the 16-bit matrix A (row stride lda elements; the hi16 plane of a pair row), written as bf16 [groups, K] -- the A operand of
 * bs_rank1_bias, which forms bias2 above (mean over the patch tokens of an image; a sample of every 8th token changes the depth
 * result by < 1e-6 m, tools/probes/weight_mean_correction.py). */
int bs_col_mean(const void* A, int64_t lda, int32_t row0, int32_t rows_per_group, int32_t groups, int32_t row_step, int32_t K,
                void* out_bf16, float* zero_out, int64_t zero_n, int32_t dtype, void* stream);
/* out[g, n] += sum_k abar[g, k] * dW[n, k]: bf16 [G, K] x bf16 [N, K]^T accumulated into fp32 [G, N], which the caller has zeroed
 * (bs_col_mean clears `zero_out[0 .. zero_n)` for it).  With dW = W - round16(W) this is bias2 of the backbone's single-pass
 * products.  The K axis is cut in two halves that meet by atomics: two addends, so the result is order-independent. */
int bs_rank1_bias(const void* abar_bf16, const void* dw_bf16, float* out, int32_t G, int32_t N, int32_t K, void* stream);

/* BEiT attention: softmax(Q K^T + relpos_bias) V -------------------------------------------- *
 * HF modeling_beit.py:268-341 (eager_attention_forward with the additive relative-position bias
 * of :179-265).  The softmax is evaluated in the log2 domain: q [B,nh,Sp,64] must be pre-scaled by
 * log2(e)/sqrt(64) and bias pre-multiplied by log2(e).  k [B,nh,Sp,64], vt [B,nh,64,Sp], Sp % 64 == 0,
 * rows/cols >= S zero; bias fp32 [nh,Sp,Sp] with <= -1e30 in key columns >= S.
 * out [B*S, nh*64] token-major (the o_proj GEMM's A operand).
 * dtype | 16: out holds (hi | lo) pairs, [B*S, 2*nh*64]; dtype | 32: (hi16 | hi8 | lo8) rows of the same size. */
int bs_attention(const void* q, const void* k, const void* vt, const float* bias, void* out,
                 int32_t B, int32_t nh, int32_t S, int32_t Sp, int32_t dtype, void* stream);
/* The same attention for a window of hp x wp patches + cls with the bias taken from the per-head TABLE instead of a
 * materialised [nh,Sp,Sp] tensor: table fp32 [nh, (2hp-1)(2wp-1)+3], pre-multiplied by log2(e): HF's entries
 * (modeling_beit.py:194-218: (dy+hp-1)*(2wp-1) + (dx+wp-1) for patch pairs, then cls->patch, patch->cls, cls->cls) with the
 * (2hp-1)(2wp-1) patch-pair entries stored in REVERSED order (entry i at index nbody-1-i: the 8 consecutive keys a lane
 * handles are then 8 ascending words), the three cls entries last, unchanged.
 * q / k / vt hold the tokens of an image patches first, cls LAST (S = hp*wp + 1; bs_gemm_desc.qkv_cls_last); out rows are in
 * the usual order (cls first per image), or with grouped > 0 the B cls rows first and the hp*wp patch rows of image b from row
 * grouped + b*hp*wp on (bs_gemm_desc.qkv_cls_rows / qkv_patch_row0).  Built for wp == 32 (every 512-wide network input). */
int bs_attention_table(const void* q, const void* k, const void* vt, const float* table, void* out,
                       int32_t B, int32_t nh, int32_t hp, int32_t wp, int32_t Sp, int32_t grouped, int32_t dtype, void* stream);
/* The same with SPLIT-PRECISION operands: q_lo / k_lo / vt_lo hold x - round16(x) of the respective tensor (same shapes; written by
 * bs_gemm with qkv_lo_off), and the kernel evaluates  S = Q K^T + Q_lo K^T + Q K_lo^T,  O = V P + V P_lo + V_lo P  on the 16-bit MFMA
 * (three passes into the same fp32 accumulators; P split in registers).  For weights whose LayerNorm outputs carry outlier channels
 * -- trained BEiT checkpoints -- the single 16-bit Q / K / V / P of bs_attention_table cost 1-3e-4 m of depth
 * (tools/probes/outlier_rounding_study.py); ZoeDepthEngine.calibrate chooses between the two per weight set.  hp must be even. */
int bs_attention_table_corr(const void* q, const void* k, const void* vt, const void* q_lo, const void* k_lo, const void* vt_lo,
                            const float* table, void* out, int32_t B, int32_t nh, int32_t hp, int32_t wp, int32_t Sp, int32_t grouped,
                            int32_t dtype, void* stream);

/* LayerNorm over the last dim, fp32 in; out16 (fp16/bf16, nullable) and out32 (fp32, nullable, may
 * alias x) -- HF modeling_beit.py:418,432; post-norm of the router HF modeling_zoedepth.py:876-881
 * dtype | 16: out16 holds (hi | lo) pairs, [rows, 2*cols] (bs_cast_split's format); dtype | 32: (hi16 | hi8 | lo8) rows. */
int bs_layernorm(const float* x, const float* gamma, const float* beta, void* out16, float* out32, int32_t rows,
                 int32_t cols, float eps, int32_t dtype, void* stream);

/* device-to-device copy on the stream (re-arming the router's positional-encoding buffer) */
int bs_copy_f32(const float* src, float* dst, int64_t n, void* stream);

/* split-precision helpers: fp32 [rows, cols] -> 16-bit (hi | lo) pairs [rows, 2*cols] with x = hi + lo to ~22 bits; ReLU of such a
 * tensor (the pre-activation residual units, HF modeling_zoedepth.py:225-241).  bs_resize_bilinear_nhwc takes such tensors when
 * bit 1 of align_corners is set; bs_logbinom_depth takes a (hi | lo) `last` when bit 4 of dtype is set.
 * With `| 32` on the dtype argument (bit 2 of align_corners for the resize) the same calls work on the (hi16 | hi8 | lo8)
 * format of the FP8 correction passes (BS_F8_ACT_*_EXP above); bs_relu_split with `| 32 | 64` leaves the lo8 plane alone (neither read
 * nor written: an output whose only reader runs the weight-rounding correction only), with `| 32 | 128` both FP8 planes (its only reader
 * runs one 16-bit pass). */
int bs_cast_split(const float* x, void* out, int64_t rows, int32_t cols, int32_t out_dtype, void* stream);
int bs_relu_split(const void* x, void* out, int64_t rows, int32_t cols, int32_t dtype, void* stream);

/* fp32 -> fp16/bf16 cast (tap copies of the residual stream) */
int bs_cast(const float* x, void* out, int64_t n, int32_t out_dtype, void* stream);

/* MDEM pre-processing: uint8 frames -> im2col'ed patch matrix --------------------------------- *
 * HF image_processing_pil_zoedepth.py:181-232 (upstream depth_model.py#L57): /255, reflect pad,
 * bilinear align_corners=True resize to (nh, nw), (x-0.5)/0.5; fused with the 16x16 patch gather
 * of HF modeling_beit.py:83-90.  frames [B,H,W,3] u8; out [2B or B][(nh/16)*(nw/16)][3*16*16]
 * (k = c*256 + ky*16 + kx); images B..2B-1 are the W-flipped copies when flip != 0.
 * out_dtype | 16: rows are (hi | lo) pairs, [., 2*768]; out_dtype | 32: (hi16 | hi8 | lo8) rows. */
int bs_preprocess_patches(const uint8_t* frames, void* out, int32_t B, int32_t H, int32_t W, int32_t nh,
                          int32_t nw, int32_t flip, int32_t out_dtype, void* stream);
/* the same pre-processing to a plain NCHW fp32 image (tests) */
int bs_preprocess_image(const uint8_t* frames, float* out, int32_t B, int32_t H, int32_t W, int32_t nh,
                        int32_t nw, int32_t flip, void* stream);

/* rows[b*rows_per_image + 0, :] = v  (cls token, HF modeling_beit.py:166-167) */
int bs_fill_rows(float* x, const float* v, int32_t B, int32_t rows_per_image, int32_t cols, void* stream);

/* bilinear resize of an NHWC fp16/bf16 map -- F.interpolate calls at HF modeling_zoedepth.py:259,319,360.  `align_corners`: bit 0 the flag itself,
 * bit 1 the tensors hold (hi | lo) 16-bit pairs, bit 2 (hi16 | hi8 | lo8) rows, bit 3 (with bit 2) the OUTPUT's lo8 plane is not written (for a map
 * whose every consumer runs the weight-rounding correction only: nobody reads that plane), bit 4 (with bit 2) neither FP8 plane is (every consumer
 * runs one 16-bit pass) */
int bs_resize_bilinear_nhwc(const void* x, void* out, int32_t B, int32_t Hin, int32_t Win, int32_t C,
                            int32_t Hout, int32_t Wout, int32_t align_corners, int32_t dtype, void* stream);

/* out = relu(bilinear resize(x, align_corners) + bias[c]): the second half of a 1x1 convolution + ReLU whose input is an upsampled map -- the
 * bins head's projectors read the fusion stage's x2 outputs (HF modeling_zoedepth.py:749-772 on :316-322's maps); the convolution is linear and
 * commutes with the resize, so it runs at the LOW resolution (a quarter of the pixels, bs_gemm without bias) and this call upsamples its output.
 * x, out: NHWC 16-bit rows of C values, or (hi | lo) pairs (flags bit 1); flags bit 0 (align_corners) is required; bias fp32 [C]. */
int bs_resize_bias_relu_nhwc(const void* x, const float* bias, void* out, int32_t B, int32_t Hin, int32_t Win, int32_t C, int32_t Hout,
                             int32_t Wout, int32_t flags, int32_t dtype, void* stream);

/* conv3x3(pad 1)(interpolate x2(x)) evaluated from tap products taken at the low resolution -- the relative head's
 * `upsample` + `conv2` (+ ReLU), HF modeling_zoedepth.py:358-362.  y fp32 [B, Hin, Win, 9*Cout] holds, per low-resolution pixel,
 * y[(ky*3+kx)*Cout + o] = sum_c W[o, c, ky, kx] x[c] (one bs_gemm over the low-resolution map); this call gathers, per output pixel
 * p and tap d, the bilinear sample of y's tap plane at p + d (zero when p + d lies outside the Hout x Wout map: the conv's zero
 * padding), adds bias fp32 [Cout], applies ReLU when `relu`, and writes an NHWC 16-bit map [B, Hout, Wout, Cout] (`align_corners`
 * bits 1 / 2 select the (hi | lo) / (hi16 | hi8 | lo8) pair formats as for bs_resize_bilinear_nhwc). */
int bs_upconv_tapsum(const float* y, const float* bias, void* out, int32_t B, int32_t Hin, int32_t Win, int32_t Cout,
                     int32_t Hout, int32_t Wout, int32_t align_corners, int32_t relu, int32_t dtype, void* stream);

/* The same relu(conv3x3(pad 1)(interpolate x2, align_corners(x))) in ONE launch, from the low-resolution input itself (round 5): the tap
 * products exist only as LDS tiles (bodyslam_amd/csrc/upconv_fused.hip).  Replaces bs_gemm (the tap products) + bs_upconv_tapsum for the
 * relative head's geometry: Cin = 128, Cout = 32, Hout = 2 Hin, Wout = 2 Win.  x: NHWC rows of Cin 16-bit values (mode 0) or
 * (hi16 | hi8 | lo8) rows (mode 1: weight-rounding correction only, mode 2: both corrections); w: [9*Cout] rows, n = (ky*3+kx)*Cout + o, of
 * Cin 16-bit values (mode 0) or [W_hi16 | W_lo8 | W_hi8] (bs_gemm's FP8 correction packing; sa0, sb0, sa1, sb1 = its f8_scales bytes).
 * flags: bit 0 align_corners (required), bits 1 / 2 the output pair format as for bs_upconv_tapsum (modes 1, 2 need one). */
int bs_upconv_fused(const void* x, const void* w, const float* bias, void* out, int32_t B, int32_t Hin, int32_t Win, int32_t Cin,
                    int32_t Cout, int32_t Hout, int32_t Wout, int32_t flags, int32_t relu, int32_t mode, int32_t sa0, int32_t sb0,
                    int32_t sa1, int32_t sb1, int32_t dtype, void* stream);

/* metric-bins head ---------------------------------------------------------------------------- *
 * Both heads (nyu | kitti) are carried side by side as channel groups; `route` int32 [B] (from
 * bs_route_argmax) says which group an image uses -- per image, because the reference always runs
 * the network with batch 1 (interface.py:61), never HF's batch-summed vote (modeling_zoedepth.py:1063-1067).
 *
 * attractor step, HF modeling_zoedepth.py:665-746 (unnormed, memory_efficient, kind "mean",
 * inv_attractor defaults alpha=300 gamma=2): bins_out[b,y,x,g,:] = c + mean_a dx/(1+300 dx^2),
 * c = bilinear(align_corners=True) resample of bins_prev to (H, W), dx = A[b,y,x,g,a] - c.
 * A fp32 [B,H,W,groups*n_attr] (softplus already applied), bins fp32 NHWC [.,groups*n_bins];
 * route nullable (null: every group is computed). */
int bs_attractor_step(const float* A, const float* bins_prev, float* bins_out, const int32_t* route, int32_t B,
                      int32_t Hp, int32_t Wp, int32_t H, int32_t W, int32_t groups, int32_t n_bins, int32_t n_attr,
                      void* stream);
/* The attractor MLP in one launch, HF modeling_zoedepth.py:665-700 (`Conv2d(E, 2E, 1)`, ReLU, `Conv2d(2E, n_attractors, 1)`, softplus;
 * both metric heads' MLPs stacked):  out = act2(round16(relu(x W1^T + b1)) W2^T + b2).
 * x rows of ldx 16-bit values (the first K1 are read), W1 [N1, K1], W2 [N2, N1] 16-bit, b1 / b2 fp32, out fp32 [M, N2].
 * Built for K1 = 128, N1 = 256, N2 a multiple of 4 up to 32; bit-identical to bs_gemm(act = ReLU, 16-bit out) followed by
 * bs_gemm(act = act2, fp32 out) -- the hidden map (M x 256) never reaches memory. */
int bs_mlp2(const void* x, int32_t ldx, const void* W1, const float* b1, const void* W2, const float* b2, float* out, int32_t M,
            int32_t K1, int32_t N1, int32_t N2, int32_t act2, int32_t dtype, void* stream);
/* bs_add_resized + bs_mlp2 in one launch (an attractor level, HF modeling_zoedepth.py:726-730 then :665-700): the MLP's input row is
 * formed in the kernel, x[m] = round16(emb[m] + bilinear_align_corners(prev)[m]), and never stored.  emb [B,H,W,K1] and prev [B,Hp,Wp,K1]
 * 16-bit NHWC; dtype bit 4 (| 16): both hold (hi | lo) pairs of K1 channels each (pixel stride 2 K1), the sum uses hi + lo.
 * Bit-identical to bs_add_resized followed by bs_mlp2 on its hi half. */
int bs_mlp2_add(const void* emb, const void* prev, const void* W1, const float* b1, const void* W2, const float* b2, float* out,
                int32_t B, int32_t Hp, int32_t Wp, int32_t H, int32_t W, int32_t K1, int32_t N1, int32_t N2, int32_t act2, int32_t dtype,
                void* stream);
/* One level of the bins head's projector path in one launch (round 6; replaces, for that level, bs_resize_bias_relu_nhwc + the projector's
 * second 1x1 convolution as a 3-pass bs_gemm + the sum inside bs_mlp2_add + -- at the last level -- the log-binomial embedding bs_gemm; HF
 * modeling_zoedepth.py:749-772 (Projector), :726-730 (the attractor level's input), :376-491 (the conditional log-binomial MLP's first layer):
 *   e1 = relu(bilinear_align_corners(z) + b_c1);  emb = W_c2 e1 + b_c2;  x = round16(emb + bilinear_align_corners(emb_prev));  Eh = W_e e1 + b_e
 * z [B,Hl,Wl,2 PM] and emb_prev [B,Hl,Wl,2 E]: (hi | lo) 16-bit pairs on the SAME low-resolution grid; W_c2 [E, 3 PM], W_e [NE, 3 PM]: rows
 * [W_hi | W_hi | W_lo] (the 3-pass pair packing of bs_gemm's seg1 form); biases fp32.  Outputs: x_out [B,H,W,E] 16-bit (the attractor MLP's
 * input: bs_mlp2 with ldx = E), emb_out [B,H,W,2 E] (hi | lo) pairs or null (the last level hands no embedding on), eh_out fp32 [B,H,W,NE] or
 * null (then W_e / b_e are null too).  Built for PM = 64, E = 128, NE a multiple of 16 up to 80, W a multiple of 32.  The products run the K
 * order of the bs_gemm they replace: Eh carries the same bits; x differs from the two-launch path by the pair rounding of emb it no longer makes. */
int bs_projector_level(const void* z, const float* b_c1, const void* emb_prev, const void* Wc2, const float* b_c2, const void* We,
                       const float* b_e, void* x_out, void* emb_out, float* eh_out, int32_t B, int32_t Hl, int32_t Wl, int32_t H, int32_t W,
                       int32_t PM, int32_t E, int32_t NE, int32_t dtype, void* stream);
/* out[b,y,x,:] = x[b,y,x,:] + bilinear_align_corners(prev)[b,y,x,:] (fp16/bf16 NHWC); HF :726-730.
 * dtype bit 4 (| 16): x, prev and out hold (hi | lo) pairs of C channels each (pixel stride 2C), see bs_cast_split. */
int bs_add_resized(const void* x, const void* prev, void* out, int32_t B, int32_t Hp, int32_t Wp, int32_t H,
                   int32_t W, int32_t C, int32_t dtype, void* stream);
/* conditional log-binomial + expectation, HF modeling_zoedepth.py:376-491,1086-1101, per output pixel:
 *   h = gelu(bilinear(Eh)[g] + W0_last[g] . last)     Eh fp32 [B,He,We,2*40] = W0_emb . emb + b0 (a bs_gemm; the
 *                                                      1x1 conv commutes with the bilinear resample)
 *   pt = softplus(W2[g] h + b2[g]); p, T; y_k = logC(63,k) + k log p + (63-k) log(1-p)
 *   depth = sum_k softmax(y/T)_k * bilinear(bins)[g,k]
 * last 16-bit [B,H,W,32]; bins fp32 [B,He,We,2*64]; w0_last [2,40,32], w2 [2,4,40], b2 [2,4] fp32. */
int bs_logbinom_depth(const void* last, const float* Eh, const float* bins, const float* w0_last, const float* w2,
                      const float* b2, const int32_t* route, float* depth, int32_t B, int32_t H, int32_t W,
                      int32_t He, int32_t We, float min_temp, float max_temp, int32_t dtype, void* stream);
/* The same with the hidden width as an argument (40: NK head, 80: the single-head ZoeD_N / ZoeD_K models, whose Eh is
 * [B,He,We,2*80] and w0_last / w2 are [2,hid,32] / [2,4,hid]) and the single head's 33rd MLP input, the relative depth
 * relu(conv3(last)) (HF modeling_zoedepth.py:367-371,1186-1191): rel_w (nullable) = per group
 * [W0 column of that input x hid | conv3 weight x 32 | conv3 bias]. */
int bs_logbinom_depth_ex(const void* last, const float* Eh, const float* bins, const float* w0_last, const float* w2,
                         const float* b2, const float* rel_w, int32_t hid, const int32_t* route, float* depth, int32_t B,
                         int32_t H, int32_t W, int32_t He, int32_t We, float min_temp, float max_temp, int32_t dtype,
                         void* stream);
/* domain router pieces, HF modeling_zoedepth.py:775-962: multi-head attention of the 4-layer patch
 * transformer (head_dim 32, S = 1 + h*w tokens, no mask) on fused fp32 qkv [B*S, 3*D] -> out 16-bit
 * [B*S, D]; the linear layers run on bs_gemm and the post-norms on bs_layernorm. */
int bs_small_attention(const float* qkv, void* out, int32_t B, int32_t S, int32_t nheads, int32_t dtype, void* stream);
/* route[b] = argmax(logits[b, 0:2]) (first maximal index, as torch.argmax), HF :1066-1067 */
int bs_route_argmax(const float* logits, int32_t ld, int32_t* route, int32_t B, void* stream);

/* MDEM post-processing: HF image_processing_pil_zoedepth.py:234-341 + upstream infer_pil:
 * average d[b] with the un-flipped d[B+b], bicubic (A=-0.75, align_corners=False) resize of the
 * (nh, nw) map to the padded size, crop, write metres fp32 [B,H,W] and (nullable) uint16 = trunc(m*256). */
int bs_postprocess_depth(const float* depth_net, float* depth_m, uint16_t* depth_u16, int32_t B, int32_t H,
                         int32_t W, int32_t nh, int32_t nw, int32_t flip, void* stream);

/* MPEM (CyclePose pose branch) ---------------------------------------------------------------- */
/* mpem_interface.py:40-44,85-94 + architecture_v3.py:120-122: center-crop 128, /255, (x-.5)/.5, pair
 * concat, ReflectionPad(3) and 7x7 im2col: out [P*128*128, 320] (k = (ky*7+kx)*6 + c, zero padded 294..319) */
int bs_cyclepose_im2col(const uint8_t* frames, const int32_t* pairs, void* out, int32_t P, int32_t H, int32_t W,
                        int32_t dtype, void* stream);
/* the same over an arbitrary CH x CW window at (top, left) of every frame (type_of_trans='resize', mpem_interface.py:45-50,88-90:
 * the frames arrive already resized to 128 x W' by PIL and the window is the whole frame); rows [P*CH*CW, 320 (x2 with | 16)] */
int bs_cyclepose_im2col_window(const uint8_t* frames, const int32_t* pairs, void* out, int32_t P, int32_t H, int32_t W,
                               int32_t top, int32_t left, int32_t CH, int32_t CW, int32_t dtype, void* stream);
/* InstanceNorm2d(eps, no affine) + ReLU on an NHWC map, fp32 in -> fp16/bf16 out (+ optional fp32 copy)
 * architecture_v3.py:123-124,134-137.  scratch: >= P * ceil(HW/256) * 2 * C floats (per-chunk mean / M2). */
int bs_instnorm_relu_nhwc(const float* x, void* out, float* out_f32, float* scratch, int32_t P, int32_t HW, int32_t C,
                          float eps, int32_t dtype, void* stream);
/* AdaptiveAvgPool2d(1) of an NHWC fp32 map -> [P, C] fp32 (architecture_v3.py:146) */
int bs_avgpool_nhwc(const float* x, float* out, int32_t P, int32_t HW, int32_t C, void* stream);
/* pose head: skip_linear(cat[pooled, flatten_NCHW(x2)]) + pose_dense(pooled) -> quaternion normalise
 * -> 4x4 (architecture_v3.py:205-226, geometry_utils.py:230-265).  x2 is NHWC fp32 [P,32,32,256]; w_skip_x2 is
 * the skip weight re-laid to [7][HW][C]; outputs pose7 [P,7] and T [P,16] fp32.  The 262 144-long dot
 * products are split-K partial sums combined in a fixed order (bitwise reproducible). */
int bs_cyclepose_head(const float* pooled, const float* x2, const float* w_skip_pool, const float* w_skip_x2,
                      const float* b_skip, const float* w1, const float* b1, const float* w2, const float* b2,
                      float* pose7, float* T, float* scratch /* >= P*ceil(HW*C/4096)*8 floats */, int32_t P,
                      int32_t HW, int32_t C, void* stream);

/* 3DM ----------------------------------------------------------------------------------------- */
/* 3DM/scaling_system.py:72-77 pixel_to_3d + RGBD constants of 3DM/slam_utils.py:173,212-220,232.
 * depth u16 [B,H,W]; K = fx,fy,cx,cy (host); poses fp64 [B,16] device, nullable; xyz fp32 [B,H*W,3],
 * idx int32 [B,H*W] (row-major order of the valid pixels), count int32 [B]; scratch int32 >= B*(H*W/256+2). */
int bs_backproject(const uint16_t* depth, int32_t B, int32_t H, int32_t W, const double* K_host, double depth_scale,
                   double depth_trunc, const double* poses, float* xyz, int32_t* idx, int32_t* count,
                   int32_t* scratch, void* stream);
/* 3DM/scaling_system.py:72-77 pixel_to_3d verbatim on n (u, v, depth) fp64 triples -> (x, y, z) fp64 */
int bs_pixel_to_3d(const double* uvd, int64_t n, const double* K_host, double* out, void* stream);
/* 3DM/slam_utils.py:110-122 compute_curr_estimate_global_pose chained over N relatives (fp32 [N,16]) from
 * g0 (fp64 [16], host, nullable = identity) -> g_abs fp64 [N+1,16]; per-step SO(3) projection
 * (slam_utils.py:93-108).  With G_{i-1} a rotation, proj(G_{i-1}.R R_i) = G_{i-1}.R proj(R_i): every relative pose is
 * projected independently (one lane per pose), then the chain is a wave-shuffle prefix product over SE(3); matrices whose
 * bottom row is not [0 0 0 1] take the strict one-lane recurrence. */
int bs_pose_chain(const float* t_rel, int32_t N, const double* g0_host, double* g_abs, void* stream);
/* the same with g0 on the DEVICE (fp64 [16]; it may be the last pose of a previous call's g_abs: a sequence processed in
 * steps continues its chain without a host round trip -- 3DM/slam.py:148-153 keeps current_global_extrinsic_matrix) */
int bs_pose_chain_from(const float* t_rel, int32_t N, const double* g0_dev, double* g_abs, void* stream);

/* TSDF map (N4) ---------------------------------------------------------------------------------- *
 * Open3D ScalableTSDFVolume as wrapped by BodySLAM_not_refactored/3DM/tsdf.py:5-52 (voxel_length 0.001, sdf_trunc 0.1, RGB8,
 * volume_unit_resolution 32, depth_sampling_stride 8; integrate per frame at 3DM/slam.py:117,179, extract_point_cloud at :126,195).
 * State, all in device memory and owned by the caller:
 *   blocks      one per volume unit: res^3 voxels x 5 fp32 (tsdf, weight, r, g, b), voxel x*res^2 + y*res + z, zero-filled; block s
 *               lives at slab_base[s / slab_units] + (s % slab_units) * res^3 * 20 bytes (slab_base: int64 device array)
 *   unit table  open addressing: table_keys int64 [table_cap] (-1 = empty; table_cap a power of two), table_slots int32 [table_cap]
 *               (-1 until a block is assigned), table_stamp int32 [table_cap]; unit_index int32 [max_units, 3]
 *   counters    int32 [3]: number of units (persists across calls), units touched by the last bs_tsdf_touch, overflow flag
 *               (1: table full, 2: more new units than blocks -- sticky: set by any frame since the caller last cleared it, so a
 *               stream of frames can be checked once at its end)
 *
 * bs_tsdf_touch: ScalableTSDFVolume::Integrate's unit discovery -- every unit that meets the +-sdf_trunc box of a point of the
 *   depth image sampled every `stride` pixels (depth fp32 [H, W] metres, <= 0 invalid; pose = rows 0..2 of the camera->world 4x4,
 *   i.e. extrinsic^-1; K = (fx, fy, cx, cy); both host doubles) is inserted, gets a block number and -- unless its bounding sphere
 *   lies wholly outside the view frustum, where the integration updates nothing -- goes on `touched` (int32 [max_units]);
 *   frame_id must differ from call to call.
 * bs_tsdf_integrate: UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier on the units of `touched`
 *   (extrinsic = rows 0..2 of the world->camera 4x4; color u8 [H, W, 3] or NULL).  Their number is read on the device from
 *   n_touched_dev (= counters + 1); the host value n_touched only sizes the grid (any value >= 1 works), so a stream of frames is
 *   integrated without reading anything back.  The blocks handed out must exist: bs_tsdf_touch's max_units is the number of blocks
 *   the caller has allocated (it raises overflow flag 2 rather than hand out a block that does not exist).
 * bs_tsdf_extract: ScalableTSDFVolume::ExtractPointCloud (points, colours, normals) over blocks 0 .. units-1 in two passes:
 *   points == NULL counts into unit_count int32 [units]; otherwise unit_offset int64 [units] (exclusive prefix sums of the counts)
 *   places each unit's points, points / colors fp32 [total, 3]; normals fp32 [total, 3] or NULL: ScalableTSDFVolume::GetNormalAt,
 *   the normalised central difference of the trilinearly interpolated tsdf (GetTSDFAt) at +-0.99 voxel_length. */
int bs_tsdf_touch(const float* depth, int32_t H, int32_t W, int32_t stride, const double* K, const double* pose, double unit_length,
                  double sdf_trunc, void* table_keys, int32_t* table_slots, int32_t* table_stamp, int32_t table_cap, int32_t frame_id,
                  int32_t* unit_index, int32_t max_units, int32_t* counters, int32_t* touched, void* stream);
int bs_tsdf_integrate(const float* depth, const uint8_t* color, int32_t H, int32_t W, const double* K, const double* extrinsic,
                      const int32_t* unit_index, const int32_t* touched, int32_t n_touched, const int64_t* slab_base, int32_t slab_units,
                      int32_t res, double voxel_length, double sdf_trunc, const int32_t* n_touched_dev, void* stream);
/* A batch of frames at once (the reference integrates frame by frame, 3DM/slam.py:179; a voxel's running mean takes the frames in
 * order and nothing else orders them, so up to BS_TSDF_BATCH_MAX frames are integrated by loading every touched voxel once, applying
 * the frames that touch its unit in ascending order in registers and storing it once -- bit for bit what that many
 * bs_tsdf_touch + bs_tsdf_integrate calls leave in the blocks, in three launches).
 *   bs_tsdf_frames_upload    writes the batch's frame records (BS_TSDF_FRAME_BYTES each) to frames_dev: depth[f] / color[f] = the frames' device
 *                            images (color NULL or all entries NULL: no colours), K = (fx, fy, cx, cy), extrinsics / poses = host
 *                            doubles [n_frames, 16]: the world->camera 4x4 of every frame and its inverse
 *   bs_tsdf_touch_batch      unit discovery of all frames: inserts the units and ORs bit f into table_fmask (uint64 [table_cap], zero
 *                            between batches) of every unit frame f touches.  No blocks are handed out: the caller reads counters[0]
 *                            and counts the entries with a key but no slot, makes the missing blocks, then calls
 *   bs_tsdf_integrate_batch  block assignment (max_units = the blocks that exist; overflow flag 2 otherwise), the per-frame frustum
 *                            test, unit_mask (uint64 [max_units]) and `touched`, then the integration; counters[1] = the units updated;
 *                            table_fmask is zero again afterwards */
#define BS_TSDF_BATCH_MAX 64
#define BS_TSDF_FRAME_BYTES 256
int bs_tsdf_frames_upload(const float* const* depth, const uint8_t* const* color, const double* K, const double* extrinsics, const double* poses,
                          int32_t n_frames, void* frames_dev, void* stream);
int bs_tsdf_touch_batch(const void* frames_dev, int32_t n_frames, int32_t H, int32_t W, int32_t stride, double unit_length, double sdf_trunc,
                        void* table_keys, void* table_fmask, int32_t table_cap, int32_t* counters, void* stream);
int bs_tsdf_integrate_batch(const void* frames_dev, int32_t n_frames, int32_t H, int32_t W, const void* table_keys, int32_t* table_slots,
                            void* table_fmask, int32_t table_cap, int32_t* unit_index, int32_t max_units, int32_t* counters, int32_t* touched,
                            void* unit_mask, const int64_t* slab_base, int32_t slab_units, int32_t res, double voxel_length, double sdf_trunc,
                            void* stream);
int bs_tsdf_extract(const int32_t* unit_index, int32_t units, const void* table_keys, const int32_t* table_slots, int32_t table_cap,
                    const int64_t* slab_base, int32_t slab_units, int32_t res, double voxel_length, int32_t* unit_count,
                    const int64_t* unit_offset, float* points, float* colors, float* normals, void* stream);

/* ScalableTSDFVolume::ExtractTriangleMesh (BodySLAM_not_refactored/3DM/tsdf.py:42-52, called on the last frame at 3DM/slam.py:189-193):
 * marching cubes over every voxel cube of blocks 0 .. units-1, in two passes like bs_tsdf_extract.  mc_tab (device int32) = the case
 * table [256][tri_width] (-1 terminated lists of cube-edge ids, three per triangle) followed by the edge corners [12][2]
 * (bodyslam_amd/marching_cubes.py: corner c at offset (c & 1, c >> 1 & 1, c >> 2 & 1), bit c of the case = tsdf < 0).  Count pass
 * (vertices == NULL): unit_count int32 [units] = triangles per unit.  Write pass: unit_offset int64 [units] (exclusive prefix sums,
 * in triangles); per triangle corner k of triangle t: vertices / colors fp32 [3 * total, 3] and vertex_keys int64 [3 * total], the
 * identity of the cube edge the vertex lies on -- equal keys are one vertex of the mesh.  *err is set when a voxel coordinate does
 * not fit the key's 20 bits per axis. */
int bs_tsdf_mesh(const int32_t* unit_index, int32_t units, const void* table_keys, const int32_t* table_slots, int32_t table_cap,
                 const int64_t* slab_base, int32_t slab_units, int32_t res, double voxel_length, const int32_t* mc_tab, int32_t tri_width,
                 int32_t* unit_count, const int64_t* unit_offset, float* vertices, float* colors, int64_t* vertex_keys, int32_t* err, void* stream);

/* dense RGB-D odometry (N3) -------------------------------------------------------------------- *
 * The role of Open3D's rgbd_odometry_multi_scale (Method.Hybrid, 20 / 10 / 5 iterations) at
 * BodySLAM_not_refactored/3DM/visual_odometry.py:97-120; the algorithm is stated in oracle/rgbd_odometry_ref.py (parity with Open3D
 * unpinned).  Images are fp32 [H, W] in device memory, NaN = invalid depth; K = (fx, fy, cx, cy) and T (rows 0..2 of the
 * source -> target 4x4) are host doubles.
 *   bs_odo_prepare     intensity = (0.299 R + 0.587 G + 0.114 B) / 255 of color u8 [H, W, 3]; depth metres, <= 0 or > depth_max -> NaN
 *   bs_odo_pyrdown     next pyramid level [(H+1)/2, (W+1)/2]: [1 4 6 4 1]^2 / 256 at the even pixels, replicate borders; is_depth: the
 *                      weights run over the neighbours within depth_threshold of the centre, NaN where the centre is invalid
 *   bs_odo_sobel       3x3 Sobel / 8 in x and y, replicate borders, NaN propagates
 *   bs_odo_accumulate  the sums of one Gauss-Newton step at pose T: out29 = [21 upper-triangle terms of sum w J^T J (row-major),
 *                      6 terms of sum w J^T r, the weighted cost, the inlier count]; hybrid residuals (intensity + depth), Huber
 *                      weights, target sampled bilinearly; partial = scratch double [ceil(H*W/256), 29]; deterministic.
 *                      flags bit 0: the target images are read at the NEAREST pixel of the projected point (round half away from zero),
 *                      Open3D's association; bit 1: Open3D's robust step -- sum J^T J unweighted, sum J^T huber'(r) with
 *                      huber'(r) = r clipped to +-delta, cost = sum huber(r)
 * `batch` (prepare / pyrdown / sobel / step): the images are `batch` contiguous [H, W] images, processed by ONE launch per stage.  The
 * frame-to-frame pairs of a sequence are independent of each other (every pair starts from the identity), so a batch of frames
 * is tracked as `batch` simultaneous pairs: bs_odo_step's pair p reads source image p of the src_* stacks and target image p of the
 * tgt_* stacks (for consecutive frames held in one [slots, H, W] stack: src = the stack + H*W, tgt = the stack), its pose is
 * T_dev[12 p ..], its scratch partial[p][.][29], its sums out29[29 p ..].  The sums of a pair do not depend on the batch. */
/* the pseudo-RGBD depth of the 3DM loop (3DM/slam_utils.py:212-220): out = u16 / depth_scale as fp32 metres, values >= depth_trunc -> 0 */
int bs_depth_u16_to_m(const uint16_t* depth_u16, int64_t n, double depth_scale, double depth_trunc, float* out, void* stream);
/* `iterations` Gauss-Newton steps of one pyramid level entirely on the device: bs_odo_accumulate's sums at the pose in T_dev (12
 * doubles in device memory, rows 0..2 of source -> target), delta = -(A + 1e-12 I)^-1 b, T_dev <- exp(delta) T_dev; a step with
 * fewer than 6 inliers leaves T_dev alone.  No host round trip between steps. */
int bs_odo_step(const float* src_intensity, const float* src_depth, const float* tgt_intensity, const float* tgt_depth,
                const float* tgt_dIx, const float* tgt_dIy, const float* tgt_dDx, const float* tgt_dDy, int32_t batch, int32_t H, int32_t W,
                const double* K, double* T_dev, int32_t iterations, double depth_outlier_trunc, double depth_huber,
                double intensity_huber, double* partial, double* out29, int32_t flags, void* stream);
int bs_odo_prepare(const uint8_t* color, const float* depth, int32_t batch, int32_t H, int32_t W, double depth_max, float* intensity,
                   float* depth_out, void* stream);
int bs_odo_pyrdown(const float* src, int32_t batch, int32_t H, int32_t W, float* dst, int32_t is_depth, double depth_threshold, void* stream);
int bs_odo_sobel(const float* img, int32_t batch, int32_t H, int32_t W, float* gx, float* gy, void* stream);
int bs_odo_accumulate(const float* src_intensity, const float* src_depth, const float* tgt_intensity, const float* tgt_depth,
                      const float* tgt_dIx, const float* tgt_dIy, const float* tgt_dDx, const float* tgt_dDy, int32_t H, int32_t W,
                      const double* K, const double* T, double depth_outlier_trunc, double depth_huber, double intensity_huber,
                      double* partial, double* out29, int32_t flags, void* stream);

/* ---- engine files: the forward of a whole model for a host without Python ------------------------------------
 * The reference's hosts are DepthEstimator.infer_depth_map (BodySLAM_Refactored/src/depth_estimation/interface.py:39-45) and
 * MPEMInterface.infer_relative_pose_between (BodySLAM_not_refactored/MPEM/mpem_interface.py:61-99), both Python.  A plan -- the launch
 * sequence of one (model, batch, frame size, precision, dtype) over static device buffers -- is compiled once by the Python builder and
 * written to an engine file (python -m bodyslam_amd.engine_export); any host then runs it with the calls below.  bs_engine_load
 * allocates every buffer and uploads the constants (weights in their packed device form); nothing is allocated per call.
 *   bs_engine_io          the device address / size of a named static input or output ("frames", "depth_m", "depth_u16"; "pairs", "T")
 *   bs_engine_run         issues the launches on `stream` (and the engine's own side stream, forked from and joined to it)
 *   bs_engine_upload / _download   host <-> named buffer, synchronous: for hosts that do not link the HIP runtime themselves
 *   bs_zoedepth_forward   frames u8 [B,H,W,3] (device) -> depth metres fp32 [B,H,W] and / or uint16 metres*256 [B,H,W] (either may be
 *                         NULL); flip augmentation, routing, post-processing as the engine was exported
 *   bs_cyclepose_forward  frames u8 [n_frames,H,W,3], pairs int32 [P,2] (indices into frames) -> T fp32 [P,16] (row-major 4x4)
 * Shapes must be the ones the engine was exported for (checked). */
typedef struct bs_engine bs_engine;
int bs_engine_load(const char* path, bs_engine** out);
int bs_engine_destroy(bs_engine* e);
int bs_engine_io(const bs_engine* e, const char* name, void** dev_ptr, int64_t* nbytes);
int64_t bs_engine_device_bytes(const bs_engine* e);
int bs_engine_run(bs_engine* e, void* stream);
int bs_engine_upload(bs_engine* e, const char* name, const void* host, int64_t nbytes);
int bs_engine_download(bs_engine* e, const char* name, void* host, int64_t nbytes);
int bs_zoedepth_forward(bs_engine* e, const uint8_t* frames_dev, int32_t B, int32_t H, int32_t W, float* depth_m_dev, uint16_t* depth_u16_dev,
                        void* stream);
int bs_cyclepose_forward(bs_engine* e, const uint8_t* frames_dev, int32_t n_frames, int32_t H, int32_t W, const int32_t* pairs_dev, int32_t P,
                         float* T_rel_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BODYSLAM_HIP_H */
