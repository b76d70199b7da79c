// MPEM (CyclePose generator, mode="pose") kernels that are not GEMM-shaped.  The four
// convolutions run on bs_gemm (igemm.hip); this file holds the input transform + 7x7 patch
// gather, InstanceNorm+ReLU, global average pool and the pose head (skip GEMV over 262 656
// features as a split-K wavefront reduction, dense head, quaternion -> SE(3)).
//   BodySLAM_not_refactored/MPEM/mpem_interface.py:40-44,85-94
//   BodySLAM_not_refactored/MPEM/architecture_v3.py:120-155,195-226
//   BodySLAM_not_refactored/UTILS/geometry_utils.py:230-265
#include "common.h"

namespace bs {

constexpr int CP_CROP = 128;
constexpr int CP_K = 320;  // 7*7*6 = 294 padded to a multiple of 64

// one thread per (row m, tap): writes the 6 channels of that tap (12 bytes) -- consecutive threads
// cover consecutive taps of one row, so a row's 640 bytes are written by 54 neighbouring lanes
// SPLIT: a row is [hi x 320 | lo x 320], value = hi + lo to ~22 bits (the A operand of a split-precision GEMM, bs_gemm seg1)
template <typename T, bool SPLIT>
__global__ __launch_bounds__(256) void cp_im2col_kernel(const uint8_t* frames, const int32_t* pairs, T* out, int P, int H, int W,
                                                         int top, int left, int CH, int CW) {
    // the network sees the CH x CW window of every frame that starts at (top, left): 128 x 128 centre crop, or a whole
    // (already resized) frame
    const int64_t total = (int64_t)P * CH * CW * 54;  // 49 taps + 5 pad slots (6 elems each -> 324 >= 320)
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int slot = (int)(gid % 54);
    const int64_t m = gid / 54;
    T* row = out + m * (SPLIT ? 2 * CP_K : CP_K);
    if (slot >= 49) {
        // zero padding 294..319 (26 elements): slots 49..53 write 6,6,6,6,2
        const int k0 = 294 + (slot - 49) * 6;
        for (int i = 0; i < 6 && k0 + i < CP_K; ++i) {
            row[k0 + i] = T16<T>::from_f32(0.0f);
            if (SPLIT) row[CP_K + k0 + i] = T16<T>::from_f32(0.0f);
        }
        return;
    }
    const int x = (int)(m % CW);
    const int y = (int)((m / CW) % CH);
    const int p = (int)(m / ((int64_t)CW * CH));
    const int ky = slot / 7, kx = slot - ky * 7;
    int yy = y + ky - 3, xx = x + kx - 3;
    yy = yy < 0 ? -yy : (yy >= CH ? 2 * CH - 2 - yy : yy);  // ReflectionPad2d(3)
    xx = xx < 0 ? -xx : (xx >= CW ? 2 * CW - 2 - xx : xx);
    const int f0 = pairs[2 * p], f1 = pairs[2 * p + 1];
    const uint8_t* p0 = frames + (((int64_t)f0 * H + top + yy) * W + left + xx) * 3;
    const uint8_t* p1 = frames + (((int64_t)f1 * H + top + yy) * W + left + xx) * 3;
    T* o = row + slot * 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // ToTensor (/255) then Normalize(0.5, 0.5), both in fp32 as torchvision does
        const float v0 = (__fdiv_rn((float)p0[c], 255.0f) - 0.5f) / 0.5f, v1 = (__fdiv_rn((float)p1[c], 255.0f) - 0.5f) / 0.5f;
        o[c] = T16<T>::from_f32(v0);
        o[3 + c] = T16<T>::from_f32(v1);
        if (SPLIT) {
            o[CP_K + c] = T16<T>::from_f32(v0 - T16<T>::to_f32(o[c]));
            o[CP_K + 3 + c] = T16<T>::from_f32(v1 - T16<T>::to_f32(o[3 + c]));
        }
    }
}

// InstanceNorm2d (biased variance, eps, no affine) + ReLU over an NHWC fp32 map, in two fully parallel passes:
//   (1) per (image, 256-row chunk): chunk mean and M2 (sum of squared deviations about the chunk mean) per channel;
//   (2) per chunk: combine the partials (Chan et al.: equal-weight chunks, numerically as good as two-pass over
//       the whole map), normalise + ReLU, write 16-bit (and optionally fp32).
// thread = 4 consecutive channels (16-byte loads); a 256-thread block covers 1024/C rows per sweep.
constexpr int IN_CHUNK = 256;   // rows (pixels) per block

__global__ __launch_bounds__(256) void instnorm_stats_kernel(const float* x, float* part, int HW, int C, int nchunk) {
    const int p = blockIdx.y, ch = blockIdx.x;
    const int tpr = C >> 2, rps = 256 / tpr;            // threads per row, rows per sweep
    const int c4 = (threadIdx.x % tpr) * 4, rg = threadIdx.x / tpr;
    const int r0 = ch * IN_CHUNK, r1 = min(HW, r0 + IN_CHUNK);
    const float* xp = x + ((int64_t)p * HW) * C + c4;
    __shared__ float red[256 * 4];
    __shared__ float mean_s[256];
    f32x4 s = {0, 0, 0, 0};
    for (int r = r0 + rg; r < r1; r += rps) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xp + (int64_t)r * C);
        s += v;
    }
    *reinterpret_cast<f32x4*>(red + threadIdx.x * 4) = s;
    __syncthreads();
    if (threadIdx.x < C) {
        const int c = threadIdx.x;
        float a = 0.f;
        for (int g = 0; g < rps; ++g) a += red[(g * tpr + (c >> 2)) * 4 + (c & 3)];
        mean_s[c] = a / (float)(r1 - r0);
    }
    __syncthreads();
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean_s + c4);
    f32x4 q = {0, 0, 0, 0};
    for (int r = r0 + rg; r < r1; r += rps) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xp + (int64_t)r * C) - mu;
        q += v * v;
    }
    __syncthreads();
    *reinterpret_cast<f32x4*>(red + threadIdx.x * 4) = q;
    __syncthreads();
    if (threadIdx.x < C) {
        const int c = threadIdx.x;
        float a = 0.f;
        for (int g = 0; g < rps; ++g) a += red[(g * tpr + (c >> 2)) * 4 + (c & 3)];
        float* o = part + (((int64_t)p * nchunk + ch) * 2) * C;
        o[c] = mean_s[c];
        o[C + c] = a;
    }
}

template <typename T, bool SPLIT>
__global__ __launch_bounds__(256) void instnorm_apply_kernel(const float* x, const float* part, T* out, float* out_f32, int HW, int C,
                                                              int nchunk, float eps) {
    const int p = blockIdx.y, ch = blockIdx.x;
    const int tpr = C >> 2, rps = 256 / tpr;
    const int c4 = (threadIdx.x % tpr) * 4, rg = threadIdx.x / tpr;
    const int r0 = ch * IN_CHUNK, r1 = min(HW, r0 + IN_CHUNK);
    __shared__ float mean_s[256], rstd_s[256];
    if (threadIdx.x < C) {
        const int c = threadIdx.x;
        const float* pp = part + ((int64_t)p * nchunk * 2) * C + c;
        // all chunks but possibly the last hold IN_CHUNK rows; weight by the true counts
        float msum = 0.f;
        for (int k = 0; k < nchunk; ++k) msum += pp[(int64_t)k * 2 * C] * (float)(min(HW, (k + 1) * IN_CHUNK) - k * IN_CHUNK);
        const float mean = msum / (float)HW;
        float m2 = 0.f;
        for (int k = 0; k < nchunk; ++k) {
            const float d = pp[(int64_t)k * 2 * C] - mean;
            m2 += pp[(int64_t)k * 2 * C + C] + d * d * (float)(min(HW, (k + 1) * IN_CHUNK) - k * IN_CHUNK);
        }
        mean_s[c] = mean;
        rstd_s[c] = 1.0f / sqrtf(m2 / (float)HW + eps);
    }
    __syncthreads();
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean_s + c4), rs = *reinterpret_cast<const f32x4*>(rstd_s + c4);
    const int64_t base = ((int64_t)p * HW) * C + c4;
    for (int r = r0 + rg; r < r1; r += rps) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + base + (int64_t)r * C);
        f32x4 y;
        typename T16<T>::v4 o, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] = fmaxf((v[e] - mu[e]) * rs[e], 0.0f);
            o[e] = T16<T>::from_f32(y[e]);
            ol[e] = T16<T>::from_f32(y[e] - T16<T>::to_f32(o[e]));
        }
        if (SPLIT) {     // pixel vector [hi x C | lo x C]
            T* op = out + (((int64_t)p * HW + r) * 2) * C + c4;
            *reinterpret_cast<typename T16<T>::v4*>(op) = o;
            *reinterpret_cast<typename T16<T>::v4*>(op + C) = ol;
        } else {
            *reinterpret_cast<typename T16<T>::v4*>(out + base + (int64_t)r * C) = o;
        }
        if (out_f32) *reinterpret_cast<f32x4*>(out_f32 + base + (int64_t)r * C) = y;
    }
}

__global__ __launch_bounds__(256) void avgpool_kernel(const float* x, float* out, int HW, int C) {
    const int p = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const float* xp = x + (int64_t)p * HW * C + c;
    __shared__ float red[4][64];
    float s = 0.f;
    for (int i = g; i < HW; i += 4) s += xp[(int64_t)i * C];
    red[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0) {
        const int l = threadIdx.x;
        out[(int64_t)p * C + c] = (red[0][l] + red[1][l] + red[2][l] + red[3][l]) / (float)HW;
    }
}

// split-K partial dot products of the skip GEMV: partial[p][chunk][7]
constexpr int SKIP_CHUNK = 4096;
__global__ __launch_bounds__(256) void skip_partial_kernel(const float* x2, const float* w, float* partial, int K, int nchunk) {
    const int chunk = blockIdx.x, p = blockIdx.y;
    const int k0 = chunk * SKIP_CHUNK;
    const float* xp = x2 + (int64_t)p * K + k0;
    float acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x * 4; i < SKIP_CHUNK && k0 + i < K; i += 256 * 4) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + i);
#pragma unroll
        for (int o = 0; o < 7; ++o) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (int64_t)o * K + k0 + i);
            acc[o] += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
        }
    }
    __shared__ float red[4][7];
#pragma unroll
    for (int o = 0; o < 7; ++o) {
        float a = acc[o];
        for (int s = 32; s > 0; s >>= 1) a += __shfl_down(a, s, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][o] = a;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int o = threadIdx.x;
        partial[((int64_t)p * nchunk + chunk) * 8 + o] = red[0][o] + red[1][o] + red[2][o] + red[3][o];
    }
}

// per pair: finish the skip sum, dense head, quaternion -> 4x4.  One wave per pair.
__global__ __launch_bounds__(64) void pose_head_kernel(const float* pooled, const float* partial, int nchunk, const float* w_skip_pool,
                                                        const float* b_skip, const float* w1, const float* b1, const float* w2,
                                                        const float* b2, float* pose7, float* Tout, int CP) {
    const int p = blockIdx.x, lane = threadIdx.x;
    __shared__ float hid[128];
    __shared__ float pv[512];
    for (int i = lane; i < CP; i += 64) pv[i] = pooled[(int64_t)p * CP + i];
    __syncthreads();
    // pose_dense.1: Linear(512,128) + ReLU
    for (int o = lane; o < 128; o += 64) {
        float a = 0.f;
        const float* wr = w1 + (int64_t)o * CP;
        for (int k = 0; k < CP; ++k) a += wr[k] * pv[k];
        hid[o] = fmaxf(a + b1[o], 0.f);
    }
    __syncthreads();
    __shared__ float out7[8];
    if (lane < 7) {
        float dense = 0.f;
        for (int k = 0; k < 128; ++k) dense += w2[lane * 128 + k] * hid[k];
        dense += b2[lane];
        float skip = 0.f;
        for (int k = 0; k < CP; ++k) skip += w_skip_pool[(int64_t)lane * CP + k] * pv[k];
        for (int c = 0; c < nchunk; ++c) skip += partial[((int64_t)p * nchunk + c) * 8 + lane];
        skip += b_skip[lane];
        out7[lane] = dense + skip;
        pose7[(int64_t)p * 7 + lane] = dense + skip;
    }
    __syncthreads();
    if (lane == 0) {
        const float tx = out7[0], ty = out7[1], tz = out7[2];
        float r = out7[3], i = out7[4], j = out7[5], k = out7[6];
        const float nrm = sqrtf(r * r + i * i + j * j + k * k);  // normalize_quaternion: q / ||q||, no eps
        r /= nrm; i /= nrm; j /= nrm; k /= nrm;
        const float two_s = 2.0f / (r * r + i * i + j * j + k * k);  // quaternion_to_matrix divides by |q|^2 again
        float* T = Tout + (int64_t)p * 16;
        T[0] = 1 - two_s * (j * j + k * k); T[1] = two_s * (i * j - k * r);     T[2] = two_s * (i * k + j * r);      T[3] = tx;
        T[4] = two_s * (i * j + k * r);     T[5] = 1 - two_s * (i * i + k * k); T[6] = two_s * (j * k - i * r);      T[7] = ty;
        T[8] = two_s * (i * k - j * r);     T[9] = two_s * (j * k + i * r);     T[10] = 1 - two_s * (i * i + j * j); T[11] = tz;
        T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
    }
}

}  // namespace bs

static int im2col_launch(const uint8_t* frames, const int32_t* pairs, void* out, int P, int H, int W, int top, int left, int CH, int CW,
                         int dtype, bool split, hipStream_t st) {
    using namespace bs;
    const int64_t total = (int64_t)P * CH * CW * 54;
    const unsigned blocks = (unsigned)cdiv64(total, 256);
    if (dtype == BS_F16 && split)
        hipLaunchKernelGGL((cp_im2col_kernel<f16, true>), dim3(blocks), dim3(256), 0, st, frames, pairs, (f16*)out, P, H, W, top, left, CH, CW);
    else if (dtype == BS_F16)
        hipLaunchKernelGGL((cp_im2col_kernel<f16, false>), dim3(blocks), dim3(256), 0, st, frames, pairs, (f16*)out, P, H, W, top, left, CH, CW);
    else if (split)
        hipLaunchKernelGGL((cp_im2col_kernel<bf16, true>), dim3(blocks), dim3(256), 0, st, frames, pairs, (bf16*)out, P, H, W, top, left, CH, CW);
    else
        hipLaunchKernelGGL((cp_im2col_kernel<bf16, false>), dim3(blocks), dim3(256), 0, st, frames, pairs, (bf16*)out, P, H, W, top, left, CH, CW);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_cyclepose_im2col(const uint8_t* frames, const int32_t* pairs, void* out, int32_t P, int32_t H, int32_t W,
                                   int32_t dtype, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("bs_cyclepose_im2col: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(frames && pairs && out && P >= 0, "bs_cyclepose_im2col: bad argument");
    BS_REQUIRE(H >= CP_CROP && W >= CP_CROP, "bs_cyclepose_im2col: frame %dx%d smaller than the 128 crop", W, H);
    const bool split = (dtype & 16) != 0;      // bit 4: rows of (hi | lo) pairs, [P*128*128, 640]
    dtype &= 15;
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "bs_cyclepose_im2col: dtype");
    if (P == 0) return BS_OK;
    // torchvision CenterCrop: top = int(round((H - 128) / 2.0)) (Python banker's rounding of x.5)
    auto pyround_half = [](int v) { const int q = v / 2; return (v % 2 == 0) ? q : ((q % 2 == 0) ? q : q + 1); };
    const int top = pyround_half(H - CP_CROP), left = pyround_half(W - CP_CROP);
    return im2col_launch(frames, pairs, out, P, H, W, top, left, CP_CROP, CP_CROP, dtype, split, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int bs_cyclepose_im2col_window(const uint8_t* frames, const int32_t* pairs, void* out, int32_t P, int32_t H, int32_t W,
                                          int32_t top, int32_t left, int32_t CH, int32_t CW, int32_t dtype, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("bs_cyclepose_im2col_window: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(frames && pairs && out && P >= 0, "bs_cyclepose_im2col_window: bad argument");
    BS_REQUIRE(CH >= 4 && CW >= 4 && top >= 0 && left >= 0 && top + CH <= H && left + CW <= W,
               "bs_cyclepose_im2col_window: window %dx%d at (%d,%d) outside the %dx%d frame (or smaller than the reflection pad)", CW, CH, left, top, W, H);
    const bool split = (dtype & 16) != 0;
    dtype &= 15;
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "bs_cyclepose_im2col_window: dtype");
    if (P == 0) return BS_OK;
    return im2col_launch(frames, pairs, out, P, H, W, top, left, CH, CW, dtype, split, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int bs_instnorm_relu_nhwc(const float* x, void* out, float* out_f32, float* scratch, int32_t P, int32_t HW, int32_t C,
                                     float eps, int32_t dtype, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("bs_instnorm_relu_nhwc: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(x && out && scratch && P >= 0 && HW > 0 && C % 4 == 0 && C >= 4 && C <= 256 && 256 % (C / 4) == 0,
               "bs_instnorm_relu_nhwc: bad argument (C must be 4..256 with C/4 dividing 256)");
    const bool split = (dtype & 16) != 0;      // bit 4: pixel vectors of (hi | lo) pairs, [P, HW, 2C]
    dtype &= 15;
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "bs_instnorm_relu_nhwc: dtype");
    if (P == 0) return BS_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nchunk = cdiv(HW, IN_CHUNK);
    dim3 grid(nchunk, P);
    hipLaunchKernelGGL(instnorm_stats_kernel, grid, dim3(256), 0, st, x, scratch, HW, C, nchunk);
    BS_CHECK_LAUNCH();
    if (dtype == BS_F16 && split)
        hipLaunchKernelGGL((instnorm_apply_kernel<f16, true>), grid, dim3(256), 0, st, x, (const float*)scratch, (f16*)out, out_f32, HW, C, nchunk, eps);
    else if (dtype == BS_F16)
        hipLaunchKernelGGL((instnorm_apply_kernel<f16, false>), grid, dim3(256), 0, st, x, (const float*)scratch, (f16*)out, out_f32, HW, C, nchunk, eps);
    else if (split)
        hipLaunchKernelGGL((instnorm_apply_kernel<bf16, true>), grid, dim3(256), 0, st, x, (const float*)scratch, (bf16*)out, out_f32, HW, C, nchunk, eps);
    else
        hipLaunchKernelGGL((instnorm_apply_kernel<bf16, false>), grid, dim3(256), 0, st, x, (const float*)scratch, (bf16*)out, out_f32, HW, C, nchunk, eps);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_avgpool_nhwc(const float* x, float* out, int32_t P, int32_t HW, int32_t C, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("bs_avgpool_nhwc: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(x && out && P >= 0 && HW > 0 && C % 64 == 0, "bs_avgpool_nhwc: bad argument");
    if (P == 0) return BS_OK;
    hipLaunchKernelGGL(avgpool_kernel, dim3(C / 64, P), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, out, HW, C);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_cyclepose_head(const float* pooled, const float* x2, const float* w_skip_pool, const float* w_skip_x2,
                                 const float* b_skip, const float* w1, const float* b1, const float* w2, const float* b2,
                                 float* pose7, float* T, float* scratch, int32_t P, int32_t HW, int32_t C, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("bs_cyclepose_head: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(pooled && x2 && w_skip_pool && w_skip_x2 && b_skip && w1 && b1 && w2 && b2 && pose7 && T && scratch,
               "bs_cyclepose_head: null argument");
    if (P == 0) return BS_OK;
    const int K = HW * C;
    BS_REQUIRE(K % 4 == 0, "bs_cyclepose_head: HW*C must be a multiple of 4");
    const int nchunk = cdiv(K, SKIP_CHUNK);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(skip_partial_kernel, dim3(nchunk, P), dim3(256), 0, st, x2, w_skip_x2, scratch, K, nchunk);
    BS_CHECK_LAUNCH();
    hipLaunchKernelGGL(pose_head_kernel, dim3(P), dim3(64), 0, st, pooled, (const float*)scratch, nchunk, w_skip_pool, b_skip, w1, b1,
                       w2, b2, pose7, T, 512);
    BS_CHECK_LAUNCH();
    return BS_OK;
}
