// Pointwise producers of the F4 activation format (include/bodyslam_hip.h: hi16 values, two e2m1 planes with one E8M0 scale per 64
// channels, an e4m3 plane of the rounding residual): fp32 rows -> F4 (the backbone taps that feed the DPT readout), ReLU of an F4
// map (pre-activation residual units, HF modeling_zoedepth.py:228-262), bilinear resize of an F4 map (fusion stage, :316-322).
// All three are HBM-bound streaming kernels: thread = 16 channels of one pixel, every access 8-32 bytes wide; the four threads of a
// 64-channel group are neighbouring lanes, so the group maxima come from two DPP quad permutes.
#include "common.h"

namespace bs {

__device__ __forceinline__ float quad_max(float v) {
    // lanes 4q .. 4q+3: xor 1 (quad_perm [1,0,3,2]) then xor 2 (quad_perm [2,3,0,1])
    int u = __builtin_bit_cast(int, v);
    int a = __builtin_amdgcn_update_dpp(u, u, 0xB1, 0xf, 0xf, false);
    v = fmaxf(v, __builtin_bit_cast(float, a));
    u = __builtin_bit_cast(int, v);
    a = __builtin_amdgcn_update_dpp(u, u, 0x4E, 0xf, 0xf, false);
    return fmaxf(v, __builtin_bit_cast(float, a));
}

// 16 channels [c, c + 16) (c % 16 == 0) of the pixel whose vector starts at `pix` (C channels): all planes + (one lane of the quad) scales
template <typename T>
__device__ __forceinline__ void f4_store16(T* pix, int C, int c, const float (&v)[16]) {
    typedef typename T16<T>::v8 v8;
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    v8 h0, h1;
    float r[16];
    float mh = 0.0f, ml = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const T h = T16<T>::from_f32(v[e]);
        if (e < 8) h0[e] = h; else h1[e - 8] = h;
        r[e] = v[e] - T16<T>::to_f32(h);
        mh = fmaxf(mh, fabsf(v[e]));
        ml = fmaxf(ml, fabsf(r[e]));
    }
    *reinterpret_cast<v8*>(pix + c) = h0;
    *reinterpret_cast<v8*>(pix + c + 8) = h1;
    mh = quad_max(mh);
    ml = quad_max(ml);
    int eh = (int)(__builtin_bit_cast(unsigned, mh) >> 23) - 2, el = (int)(__builtin_bit_cast(unsigned, ml) >> 23) - 2;
    eh = eh < 1 ? 1 : eh;
    el = el < 1 ? 1 : el;
    const float sh = __builtin_bit_cast(float, (unsigned)eh << 23), sl = __builtin_bit_cast(float, (unsigned)el << 23);
    unsigned ph0 = 0, ph1 = 0, pl0 = 0, pl1 = 0;
    ph0 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ph0, v[0], v[1], sh, 0);
    ph0 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ph0, v[2], v[3], sh, 1);
    ph0 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ph0, v[4], v[5], sh, 2);
    ph0 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ph0, v[6], v[7], sh, 3);
    ph1 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ph1, v[8], v[9], sh, 0);
    ph1 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ph1, v[10], v[11], sh, 1);
    ph1 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ph1, v[12], v[13], sh, 2);
    ph1 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ph1, v[14], v[15], sh, 3);
    pl0 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pl0, r[0], r[1], sl, 0);
    pl0 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pl0, r[2], r[3], sl, 1);
    pl0 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pl0, r[4], r[5], sl, 2);
    pl0 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pl0, r[6], r[7], sl, 3);
    pl1 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pl1, r[8], r[9], sl, 0);
    pl1 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pl1, r[10], r[11], sl, 1);
    pl1 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pl1, r[12], r[13], sl, 2);
    pl1 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pl1, r[14], r[15], sl, 3);
    // chunk position of channels [c, c + 16): unit (c >> 8), group g = (c >> 6) & 3, half (c >> 5) & 1 -> chunk g + 4 * half, 8 bytes in
    char* bytes = reinterpret_cast<char*>(pix);
    const int pos = (c >> 8) * 128 + (((c >> 6) & 3) + 4 * ((c >> 5) & 1)) * 16 + ((c & 31) >> 1);
    *reinterpret_cast<u32x2_*>(bytes + 2 * C + pos) = u32x2_{ph0, ph1};
    *reinterpret_cast<u32x2_*>(bytes + 2 * C + (C >> 1) + pos) = u32x2_{pl0, pl1};
    const float s8 = __builtin_ldexpf(1.0f, F8_ACT_LO_EXP);
    i32x4 l8;
#pragma unroll
    for (int g = 0; g < 4; ++g) l8[g] = f8_pack4(r[4 * g] * s8, r[4 * g + 1] * s8, r[4 * g + 2] * s8, r[4 * g + 3] * s8);
    *reinterpret_cast<i32x4*>(bytes + 3 * C + c) = l8;
    if ((c & 63) == 0) {
        bytes[4 * C + (c >> 6)] = (char)eh;
        bytes[4 * C + (C >> 6) + (c >> 6)] = (char)el;
    }
}

// value (hi16 + lo8 * 2^-LO_EXP) of 16 channels of an F4 (or (hi16 | hi8 | lo8): the two planes sit at the same offsets) pixel
template <typename T>
__device__ __forceinline__ void f4_load16(const T* pix, int C, int c, float (&q)[16]) {
    typedef typename T16<T>::v8 v8;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const v8 h0 = *reinterpret_cast<const v8*>(pix + c), h1 = *reinterpret_cast<const v8*>(pix + c + 8);
    const i32x4 l = *reinterpret_cast<const i32x4*>(reinterpret_cast<const char*>(pix) + 3 * C + c);
    const float sc = __builtin_ldexpf(1.0f, -F8_ACT_LO_EXP);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        q[e] = T16<T>::to_f32(h0[e]);
        q[8 + e] = T16<T>::to_f32(h1[e]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float lo[4];
        f8_unpack4(l[g], lo);
#pragma unroll
        for (int e = 0; e < 4; ++e) q[4 * g + e] += lo[e] * sc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void f4_cast_kernel(const float* x, T* out, int64_t rows, int C, int pitch) {
    const int c16n = C >> 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // (grid sized exactly; C % 64 == 0 keeps quads whole)
    if (i >= rows * c16n) return;
    const int64_t r = i / c16n;
    const int c = (int)(i - r * c16n) * 16;
    float v[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(x + r * C + c + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * g + e] = t[e];
    }
    f4_store16<T>(out + r * pitch, C, c, v);
}

// ReLU of an F4 map: the value hi16 + lo8 is rectified (its sign is the sign of hi16) and re-encoded with fresh block scales
template <typename T>
__global__ __launch_bounds__(256) void f4_relu_kernel(const T* x, T* out, int64_t rows, int C, int pitch) {
    const int c16n = C >> 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * c16n) return;
    const int64_t r = i / c16n;
    const int c = (int)(i - r * c16n) * 16;
    float v[16];
    f4_load16<T>(x + r * pitch, C, c, v);
    const typename T16<T>::v8 h0 = *reinterpret_cast<const typename T16<T>::v8*>(x + r * pitch + c),
                              h1 = *reinterpret_cast<const typename T16<T>::v8*>(x + r * pitch + c + 8);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float hv = T16<T>::to_f32(e < 8 ? h0[e] : h1[e - 8]);
        v[e] = hv > 0.0f ? v[e] : 0.0f;
    }
    f4_store16<T>(out + r * pitch, C, c, v);
}

// bilinear resize of an F4 map (input pitch `pin`; the input may also be a (hi16 | hi8 | lo8) map: same value planes)
template <typename T>
__global__ __launch_bounds__(256) void f4_resize_kernel(const T* x, T* out, int B, int Hin, int Win, int C, int Hout, int Wout, float sy, float sx,
                                                        int align, int pin, int pout) {
    const int c16n = C >> 4;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)Wout * c16n) return;
    const int c16 = idx % c16n, ox = idx / c16n;
    const int b = blockIdx.y / Hout, oy = blockIdx.y - b * Hout;
    const int64_t pix = ((int64_t)b * Hout + oy) * Wout + ox;
    float fy, fx;
    if (align) {
        fy = sy * (float)oy;
        fx = sx * (float)ox;
    } else {
        fy = fmaxf(__fmaf_rn(sy, (float)oy + 0.5f, -0.5f), 0.0f);
        fx = fmaxf(__fmaf_rn(sx, (float)ox + 0.5f, -0.5f), 0.0f);
    }
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 > Hin - 1 ? Hin - 1 : y0;
    x0 = x0 > Win - 1 ? Win - 1 : x0;
    const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
    const T* xb = x + (int64_t)b * Hin * Win * pin;
    float q00[16], q01[16], q10[16], q11[16], v[16];
    f4_load16<T>(xb + ((int64_t)y0 * Win + x0) * pin, C, c16 * 16, q00);
    f4_load16<T>(xb + ((int64_t)y0 * Win + x1) * pin, C, c16 * 16, q01);
    f4_load16<T>(xb + ((int64_t)y1 * Win + x0) * pin, C, c16 * 16, q10);
    f4_load16<T>(xb + ((int64_t)y1 * Win + x1) * pin, C, c16 * 16, q11);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = hy * (hx * q00[e] + lx * q01[e]) + ly * (hx * q10[e] + lx * q11[e]);
    f4_store16<T>(out + pix * pout, C, c16 * 16, v);
}

int f4_cast(const float* x, void* out, int64_t rows, int C, int dtype, hipStream_t st) {
    BS_REQUIRE(C % 256 == 0, "F4 format: C=%d must be a multiple of 256", C);
    const int64_t n = rows * (C / 16);
    const int pitch = BS_F4_PITCH_ELEMS(C);
    if (dtype == BS_F16)
        hipLaunchKernelGGL(f4_cast_kernel<f16>, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, x, (f16*)out, rows, C, pitch);
    else
        hipLaunchKernelGGL(f4_cast_kernel<bf16>, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, x, (bf16*)out, rows, C, pitch);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

int f4_relu(const void* x, void* out, int64_t rows, int C, int dtype, hipStream_t st) {
    BS_REQUIRE(C % 256 == 0, "F4 format: C=%d must be a multiple of 256", C);
    const int64_t n = rows * (C / 16);
    const int pitch = BS_F4_PITCH_ELEMS(C);
    if (dtype == BS_F16)
        hipLaunchKernelGGL(f4_relu_kernel<f16>, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, (const f16*)x, (f16*)out, rows, C, pitch);
    else
        hipLaunchKernelGGL(f4_relu_kernel<bf16>, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, (const bf16*)x, (bf16*)out, rows, C, pitch);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

int f4_resize(const void* x, void* out, int B, int Hin, int Win, int C, int Hout, int Wout, int align, int dtype, hipStream_t st) {
    BS_REQUIRE(C % 256 == 0, "F4 format: C=%d must be a multiple of 256", C);
    float sy, sx;
    if (align) {
        sy = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f;
        sx = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
    } else {
        sy = (float)Hin / (float)Hout;
        sx = (float)Win / (float)Wout;
    }
    const dim3 blocks(cdiv(Wout * (C / 16), 256), B * Hout);
    const int pitch = BS_F4_PITCH_ELEMS(C);
    if (dtype == BS_F16)
        hipLaunchKernelGGL(f4_resize_kernel<f16>, blocks, dim3(256), 0, st, (const f16*)x, (f16*)out, B, Hin, Win, C, Hout, Wout, sy, sx, align, pitch, pitch);
    else
        hipLaunchKernelGGL(f4_resize_kernel<bf16>, blocks, dim3(256), 0, st, (const bf16*)x, (bf16*)out, B, Hin, Win, C, Hout, Wout, sy, sx, align, pitch,
                           pitch);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

}  // namespace bs
