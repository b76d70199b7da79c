// HBM-bound helper kernels of the MDEM path: LayerNorm, casts, uint8 -> patch-matrix
// pre-processing (reflect pad + bilinear resize + normalise fused with the 16x16 patch gather),
// NHWC bilinear resampling, the final bicubic/flip-average/uint16 post-processing.
//   HF image_processing_pil_zoedepth.py:181-232,234-341; HF modeling_beit.py:63-176,418,432;
//   HF modeling_zoedepth.py:259,319,360
#include "common.h"

namespace bs {

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row; the row is cached in registers when cols <= 1024
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* gamma, const float* beta, T* out16, float* out32,
                                                         int rows, int cols, float eps, int split, int plane_rows) {
    // split: out16 is [rows, 2*cols] = (hi | lo) pairs, y = hi + lo to ~22 bits (operand of a split-precision GEMM)
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * cols;
    constexpr int VPT = 4;
    f32x4 v[VPT];
    const int nvec = cols >> 2;
    const bool cached = nvec <= 64 * VPT;
    float s = 0.f;
    if (cached) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int vi = lane + i * 64;
            v[i] = vi < nvec ? *reinterpret_cast<const f32x4*>(xr + vi * 4) : f32x4{0, 0, 0, 0};
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    } else {
        for (int vi = lane; vi < nvec; vi += 64) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(xr + vi * 4);
            s += t[0] + t[1] + t[2] + t[3];
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)cols;
    float q = 0.f;
    if (cached) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            if (lane + i * 64 < nvec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[i][e] - mean;
                    q += d * d;
                }
            }
        }
    } else {
        for (int vi = lane; vi < nvec; vi += 64) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(xr + vi * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = t[e] - mean;
                q += d * d;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / (float)cols + eps);
    for (int i = 0, vi = lane; vi < nvec; vi += 64, ++i) {
        f32x4 t;
        if (cached) {
            // static indexing only (runtime-indexed vector arrays go to scratch)
            t = i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3]));
        } else {
            t = *reinterpret_cast<const f32x4*>(xr + vi * 4);
        }
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + vi * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + vi * 4);
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (t[e] - mean) * rstd * g[e] + b[e];
        if (out16) {
            typename T16<T>::v4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = T16<T>::from_f32(y[e]);
            if (split == 2 && plane_rows > 0 && row >= plane_rows) {   // a row whose consumer runs no FP8 stage: hi16 only
                *reinterpret_cast<typename T16<T>::v4*>(out16 + (int64_t)row * cols * 2 + vi * 4) = o;
            } else if (split == 2) {   // (hi16 | hi8 | lo8) planes of a 4*cols-byte row
                T* rowp = out16 + (int64_t)row * cols * 2;
                *reinterpret_cast<typename T16<T>::v4*>(rowp + vi * 4) = o;
                const float sh = __builtin_ldexpf(1.0f, F8_ACT_HI_EXP), sl = __builtin_ldexpf(1.0f, F8_ACT_LO_EXP);
                char* planes = reinterpret_cast<char*>(rowp + cols);
                *reinterpret_cast<int*>(planes + vi * 4) = f8_pack4(y[0] * sh, y[1] * sh, y[2] * sh, y[3] * sh);
                *reinterpret_cast<int*>(planes + cols + vi * 4) =
                    f8_pack4((y[0] - T16<T>::to_f32(o[0])) * sl, (y[1] - T16<T>::to_f32(o[1])) * sl, (y[2] - T16<T>::to_f32(o[2])) * sl,
                             (y[3] - T16<T>::to_f32(o[3])) * sl);
            } else if (split) {
                typename T16<T>::v4 ol;
#pragma unroll
                for (int e = 0; e < 4; ++e) ol[e] = T16<T>::from_f32(y[e] - T16<T>::to_f32(o[e]));
                *reinterpret_cast<typename T16<T>::v4*>(out16 + (int64_t)row * cols * 2 + vi * 4) = o;
                *reinterpret_cast<typename T16<T>::v4*>(out16 + (int64_t)row * cols * 2 + cols + vi * 4) = ol;
            } else {
                *reinterpret_cast<typename T16<T>::v4*>(out16 + (int64_t)row * cols + vi * 4) = o;
            }
        }
        if (out32) *reinterpret_cast<f32x4*>(out32 + (int64_t)row * cols + vi * 4) = f32x4{y[0], y[1], y[2], y[3]};
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* x, T* out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(x + i * 4);
        typename T16<T>::v4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = T16<T>::from_f32(t[e]);
        *reinterpret_cast<typename T16<T>::v4*>(out + i * 4) = o;
    }
}

// fp32 [rows, C] -> 16-bit (hi | lo) pairs [rows, 2C]: x = hi + lo to ~22 bits (split-precision operands)
template <typename T>
__global__ __launch_bounds__(256) void cast_split_kernel(const float* x, T* out, int64_t rows, int C, int f8) {
    const int c4n = C >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * c4n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c4n;
        const int c = (int)(i - r * c4n) * 4;
        const f32x4 t = *reinterpret_cast<const f32x4*>(x + r * C + c);
        typename T16<T>::v4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = T16<T>::from_f32(t[e]);
            l[e] = T16<T>::from_f32(t[e] - T16<T>::to_f32(h[e]));
        }
        *reinterpret_cast<typename T16<T>::v4*>(out + r * 2 * C + c) = h;
        if (f8) {     // (hi16 | hi8 | lo8)
            const float sh = __builtin_ldexpf(1.0f, F8_ACT_HI_EXP), sl = __builtin_ldexpf(1.0f, F8_ACT_LO_EXP);
            char* planes = reinterpret_cast<char*>(out + r * 2 * C + C);
            *reinterpret_cast<int*>(planes + c) = f8_pack4(t[0] * sh, t[1] * sh, t[2] * sh, t[3] * sh);
            *reinterpret_cast<int*>(planes + C + c) = f8_pack4((t[0] - T16<T>::to_f32(h[0])) * sl, (t[1] - T16<T>::to_f32(h[1])) * sl,
                                                                (t[2] - T16<T>::to_f32(h[2])) * sl, (t[3] - T16<T>::to_f32(h[3])) * sl);
        } else {
            *reinterpret_cast<typename T16<T>::v4*>(out + r * 2 * C + C + c) = l;
        }
    }
}

// ReLU of a (hi | lo) tensor [rows, 2C]: the sign of hi + lo is the sign of hi (|lo| <= ulp(hi)/2)
template <typename T>
__global__ __launch_bounds__(256) void relu_split_kernel(const T* x, T* out, int64_t rows, int C, int f8) {
    typedef typename T16<T>::v8 v8;
    const int c8n = C >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * c8n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c8n;
        const int c = (int)(i - r * c8n) * 8;
        if (f8) {     // (hi16 | hi8 | lo8): the three planes are masked bytewise, nothing is decoded
            v8 h = *reinterpret_cast<const v8*>(x + r * 2 * C + c);
            const unsigned char* pin = reinterpret_cast<const unsigned char*>(x + r * 2 * C + C);
            unsigned char* pout = reinterpret_cast<unsigned char*>(out + r * 2 * C + C);
            // (f8 == 2: the lo8 plane is neither read nor written -- every consumer of the output is weight-only; f8 == 3: nor is the hi8 plane --
            // every consumer runs one 16-bit pass)
            unsigned long long h8 = f8 == 3 ? 0ull : *reinterpret_cast<const unsigned long long*>(pin + c), l8 = f8 >= 2 ? 0ull : *reinterpret_cast<const unsigned long long*>(pin + C + c);
            unsigned long long mask = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool pos = (float)h[e] > 0.0f;
                h[e] = pos ? h[e] : T16<T>::from_f32(0.0f);
                mask |= pos ? (0xffull << (8 * e)) : 0ull;
            }
            *reinterpret_cast<v8*>(out + r * 2 * C + c) = h;
            if (f8 != 3) *reinterpret_cast<unsigned long long*>(pout + c) = h8 & mask;
            if (f8 < 2) *reinterpret_cast<unsigned long long*>(pout + C + c) = l8 & mask;
            continue;
        }
        v8 h = *reinterpret_cast<const v8*>(x + r * 2 * C + c), l = *reinterpret_cast<const v8*>(x + r * 2 * C + C + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool pos = (float)h[e] > 0.0f;
            h[e] = pos ? h[e] : T16<T>::from_f32(0.0f);
            l[e] = pos ? l[e] : T16<T>::from_f32(0.0f);
        }
        *reinterpret_cast<v8*>(out + r * 2 * C + c) = h;
        *reinterpret_cast<v8*>(out + r * 2 * C + C + c) = l;
    }
}

__global__ void fill_rows_kernel(float* x, const float* v, int rows_per_image, int cols) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < cols; i += blockDim.x) x[(int64_t)b * rows_per_image * cols + i] = v[i];
}

// ---------------------------------------------------------------------------------------------
// pre-processing
// ---------------------------------------------------------------------------------------------
struct PreGeom {
    int H, W, ph, pw, nh, nw;
    float sy, sx;  // align_corners=True scales: (Hp-1)/(nh-1), (Wp-1)/(nw-1)
};

__device__ __forceinline__ float pre_sample(const uint8_t* frame, const PreGeom& g, int c, int y, int x) {
    // value of the normalised network input at (c, y, x) of the UNFLIPPED image
    const float fy = g.sy * (float)y, fx = g.sx * (float)x;
    const int Hp = g.H + 2 * g.ph, Wp = g.W + 2 * g.pw;
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 > Hp - 1 ? Hp - 1 : y0;
    x0 = x0 > Wp - 1 ? Wp - 1 : x0;
    const int y1 = y0 + (y0 < Hp - 1 ? 1 : 0), x1 = x0 + (x0 < Wp - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    auto refl = [](int i, int pad, int n) {
        int s = i - pad;
        s = s < 0 ? -s : s;
        return s >= n ? 2 * (n - 1) - s : s;
    };
    const int sy0 = refl(y0, g.ph, g.H), sy1 = refl(y1, g.ph, g.H);
    const int sx0 = refl(x0, g.pw, g.W), sx1 = refl(x1, g.pw, g.W);
    const float k = 1.0f / 255.0f;
    const float p00 = (float)frame[((int64_t)sy0 * g.W + sx0) * 3 + c] * k;
    const float p01 = (float)frame[((int64_t)sy0 * g.W + sx1) * 3 + c] * k;
    const float p10 = (float)frame[((int64_t)sy1 * g.W + sx0) * 3 + c] * k;
    const float p11 = (float)frame[((int64_t)sy1 * g.W + sx1) * 3 + c] * k;
    const float v = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
    return (v - 0.5f) / 0.5f;
}

template <typename T>
__global__ __launch_bounds__(256) void pre_patches_kernel(const uint8_t* frames, T* out, int B, int nimg, PreGeom g, int split) {
    const int hp = g.nh / 16, wp = g.nw / 16;
    const int64_t total = (int64_t)nimg * hp * wp * 768;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int k = (int)(gid % 768);
    const int64_t r = gid / 768;
    const int px = (int)(r % wp), py = (int)((r / wp) % hp), img = (int)(r / ((int64_t)wp * hp));
    const int c = k >> 8, ky = (k >> 4) & 15, kx = k & 15;
    const int y = py * 16 + ky;
    int x = px * 16 + kx;
    const int b = img >= B ? img - B : img;
    if (img >= B) x = g.nw - 1 - x;  // torch.flip(x, dims=[3]) of the network input
    const float v = pre_sample(frames + (int64_t)b * g.H * g.W * 3, g, c, y, x);
    const T hi = T16<T>::from_f32(v);
    if (split == 2) {   // rows of [768 hi16 | 768 hi8 | 768 lo8]
        out[r * 1536 + k] = hi;
        char* planes = reinterpret_cast<char*>(out + r * 1536 + 768);
        planes[k] = (char)(f8_pack4(v * __builtin_ldexpf(1.0f, F8_ACT_HI_EXP), 0.f, 0.f, 0.f) & 0xff);
        planes[768 + k] = (char)(f8_pack4((v - T16<T>::to_f32(hi)) * __builtin_ldexpf(1.0f, F8_ACT_LO_EXP), 0.f, 0.f, 0.f) & 0xff);
    } else if (split) {   // rows of [768 hi | 768 lo]
        out[r * 1536 + k] = hi;
        out[r * 1536 + 768 + k] = T16<T>::from_f32(v - T16<T>::to_f32(hi));
    } else {
        out[gid] = hi;
    }
}

__global__ __launch_bounds__(256) void pre_image_kernel(const uint8_t* frames, float* out, int B, int nimg, PreGeom g) {
    const int64_t total = (int64_t)nimg * 3 * g.nh * g.nw;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    int x = (int)(gid % g.nw);
    const int y = (int)((gid / g.nw) % g.nh), c = (int)((gid / ((int64_t)g.nw * g.nh)) % 3), img = (int)(gid / ((int64_t)3 * g.nw * g.nh));
    const int b = img >= B ? img - B : img;
    if (img >= B) x = g.nw - 1 - x;
    out[gid] = pre_sample(frames + (int64_t)b * g.H * g.W * 3, g, c, y, x);
}

// ---------------------------------------------------------------------------------------------
// (hi16 | hi8 | lo8) pixels, 16 channels per thread: every access is 16 bytes wide (two for the hi16 values, one per FP8 plane);
// the x2 upsampling of the fusion stage / relative head writes 6-13 GB per call at the bench batch
// LO: also write the lo8 plane (false: every consumer of the map drops the activation-rounding correction and never reads it)
template <typename T, bool LO = true, bool HI8 = true>
__global__ __launch_bounds__(256) void resize_nhwc_f8_kernel(const T* x, T* out, int B, int Hin, int Win, int C, int Hout, int Wout, float sy,
                                                              float sx, int align) {
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
    typedef typename T16<T>::v8 v8;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const T* xb = x + (int64_t)b * Hin * Win * C * 2;
    const float sc = __builtin_ldexpf(1.0f, -F8_ACT_LO_EXP);
    float q00[16], q01[16], q10[16], q11[16];
    auto ld = [&](int yy, int xx, float (&q)[16]) {
        const T* pp = xb + ((int64_t)yy * Win + xx) * C * 2;
        const v8 h0 = *reinterpret_cast<const v8*>(pp + c16 * 16), h1 = *reinterpret_cast<const v8*>(pp + c16 * 16 + 8);
        const i32x4 l = *reinterpret_cast<const i32x4*>(reinterpret_cast<const char*>(pp + C + (C >> 1)) + c16 * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            q[e] = (float)h0[e];
            q[8 + e] = (float)h1[e];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float lo[4];
            f8_unpack4(l[g], lo);
#pragma unroll
            for (int e = 0; e < 4; ++e) q[4 * g + e] = fmaf(lo[e], sc, q[4 * g + e]);      // (sc is a power of two: the product is exact)
        }
    };
    ld(y0, x0, q00);
    ld(y0, x1, q01);
    ld(y1, x0, q10);
    ld(y1, x1, q11);
    v8 o0, o1;
    float vv[16], rl[16];
    const f32x2_ hx2 = {hx, hx}, lx2 = {lx, lx}, hy2 = {hy, hy}, ly2 = {ly, ly};
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
        const f32x2_ v2 = bilerp2(f32x2_{q00[e], q00[e + 1]}, f32x2_{q01[e], q01[e + 1]}, f32x2_{q10[e], q10[e + 1]}, f32x2_{q11[e], q11[e + 1]}, hx2, lx2, hy2, ly2);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float v = v2[u];
            vv[e + u] = v;
            const T h = T16<T>::from_f32(v);
            if (e + u < 8) o0[e + u] = h; else o1[e + u - 8] = h;
            if (LO) rl[e + u] = v - (float)h;
        }
    }
    T* op = out + pix * C * 2;
    *reinterpret_cast<v8*>(op + c16 * 16) = o0;
    *reinterpret_cast<v8*>(op + c16 * 16 + 8) = o1;
    if (!HI8) return;       // (hi16 | - | -): every consumer of the map runs one 16-bit pass
    const float sh = __builtin_ldexpf(1.0f, F8_ACT_HI_EXP), sl = __builtin_ldexpf(1.0f, F8_ACT_LO_EXP);
    char* planes = reinterpret_cast<char*>(op + C);
    i32x4 ph, pl;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        ph[g] = f8_pack4(vv[4 * g] * sh, vv[4 * g + 1] * sh, vv[4 * g + 2] * sh, vv[4 * g + 3] * sh);
        if (LO) pl[g] = f8_pack4(rl[4 * g] * sl, rl[4 * g + 1] * sl, rl[4 * g + 2] * sl, rl[4 * g + 3] * sl);
    }
    *reinterpret_cast<i32x4*>(planes + c16 * 16) = ph;
    if (LO) *reinterpret_cast<i32x4*>(planes + C + c16 * 16) = pl;
}

// ---------------------------------------------------------------------------------------------
// conv3x3(upsample x2(x)) from tap products taken at the LOW resolution (the relative head's conv2, HF modeling_zoedepth.py:358-362).
// Both steps are linear and the bilinear resize acts on each channel alone, so with y[q, tap, o] = sum_c W[o, c, tap] x[q, c] (one plain
// GEMM over the low-resolution pixels q: a quarter of the conv's FLOPs, and N = 9 Cout instead of an N = 32 conv that the LDS fill
// rate bounds)   out(p, o) = act(bias[o] + sum_tap [p + d_tap inside] * bilinear(y[., tap, o])(p + d_tap)):   the zero padding of the
// conv applies to the UPSAMPLED map, hence the inside test on p + d.  The upsampled map (13 GB at the bench batch) is never written.
// Thread = 4 output channels of one output pixel; the 36 16-byte reads per thread hit L1 / L2 (neighbouring pixels share corners).
template <typename T, int SPLIT>
__global__ __launch_bounds__(256) void upconv_tapsum_kernel(const float* y, const float* bias, T* out, int B, int Hin, int Win, int Co, int Hout,
                                                             int Wout, float sy, float sx, int align, int relu) {
    const int groups = Co >> 2;
    // a block is a TW x TH patch of output pixels (x fastest) times the channel groups: vertical neighbours share their low-resolution
    // rows in L1 (a 32 x 1 strip re-fetched every corner row from L2 for each output row)
    const int per = 256 / groups, TW = per >= 8 ? 8 : per, TH = per / TW;
    const int g = threadIdx.x % groups, px = threadIdx.x / groups;
    const int ox = blockIdx.x * TW + px % TW;
    const int nty = (Hout + TH - 1) / TH;
    const int b = blockIdx.y / nty, oy = (blockIdx.y - b * nty) * TH + px / TW;
    if (px >= per || ox >= Wout || oy >= Hout) return;
    const int ld = 9 * Co;
    const float* yb = y + (int64_t)b * Hin * Win * ld + 4 * g;
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    f32x4_ acc = *reinterpret_cast<const f32x4_*>(bias + 4 * g);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int r = oy + ky - 1;
        if (r < 0 || r >= Hout) continue;
        const float fy = align ? sy * (float)r : fmaxf(__fmaf_rn(sy, (float)r + 0.5f, -0.5f), 0.0f);
        int y0 = (int)fy;
        y0 = y0 > Hin - 1 ? Hin - 1 : y0;
        const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0);
        const float ly = fy - (float)y0, hy = 1.0f - ly;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int c = ox + kx - 1;
            if (c < 0 || c >= Wout) continue;
            const float fx = align ? sx * (float)c : fmaxf(__fmaf_rn(sx, (float)c + 0.5f, -0.5f), 0.0f);
            int x0 = (int)fx;
            x0 = x0 > Win - 1 ? Win - 1 : x0;
            const int x1 = x0 + (x0 < Win - 1 ? 1 : 0);
            const float lx = fx - (float)x0, hx = 1.0f - lx;
            const float* yt = yb + (ky * 3 + kx) * Co;
            const f32x4_ q00 = *reinterpret_cast<const f32x4_*>(yt + ((int64_t)y0 * Win + x0) * ld);
            const f32x4_ q01 = *reinterpret_cast<const f32x4_*>(yt + ((int64_t)y0 * Win + x1) * ld);
            const f32x4_ q10 = *reinterpret_cast<const f32x4_*>(yt + ((int64_t)y1 * Win + x0) * ld);
            const f32x4_ q11 = *reinterpret_cast<const f32x4_*>(yt + ((int64_t)y1 * Win + x1) * ld);
            // (fused and two channels per instruction; the build has -ffp-contract=off: 10 instructions per channel and tap otherwise)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 hx2 = {hx, hx}, lx2 = {lx, lx}, hy2 = {hy, hy}, ly2 = {ly, ly};
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const f32x2 top = __builtin_elementwise_fma(lx2, f32x2{q01[e], q01[e + 1]}, hx2 * f32x2{q00[e], q00[e + 1]});
                const f32x2 bot = __builtin_elementwise_fma(lx2, f32x2{q11[e], q11[e + 1]}, hx2 * f32x2{q10[e], q10[e + 1]});
                const f32x2 v = __builtin_elementwise_fma(ly2, bot, hy2 * top);
                acc[e] += v[0];
                acc[e + 1] += v[1];
            }
        }
    }
    if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.0f);
    }
    const int64_t pix = ((int64_t)b * Hout + oy) * Wout + ox;
    typedef T t4 __attribute__((ext_vector_type(4)));
    t4 hi;
    float rl[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = T16<T>::from_f32(acc[e]);
        rl[e] = acc[e] - T16<T>::to_f32(hi[e]);
    }
    if (SPLIT == 0) {
        *reinterpret_cast<t4*>(out + pix * Co + 4 * g) = hi;
    } else if (SPLIT == 1) {      // (hi | lo) 16-bit pairs
        t4 lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) lo[e] = T16<T>::from_f32(rl[e]);
        *reinterpret_cast<t4*>(out + pix * 2 * Co + 4 * g) = hi;
        *reinterpret_cast<t4*>(out + pix * 2 * Co + Co + 4 * g) = lo;
    } else {                      // (hi16 | hi8 | lo8)
        T* op = out + pix * 2 * Co;
        *reinterpret_cast<t4*>(op + 4 * g) = hi;
        const float sh = __builtin_ldexpf(1.0f, F8_ACT_HI_EXP), sl = __builtin_ldexpf(1.0f, F8_ACT_LO_EXP);
        char* planes = reinterpret_cast<char*>(op + Co);
        *reinterpret_cast<int*>(planes + 4 * g) = f8_pack4(acc[0] * sh, acc[1] * sh, acc[2] * sh, acc[3] * sh);
        *reinterpret_cast<int*>(planes + Co + 4 * g) = f8_pack4(rl[0] * sl, rl[1] * sl, rl[2] * sl, rl[3] * sl);
    }
}

// The same sum with the tap products of a 16 x 16 output tile's low-resolution window staged in LDS (round 4; x2 upsampling, Co = 32: the
// relative head).  The kernel above gathers its 36 corner vectors per (pixel, channel group) through L1 / L2: 21 GB through L1 for 7.2 GB
// of products, 4.6 ms.  Here a block of 512 threads copies the window -- at most 11 x 11 low-resolution pixels x 1152 bytes, whole rows
// contiguous in memory -- by LDS-DMA and every corner read is a ds_read_b128.  The four corner weights of a tap are formed once (w = {hy, ly} x
// {hx, lx}) and the corners accumulated by fused multiply-adds, two channels per instruction: 8 v_pk_fma_f32 per tap instead of the gather
// form's 12 packed operations + 4 adds (another association of the same fp32 sum).
constexpr int TS_TW = 16, TS_TH = 16, TS_LW = 11, TS_LH = 11, TS_TAB = 1024;   // TS_TAB: bytes of the per-block tables in front of the window
template <typename T, int SPLIT>
__global__ __launch_bounds__(512) void upconv_tapsum_lds_kernel(const float* y, const float* bias, T* out, int B, int Hin, int Win, int Hout, int Wout,
                                                                 float sy, float sx, int align, int relu) {
    constexpr int Co = 32, ld = 9 * Co;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntx = (Wout + TS_TW - 1) / TS_TW, nty = (Hout + TS_TH - 1) / TS_TH;
    const int tx0 = (blockIdx.x % ntx) * TS_TW, t2 = blockIdx.x / ntx;
    const int ty0 = (t2 % nty) * TS_TH, b = t2 / nty;
    auto src = [&](int r, float s, int n) {       // source coordinate of output row / column r (the kernel above's expression)
        return align ? s * (float)r : fmaxf(__fmaf_rn(s, (float)r + 0.5f, -0.5f), 0.0f);
    };
    auto lo_idx = [&](int r, float s, int n) {
        int i0 = (int)src(r, s, n);
        return i0 > n - 1 ? n - 1 : i0;
    };
    // window: the corner rows / columns of output rows ty0 - 1 .. ty0 + TH (those inside the image)
    const int rmin = ty0 > 0 ? ty0 - 1 : 0, rmax = ty0 + TS_TH < Hout ? ty0 + TS_TH : Hout - 1;
    const int cmin = tx0 > 0 ? tx0 - 1 : 0, cmax = tx0 + TS_TW < Wout ? tx0 + TS_TW : Wout - 1;
    const int wy0 = lo_idx(rmin, sy, Hin), wx0 = lo_idx(cmin, sx, Win);
    int wy1 = lo_idx(rmax, sy, Hin), wx1 = lo_idx(cmax, sx, Win);
    wy1 += wy1 < Hin - 1 ? 1 : 0;
    wx1 += wx1 < Win - 1 ? 1 : 0;
    const int LWc = wx1 - wx0 + 1, LHc = wy1 - wy0 + 1;        // <= TS_LW, TS_LH (checked on the host for the x2 geometry)
    const int row16 = LWc * (ld * 4 / 16), total16 = LHc * row16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* ybase = reinterpret_cast<const char*>(y + (((int64_t)b * Hin + wy0) * Win + wx0) * ld);
    for (int k0 = wave * 64; k0 < total16; k0 += 512) {
        const int k = k0 + lane;
        if (k < total16) {
            const int ly = k / row16, rem = k - ly * row16;
            glds16(ybase + ((int64_t)ly * Win * ld * 4) + (int64_t)rem * 16, smem + TS_TAB + k0 * 16);
        }
    }
    // Per-block tables in front of the window: for the TH + 2 output rows and TW + 2 output columns the tile's taps touch, the byte offsets of
    // the two corner rows / columns inside the window and the two interpolation weights.  A tap outside the image gets weights 0 (its
    // offsets point at the window's first pixel): no branches in the item loop, and the eight channel groups of a pixel no longer
    // recompute what they share.
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    i32x4_* tabs = reinterpret_cast<i32x4_*>(smem);          // [0, TH + 2): rows, [TH + 2, TH + TW + 4): columns; the window starts TS_TAB bytes in
    if (tid < TS_TH + 2 + TS_TW + 2) {
        const bool isrow = tid < TS_TH + 2;
        const int r = isrow ? ty0 - 1 + tid : tx0 - 1 + (tid - (TS_TH + 2));
        const int nout = isrow ? Hout : Wout, nin = isrow ? Hin : Win, w0 = isrow ? wy0 : wx0;
        const float s = isrow ? sy : sx;
        const int unit = isrow ? LWc * ld * 4 : ld * 4;
        i32x4_ e = {0, 0, 0, 0};
        if (r >= 0 && r < nout) {
            const float f = src(r, s, nin);
            int i0 = (int)f;
            i0 = i0 > nin - 1 ? nin - 1 : i0;
            const int i1 = i0 + (i0 < nin - 1 ? 1 : 0);
            const float l = f - (float)i0, h = 1.0f - l;
            e = i32x4_{(i0 - w0) * unit, (i1 - w0) * unit, __float_as_int(h), __float_as_int(l)};
        }
        tabs[tid] = e;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int g = tid & 7;
    const f32x4_ bias4 = *reinterpret_cast<const f32x4_*>(bias + 4 * g);
    const char* yg = smem + TS_TAB + 16 * g;
    for (int it = 0; it < TS_TW * TS_TH / 64; ++it) {
        const int px = it * 64 + (tid >> 3);
        const int lx_ = px % TS_TW, ly_ = px / TS_TW;
        const int ox = tx0 + lx_, oy = ty0 + ly_;
        if (ox >= Wout || oy >= Hout) continue;
        f32x2 a01 = {bias4[0], bias4[1]}, a23 = {bias4[2], bias4[3]};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const i32x4_ ry = tabs[ly_ + ky];
            const float hy = __int_as_float(ry[2]), wy = __int_as_float(ry[3]);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const i32x4_ cx = tabs[TS_TH + 2 + lx_ + kx];
                const float hx = __int_as_float(cx[2]), wx = __int_as_float(cx[3]);
                const char* yt = yg + (ky * 3 + kx) * Co * 4;
                const f32x4_ q00 = *reinterpret_cast<const f32x4_*>(yt + ry[0] + cx[0]);
                const f32x4_ q01 = *reinterpret_cast<const f32x4_*>(yt + ry[0] + cx[1]);
                const f32x4_ q10 = *reinterpret_cast<const f32x4_*>(yt + ry[1] + cx[0]);
                const f32x4_ q11 = *reinterpret_cast<const f32x4_*>(yt + ry[1] + cx[1]);
                const float w00 = hy * hx, w01 = hy * wx, w10 = wy * hx, w11 = wy * wx;
                const f32x2 v00 = {w00, w00}, v01 = {w01, w01}, v10 = {w10, w10}, v11 = {w11, w11};
                a01 = __builtin_elementwise_fma(v00, f32x2{q00[0], q00[1]}, a01);
                a23 = __builtin_elementwise_fma(v00, f32x2{q00[2], q00[3]}, a23);
                a01 = __builtin_elementwise_fma(v01, f32x2{q01[0], q01[1]}, a01);
                a23 = __builtin_elementwise_fma(v01, f32x2{q01[2], q01[3]}, a23);
                a01 = __builtin_elementwise_fma(v10, f32x2{q10[0], q10[1]}, a01);
                a23 = __builtin_elementwise_fma(v10, f32x2{q10[2], q10[3]}, a23);
                a01 = __builtin_elementwise_fma(v11, f32x2{q11[0], q11[1]}, a01);
                a23 = __builtin_elementwise_fma(v11, f32x2{q11[2], q11[3]}, a23);
            }
        }
        f32x4_ acc = {a01[0], a01[1], a23[0], a23[1]};
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.0f);
        }
        const int64_t pix = ((int64_t)b * Hout + oy) * Wout + ox;
        typedef T t4 __attribute__((ext_vector_type(4)));
        t4 hi;
        float rl[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = T16<T>::from_f32(acc[e]);
            rl[e] = acc[e] - T16<T>::to_f32(hi[e]);
        }
        if (SPLIT == 0) {
            *reinterpret_cast<t4*>(out + pix * Co + 4 * g) = hi;
        } else if (SPLIT == 1) {      // (hi | lo) 16-bit pairs
            t4 lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) lo[e] = T16<T>::from_f32(rl[e]);
            *reinterpret_cast<t4*>(out + pix * 2 * Co + 4 * g) = hi;
            *reinterpret_cast<t4*>(out + pix * 2 * Co + Co + 4 * g) = lo;
        } else {                      // (hi16 | hi8 | lo8)
            T* op = out + pix * 2 * Co;
            *reinterpret_cast<t4*>(op + 4 * g) = hi;
            const float sh = __builtin_ldexpf(1.0f, F8_ACT_HI_EXP), sl = __builtin_ldexpf(1.0f, F8_ACT_LO_EXP);
            char* planes = reinterpret_cast<char*>(op + Co);
            *reinterpret_cast<int*>(planes + 4 * g) = f8_pack4(acc[0] * sh, acc[1] * sh, acc[2] * sh, acc[3] * sh);
            *reinterpret_cast<int*>(planes + Co + 4 * g) = f8_pack4(rl[0] * sl, rl[1] * sl, rl[2] * sl, rl[3] * sl);
        }
    }
}

// NHWC bilinear resize (+ optional add): thread = one 8-channel group of one output pixel
// ---------------------------------------------------------------------------------------------
// BIAS_RELU (bs_resize_bias_relu_nhwc): out = relu(resize(x) + bias[c]) -- a 1x1 convolution + ReLU whose input is an upsampled map, evaluated as
// the convolution at the LOW resolution (linear, commutes with the resize) and this kernel.
template <typename T, bool ADD, int SPLIT, bool BIAS_RELU = false>
__global__ __launch_bounds__(256) void resize_nhwc_kernel(const T* x, const T* addend, T* out, int B, int Hin, int Win, int C, int Hout,
                                                           int Wout, float sy, float sx, int align, const float* bias = nullptr) {
    // SPLIT: the tensors hold (hi | lo) pairs, C channels each (pixel stride 2C); the value hi + lo is resampled and re-split
    // grid.y = (image, output row); grid.x covers (column, 8-channel group) of that row
    const int c8n = C >> 3;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)Wout * c8n) return;
    const int c8 = idx % c8n, ox = idx / c8n;
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
    typedef typename T16<T>::v8 v8;
    constexpr int PS = SPLIT ? 2 : 1;                       // pixel stride in units of C (both pair formats are 4C bytes per pixel)
    const T* xb = x + (int64_t)b * Hin * Win * C * PS + c8 * 8;
    float q00[8], q01[8], q10[8], q11[8], av[8];
    auto ld = [&](const T* ptr, float (&dst)[8]) {
        const v8 h = *reinterpret_cast<const v8*>(ptr);
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = (float)h[e];
        if (SPLIT == 1) {
            const v8 l = *reinterpret_cast<const v8*>(ptr + C);
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e] += (float)l[e];
        } else if (SPLIT == 2) {      // (hi16 | hi8 | lo8) pixel: ptr points at the 8 hi16 values of channel group c8
            const char* lo8 = reinterpret_cast<const char*>(ptr - c8 * 8 + C + (C >> 1)) + c8 * 8;
            const int p0 = *reinterpret_cast<const int*>(lo8), p1 = *reinterpret_cast<const int*>(lo8 + 4);
            float l0[4], l1[4];
            f8_unpack4(p0, l0);
            f8_unpack4(p1, l1);
            const float sc = __builtin_ldexpf(1.0f, -F8_ACT_LO_EXP);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dst[e] = fmaf(l0[e], sc, dst[e]);                 // (sc is a power of two: the product is exact)
                dst[4 + e] = fmaf(l1[e], sc, dst[4 + e]);
            }
        }
    };
    ld(xb + ((int64_t)y0 * Win + x0) * C * PS, q00);
    ld(xb + ((int64_t)y0 * Win + x1) * C * PS, q01);
    ld(xb + ((int64_t)y1 * Win + x0) * C * PS, q10);
    ld(xb + ((int64_t)y1 * Win + x1) * C * PS, q11);
    if (ADD) ld(addend + pix * C * PS + c8 * 8, av);
    v8 o, ol;
    float vv[8];
    const f32x2_ hx2 = {hx, hx}, lx2 = {lx, lx}, hy2 = {hy, hy}, ly2 = {ly, ly};
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        f32x2_ v = bilerp2(f32x2_{q00[e], q00[e + 1]}, f32x2_{q01[e], q01[e + 1]}, f32x2_{q10[e], q10[e + 1]}, f32x2_{q11[e], q11[e + 1]}, hx2, lx2, hy2, ly2);
        if (ADD) v += f32x2_{av[e], av[e + 1]};
        if (BIAS_RELU) {
            v += *reinterpret_cast<const f32x2_*>(bias + c8 * 8 + e);
            v = f32x2_{fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f)};
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            vv[e + u] = v[u];
            o[e + u] = T16<T>::from_f32(v[u]);
            if (SPLIT == 1) ol[e + u] = T16<T>::from_f32(v[u] - (float)o[e + u]);
        }
    }
    *reinterpret_cast<v8*>(out + pix * C * PS + c8 * 8) = o;
    if (SPLIT == 1) *reinterpret_cast<v8*>(out + pix * C * PS + C + c8 * 8) = ol;
    if (SPLIT == 2) {
        const float sh = __builtin_ldexpf(1.0f, F8_ACT_HI_EXP), sl = __builtin_ldexpf(1.0f, F8_ACT_LO_EXP);
        char* planes = reinterpret_cast<char*>(out + pix * C * 2 + C);
        typedef int i32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<i32x2*>(planes + c8 * 8) = i32x2{f8_pack4(vv[0] * sh, vv[1] * sh, vv[2] * sh, vv[3] * sh),
                                                           f8_pack4(vv[4] * sh, vv[5] * sh, vv[6] * sh, vv[7] * sh)};
        *reinterpret_cast<i32x2*>(planes + C + c8 * 8) =
            i32x2{f8_pack4((vv[0] - (float)o[0]) * sl, (vv[1] - (float)o[1]) * sl, (vv[2] - (float)o[2]) * sl, (vv[3] - (float)o[3]) * sl),
                  f8_pack4((vv[4] - (float)o[4]) * sl, (vv[5] - (float)o[5]) * sl, (vv[6] - (float)o[6]) * sl, (vv[7] - (float)o[7]) * sl)};
    }
}

// ---------------------------------------------------------------------------------------------
// post-processing: flip-average + bicubic (A = -0.75, align_corners=False) + crop + x256 -> u16
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
    const float A = -0.75f;
    auto c1 = [&](float x) { return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f; };
    auto c2 = [&](float x) { return ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A; };
    c[0] = c2(t + 1.0f);
    c[1] = c1(t);
    c[2] = c1(1.0f - t);
    c[3] = c2(2.0f - t);
}

__global__ __launch_bounds__(256) void postprocess_kernel(const float* dnet, float* depth_m, uint16_t* depth_u16, int B, int H, int W,
                                                           int nh, int nw, int ph, int pw, float sy, float sx, int flip) {
    const int64_t total = (int64_t)B * H * W;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int x = (int)(gid % W), y = (int)((gid / W) % H), b = (int)(gid / ((int64_t)W * H));
    const int yp = y + ph, xp = x + pw;  // position in the padded (resize target) frame
    // torch evaluates the source index with one rounding (FMA contraction, CPU and CUDA builds alike); near
    // index 384 a float ulp is 3e-5, so two roundings would move the cubic weights visibly
    const float fy = __fmaf_rn(sy, (float)yp + 0.5f, -0.5f), fx = __fmaf_rn(sx, (float)xp + 0.5f, -0.5f);
    const float fly = floorf(fy), flx = floorf(fx);
    const int iy = (int)fly, ix = (int)flx;
    float cy[4], cx[4];
    cubic_coeffs(fy - fly, cy);
    cubic_coeffs(fx - flx, cx);
    const float* d0 = dnet + (int64_t)b * nh * nw;
    const float* d1 = dnet + (int64_t)(B + b) * nh * nw;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int yy = iy - 1 + i;
        yy = yy < 0 ? 0 : (yy > nh - 1 ? nh - 1 : yy);
        float row = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int xx = ix - 1 + j;
            xx = xx < 0 ? 0 : (xx > nw - 1 ? nw - 1 : xx);
            float v = d0[(int64_t)yy * nw + xx];
            if (flip) v = (v + d1[(int64_t)yy * nw + (nw - 1 - xx)]) / 2.0f;
            row += v * cx[j];
        }
        acc += row * cy[i];
    }
    depth_m[gid] = acc;
    if (depth_u16) {
        float s = acc * 256.0f;  // numpy float32 -> uint16 cast: truncation toward zero
        s = s < 0.f ? 0.f : (s > 65535.f ? 65535.f : s);
        depth_u16[gid] = (uint16_t)s;
    }
}

static PreGeom make_geom(int H, int W, int nh, int nw) {
    PreGeom g;
    g.H = H; g.W = W; g.nh = nh; g.nw = nw;
    g.ph = (int)(sqrt((double)H / 2.0) * 3.0);
    g.pw = (int)(sqrt((double)W / 2.0) * 3.0);
    g.sy = nh > 1 ? (float)(H + 2 * g.ph - 1) / (float)(nh - 1) : 0.f;
    g.sx = nw > 1 ? (float)(W + 2 * g.pw - 1) / (float)(nw - 1) : 0.f;
    return g;
}

}  // namespace bs

using namespace bs;
#define BS_ENTRY(name) \
    if (!initialized()) { set_error(name ": call bs_init first"); return BS_ERR_NOT_INIT; }

extern "C" int bs_layernorm(const float* x, const float* gamma, const float* beta, void* out16, float* out32, int32_t rows,
                            int32_t cols, float eps, int32_t dtype, void* stream) {
    BS_ENTRY("bs_layernorm");
    BS_REQUIRE(x && gamma && beta && (out16 || out32) && rows >= 0 && cols > 0 && cols % 4 == 0, "bs_layernorm: bad argument");
    const int split = (dtype & 32) ? 2 : ((dtype & 16) ? 1 : 0);   // bit 4: (hi | lo) 16-bit pairs; bit 5: (hi16 | hi8 | lo8)
    const int plane_rows = dtype >> 8;     // bits 8..: with bit 5, > 0: only rows below this index write their FP8 planes
    dtype &= 15;
    BS_REQUIRE(!out16 || dtype == BS_F16 || dtype == BS_BF16, "bs_layernorm: dtype");
    if (rows == 0) return BS_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(cdiv(rows, 4));
    if (dtype == BS_BF16)
        hipLaunchKernelGGL(layernorm_kernel<bf16>, grid, dim3(256), 0, st, x, gamma, beta, (bf16*)out16, out32, rows, cols, eps, split, plane_rows);
    else
        hipLaunchKernelGGL(layernorm_kernel<f16>, grid, dim3(256), 0, st, x, gamma, beta, (f16*)out16, out32, rows, cols, eps, split, plane_rows);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_cast(const float* x, void* out, int64_t n, int32_t out_dtype, void* stream) {
    BS_ENTRY("bs_cast");
    BS_REQUIRE(x && out && n >= 0 && n % 4 == 0, "bs_cast: n must be a multiple of 4");
    BS_REQUIRE(out_dtype == BS_F16 || out_dtype == BS_BF16, "bs_cast: dtype");
    if (n == 0) return BS_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t n4 = n / 4;
    const unsigned blocks = (unsigned)(cdiv64(n4, 256) < 8192 ? cdiv64(n4, 256) : 8192);
    if (out_dtype == BS_F16)
        hipLaunchKernelGGL(cast_kernel<f16>, dim3(blocks), dim3(256), 0, st, x, (f16*)out, n4);
    else
        hipLaunchKernelGGL(cast_kernel<bf16>, dim3(blocks), dim3(256), 0, st, x, (bf16*)out, n4);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_fill_rows(float* x, const float* v, int32_t B, int32_t rows_per_image, int32_t cols, void* stream) {
    BS_ENTRY("bs_fill_rows");
    BS_REQUIRE(x && v && B >= 0 && rows_per_image > 0 && cols > 0, "bs_fill_rows: bad argument");
    if (B == 0) return BS_OK;
    hipLaunchKernelGGL(fill_rows_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, v, rows_per_image, cols);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_preprocess_patches(const uint8_t* frames, void* out, int32_t B, int32_t H, int32_t W, int32_t nh, int32_t nw,
                                     int32_t flip, int32_t out_dtype, void* stream) {
    BS_ENTRY("bs_preprocess_patches");
    BS_REQUIRE(frames && out && B >= 0 && H > 1 && W > 1 && nh % 16 == 0 && nw % 16 == 0 && nh > 0 && nw > 0,
               "bs_preprocess_patches: bad geometry");
    const int split = (out_dtype & 32) ? 2 : ((out_dtype & 16) ? 1 : 0);   // bit 4: (hi | lo) pairs, bit 5: (hi16 | hi8 | lo8); [., 2*768]
    out_dtype &= 15;
    BS_REQUIRE(out_dtype == BS_F16 || out_dtype == BS_BF16, "bs_preprocess_patches: dtype");
    if (B == 0) return BS_OK;
    const PreGeom g = make_geom(H, W, nh, nw);
    BS_REQUIRE(g.ph < H && g.pw < W, "bs_preprocess_patches: reflect pad exceeds the frame");
    const int nimg = flip ? 2 * B : B;
    const int64_t total = (int64_t)nimg * (nh / 16) * (nw / 16) * 768;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (out_dtype == BS_F16)
        hipLaunchKernelGGL(pre_patches_kernel<f16>, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, frames, (f16*)out, B, nimg, g, split);
    else
        hipLaunchKernelGGL(pre_patches_kernel<bf16>, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, frames, (bf16*)out, B, nimg, g, split);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_preprocess_image(const uint8_t* frames, float* out, int32_t B, int32_t H, int32_t W, int32_t nh, int32_t nw,
                                   int32_t flip, void* stream) {
    BS_ENTRY("bs_preprocess_image");
    BS_REQUIRE(frames && out && B >= 0 && H > 1 && W > 1 && nh > 0 && nw > 0, "bs_preprocess_image: bad geometry");
    if (B == 0) return BS_OK;
    const PreGeom g = make_geom(H, W, nh, nw);
    BS_REQUIRE(g.ph < H && g.pw < W, "bs_preprocess_image: reflect pad exceeds the frame");
    const int nimg = flip ? 2 * B : B;
    const int64_t total = (int64_t)nimg * 3 * nh * nw;
    hipLaunchKernelGGL(pre_image_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), frames,
                       out, B, nimg, g);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

template <typename T>
static int launch_resize(const void* x, const void* addend, void* out, int B, int Hin, int Win, int C, int Hout, int Wout, int align,
                         hipStream_t st, int split = 0) {
    float sy, sx;
    if (align) {
        sy = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f;
        sx = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
    } else {
        sy = (float)Hin / (float)Hout;
        sx = (float)Win / (float)Wout;
    }
    const dim3 blocks(cdiv(Wout * (C / 8), 256), B * Hout);
    if (split >= 2 && split <= 4) {      // (hi16 | hi8 | lo8) pixels (no add variant: the bins head's embeddings stay 16-bit pairs); 3: no lo8 plane out; 4: no plane out
        const dim3 blocks16(cdiv(Wout * (C / 16), 256), B * Hout);
        if (split == 4)
            hipLaunchKernelGGL((resize_nhwc_f8_kernel<T, false, false>), blocks16, dim3(256), 0, st, (const T*)x, (T*)out, B, Hin, Win, C, Hout, Wout, sy, sx, align);
        else if (split == 3)
            hipLaunchKernelGGL((resize_nhwc_f8_kernel<T, false>), blocks16, dim3(256), 0, st, (const T*)x, (T*)out, B, Hin, Win, C, Hout, Wout, sy, sx, align);
        else
            hipLaunchKernelGGL((resize_nhwc_f8_kernel<T, true>), blocks16, dim3(256), 0, st, (const T*)x, (T*)out, B, Hin, Win, C, Hout, Wout, sy, sx, align);
        BS_CHECK_LAUNCH();
        return BS_OK;
    }
    if (split) {
        if (addend)
            hipLaunchKernelGGL((resize_nhwc_kernel<T, true, 1>), blocks, dim3(256), 0, st, (const T*)x, (const T*)addend, (T*)out, B, Hin, Win,
                               C, Hout, Wout, sy, sx, align);
        else
            hipLaunchKernelGGL((resize_nhwc_kernel<T, false, 1>), blocks, dim3(256), 0, st, (const T*)x, (const T*)nullptr, (T*)out, B, Hin,
                               Win, C, Hout, Wout, sy, sx, align);
        BS_CHECK_LAUNCH();
        return BS_OK;
    }
    if (addend)
        hipLaunchKernelGGL((resize_nhwc_kernel<T, true, 0>), blocks, dim3(256), 0, st, (const T*)x, (const T*)addend, (T*)out, B, Hin,
                           Win, C, Hout, Wout, sy, sx, align);
    else
        hipLaunchKernelGGL((resize_nhwc_kernel<T, false, 0>), blocks, dim3(256), 0, st, (const T*)x, (const T*)nullptr, (T*)out, B,
                           Hin, Win, C, Hout, Wout, sy, sx, align);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

// ---------------------------------------------------------------------------------------------
// column means over a sample of each group's rows (bs_col_mean): block = one group x 256 columns; 32 lanes x 16 bytes cover the
// 256 columns of a row, the 8 row lanes of a block stride over the sampled rows; fp32 sums, LDS reduction over the row lanes
template <typename T>
__global__ __launch_bounds__(256) void col_mean_kernel(const T* A, int64_t lda, int row0, int rows_per_group, int row_step, int K,
                                                        bf16* out, float* zero_out, int64_t zero_n) {
    // (the accumulation target of bs_rank1_bias, cleared by the kernel that precedes it anyway)
    for (int64_t i = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < zero_n; i += (int64_t)gridDim.x * gridDim.y * 256)
        zero_out[i] = 0.0f;
    typedef typename T16<T>::v8 v8;
    __shared__ float red[8][32][8];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int g = blockIdx.y, c0 = (blockIdx.x * 32 + tx) * 8;
    const T* base = A + ((int64_t)row0 + (int64_t)g * rows_per_group) * lda;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < K) {
        for (int j = ty * row_step; j < rows_per_group; j += 8 * row_step) {
            const v8 v = *reinterpret_cast<const v8*>(base + (int64_t)j * lda + c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += T16<T>::to_f32(v[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty][tx][e] = acc[e];
    __syncthreads();
    if (ty == 0 && c0 < K) {
        const int count = (rows_per_group + row_step - 1) / row_step;
        const float inv = 1.0f / (float)count;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float s_ = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) s_ += red[r][tx][e];
            out[(int64_t)g * K + c0 + e] = T16<bf16>::from_f32(s_ * inv);
        }
    }
}

extern "C" int bs_col_mean(const void* A, int64_t lda, int32_t row0, int32_t rows_per_group, int32_t groups, int32_t row_step, int32_t K,
                           void* out_bf16, float* zero_out, int64_t zero_n, int32_t dtype, void* stream) {
    BS_ENTRY("bs_col_mean");
    BS_REQUIRE(A && out_bf16, "bs_col_mean: null operand");
    BS_REQUIRE(row0 >= 0 && rows_per_group > 0 && groups > 0 && row_step > 0 && K > 0 && K % 8 == 0 && lda >= K && lda % 8 == 0,
               "bs_col_mean: bad geometry (K and lda must be multiples of 8)");
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "bs_col_mean: dtype must be f16 or bf16");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(cdiv(K, 256), groups);
    if (dtype == BS_F16)
        hipLaunchKernelGGL((col_mean_kernel<f16>), grid, dim3(256), 0, st, (const f16*)A, lda, row0, rows_per_group, row_step, K,
                           (bf16*)out_bf16, zero_out, zero_out ? zero_n : 0);
    else
        hipLaunchKernelGGL((col_mean_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)A, lda, row0, rows_per_group, row_step, K,
                           (bf16*)out_bf16, zero_out, zero_out ? zero_n : 0);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

// ---------------------------------------------------------------------------------------------
// bs_rank1_bias: out[g, n] += sum_k abar[g, k] dW[n, k] -- a [G, K] x [K, N] product with G ~ 128 rows.  As a tile GEMM it is 8-32
// blocks walking a long K (latency-bound, ~100 us).  Round 2: a block owned 32 columns x half of K, the halves met by atomics (21 us:
// 16-64 dependent fragment loads per wave).  Round 3: a block owns 16 columns x 64 rows x ALL of K, its four waves each take a
// QUARTER of K (8-32 steps, four in flight), fragments go straight from L2 to registers (both operands are a few MB), 16x16x32 bf16
// MFMA, and the four partial tiles meet in LDS in wave order: no atomics, the sum has a fixed order.
#ifdef BS_DIAG
#define R1_ABL(x) (abl & (x))
#else
#define R1_ABL(x) 0
#endif
__global__ __launch_bounds__(256) void rank1_bias_kernel(const bf16* abar, const bf16* dW, float* out, int G, int N, int K, int abl) {
    // abl: diagnostics build only (BS_RANK1_ABLATE: 1 = no MFMA, 2 = no LDS exchange, 4 = no read-modify-write of the output, 8 = the f16 MFMA in place of the bf16 one, 16 = 120 KiB of unused dynamic LDS per block) -- wrong results, for
    // tools/probes/rank1_ablate.sh: which part of this kernel disturbs a kernel of another stream (the MFMA: profiles/r06_reproducibility.txt (8))
    typedef T16<bf16>::v8 v8;
    __shared__ float red[4][64][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16, g0 = blockIdx.y * 64;
    // k range of this wave: whole 32-element steps, the last wave takes the remainder
    const int steps = K / 32, per = steps / 4, s0 = wave * per, s1 = wave == 3 ? steps : s0 + per;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16* ap[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int g = g0 + i * 16 + frow;
        g = g < G ? g : G - 1;
        ap[i] = abar + (int64_t)g * K + fq * 8;
    }
    int n = n0 + frow;
    n = n < N ? n : N - 1;
    const bf16* bp = dW + (int64_t)n * K + fq * 8;
    auto step = [&](int k) {
        v8 af[4];
        const v8 bf = *reinterpret_cast<const v8*>(bp + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const v8*>(ap[i] + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (R1_ABL(1)) acc[i][0] += (float)af[i][0] + (float)bf[0];
            else if (R1_ABL(8)) acc[i] = T16<f16>::mfma16(__builtin_bit_cast(T16<f16>::v8, bf), __builtin_bit_cast(T16<f16>::v8, af[i]), acc[i]);   // the f16 instruction on the same bits
            else acc[i] = T16<bf16>::mfma16(bf, af[i], acc[i]);
        }
    };
    int st = s0;
    for (; st + 4 <= s1; st += 4) {            // four steps' loads in flight
#pragma unroll
        for (int u = 0; u < 4; ++u) step((st + u) * 32);
    }
    for (; st < s1; ++st) step(st * 32);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (!R1_ABL(2)) red[wave][i * 16 + frow][fq * 4 + e] = acc[i][e];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int idx = r * 256 + threadIdx.x, gl = idx >> 4, c = idx & 15;
        const int g = g0 + gl, nn = n0 + c;
        if (g < G && nn < N) {
            float* o = out + (int64_t)g * N + nn;
            if (R1_ABL(2)) *o += acc[0][0];
            else if (R1_ABL(4)) *o = ((red[0][gl][c] + red[1][gl][c]) + red[2][gl][c]) + red[3][gl][c];
            else *o += ((red[0][gl][c] + red[1][gl][c]) + red[2][gl][c]) + red[3][gl][c];
        }
    }
}

extern "C" int bs_rank1_bias(const void* abar_bf16, const void* dw_bf16, float* out, int32_t G, int32_t N, int32_t K, void* stream) {
    BS_ENTRY("bs_rank1_bias");
    BS_REQUIRE(abar_bf16 && dw_bf16 && out, "bs_rank1_bias: null operand");
    BS_REQUIRE(G > 0 && N > 0 && K > 0 && K % 64 == 0, "bs_rank1_bias: K=%d must be a multiple of 64", K);
    static const int abl = diag_env("BS_RANK1_ABLATE") ? atoi(diag_env("BS_RANK1_ABLATE")) : 0;
    size_t lds = 0;
#ifdef BS_DIAG
    if (abl & 16) {       // diagnostics: 120 KiB of (unused) dynamic LDS per block -- no other kernel's 42-KiB block fits on the CU beside it
        lds = 120 * 1024;
        BS_MAX_DYNAMIC_LDS(reinterpret_cast<const void*>(&rank1_bias_kernel), 120 * 1024);
    }
#endif
    hipLaunchKernelGGL(rank1_bias_kernel, dim3(cdiv(N, 16), cdiv(G, 64)), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), (const bf16*)abar_bf16,
                       (const bf16*)dw_bf16, out, G, N, K, abl);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

template <typename T>
static int launch_tapsum(const float* y, const float* bias, void* out, int B, int Hin, int Win, int Co, int Hout, int Wout, int align, int split,
                         int relu, hipStream_t st) {
    float sy, sx;
    if (align) {
        sy = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f;
        sx = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
    } else {
        sy = (float)Hin / (float)Hout;
        sx = (float)Win / (float)Wout;
    }
    // the LDS-staged form: the relative head's geometry (x2, 32 channels; its window bound of 11 x 11 holds for scale factors >= 0.49)
    static const bool no_lds = diag_env("BS_TAPSUM_NO_LDS") != nullptr;     // diagnostics
    if (!no_lds && Co == 32 && Hout == 2 * Hin && Wout == 2 * Win && Hin >= 2 && Win >= 2) {
        constexpr int smem = TS_TAB + TS_LW * TS_LH * 9 * 32 * 4;
        static_assert((TS_TH + 2 + TS_TW + 2) * 16 <= TS_TAB, "tables");
        const dim3 grid(cdiv(Wout, TS_TW) * cdiv(Hout, TS_TH) * B);
#define BS_TS(SP)                                                                                                                          \
    do {                                                                                                                                   \
        BS_MAX_DYNAMIC_LDS(((const void*)upconv_tapsum_lds_kernel<T, SP>), smem); \
        hipLaunchKernelGGL((upconv_tapsum_lds_kernel<T, SP>), grid, dim3(512), smem, st, y, bias, (T*)out, B, Hin, Win, Hout, Wout, sy, sx, align, \
                           relu);                                                                                                          \
    } while (0)
        if (split == 2) BS_TS(2);
        else if (split == 1) BS_TS(1);
        else BS_TS(0);
#undef BS_TS
        BS_CHECK_LAUNCH();
        return BS_OK;
    }
    const int groups = Co / 4, per = 256 / groups, TW = per >= 8 ? 8 : per, TH = per / TW;
    const dim3 blocks(cdiv(Wout, TW), B * cdiv(Hout, TH));
    if (split == 2)
        hipLaunchKernelGGL((upconv_tapsum_kernel<T, 2>), blocks, dim3(256), 0, st, y, bias, (T*)out, B, Hin, Win, Co, Hout, Wout, sy, sx, align, relu);
    else if (split == 1)
        hipLaunchKernelGGL((upconv_tapsum_kernel<T, 1>), blocks, dim3(256), 0, st, y, bias, (T*)out, B, Hin, Win, Co, Hout, Wout, sy, sx, align, relu);
    else
        hipLaunchKernelGGL((upconv_tapsum_kernel<T, 0>), blocks, dim3(256), 0, st, y, bias, (T*)out, B, Hin, Win, Co, Hout, Wout, sy, sx, align, relu);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_upconv_tapsum(const float* y, const float* bias, void* out, int32_t B, int32_t Hin, int32_t Win, int32_t Cout, int32_t Hout,
                                int32_t Wout, int32_t align_corners, int32_t relu, int32_t dtype, void* stream) {
    BS_ENTRY("bs_upconv_tapsum");
    BS_REQUIRE(y && bias && out, "bs_upconv_tapsum: null operand");
    BS_REQUIRE(B > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "bs_upconv_tapsum: empty problem");
    BS_REQUIRE(Cout > 0 && Cout % 4 == 0, "bs_upconv_tapsum: Cout=%d must be a multiple of 4", Cout);
    BS_REQUIRE((int64_t)B * Hout <= 0x7fffffffll && (int64_t)Wout * (Cout / 4) <= 0x7fffffffll, "bs_upconv_tapsum: grid too large");
    BS_REQUIRE(Cout / 4 <= 256 && 256 % (Cout / 4) == 0, "bs_upconv_tapsum: Cout / 4 = %d must divide 256", Cout / 4);
    const int split = (align_corners & 4) ? 2 : ((align_corners & 2) ? 1 : 0);
    const int align = align_corners & 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "bs_upconv_tapsum: dtype must be f16 or bf16");
    return dtype == BS_F16 ? launch_tapsum<f16>(y, bias, out, B, Hin, Win, Cout, Hout, Wout, align, split, relu, st)
                           : launch_tapsum<bf16>(y, bias, out, B, Hin, Win, Cout, Hout, Wout, align, split, relu, st);
}

extern "C" int bs_resize_bilinear_nhwc(const void* x, void* out, int32_t B, int32_t Hin, int32_t Win, int32_t C, int32_t Hout,
                                       int32_t Wout, int32_t align_corners, int32_t dtype, void* stream) {
    BS_ENTRY("bs_resize_bilinear_nhwc");
    BS_REQUIRE(x && out && B >= 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && C > 0 && C % 8 == 0, "bs_resize_bilinear_nhwc: bad argument");
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "bs_resize_bilinear_nhwc: dtype");
    if (B == 0) return BS_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // bit 1: (hi | lo) 16-bit pairs; bit 2: (hi16 | hi8 | lo8); bit 3 (with bit 2): the output's lo8 plane is not written (no consumer reads it);
    // bit 4 (with bit 2): neither plane is (every consumer runs one 16-bit pass)
    const int split = (align_corners & 4) ? ((align_corners & 16) ? 4 : ((align_corners & 8) ? 3 : 2)) : ((align_corners & 2) ? 1 : 0);
    const int ac = align_corners & 1;
    BS_REQUIRE(split < 2 || C % 16 == 0, "bs_resize_bilinear_nhwc: the FP8 pair format needs C %% 16 == 0");
    return dtype == BS_F16 ? launch_resize<f16>(x, nullptr, out, B, Hin, Win, C, Hout, Wout, ac, st, split)
                           : launch_resize<bf16>(x, nullptr, out, B, Hin, Win, C, Hout, Wout, ac, st, split);
}

extern "C" int bs_add_resized(const void* x, const void* prev, void* out, int32_t B, int32_t Hp, int32_t Wp, int32_t H, int32_t W,
                              int32_t C, int32_t dtype, void* stream) {
    BS_ENTRY("bs_add_resized");
    BS_REQUIRE(x && prev && out && B >= 0 && Hp > 0 && Wp > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "bs_add_resized: bad argument");
    const int split = (dtype & 16) ? 1 : 0;           // bit 4: x, prev and out hold (hi | lo) pairs of C channels each
    dtype &= 15;
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "bs_add_resized: dtype");
    if (B == 0) return BS_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dtype == BS_F16 ? launch_resize<f16>(prev, x, out, B, Hp, Wp, C, H, W, 1, st, split)
                           : launch_resize<bf16>(prev, x, out, B, Hp, Wp, C, H, W, 1, st, split);
}

extern "C" int bs_resize_bias_relu_nhwc(const void* x, const float* bias, void* out, int32_t B, int32_t Hin, int32_t Win, int32_t C, int32_t Hout,
                                        int32_t Wout, int32_t flags, int32_t dtype, void* stream) {
    BS_ENTRY("bs_resize_bias_relu_nhwc");
    BS_REQUIRE(x && bias && out, "bs_resize_bias_relu_nhwc: null operand");
    BS_REQUIRE(B > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && C > 0 && C % 8 == 0, "bs_resize_bias_relu_nhwc: bad shape (C %% 8 == 0)");
    BS_REQUIRE((flags & 1) && !(flags & 4), "bs_resize_bias_relu_nhwc: align_corners, single 16-bit or (hi | lo) pair rows");
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "bs_resize_bias_relu_nhwc: dtype must be f16 or bf16");
    const float sy = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f, sx = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
    const dim3 blocks(cdiv(Wout * (C / 8), 256), B * Hout);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define BS_RBR(TT)                                                                                                                        \
    do {                                                                                                                                  \
        if (flags & 2)                                                                                                                    \
            hipLaunchKernelGGL((resize_nhwc_kernel<TT, false, 1, true>), blocks, dim3(256), 0, st, (const TT*)x, (const TT*)nullptr, (TT*)out, B, Hin, \
                               Win, C, Hout, Wout, sy, sx, 1, bias);                                                                     \
        else                                                                                                                              \
            hipLaunchKernelGGL((resize_nhwc_kernel<TT, false, 0, true>), blocks, dim3(256), 0, st, (const TT*)x, (const TT*)nullptr, (TT*)out, B, Hin, \
                               Win, C, Hout, Wout, sy, sx, 1, bias);                                                                     \
    } while (0)
    if (dtype == BS_F16) BS_RBR(f16);
    else BS_RBR(bf16);
#undef BS_RBR
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_postprocess_depth(const float* depth_net, float* depth_m, uint16_t* depth_u16, int32_t B, int32_t H, int32_t W,
                                    int32_t nh, int32_t nw, int32_t flip, void* stream) {
    BS_ENTRY("bs_postprocess_depth");
    BS_REQUIRE(depth_net && depth_m && B >= 0 && H > 1 && W > 1 && nh > 0 && nw > 0, "bs_postprocess_depth: bad argument");
    if (B == 0) return BS_OK;
    const PreGeom g = make_geom(H, W, nh, nw);
    const int Hp = H + 2 * g.ph, Wp = W + 2 * g.pw;
    const float sy = (float)nh / (float)Hp, sx = (float)nw / (float)Wp;
    const int64_t total = (int64_t)B * H * W;
    hipLaunchKernelGGL(postprocess_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       depth_net, depth_m, depth_u16, B, H, W, nh, nw, g.ph, g.pw, sy, sx, flip);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_cast_split(const float* x, void* out, int64_t rows, int32_t cols, int32_t out_dtype, void* stream) {
    BS_ENTRY("bs_cast_split");
    BS_REQUIRE(x && out && rows >= 0 && cols > 0 && cols % 4 == 0, "bs_cast_split: cols must be a multiple of 4");
    const int f8 = (out_dtype & 32) ? 1 : 0;          // bit 5: (hi16 | hi8 | lo8) instead of (hi | lo) 16-bit pairs
    out_dtype &= 15;
    BS_REQUIRE(out_dtype == BS_F16 || out_dtype == BS_BF16, "bs_cast_split: dtype");
    if (rows == 0) return BS_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t n = rows * (cols / 4);
    const unsigned blocks = (unsigned)(cdiv64(n, 256) < 8192 ? cdiv64(n, 256) : 8192);
    if (out_dtype == BS_F16)
        hipLaunchKernelGGL(cast_split_kernel<f16>, dim3(blocks), dim3(256), 0, st, x, (f16*)out, rows, cols, f8);
    else
        hipLaunchKernelGGL(cast_split_kernel<bf16>, dim3(blocks), dim3(256), 0, st, x, (bf16*)out, rows, cols, f8);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_relu_split(const void* x, void* out, int64_t rows, int32_t cols, int32_t dtype, void* stream) {
    BS_ENTRY("bs_relu_split");
    BS_REQUIRE(x && out && rows >= 0 && cols > 0 && cols % 8 == 0, "bs_relu_split: cols must be a multiple of 8");
    // bit 5: (hi16 | hi8 | lo8) rows; bit 6 (with bit 5): without the lo8 plane; bit 7 (with bit 5): without either plane
    const int f8 = (dtype & 32) ? ((dtype & 128) ? 3 : ((dtype & 64) ? 2 : 1)) : 0;
    dtype &= 15;
    BS_REQUIRE(dtype == BS_F16 || dtype == BS_BF16, "bs_relu_split: dtype");
    if (rows == 0) return BS_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t n = rows * (cols / 8);
    const unsigned blocks = (unsigned)(cdiv64(n, 256) < 16384 ? cdiv64(n, 256) : 16384);
    if (dtype == BS_F16)
        hipLaunchKernelGGL(relu_split_kernel<f16>, dim3(blocks), dim3(256), 0, st, (const f16*)x, (f16*)out, rows, cols, f8);
    else
        hipLaunchKernelGGL(relu_split_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)x, (bf16*)out, rows, cols, f8);
    BS_CHECK_LAUNCH();
    return BS_OK;
}
