// Dense RGB-D odometry of the VO step: the role Open3D's rgbd_odometry_multi_scale (Hybrid, 20 / 10 / 5
// iterations) plays at BodySLAM_not_refactored/3DM/visual_odometry.py:97-120.  Open3D is un-vendored: parity unpinned; the algorithm
// is the statement in oracle/rgbd_odometry_ref.py (hybrid photometric + geometric Gauss-Newton on a 3-level pyramid, Huber losses),
// which these kernels implement term by term.  Two switches (the `flags` of bs_odo_accumulate / bs_odo_step): bit 0 = the target is
// read at the NEAREST pixel of the projected point (Open3D's association; default of the product) instead of bilinearly; bit 1 =
// Open3D's form of the robust step (J^T J unweighted, J^T applied to the Huber-clipped residual) instead of IRLS weights.
//
// All of it is HBM-bound streaming / reduction work on 1.2 MB images: images are fp32 in HBM (NaN = invalid depth), the per-pixel
// arithmetic runs in fp64 (the vector fp64 rate is not the limit), and one Gauss-Newton step is ONE kernel that reduces 29 sums
// (21 of the symmetric 6x6, 6 of the right-hand side, the cost, the inlier count) per block into a [blocks, 29] buffer plus a
// second, single-block kernel that adds the partials in a fixed order -- the result does not depend on scheduling.
#include "common.h"

namespace bs {

constexpr int ODO_TERMS = 29;
constexpr int ODO_GRID = 256;     // blocks of a Gauss-Newton step (one per CU; each walks its pixels in a fixed order: deterministic partials)

__global__ __launch_bounds__(256) void odo_prepare_kernel(const uint8_t* __restrict__ color, const float* __restrict__ depth, int64_t n, float depth_max,
                                                           float* __restrict__ inten, float* __restrict__ dout) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = (0.299 * (double)color[3 * i] + 0.587 * (double)color[3 * i + 1] + 0.114 * (double)color[3 * i + 2]) / 255.0;
    inten[i] = (float)v;
    const float d = depth[i];
    dout[i] = (d > 0.0f && d <= depth_max) ? d : __builtin_nanf("");
}

// pseudo-RGBD depth of the 3DM loop (3DM/slam_utils.py:212-220): z = u16 / depth_scale as fp32, z >= depth_trunc -> 0 (no measurement)
__global__ __launch_bounds__(256) void depth_u16_to_m_kernel(const uint16_t* __restrict__ d, int64_t n, float scale, float trunc, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float z = __fdiv_rn((float)d[i], scale);
    out[i] = z >= trunc ? 0.0f : z;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// level l -> l + 1: [1 4 6 4 1]^2 / 256 at every even pixel, replicate borders.  DEPTH: the weights run over the neighbours whose
// depth is within `thr` of the centre's (NaN neighbours drop out), NaN when the centre is invalid.
template <bool DEPTH>
__global__ __launch_bounds__(256) void odo_pyrdown_kernel(const float* __restrict__ src, int H, int W, float* __restrict__ dst, int h2, int w2, double thr) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w2 || y >= h2) return;
    src += (int64_t)blockIdx.z * H * W;            // image blockIdx.z of a batch of contiguous [H, W] images
    dst += (int64_t)blockIdx.z * h2 * w2;
    const double k5[5] = {1.0 / 16, 4.0 / 16, 6.0 / 16, 4.0 / 16, 1.0 / 16};
    const double centre = (double)src[(int64_t)clampi(2 * y, 0, H - 1) * W + clampi(2 * x, 0, W - 1)];
    if (DEPTH && centre != centre) {
        dst[(int64_t)y * w2 + x] = __builtin_nanf("");
        return;
    }
    double s = 0.0, wsum = 0.0;
    for (int dy = 0; dy < 5; ++dy)
        for (int dx = 0; dx < 5; ++dx) {
            const double v = (double)src[(int64_t)clampi(2 * y + dy - 2, 0, H - 1) * W + clampi(2 * x + dx - 2, 0, W - 1)];
            const double w = k5[dy] * k5[dx];
            if (DEPTH) {
                if (fabs(v - centre) <= thr) {        // false for NaN
                    s += w * v;
                    wsum += w;
                }
            } else {
                s += w * v;
            }
        }
    dst[(int64_t)y * w2 + x] = (float)(DEPTH ? s / wsum : s);
}

// 3x3 Sobel / 8, replicate borders; NaN propagates
__global__ __launch_bounds__(256) void odo_sobel_kernel(const float* __restrict__ img, int H, int W, float* __restrict__ gx, float* __restrict__ gy) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    img += (int64_t)blockIdx.z * H * W;
    gx += (int64_t)blockIdx.z * H * W;
    gy += (int64_t)blockIdx.z * H * W;
    auto at = [&](int yy, int xx) { return (double)img[(int64_t)clampi(yy, 0, H - 1) * W + clampi(xx, 0, W - 1)]; };
    const double a = at(y - 1, x - 1), b = at(y - 1, x), c = at(y - 1, x + 1), d = at(y, x - 1), f = at(y, x + 1), g = at(y + 1, x - 1), h = at(y + 1, x),
                 i = at(y + 1, x + 1);
    gx[(int64_t)y * W + x] = (float)(((c + 2.0 * f + i) - (a + 2.0 * d + g)) / 8.0);
    gy[(int64_t)y * W + x] = (float)(((g + 2.0 * h + i) - (a + 2.0 * b + c)) / 8.0);
}

struct OdoPose {
    double t[12];     // rows 0..2 of the source -> target 4x4
    double fx, fy, cx, cy;
};

// one Gauss-Newton step's sums, per block.  FLAGS bit 0: nearest-pixel association; bit 1: Open3D's Huber form
template <int FLAGS>
__global__ __launch_bounds__(256) void odo_accumulate_kernel(const float* __restrict__ Is, const float* __restrict__ Ds, const float* __restrict__ It,
                                                              const float* __restrict__ Dt, const float* __restrict__ dIx, const float* __restrict__ dIy,
                                                              const float* __restrict__ dDx, const float* __restrict__ dDy, int H, int W, OdoPose P,
                                                              const double* __restrict__ T_dev, double outlier, double huber_d, double huber_i,
                                                              double* __restrict__ partial) {
    {                     // pair blockIdx.y of a batch: source image y and target image y of two [batch, H, W] stacks, its own pose and partials
        const int64_t o = (int64_t)blockIdx.y * H * W;
        Is += o; Ds += o; It += o; Dt += o; dIx += o; dIy += o; dDx += o; dDy += o;
        partial += (int64_t)blockIdx.y * gridDim.x * ODO_TERMS;
    }
    if (T_dev) {          // the pose lives on the device (bs_odo_step chains steps without a host round trip)
#pragma unroll
        for (int i = 0; i < 12; ++i) P.t[i] = T_dev[(int64_t)blockIdx.y * 12 + i];
    }
    double acc[ODO_TERMS], tot[ODO_TERMS];
#pragma unroll
    for (int k = 0; k < ODO_TERMS; ++k) tot[k] = 0.0;
    for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < (int64_t)H * W; pix += (int64_t)gridDim.x * 256) {
#pragma unroll
    for (int k = 0; k < ODO_TERMS; ++k) acc[k] = 0.0;
    {
        const int v = (int)(pix / W), u = (int)(pix % W);
        const double z = (double)Ds[pix];
        bool ok = z == z;
        if (ok) {
            const double X = ((double)u - P.cx) * z / P.fx, Y = ((double)v - P.cy) * z / P.fy;
            const double px = P.t[0] * X + P.t[1] * Y + P.t[2] * z + P.t[3];
            const double py = P.t[4] * X + P.t[5] * Y + P.t[6] * z + P.t[7];
            const double pz = P.t[8] * X + P.t[9] * Y + P.t[10] * z + P.t[11];
            ok = pz > 0.0;
            if (ok) {
                const double uf = P.fx * px / pz + P.cx, vf = P.fy * py / pz + P.cy;
                constexpr bool NEAREST = (FLAGS & 1) != 0;
                const double ur = round(uf), vr = round(vf);           // (round half away from zero, as roundf)
                ok = NEAREST ? (ur >= 0.0 && ur <= (double)(W - 1) && vr >= 0.0 && vr <= (double)(H - 1))
                             : (uf >= 0.0 && uf <= (double)(W - 1) && vf >= 0.0 && vf <= (double)(H - 1));
                if (ok) {
                    int u0 = (int)floor(uf), v0 = (int)floor(vf);
                    u0 = u0 > W - 2 ? W - 2 : u0;
                    v0 = v0 > H - 2 ? H - 2 : v0;
                    u0 = u0 < 0 ? 0 : u0;
                    v0 = v0 < 0 ? 0 : v0;
                    const double au = uf - (double)u0, av = vf - (double)v0;
                    const int64_t o00 = (int64_t)v0 * W + u0;
                    const int64_t onn = (int64_t)(int)vr * W + (int)ur;
                    auto bil = [&](const float* __restrict__ img) {
                        if (NEAREST) return (double)img[onn];
                        return (1.0 - av) * ((1.0 - au) * (double)img[o00] + au * (double)img[o00 + 1]) +
                               av * ((1.0 - au) * (double)img[o00 + W] + au * (double)img[o00 + W + 1]);
                    };
                    const double dt = bil(Dt), hx = bil(dDx), hy = bil(dDy);
                    const double rD = dt - pz;
                    ok = dt == dt && hx == hx && hy == hy && fabs(rD) <= outlier;
                    if (ok) {
                        const double gx = bil(dIx), gy = bil(dIy);
                        const double rI = bil(It) - (double)Is[pix];
                        const double iz = 1.0 / pz;
                        const double c0 = gx * P.fx * iz, c1 = gy * P.fy * iz, c2 = -(c0 * px + c1 * py) * iz;
                        const double d0 = hx * P.fx * iz, d1 = hy * P.fy * iz, d2 = -(d0 * px + d1 * py) * iz;
                        const double JI[6] = {-pz * c1 + py * c2, pz * c0 - px * c2, -py * c0 + px * c1, c0, c1, c2};
                        const double JD[6] = {(-pz * d1 + py * d2) - py, (pz * d0 - px * d2) + px, -py * d0 + px * d1, d0, d1, d2 - 1.0};
                        if constexpr ((FLAGS & 2) != 0) {
                            // Open3D: sum J^T J unweighted, sum J^T huber'(r), cost = sum huber(r)
                            const double qI = fabs(rI) < huber_i ? rI : copysign(huber_i, rI), qD = fabs(rD) < huber_d ? rD : copysign(huber_d, rD);
                            int k = 0;
#pragma unroll
                            for (int a = 0; a < 6; ++a)
#pragma unroll
                                for (int b = a; b < 6; ++b) acc[k++] = JI[a] * JI[b] + JD[a] * JD[b];
#pragma unroll
                            for (int a = 0; a < 6; ++a) acc[21 + a] = JI[a] * qI + JD[a] * qD;
                            acc[27] = (fabs(rI) < huber_i ? 0.5 * rI * rI : huber_i * (fabs(rI) - 0.5 * huber_i)) +
                                      (fabs(rD) < huber_d ? 0.5 * rD * rD : huber_d * (fabs(rD) - 0.5 * huber_d));
                        } else {
                            const double wI = fabs(rI) <= huber_i ? 1.0 : huber_i / fabs(rI);
                            const double wD = fabs(rD) <= huber_d ? 1.0 : huber_d / fabs(rD);
                            int k = 0;
#pragma unroll
                            for (int a = 0; a < 6; ++a)
#pragma unroll
                                for (int b = a; b < 6; ++b) acc[k++] = wI * JI[a] * JI[b] + wD * JD[a] * JD[b];
#pragma unroll
                            for (int a = 0; a < 6; ++a) acc[21 + a] = wI * JI[a] * rI + wD * JD[a] * rD;
                            acc[27] = wI * rI * rI + wD * rD * rD;
                        }
                        acc[28] = 1.0;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < ODO_TERMS; ++k) tot[k] += acc[k];
    }
    __shared__ double red[4][ODO_TERMS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < ODO_TERMS; ++k) {
        double s = tot[k];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < ODO_TERMS) partial[(int64_t)blockIdx.x * ODO_TERMS + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// the partials of all blocks, added in a fixed order: thread k owns term k; 256 threads = 8 slices of the blocks per term... kept simple:
// one wave per term group would not be faster for a few thousand blocks
__global__ __launch_bounds__(64) void odo_finish_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ out) {
    const int k = blockIdx.x;                 // one block (one wave) per term
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partial[(int64_t)b * ODO_TERMS + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (threadIdx.x == 0) out[k] = s;
}

// delta = -(A + 1e-12 I)^-1 b by Gaussian elimination with partial pivoting, T <- exp(delta) T (left twist: omega = delta[0:3],
// nu = delta[3:6]); fewer than 6 inliers or a singular system leave T alone.  One thread: 6x6.
__device__ void odo_solve(const double* __restrict__ s29, double* __restrict__ T) {
    if (s29[28] < 6.0) return;
    double M[6][7];
    int k = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) {
            M[a][b] = s29[k];
            M[b][a] = s29[k];
            ++k;
        }
    for (int a = 0; a < 6; ++a) {
        M[a][a] += 1e-12;
        M[a][6] = -s29[21 + a];
    }
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        for (int r = c + 1; r < 6; ++r)
            if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
        if (!(fabs(M[piv][c]) > 0.0)) return;
        if (piv != c)
            for (int j = 0; j < 7; ++j) {
                const double t = M[c][j];
                M[c][j] = M[piv][j];
                M[piv][j] = t;
            }
        for (int r = c + 1; r < 6; ++r) {
            const double f = M[r][c] / M[c][c];
            for (int j = c; j < 7; ++j) M[r][j] -= f * M[c][j];
        }
    }
    double d[6];
    for (int r = 5; r >= 0; --r) {
        double v = M[r][6];
        for (int j = r + 1; j < 6; ++j) v -= M[r][j] * d[j];
        d[r] = v / M[r][r];
    }
    const double wx = d[0], wy = d[1], wz = d[2];
    const double th = sqrt(wx * wx + wy * wy + wz * wz);
    const double Wx[3][3] = {{0.0, -wz, wy}, {wz, 0.0, -wx}, {-wy, wx, 0.0}};
    double W2[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) W2[i][j] = Wx[i][0] * Wx[0][j] + Wx[i][1] * Wx[1][j] + Wx[i][2] * Wx[2][j];
    double a, b, c;
    if (th < 1e-12) {
        a = 1.0; b = 0.5; c = 0.0;                 // R = I + Wx, V = I + Wx / 2 (the oracle's small-angle branch)
    } else {
        a = sin(th) / th;
        b = (1.0 - cos(th)) / (th * th);
        c = (th - sin(th)) / (th * th * th);
    }
    double R[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double I = i == j ? 1.0 : 0.0;
            R[i][j] = I + a * Wx[i][j] + (th < 1e-12 ? 0.0 : b * W2[i][j]);
            V[i][j] = I + b * Wx[i][j] + c * W2[i][j];
        }
    const double t[3] = {V[0][0] * d[3] + V[0][1] * d[4] + V[0][2] * d[5], V[1][0] * d[3] + V[1][1] * d[4] + V[1][2] * d[5],
                         V[2][0] * d[3] + V[2][1] * d[4] + V[2][2] * d[5]};
    double N[12];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 4; ++j) N[4 * i + j] = R[i][0] * T[j] + R[i][1] * T[4 + j] + R[i][2] * T[8 + j];
        N[4 * i + 3] += t[i];
    }
    for (int i = 0; i < 12; ++i) T[i] = N[i];
}

__global__ void odo_solve_kernel(const double* __restrict__ s29, double* __restrict__ T) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    odo_solve(s29, T);
}

// the fixed-order reduction of the block partials (wave w: terms w and w + 16, the same lane-strided order as odo_finish_kernel: bit-equal
// sums) and the 6x6 solve + pose update in ONE launch: a Gauss-Newton step is two dependent launches instead of three
__global__ __launch_bounds__(1024) void odo_finish_solve_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ out, double* __restrict__ T) {
    __shared__ double s29[ODO_TERMS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    partial += (int64_t)blockIdx.x * nblocks * ODO_TERMS;      // pair blockIdx.x of a batch
    out += (int64_t)blockIdx.x * ODO_TERMS;
    T += (int64_t)blockIdx.x * 12;
    for (int k = wave; k < ODO_TERMS; k += 16) {
        double s = 0.0;
        for (int b = lane; b < nblocks; b += 64) s += partial[(int64_t)b * ODO_TERMS + k];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) {
            s29[k] = s;
            out[k] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) odo_solve(s29, T);
}

template <int FLAGS>
static void launch_accumulate(dim3 nblocks, hipStream_t st, const float* a, const float* b, const float* c, const float* d, const float* e, const float* f,
                              const float* g, const float* h, int H, int W, const OdoPose& P, const double* T_dev, double o, double hd, double hi,
                              double* partial) {
    hipLaunchKernelGGL(odo_accumulate_kernel<FLAGS>, nblocks, dim3(256), 0, st, a, b, c, d, e, f, g, h, H, W, P, T_dev, o, hd, hi, partial);
}
static void launch_accumulate_flags(int flags, dim3 nblocks, hipStream_t st, const float* a, const float* b, const float* c, const float* d, const float* e,
                                    const float* f, const float* g, const float* h, int H, int W, const OdoPose& P, const double* T_dev, double o,
                                    double hd, double hi, double* partial) {
    switch (flags & 3) {
        case 0: launch_accumulate<0>(nblocks, st, a, b, c, d, e, f, g, h, H, W, P, T_dev, o, hd, hi, partial); break;
        case 1: launch_accumulate<1>(nblocks, st, a, b, c, d, e, f, g, h, H, W, P, T_dev, o, hd, hi, partial); break;
        case 2: launch_accumulate<2>(nblocks, st, a, b, c, d, e, f, g, h, H, W, P, T_dev, o, hd, hi, partial); break;
        default: launch_accumulate<3>(nblocks, st, a, b, c, d, e, f, g, h, H, W, P, T_dev, o, hd, hi, partial); break;
    }
}

}  // namespace bs

using namespace bs;
#define ODO_ENTRY(name) \
    if (!initialized()) { set_error(name ": call bs_init first"); return BS_ERR_NOT_INIT; }

extern "C" int bs_depth_u16_to_m(const uint16_t* depth_u16, int64_t n, double depth_scale, double depth_trunc, float* out, void* stream) {
    ODO_ENTRY("bs_depth_u16_to_m");
    BS_REQUIRE(depth_u16 && out && n >= 0 && depth_scale > 0.0, "bs_depth_u16_to_m: bad argument");
    if (n == 0) return BS_OK;
    hipLaunchKernelGGL(depth_u16_to_m_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), depth_u16, n,
                       (float)depth_scale, (float)depth_trunc, out);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_odo_prepare(const uint8_t* color, const float* depth, int32_t batch, int32_t H, int32_t W, double depth_max, float* intensity,
                              float* depth_out, void* stream) {
    ODO_ENTRY("bs_odo_prepare");
    BS_REQUIRE(color && depth && intensity && depth_out && H > 0 && W > 0 && batch > 0, "bs_odo_prepare: bad argument");
    const int64_t n = (int64_t)batch * H * W;
    hipLaunchKernelGGL(odo_prepare_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), color, depth, n,
                       (float)depth_max, intensity, depth_out);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_odo_pyrdown(const float* src, int32_t batch, int32_t H, int32_t W, float* dst, int32_t is_depth, double depth_threshold,
                              void* stream) {
    ODO_ENTRY("bs_odo_pyrdown");
    BS_REQUIRE(src && dst && H > 1 && W > 1 && batch > 0 && batch <= 65535, "bs_odo_pyrdown: bad argument");
    const int h2 = (H + 1) / 2, w2 = (W + 1) / 2;
    const dim3 grid(cdiv(w2, 16), cdiv(h2, 16), batch);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (is_depth)
        hipLaunchKernelGGL(odo_pyrdown_kernel<true>, grid, dim3(256), 0, st, src, H, W, dst, h2, w2, depth_threshold);
    else
        hipLaunchKernelGGL(odo_pyrdown_kernel<false>, grid, dim3(256), 0, st, src, H, W, dst, h2, w2, 0.0);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_odo_sobel(const float* img, int32_t batch, int32_t H, int32_t W, float* gx, float* gy, void* stream) {
    ODO_ENTRY("bs_odo_sobel");
    BS_REQUIRE(img && gx && gy && H > 0 && W > 0 && batch > 0 && batch <= 65535, "bs_odo_sobel: bad argument");
    hipLaunchKernelGGL(odo_sobel_kernel, dim3(cdiv(W, 16), cdiv(H, 16), batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), img, H, W, gx, gy);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_odo_accumulate(const float* src_intensity, const float* src_depth, const float* tgt_intensity, const float* tgt_depth,
                                 const float* tgt_dIx, const float* tgt_dIy, const float* tgt_dDx, const float* tgt_dDy, int32_t H, int32_t W,
                                 const double* K, const double* T, double depth_outlier_trunc, double depth_huber, double intensity_huber,
                                 double* partial, double* out29, int32_t flags, void* stream) {
    ODO_ENTRY("bs_odo_accumulate");
    BS_REQUIRE(src_intensity && src_depth && tgt_intensity && tgt_depth && tgt_dIx && tgt_dIy && tgt_dDx && tgt_dDy && K && T && partial && out29,
               "bs_odo_accumulate: null argument");
    BS_REQUIRE(H > 1 && W > 1, "bs_odo_accumulate: image too small");
    OdoPose P;
    for (int i = 0; i < 12; ++i) P.t[i] = T[i];
    P.fx = K[0]; P.fy = K[1]; P.cx = K[2]; P.cy = K[3];
    const int nblocks = (int)(cdiv64((int64_t)H * W, 256) < ODO_GRID ? cdiv64((int64_t)H * W, 256) : ODO_GRID);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    launch_accumulate_flags(flags, dim3(nblocks), st, src_intensity, src_depth, tgt_intensity, tgt_depth, tgt_dIx, tgt_dIy, tgt_dDx, tgt_dDy, H, W, P,
                            (const double*)nullptr, depth_outlier_trunc, depth_huber, intensity_huber, partial);
    BS_CHECK_LAUNCH();
    hipLaunchKernelGGL(odo_finish_kernel, dim3(ODO_TERMS), dim3(64), 0, st, partial, nblocks, out29);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_odo_step(const float* src_intensity, const float* src_depth, const float* tgt_intensity, const float* tgt_depth,
                           const float* tgt_dIx, const float* tgt_dIy, const float* tgt_dDx, const float* tgt_dDy, int32_t batch, int32_t H, int32_t W,
                           const double* K, double* T_dev, int32_t iterations, double depth_outlier_trunc, double depth_huber,
                           double intensity_huber, double* partial, double* out29, int32_t flags, void* stream) {
    ODO_ENTRY("bs_odo_step");
    BS_REQUIRE(src_intensity && src_depth && tgt_intensity && tgt_depth && tgt_dIx && tgt_dIy && tgt_dDx && tgt_dDy && K && T_dev && partial && out29,
               "bs_odo_step: null argument");
    BS_REQUIRE(H > 1 && W > 1 && iterations >= 0 && batch > 0 && batch <= 65535, "bs_odo_step: bad geometry");
    OdoPose P;
    for (int i = 0; i < 12; ++i) P.t[i] = 0.0;
    P.fx = K[0]; P.fy = K[1]; P.cx = K[2]; P.cy = K[3];
    const int nblocks = (int)(cdiv64((int64_t)H * W, 256) < ODO_GRID ? cdiv64((int64_t)H * W, 256) : ODO_GRID);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int it = 0; it < iterations; ++it) {
        launch_accumulate_flags(flags, dim3(nblocks, batch), st, src_intensity, src_depth, tgt_intensity, tgt_depth, tgt_dIx, tgt_dIy, tgt_dDx, tgt_dDy, H, W,
                                P, (const double*)T_dev, depth_outlier_trunc, depth_huber, intensity_huber, partial);
        BS_CHECK_LAUNCH();
        hipLaunchKernelGGL(odo_finish_solve_kernel, dim3(batch), dim3(1024), 0, st, partial, nblocks, out29, T_dev);
        BS_CHECK_LAUNCH();
    }
    return BS_OK;
}
