// 3DM: depth -> point cloud back-projection with order-preserving compaction, and the SE(3) pose
// chain with per-step SO(3) projection.  HBM-bound scan work: 16-byte loads of the uint16 depth,
// wave ballots + block prefix sums for the compaction, 12 + 4 bytes written per valid pixel.
//   pixel_to_3d            BodySLAM_not_refactored/3DM/scaling_system.py:72-77
//   RGBD depth constants   BodySLAM_not_refactored/3DM/slam_utils.py:173,212-220,232
//   compute_curr_estimate_global_pose / ensure_so3_v2   3DM/slam_utils.py:93-122
#include "common.h"

namespace bs {

constexpr int BP_THREADS = 256;
constexpr int BP_PER_THREAD = 8;
constexpr int BP_CHUNK = BP_THREADS * BP_PER_THREAD;  // 2048 pixels per block

// validity exactly as the reference's float32 image path: z = d / scale (fp32), z >= trunc -> 0, valid iff z > 0
__device__ __forceinline__ float depth_z32(unsigned d, float scale32, float trunc32) {
    float z = __fdiv_rn((float)d, scale32);
    return z >= trunc32 ? 0.0f : z;
}

__device__ __forceinline__ void load8_u16(const uint16_t* img, int64_t base, int64_t npix, unsigned (&d)[8]) {
    if (base + 8 <= npix && ((reinterpret_cast<uintptr_t>(img + base) & 15) == 0)) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(img + base);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d[2 * i] = v[i] & 0xFFFFu;
            d[2 * i + 1] = v[i] >> 16;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = (base + i < npix) ? img[base + i] : 0u;
    }
}

__global__ __launch_bounds__(BP_THREADS) void bp_count_kernel(const uint16_t* depth, int64_t npix, int nchunk, float scale32,
                                                               float trunc32, int32_t* chunk_count) {
    BS_ARG_NOW(chunk_count);
    const int b = blockIdx.y, c = blockIdx.x;
    const uint16_t* img = depth + (int64_t)b * npix;
    const int64_t base = (int64_t)c * BP_CHUNK + threadIdx.x * BP_PER_THREAD;
    unsigned d[8];
    load8_u16(img, base, npix, d);
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) cnt += depth_z32(d[i], scale32, trunc32) > 0.0f ? 1 : 0;
    // block reduce
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    __shared__ int ws[BP_THREADS / 64];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int i = 0; i < BP_THREADS / 64; ++i) s += ws[i];
        chunk_count[(int64_t)b * (nchunk + 1) + c] = s;
    }
}

// one block per image: exclusive scan of the chunk counts (in place), total -> count[b]
__global__ __launch_bounds__(256) void bp_scan_kernel(int32_t* chunk_count, int nchunk, int32_t* count) {
    BS_ARG_NOW(count);
    const int b = blockIdx.x;
    int32_t* cc = chunk_count + (int64_t)b * (nchunk + 1);
    __shared__ int carry_s;
    __shared__ int ws[4];
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nchunk; base += 256) {
        const int i = base + threadIdx.x;
        const int v = i < nchunk ? cc[i] : 0;
        int incl = v;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) ws[w] = incl;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += ws[k];
        const int carry = carry_s;
        if (i < nchunk) cc[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cc[nchunk] = carry_s;
        count[b] = carry_s;
    }
}

__global__ __launch_bounds__(BP_THREADS) void bp_write_kernel(const uint16_t* depth, int H, int W, int nchunk, float scale32,
                                                               float trunc32, double fx, double fy, double cx, double cy,
                                                               const double* poses, const int32_t* chunk_off, float* xyz,
                                                               int32_t* idx) {
    const int b = blockIdx.y, c = blockIdx.x;
    const int64_t npix = (int64_t)H * W;
    const uint16_t* img = depth + (int64_t)b * npix;
    const int64_t base = (int64_t)c * BP_CHUNK + threadIdx.x * BP_PER_THREAD;
    unsigned d[8];
    load8_u16(img, base, npix, d);
    float z[8];
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        z[i] = depth_z32(d[i], scale32, trunc32);
        cnt += z[i] > 0.0f ? 1 : 0;
    }
    // exclusive scan of cnt over the block in thread order (= row-major pixel order)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __shared__ int ws[BP_THREADS / 64];
    if (lane == 63) ws[w] = incl;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < w; ++k) woff += ws[k];
    int64_t o = (int64_t)b * npix + chunk_off[(int64_t)b * (nchunk + 1) + c] + woff + incl - cnt;
    double P[12];
    const bool has_pose = poses != nullptr;
    if (has_pose) {
#pragma unroll
        for (int i = 0; i < 12; ++i) P[i] = poses[(int64_t)b * 16 + i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (!(z[i] > 0.0f)) continue;
        const int64_t pix = base + i;
        const int v = (int)(pix / W), u = (int)(pix - (int64_t)v * W);
        double zz = (double)z[i];
        double x = ((double)u - cx) * zz / fx;
        double y = ((double)v - cy) * zz / fy;
        if (has_pose) {
            const double wx = P[0] * x + P[1] * y + P[2] * zz + P[3];
            const double wy = P[4] * x + P[5] * y + P[6] * zz + P[7];
            const double wz = P[8] * x + P[9] * y + P[10] * zz + P[11];
            x = wx; y = wy; zz = wz;
        }
        xyz[3 * o + 0] = (float)x;
        xyz[3 * o + 1] = (float)y;
        xyz[3 * o + 2] = (float)zz;
        idx[o] = (int32_t)pix;
        ++o;
    }
}

// ---------------------------------------------------------------------------------------------
// pose chain: G_i = G_{i-1} * T_i (fp64), R <- U diag(1, 1, det(U) det(V^T)) V^T.
// One lane walks the chain (it is a strict recurrence); the 3x3 SVD is a one-sided Jacobi in fp64.
// ---------------------------------------------------------------------------------------------
__device__ void svd3_project_so3(double (&M)[3][3]) {
    // one-sided Jacobi on the columns of A = M: A V = U S
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = M[i][j];
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; ++i) {
                    alpha += A[i][p] * A[i][p];
                    beta += A[i][q] * A[i][q];
                    gamma += A[i][p] * A[i][q];
                }
                off = fmax(off, fabs(gamma) / sqrt(alpha * beta + 1e-300));
                if (fabs(gamma) <= 1e-300) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < 3; ++i) {
                    const double ap = A[i][p], aq = A[i][q];
                    A[i][p] = cs * ap - sn * aq;
                    A[i][q] = sn * ap + cs * aq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = cs * vp - sn * vq;
                    V[i][q] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-17) break;
    }
    double s[3], U[3][3];
    for (int j = 0; j < 3; ++j) {
        s[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
        const double inv = s[j] > 0 ? 1.0 / s[j] : 0.0;
        for (int i = 0; i < 3; ++i) U[i][j] = A[i][j] * inv;
    }
    // the reflection correction acts on the SMALLEST singular direction (LAPACK orders them descending)
    // (ties -> the last index, which is where a descending LAPACK ordering leaves the flipped direction)
    int kmin = 0;
    if (s[1] <= s[kmin]) kmin = 1;
    if (s[2] <= s[kmin]) kmin = 2;
    auto det3 = [](const double (&X)[3][3]) {
        return X[0][0] * (X[1][1] * X[2][2] - X[1][2] * X[2][1]) - X[0][1] * (X[1][0] * X[2][2] - X[1][2] * X[2][0]) +
               X[0][2] * (X[1][0] * X[2][1] - X[1][1] * X[2][0]);
    };
    const double dd = det3(U) * det3(V);  // det(V^T) == det(V)
    double D[3] = {1.0, 1.0, 1.0};
    D[kmin] = dd;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += U[i][k] * D[k] * V[j][k];
            M[i][j] = acc;
        }
}

// The strict recurrence, one lane (kept for inputs whose 4x4 matrices are not affine: a bottom row other than [0 0 0 1]).
__device__ void pose_chain_sequential(const float* t_rel, int N, const double* g0, double* g_abs) {
    double G[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) G[i][j] = g0[i * 4 + j];
    for (int n = 0; n < N; ++n) {
        double T[4][4], C[4][4];
        for (int i = 0; i < 16; ++i) T[i / 4][i % 4] = (double)t_rel[(int64_t)n * 16 + i];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double acc = 0;
                for (int k = 0; k < 4; ++k) acc += G[i][k] * T[k][j];
                C[i][j] = acc;
            }
        double R[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[i][j] = C[i][j];
        svd3_project_so3(R);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) C[i][j] = R[i][j];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                G[i][j] = C[i][j];
                g_abs[(int64_t)(n + 1) * 16 + i * 4 + j] = C[i][j];
            }
    }
}

// Wavefront formulation of the chain.  With G_{i-1} a rotation, the SO(3) projection commutes with the product:
//   proj(G_{i-1}.R * R_i) = G_{i-1}.R * proj(R_i)      (SVD of G R = (G U) S V^T; same singular values, same determinant sign)
// so the recurrence  G_i = proj(G_{i-1} T_i)  is the prefix product of the independently projected elements
// E_i = (proj(R_i), t_i) under (Ra, ta) o (Rb, tb) = (Ra Rb, Ra tb + ta).  Phase 1 projects every relative pose in parallel
// (one lane per pose, fp64 one-sided Jacobi); phase 2 is a wave-shuffle prefix product (Kogge-Stone over 64 lanes, wave
// totals combined through LDS, 256-pose chunks carried by the block).  Pose 1 is formed from g0 exactly as the reference does
// (g0 need not be a rotation: ensure_so3_v2 goes through here); reassociation moves results by ~1e-15 (tolerance 1e-9).
struct Se3 {
    double r[9], t[3];
};
__device__ __forceinline__ Se3 se3_compose(const Se3& a, const Se3& b) {
    Se3 c;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) c.r[i * 3 + j] = a.r[i * 3] * b.r[j] + a.r[i * 3 + 1] * b.r[3 + j] + a.r[i * 3 + 2] * b.r[6 + j];
        c.t[i] = a.r[i * 3] * b.t[0] + a.r[i * 3 + 1] * b.t[1] + a.r[i * 3 + 2] * b.t[2] + a.t[i];
    }
    return c;
}
__device__ __forceinline__ Se3 se3_shfl_up(const Se3& a, int d) {
    Se3 o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.r[i] = __shfl_up(a.r[i], d, 64);
#pragma unroll
    for (int i = 0; i < 3; ++i) o.t[i] = __shfl_up(a.t[i], d, 64);
    return o;
}

// phase 1: element n (pose n + 1) -> g_abs[(n + 1) * 16 ..]: rows 0-2 = [proj(R) | t], slot 12 = 1.0 when the input is not affine
__global__ __launch_bounds__(64) void pose_project_kernel(const float* t_rel, int N, const double* g0, double* g_abs) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    double T[4][4];
    for (int i = 0; i < 16; ++i) T[i / 4][i % 4] = (double)t_rel[(int64_t)n * 16 + i];
    bool bad = !(T[3][0] == 0.0 && T[3][1] == 0.0 && T[3][2] == 0.0 && T[3][3] == 1.0);
    double C[3][4];
    if (n == 0) {   // the first step multiplies by g0 in full, as the reference does
        double G[4][4];
        for (int i = 0; i < 16; ++i) G[i / 4][i % 4] = g0[i];
        bad = bad || !(G[3][0] == 0.0 && G[3][1] == 0.0 && G[3][2] == 0.0 && G[3][3] == 1.0);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) C[i][j] = G[i][0] * T[0][j] + G[i][1] * T[1][j] + G[i][2] * T[2][j] + G[i][3] * T[3][j];
    } else {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) C[i][j] = T[i][j];
    }
    double R[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i][j] = C[i][j];
    svd3_project_so3(R);
    double* o = g_abs + (int64_t)(n + 1) * 16;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) o[i * 4 + j] = R[i][j];
        o[i * 4 + 3] = C[i][3];
    }
    o[12] = bad ? 1.0 : 0.0;
}

// phase 2: in-place prefix product over the projected elements (one block; 256 poses per round)
constexpr int CHAIN_THREADS = 256;
__global__ __launch_bounds__(CHAIN_THREADS) void pose_scan_kernel(const float* t_rel, int N, double* g_abs) {
    __shared__ Se3 wave_tot[CHAIN_THREADS / 64];
    __shared__ Se3 carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // any non-affine input: the strict recurrence on one lane (g_abs[0:16] holds g0)
    int bad = 0;
    for (int n = tid; n < N; n += CHAIN_THREADS) bad |= g_abs[(int64_t)(n + 1) * 16 + 12] != 0.0;
    if (__syncthreads_or(bad)) {
        if (tid == 0) pose_chain_sequential(t_rel, N, g_abs, g_abs);
        return;
    }
    for (int base = 0; base < N; base += CHAIN_THREADS) {
        const int n = base + tid;
        Se3 e;
        if (n < N) {
            const double* s = g_abs + (int64_t)(n + 1) * 16;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) e.r[i * 3 + j] = s[i * 4 + j];
                e.t[i] = s[i * 4 + 3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) e.r[i] = (i % 4 == 0) ? 1.0 : 0.0;
            e.t[0] = e.t[1] = e.t[2] = 0.0;
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const Se3 o = se3_shfl_up(e, d);
            if (lane >= d) e = se3_compose(o, e);
        }
        if (lane == 63) wave_tot[wave] = e;
        __syncthreads();
        Se3 pre;
        bool has_pre = false;
        if (base > 0) { pre = carry; has_pre = true; }
        for (int w = 0; w < wave; ++w) {
            pre = has_pre ? se3_compose(pre, wave_tot[w]) : wave_tot[w];
            has_pre = true;
        }
        if (has_pre) e = se3_compose(pre, e);
        __syncthreads();                       // everyone has read carry / wave_tot
        if (tid == CHAIN_THREADS - 1) carry = e;
        if (n < N) {
            double* o = g_abs + (int64_t)(n + 1) * 16;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) o[i * 4 + j] = e.r[i * 3 + j];
                o[i * 4 + 3] = e.t[i];
            }
            o[12] = 0.0; o[13] = 0.0; o[14] = 0.0; o[15] = 1.0;
        }
        __syncthreads();
    }
}

__global__ void pixel_to_3d_kernel(const double* uvd, int64_t n, double fx, double fy, double cx, double cy, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double u = uvd[3 * i], v = uvd[3 * i + 1], d = uvd[3 * i + 2];
    out[3 * i] = (u - cx) * d / fx;
    out[3 * i + 1] = (v - cy) * d / fy;
    out[3 * i + 2] = d;
}

}  // namespace bs

extern "C" int bs_pixel_to_3d(const double* uvd, int64_t n, const double* K, double* out, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("bs_pixel_to_3d: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(uvd && K && out && n >= 0, "bs_pixel_to_3d: bad argument");
    if (n == 0) return BS_OK;
    hipLaunchKernelGGL(pixel_to_3d_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), uvd, n, K[0],
                       K[1], K[2], K[3], out);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_backproject(const uint16_t* depth, int32_t B, int32_t H, int32_t W, const double* K, double depth_scale,
                              double depth_trunc, const double* poses, float* xyz, int32_t* idx, int32_t* count,
                              int32_t* scratch, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("bs_backproject: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(B >= 0 && H >= 0 && W >= 0, "bs_backproject: negative size");
    if (B == 0) return BS_OK;                       // an empty batch has no buffers to check (their pointers may be null)
    BS_REQUIRE(depth && K && xyz && idx && count && scratch, "bs_backproject: null argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t npix = (int64_t)H * W;
    if (npix == 0) {
        BS_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int32_t) * B, st));
        return BS_OK;
    }
    BS_REQUIRE(npix < (1ll << 31), "bs_backproject: image too large");
    const int nchunk = (int)cdiv64(npix, BP_CHUNK);
    const float s32 = (float)depth_scale, t32 = (float)depth_trunc;
    dim3 grid(nchunk, B);
    hipLaunchKernelGGL(bp_count_kernel, grid, dim3(BP_THREADS), 0, st, depth, npix, nchunk, s32, t32, scratch);
    BS_CHECK_LAUNCH();
    hipLaunchKernelGGL(bp_scan_kernel, dim3(B), dim3(256), 0, st, scratch, nchunk, count);
    BS_CHECK_LAUNCH();
    hipLaunchKernelGGL(bp_write_kernel, grid, dim3(BP_THREADS), 0, st, depth, H, W, nchunk, s32, t32, K[0], K[1], K[2], K[3],
                       poses, (const int32_t*)scratch, xyz, idx);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

static int pose_chain_launch(const float* t_rel, int32_t N, double* g_abs, hipStream_t st) {
    using namespace bs;
    if (N > 0) {   // g_abs[0:16] already holds g0
        hipLaunchKernelGGL(pose_project_kernel, dim3(cdiv(N, 64)), dim3(64), 0, st, t_rel, N, (const double*)g_abs, g_abs);
        BS_CHECK_LAUNCH();
        hipLaunchKernelGGL(pose_scan_kernel, dim3(1), dim3(CHAIN_THREADS), 0, st, t_rel, N, g_abs);
        BS_CHECK_LAUNCH();
    }
    return BS_OK;
}

extern "C" int bs_pose_chain(const float* t_rel, int32_t N, const double* g0_host, double* g_abs, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("bs_pose_chain: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(g_abs && (t_rel || N == 0) && N >= 0, "bs_pose_chain: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static const double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    // g_abs[0:16] is the device slot of g0 (pose 0 of the output)
    BS_CHECK_HIP(hipMemcpyAsync(g_abs, g0_host ? g0_host : eye, 16 * sizeof(double), hipMemcpyHostToDevice, st));
    return pose_chain_launch(t_rel, N, g_abs, st);
}

extern "C" int bs_pose_chain_from(const float* t_rel, int32_t N, const double* g0_dev, double* g_abs, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("bs_pose_chain_from: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(g_abs && g0_dev && (t_rel || N == 0) && N >= 0, "bs_pose_chain_from: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (g0_dev != g_abs) BS_CHECK_HIP(hipMemcpyAsync(g_abs, g0_dev, 16 * sizeof(double), hipMemcpyDeviceToDevice, st));
    return pose_chain_launch(t_rel, N, g_abs, st);
}
