// 3DM TSDF map: voxel integration and point-cloud extraction of Open3D's ScalableTSDFVolume, the
// class BodySLAM_not_refactored/3DM/tsdf.py:5-52 wraps (voxel_length 1 mm, sdf_trunc 0.1 m, RGB8, volume_unit_resolution 32,
// depth_sampling_stride 8; called per frame at 3DM/slam.py:117,179).  Open3D is an un-vendored C++ dependency: the algorithm is
// restated from its published form (oracle/tsdf_ref.py carries the same restatement; parity unpinned).
//
// HBM-bound streaming work, laid out for it: a volume unit is one contiguous block of res^3 voxels x 5 floats
// (tsdf, weight, r, g, b), voxel index x*res^2 + y*res + z as Open3D's IndexOf; a thread owns one voxel with z fastest, so a wave
// reads and writes 64 x 20 = 1280 contiguous bytes; the depth / colour images (1.2 MB) stay in L2.  The unit table (which units
// exist, where their blocks live) is host state, as in Open3D (an unordered_map); the kernels get per-call pointer lists.
#include "common.h"

namespace bs {

struct TsdfCam {
    double fx, fy, cx, cy;
    double e[12];            // extrinsic, rows 0..2 of the 4x4 (world -> camera)
};

// Open3D UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier, one thread per voxel
__global__ __launch_bounds__(256) void tsdf_integrate_kernel(const float* __restrict__ depth, const uint8_t* __restrict__ color, int H, int W,
                                                              TsdfCam cam, const int32_t* __restrict__ unit_index, float* const* __restrict__ unit_ptr,
                                                              int res, double voxel_length, double sdf_trunc) {
    const int u = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int nvox = res * res * res;
    if (v >= nvox) return;
    const int x = v / (res * res), y = (v / res) % res, z = v % res;
    const double unit_len = voxel_length * res, half = voxel_length * 0.5;
    const double px = half + voxel_length * x + unit_len * unit_index[3 * u + 0];
    const double py = half + voxel_length * y + unit_len * unit_index[3 * u + 1];
    const double pz = half + voxel_length * z + unit_len * unit_index[3 * u + 2];
    const double cx_ = cam.e[0] * px + cam.e[1] * py + cam.e[2] * pz + cam.e[3];
    const double cy_ = cam.e[4] * px + cam.e[5] * py + cam.e[6] * pz + cam.e[7];
    const double cz_ = cam.e[8] * px + cam.e[9] * py + cam.e[10] * pz + cam.e[11];
    if (!(cz_ > 0.0)) return;
    const double u_f = cx_ * cam.fx / cz_ + cam.cx + 0.5, v_f = cy_ * cam.fy / cz_ + cam.cy + 0.5;
    if (!(u_f >= 0.0001 && u_f < (double)W - 0.0001 && v_f >= 0.0001 && v_f < (double)H - 0.0001)) return;
    const int ui = (int)u_f, vi = (int)v_f;
    const float d = depth[(int64_t)vi * W + ui];
    if (!(d > 0.0f)) return;
    // depth-to-camera-distance multiplier of the pixel (Open3D keeps it as a float image)
    const float xx = (float)(((double)ui - cam.cx) / cam.fx), yy = (float)(((double)vi - cam.cy) / cam.fy);
    const float mult = sqrtf(xx * xx + yy * yy + 1.0f);
    const float sdf = (float)(((double)d - cz_) * (double)mult);
    if (!(sdf > -(float)sdf_trunc)) return;
    const float tsdf = fminf(1.0f, sdf * (float)(1.0 / sdf_trunc));
    float* vox = unit_ptr[u] + (int64_t)v * 5;
    const float w0 = vox[1], w1 = w0 + 1.0f;
    vox[0] = (vox[0] * w0 + tsdf) / w1;
    if (color) {
        const uint8_t* c = color + ((int64_t)vi * W + ui) * 3;
        vox[2] = (vox[2] * w0 + (float)c[0]) / w1;
        vox[3] = (vox[3] * w0 + (float)c[1]) / w1;
        vox[4] = (vox[4] * w0 + (float)c[2]) / w1;
    }
    vox[1] = w1;
}

// Open3D ScalableTSDFVolume::ExtractPointCloud without the normals: a voxel with weight != 0 and |tsdf| < 0.98 looks at its +x, +y,
// +z neighbour (possibly in the neighbouring unit); a sign change puts a point at the linear zero crossing, colour interpolated
// alike.  WRITE = false counts per unit, WRITE = true writes at unit_offset[u] + a per-unit cursor (order inside a unit is arbitrary,
// as the order of units is in Open3D's hash map).
template <bool WRITE>
__global__ __launch_bounds__(256) void tsdf_extract_kernel(const int32_t* __restrict__ unit_index, float* const* __restrict__ unit_ptr,
                                                            float* const* __restrict__ nbr_ptr, int res, double voxel_length,
                                                            int32_t* __restrict__ unit_count, const int64_t* __restrict__ unit_offset,
                                                            float* __restrict__ points, float* __restrict__ colors) {
    const int u = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int nvox = res * res * res;
    int found = 0;
    float pts[3][3], cols[3][3];
    if (v < nvox) {
        const float* base = unit_ptr[u];
        const float f0 = base[(int64_t)v * 5], w0 = base[(int64_t)v * 5 + 1];
        if (w0 != 0.0f && f0 < 0.98f && f0 >= -0.98f) {
            const int idx[3] = {v / (res * res), (v / res) % res, v % res};
            const double unit_len = voxel_length * res, half = voxel_length * 0.5;
            double p0[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) p0[a] = half + voxel_length * idx[a] + unit_len * unit_index[3 * u + a];
            const int stride[3] = {res * res, res, 1};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float* nb = base;
                int64_t nv = (int64_t)v + stride[a];
                if (idx[a] + 1 >= res) {             // the neighbour lives in the next unit along this axis
                    nb = nbr_ptr[3 * u + a];
                    nv = (int64_t)v - (int64_t)(res - 1) * stride[a];
                }
                if (!nb) continue;
                const float f1 = nb[nv * 5], w1 = nb[nv * 5 + 1];
                if (w1 != 0.0f && f1 < 0.98f && f1 >= -0.98f && f0 * f1 < 0.0f) {
                    const float r0 = fabsf(f0), r1 = fabsf(f1);
                    if (WRITE) {
                        double p[3] = {p0[0], p0[1], p0[2]};
                        p[a] = (p0[a] * (double)r1 + (p0[a] + voxel_length) * (double)r0) / ((double)r0 + (double)r1);
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            pts[found][k] = (float)p[k];
                            const float c0 = base[(int64_t)v * 5 + 2 + k], c1 = nb[nv * 5 + 2 + k];
                            cols[found][k] = (c0 * r1 + c1 * r0) / (r0 + r1) / 255.0f;
                        }
                    }
                    ++found;
                }
            }
        }
    }
    if (!WRITE) {
        int s = found;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(unit_count + u, s);
    } else if (found) {
        const int64_t at = unit_offset[u] + atomicAdd(unit_count + u, found);
        for (int k = 0; k < found; ++k) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                points[(at + k) * 3 + c] = pts[k][c];
                colors[(at + k) * 3 + c] = cols[k][c];
            }
        }
    }
}

}  // namespace bs

using namespace bs;

extern "C" int bs_tsdf_integrate(const float* depth, const uint8_t* color, int32_t H, int32_t W, const double* K, const double* extrinsic,
                                 const int32_t* unit_index, const void* unit_ptr, int32_t units, int32_t res, double voxel_length,
                                 double sdf_trunc, void* stream) {
    if (!initialized()) { set_error("bs_tsdf_integrate: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(units >= 0 && H > 0 && W > 0 && res > 0 && res <= 64 && voxel_length > 0.0 && sdf_trunc > 0.0, "bs_tsdf_integrate: bad geometry");
    if (units == 0) return BS_OK;
    BS_REQUIRE(depth && K && extrinsic && unit_index && unit_ptr, "bs_tsdf_integrate: null argument");
    TsdfCam cam;
    cam.fx = K[0]; cam.fy = K[1]; cam.cx = K[2]; cam.cy = K[3];
    for (int i = 0; i < 12; ++i) cam.e[i] = extrinsic[i];
    const int nvox = res * res * res;
    hipLaunchKernelGGL(tsdf_integrate_kernel, dim3(cdiv(nvox, 256), units), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), depth, color, H, W,
                       cam, unit_index, reinterpret_cast<float* const*>(unit_ptr), res, voxel_length, sdf_trunc);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_tsdf_extract(const int32_t* unit_index, const void* unit_ptr, const void* nbr_ptr, int32_t units, int32_t res,
                               double voxel_length, int32_t* unit_count, const int64_t* unit_offset, float* points, float* colors,
                               void* stream) {
    if (!initialized()) { set_error("bs_tsdf_extract: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(units >= 0 && res > 0 && res <= 64 && voxel_length > 0.0, "bs_tsdf_extract: bad geometry");
    if (units == 0) return BS_OK;
    BS_REQUIRE(unit_index && unit_ptr && nbr_ptr && unit_count, "bs_tsdf_extract: null argument");
    BS_REQUIRE((points == nullptr) == (unit_offset == nullptr) && (points == nullptr) == (colors == nullptr),
               "bs_tsdf_extract: the write pass needs unit_offset, points and colors; the count pass none of them");
    const int nvox = res * res * res;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    BS_CHECK_HIP(hipMemsetAsync(unit_count, 0, sizeof(int32_t) * units, st));
    const dim3 grid(cdiv(nvox, 256), units);
    if (!points)
        hipLaunchKernelGGL(tsdf_extract_kernel<false>, grid, dim3(256), 0, st, unit_index, reinterpret_cast<float* const*>(unit_ptr),
                           reinterpret_cast<float* const*>(nbr_ptr), res, voxel_length, unit_count, unit_offset, points, colors);
    else
        hipLaunchKernelGGL(tsdf_extract_kernel<true>, grid, dim3(256), 0, st, unit_index, reinterpret_cast<float* const*>(unit_ptr),
                           reinterpret_cast<float* const*>(nbr_ptr), res, voxel_length, unit_count, unit_offset, points, colors);
    BS_CHECK_LAUNCH();
    return BS_OK;
}
