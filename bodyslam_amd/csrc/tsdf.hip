// 3DM TSDF map: voxel integration and point-cloud extraction of Open3D's ScalableTSDFVolume, the
// class BodySLAM_not_refactored/3DM/tsdf.py:5-52 wraps (voxel_length 1 mm, sdf_trunc 0.1 m, RGB8, volume_unit_resolution 32,
// depth_sampling_stride 8; called per frame at 3DM/slam.py:117,179).  Open3D is an un-vendored C++ dependency: the algorithm is
// restated from its published form (oracle/tsdf_ref.py carries the same restatement; parity unpinned).
//
// HBM-bound streaming work, laid out for it: a volume unit is one contiguous block of res^3 voxels x 5 floats
// (tsdf, weight, r, g, b), voxel index x*res^2 + y*res + z as Open3D's IndexOf; a thread owns one voxel with z fastest, so a wave
// reads and writes 64 x 20 = 1280 contiguous bytes; the depth / colour images (1.2 MB) stay in L2.  The unit table (which units
// exist, which block each owns) is an open-addressing hash table in HBM filled by atomicCAS (Open3D: an unordered_map on the host);
// blocks are carved from zero-filled slabs whose base addresses the kernels get as a small device array.
#include <stdlib.h>

#include <mutex>

#include "common.h"

namespace bs {

struct TsdfCam {
    double fx, fy, cx, cy;
    double e[12];            // extrinsic, rows 0..2 of the 4x4 (world -> camera)
    double ifx, ify;         // 1 / fx, 1 / fy (host-computed: the per-voxel code multiplies instead of dividing)
};

// ---- the table of volume units: an open-addressing hash table in HBM (Open3D: std::unordered_map<Vector3i, VolumeUnit>) ----------
// key = three 21-bit biased unit indices; keys[h] == -1 is empty; slots[h] is the unit's block number (assigned in the second pass),
// stamp[h] the id of the last frame that touched it.
constexpr long long TS_EMPTY = -1ll;
constexpr int TS_OFF = 1 << 20;

__device__ __forceinline__ long long ts_pack(int ix, int iy, int iz) {
    return ((long long)(ix + TS_OFF) << 42) | ((long long)(iy + TS_OFF) << 21) | (long long)(iz + TS_OFF);
}
__device__ __forceinline__ unsigned ts_hash(long long k, unsigned mask) {
    unsigned long long x = (unsigned long long)k;
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (unsigned)x & mask;
}
__device__ __forceinline__ int ts_find(const long long* __restrict__ keys, const int32_t* __restrict__ slots, unsigned mask, long long k) {
    for (unsigned h = ts_hash(k, mask), n = 0; n <= mask; h = (h + 1) & mask, ++n) {
        const long long cur = keys[h];
        if (cur == k) return slots[h];
        if (cur == TS_EMPTY) return -1;
    }
    return -1;
}
__device__ __forceinline__ float* ts_block(const int64_t* __restrict__ slab_base, int slab_units, int64_t unit_bytes, int slot) {
    return reinterpret_cast<float*>(slab_base[slot / slab_units] + (int64_t)(slot % slab_units) * unit_bytes);
}

// ScalableTSDFVolume::Integrate, first half: every unit that meets the +-sdf_trunc box of a point of the strided depth sample is
// looked up / inserted and stamped with this frame.  Thread = (sampled pixel, candidate offset inside the (span)^3 box).
__global__ __launch_bounds__(256) void tsdf_touch_kernel(const float* __restrict__ depth, int H, int W, int stride, TsdfCam cam /* e = camera -> world */,
                                                          double unit_length, double sdf_trunc, int span, long long* keys, int32_t* stamp,
                                                          unsigned mask, int frame_id, int32_t* counters) {
    const int cand = span * span * span;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int ws = (W + stride - 1) / stride, hs = (H + stride - 1) / stride;
    if (t >= (int64_t)ws * hs * cand) return;
    const int c = (int)(t % cand), pix = (int)(t / cand);
    const int i = (pix / ws) * stride, j = (pix % ws) * stride;
    const double z = (double)depth[(int64_t)i * W + j];
    if (!(z > 0.0)) return;
    const double x = ((double)j - cam.cx) * z / cam.fx, y = ((double)i - cam.cy) * z / cam.fy;
    const double p[3] = {cam.e[0] * x + cam.e[1] * y + cam.e[2] * z + cam.e[3], cam.e[4] * x + cam.e[5] * y + cam.e[6] * z + cam.e[7],
                         cam.e[8] * x + cam.e[9] * y + cam.e[10] * z + cam.e[11]};
    const int o[3] = {c / (span * span), (c / span) % span, c % span};
    int u[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int lo = (int)floor((p[a] - sdf_trunc) / unit_length), hi = (int)floor((p[a] + sdf_trunc) / unit_length);
        u[a] = lo + o[a];
        if (u[a] > hi) return;
    }
    const long long k = ts_pack(u[0], u[1], u[2]);
    for (unsigned h = ts_hash(k, mask), n = 0; n <= mask; h = (h + 1) & mask, ++n) {
        long long cur = keys[h];
        if (cur == TS_EMPTY) cur = (long long)atomicCAS(reinterpret_cast<unsigned long long*>(keys + h), (unsigned long long)TS_EMPTY, (unsigned long long)k);
        if (cur == TS_EMPTY || cur == k) {
            stamp[h] = frame_id;
            return;
        }
    }
    counters[2] = 1;      // table full
}

// second half: entries stamped by this frame get a block number if they have none, and go on the frame's list -- unless the unit lies
// wholly outside the view frustum.  Units are opened in a +-sdf_trunc box around every sampled point, far wider than the frustum at
// endoscopic range (round 2, PMC: 90 % of the integrate kernel's threads projected outside the image and left): a unit whose
// bounding sphere is behind the camera or projects wholly outside the image holds no voxel the integration would update, so it
// is opened (as in Open3D) but not put on the list.  `view` = world -> camera, W x H the image; conservative bound, see below.
__global__ __launch_bounds__(256) void tsdf_assign_kernel(const long long* __restrict__ keys, int32_t* slots, const int32_t* __restrict__ stamp,
                                                           unsigned cap, int frame_id, int32_t* unit_index, int max_units, int32_t* counters,
                                                           int32_t* touched, TsdfCam view, double unit_length, int W, int H, int cull) {
    const unsigned h = blockIdx.x * 256 + threadIdx.x;
    if (h >= cap || keys[h] == TS_EMPTY || stamp[h] != frame_id) return;
    int s = slots[h];
    if (s < 0) {
        s = atomicAdd(counters + 0, 1);
        if (s >= max_units) {
            // more units than blocks (max_units = the blocks that EXIST: the caller keeps slabs allocated ahead of the map).  The
            // count is put back, so the table entry stays without a block and a later frame -- after the caller has added slabs or
            // raised the error -- can still assign one.
            atomicSub(counters + 0, 1);
            counters[2] = 2;
            return;
        }
        slots[h] = s;
        const long long k = keys[h];
        unit_index[3 * s + 0] = (int)(k >> 42) - TS_OFF;
        unit_index[3 * s + 1] = (int)((k >> 21) & ((1 << 21) - 1)) - TS_OFF;
        unit_index[3 * s + 2] = (int)(k & ((1 << 21) - 1)) - TS_OFF;
    }
    if (cull) {
        const long long k = keys[h];
        const double c[3] = {((double)((int)(k >> 42) - TS_OFF) + 0.5) * unit_length, ((double)((int)((k >> 21) & ((1 << 21) - 1)) - TS_OFF) + 0.5) * unit_length,
                             ((double)((int)(k & ((1 << 21) - 1)) - TS_OFF) + 0.5) * unit_length};
        const double r = 0.8660254037844387 * unit_length;           // half the cube's diagonal
        const double qx = view.e[0] * c[0] + view.e[1] * c[1] + view.e[2] * c[2] + view.e[3];
        const double qy = view.e[4] * c[0] + view.e[5] * c[1] + view.e[6] * c[2] + view.e[7];
        const double qz = view.e[8] * c[0] + view.e[9] * c[1] + view.e[10] * c[2] + view.e[11];
        if (qz + r <= 0.0) return;                                   // wholly behind the camera
        if (qz > r) {                                                // (a sphere that reaches the camera plane is never culled)
            // a point p = q + d, |d| <= r, projects within  f r / (qz - r) * (1 + |qx| / qz)  pixels of q's projection:
            // |px / pz - qx / qz| = |dx qz - qx dz| / (pz qz) <= r (qz + |qx|) / ((qz - r) qz)
            const double inv = 1.0 / (qz - r);
            const double ru = view.fx * r * inv * (1.0 + fabs(qx) / qz) + 1.0, rv = view.fy * r * inv * (1.0 + fabs(qy) / qz) + 1.0;
            const double uc = qx * view.fx / qz + view.cx + 0.5, vc = qy * view.fy / qz + view.cy + 0.5;
            if (uc + ru < 0.0 || uc - ru > (double)W || vc + rv < 0.0 || vc - rv > (double)H) return;
        }
    }
    touched[atomicAdd(counters + 1, 1)] = s;
}

// Open3D UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier, one thread per voxel.  Grid: blocks_per_unit x an UPPER
// BOUND of the touched units, folded into blockIdx.x (no 65 535 limit); the number of units this frame really touched is read
// from device memory (counters[1], written by tsdf_assign_kernel), so the host never has to wait for it.
// (Units wholly outside the view frustum never reach `touched`: tsdf_assign_kernel.  A test per 256-voxel block was tried first and
// cost what it saved: the block-uniform fp64 test is as many vector instructions as the per-voxel projection it replaces.)
// what one frame says about one voxel (world position p): false = the voxel is not updated; otherwise the truncated sdf and the
// pixel it was read at.  Shared by the frame-at-a-time and the frame-batched kernels: the same instructions, the same bits.
__device__ __forceinline__ bool tsdf_observe(double px, double py, double pz, const TsdfCam& cam, const float* __restrict__ depth, int H, int W,
                                             double sdf_trunc, float& tsdf, int64_t& pix) {
    // (fused multiply-adds and ONE division: this function is all the frame-batched kernel does per voxel and frame -- 15 ms per 64
    // frames were fp64 multiplies, adds and two divisions issued one by one, the build has -ffp-contract=off)
    const double cz_ = fma(cam.e[8], px, fma(cam.e[9], py, fma(cam.e[10], pz, cam.e[11])));
    if (!(cz_ > 0.0)) return false;
    const double cx_ = fma(cam.e[0], px, fma(cam.e[1], py, fma(cam.e[2], pz, cam.e[3])));
    const double cy_ = fma(cam.e[4], px, fma(cam.e[5], py, fma(cam.e[6], pz, cam.e[7])));
    // 1 / cz: v_rcp_f64 and two Newton steps (five instructions, <= 1 ulp) instead of the fifteen-instruction IEEE division
    double iz = __builtin_amdgcn_rcp(cz_);
    iz = fma(iz, fma(-cz_, iz, 1.0), iz);
    iz = fma(iz, fma(-cz_, iz, 1.0), iz);
    const double u_f = fma(cx_ * cam.fx, iz, cam.cx + 0.5), v_f = fma(cy_ * cam.fy, iz, cam.cy + 0.5);
    if (!(u_f >= 0.0001 && u_f < (double)W - 0.0001 && v_f >= 0.0001 && v_f < (double)H - 0.0001)) return false;
    const int ui = (int)u_f, vi = (int)v_f;
    pix = (int64_t)vi * W + ui;
    const float d = depth[pix];
    if (!(d > 0.0f)) return false;
    // depth-to-camera-distance multiplier of the pixel (Open3D keeps it as a float image)
    const float xx = (float)(((double)ui - cam.cx) * cam.ifx), yy = (float)(((double)vi - cam.cy) * cam.ify);
    const float mult = sqrtf(xx * xx + yy * yy + 1.0f);
    const float sdf = (float)(((double)d - cz_) * (double)mult);
    if (!(sdf > -(float)sdf_trunc)) return false;
    tsdf = fminf(1.0f, sdf * (float)(1.0 / sdf_trunc));
    return true;
}
// the running weighted mean of a voxel (tsdf, weight, r, g, b) takes one observation
__device__ __forceinline__ void tsdf_blend(float vox[5], float tsdf, const uint8_t* __restrict__ c) {
    // (one division, by the weight -- a small integer -- and four multiplies: the four IEEE divisions were a third of the per-voxel work)
    const float w0 = vox[1], w1 = w0 + 1.0f, rw = 1.0f / w1;
    vox[0] = fmaf(vox[0], w0, tsdf) * rw;
    if (c) {
        vox[2] = fmaf(vox[2], w0, (float)c[0]) * rw;
        vox[3] = fmaf(vox[3], w0, (float)c[1]) * rw;
        vox[4] = fmaf(vox[4], w0, (float)c[2]) * rw;
    }
    vox[1] = w1;
}

__device__ __forceinline__ void tsdf_integrate_block(int work, const float* __restrict__ depth, const uint8_t* __restrict__ color, int H, int W,
                                                     const TsdfCam& cam, const int32_t* __restrict__ unit_index, const int32_t* __restrict__ touched,
                                                     int blocks_per_unit, const int64_t* __restrict__ slab_base, int slab_units, int res,
                                                     double voxel_length, double sdf_trunc, int cull) {
    const int ti = work / blocks_per_unit, bx = work - ti * blocks_per_unit;
    const int u = touched[ti];
    const int v = bx * 256 + threadIdx.x;
    const int nvox = res * res * res;
    const double unit_len = voxel_length * res, half = voxel_length * 0.5;
    if (v >= nvox) return;
    const int x = v / (res * res), y = (v / res) % res, z = v % res;
    const double px = half + voxel_length * x + unit_len * unit_index[3 * u + 0];
    const double py = half + voxel_length * y + unit_len * unit_index[3 * u + 1];
    const double pz = half + voxel_length * z + unit_len * unit_index[3 * u + 2];
    float tsdf;
    int64_t pix;
    if (!tsdf_observe(px, py, pz, cam, depth, H, W, sdf_trunc, tsdf, pix)) return;
    float* vox = ts_block(slab_base, slab_units, (int64_t)nvox * 20, u) + (int64_t)v * 5;
    float s[5] = {vox[0], vox[1], 0.0f, 0.0f, 0.0f};
    if (color) {
        s[2] = vox[2]; s[3] = vox[3]; s[4] = vox[4];
    }
    tsdf_blend(s, tsdf, color ? color + pix * 3 : nullptr);
    vox[0] = s[0];
    vox[1] = s[1];
    if (color) {
        vox[2] = s[2]; vox[3] = s[3]; vox[4] = s[4];
    }
}

// a fixed grid walks the (touched unit, 256-voxel block) work items; their number comes from device memory
__global__ __launch_bounds__(256) void tsdf_integrate_kernel(const float* __restrict__ depth, const uint8_t* __restrict__ color, int H, int W,
                                                              TsdfCam cam, const int32_t* __restrict__ unit_index, const int32_t* __restrict__ touched,
                                                              const int32_t* __restrict__ n_touched_dev, int blocks_per_unit,
                                                              const int64_t* __restrict__ slab_base, int slab_units, int res, double voxel_length,
                                                              double sdf_trunc, int cull) {
    const long long total = (long long)(*n_touched_dev) * blocks_per_unit;
    for (long long w = blockIdx.x; w < total; w += gridDim.x)
        tsdf_integrate_block((int)w, depth, color, H, W, cam, unit_index, touched, blocks_per_unit, slab_base, slab_units, res, voxel_length, sdf_trunc, cull);
}

// ---- a batch of frames at once ----------------------------------------------------------------------------------------------------
// A voxel's running mean takes the frames in order, but nothing else orders them: a batch of up to 64 frames is integrated by loading
// every touched voxel ONCE, applying the frames that touch its unit in ascending order in registers, and storing it once -- the
// same values bit for bit as 64 frame-at-a-time passes (each of which reads and writes the voxel through HBM), in three launches
// instead of 192.  Which frames touch a unit is a 64-bit mask: the discovery kernel ORs the frame's bit into the unit's table entry,
// the assignment kernel hands out blocks, applies the per-frame frustum test and leaves the surviving bits in unit_mask[block].
struct TsdfFrame {            // one record per frame in device memory (bs_tsdf_frames_upload): BS_TSDF_FRAME_BYTES
    const float* depth;
    const uint8_t* color;
    double fx, fy, cx, cy;
    double pose[12];          // camera -> world, rows 0..2 (unit discovery back-projects with it)
    double view[12];          // world -> camera, rows 0..2: the `extrinsic` of ScalableTSDFVolume::Integrate
    double ifx, ify;          // 1 / fx, 1 / fy
};
static_assert(sizeof(TsdfFrame) == BS_TSDF_FRAME_BYTES, "TsdfFrame layout is part of the C ABI");

__global__ __launch_bounds__(256) void tsdf_touch_batch_kernel(const TsdfFrame* __restrict__ frames, int H, int W, int stride, double unit_length,
                                                                double sdf_trunc, int span, long long* keys, unsigned long long* fmask, unsigned mask,
                                                                int32_t* counters) {
    const TsdfFrame& F = frames[blockIdx.y];
    const int cand = span * span * span;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int ws = (W + stride - 1) / stride, hs = (H + stride - 1) / stride;
    if (t >= (int64_t)ws * hs * cand) return;
    const int c = (int)(t % cand), pix = (int)(t / cand);
    const int i = (pix / ws) * stride, j = (pix % ws) * stride;
    const double z = (double)F.depth[(int64_t)i * W + j];
    if (!(z > 0.0)) return;
    const double x = ((double)j - F.cx) * z / F.fx, y = ((double)i - F.cy) * z / F.fy;
    const double p[3] = {F.pose[0] * x + F.pose[1] * y + F.pose[2] * z + F.pose[3], F.pose[4] * x + F.pose[5] * y + F.pose[6] * z + F.pose[7],
                         F.pose[8] * x + F.pose[9] * y + F.pose[10] * z + F.pose[11]};
    const int o[3] = {c / (span * span), (c / span) % span, c % span};
    int u[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int lo = (int)floor((p[a] - sdf_trunc) / unit_length), hi = (int)floor((p[a] + sdf_trunc) / unit_length);
        u[a] = lo + o[a];
        if (u[a] > hi) return;
    }
    const long long k = ts_pack(u[0], u[1], u[2]);
    const unsigned long long bit = 1ull << blockIdx.y;
    for (unsigned h = ts_hash(k, mask), n = 0; n <= mask; h = (h + 1) & mask, ++n) {
        long long cur = keys[h];
        if (cur == TS_EMPTY) cur = (long long)atomicCAS(reinterpret_cast<unsigned long long*>(keys + h), (unsigned long long)TS_EMPTY, (unsigned long long)k);
        if (cur == TS_EMPTY || cur == k) {
            if (!(__atomic_load_n(fmask + h, __ATOMIC_RELAXED) & bit)) atomicOr(fmask + h, bit);     // (thousands of candidates name the same unit)
            return;
        }
    }
    counters[2] = 1;      // table full
}

__global__ __launch_bounds__(256) void tsdf_assign_batch_kernel(const TsdfFrame* __restrict__ frames, const long long* __restrict__ keys, int32_t* slots,
                                                                 unsigned long long* fmask, unsigned cap, int32_t* unit_index, int max_units,
                                                                 int32_t* counters, int32_t* touched, unsigned long long* unit_mask, double unit_length,
                                                                 int W, int H, int cull) {
    const unsigned h = blockIdx.x * 256 + threadIdx.x;
    if (h >= cap || keys[h] == TS_EMPTY) return;
    const unsigned long long m = fmask[h];
    if (m == 0ull) return;
    const long long k = keys[h];
    const int ix = (int)(k >> 42) - TS_OFF, iy = (int)((k >> 21) & ((1 << 21) - 1)) - TS_OFF, iz = (int)(k & ((1 << 21) - 1)) - TS_OFF;
    int s = slots[h];
    if (s < 0) {
        s = atomicAdd(counters + 0, 1);
        if (s >= max_units) {          // (the caller reserves what the discovery found before this kernel runs: only a full map gets here)
            atomicSub(counters + 0, 1);
            counters[2] = 2;
            fmask[h] = 0ull;           // the table's frame masks are zero between batches: a stale bit would name another frame next time
            return;
        }
        slots[h] = s;
        unit_index[3 * s + 0] = ix;
        unit_index[3 * s + 1] = iy;
        unit_index[3 * s + 2] = iz;
    }
    fmask[h] = 0ull;
    unsigned long long keep = m;
    if (cull) {
        const double c[3] = {((double)ix + 0.5) * unit_length, ((double)iy + 0.5) * unit_length, ((double)iz + 0.5) * unit_length};
        const double r = 0.8660254037844387 * unit_length;
        for (unsigned long long rest = m; rest; rest &= rest - 1) {          // the per-frame frustum test of tsdf_assign_kernel
            const int f = __builtin_ctzll(rest);
            const TsdfFrame& F = frames[f];
            const double qx = F.view[0] * c[0] + F.view[1] * c[1] + F.view[2] * c[2] + F.view[3];
            const double qy = F.view[4] * c[0] + F.view[5] * c[1] + F.view[6] * c[2] + F.view[7];
            const double qz = F.view[8] * c[0] + F.view[9] * c[1] + F.view[10] * c[2] + F.view[11];
            bool out = qz + r <= 0.0;
            if (!out && qz > r) {
                const double inv = 1.0 / (qz - r);
                const double ru = F.fx * r * inv * (1.0 + fabs(qx) / qz) + 1.0, rv = F.fy * r * inv * (1.0 + fabs(qy) / qz) + 1.0;
                const double uc = qx * F.fx / qz + F.cx + 0.5, vc = qy * F.fy / qz + F.cy + 0.5;
                out = uc + ru < 0.0 || uc - ru > (double)W || vc + rv < 0.0 || vc - rv > (double)H;
            }
            if (out) keep &= ~(1ull << f);
        }
    }
    if (keep == 0ull) return;
    unit_mask[s] = keep;
    touched[atomicAdd(counters + 1, 1)] = s;
}

__global__ __launch_bounds__(256) void tsdf_integrate_batch_kernel(const TsdfFrame* __restrict__ frames, int H, int W, const int32_t* __restrict__ unit_index,
                                                                    const int32_t* __restrict__ touched, const int32_t* __restrict__ n_touched_dev,
                                                                    const unsigned long long* __restrict__ unit_mask, int blocks_per_unit,
                                                                    const int64_t* __restrict__ slab_base, int slab_units, int res, double voxel_length,
                                                                    double sdf_trunc) {
    const long long total = (long long)(*n_touched_dev) * blocks_per_unit;
    const int nvox = res * res * res;
    const double unit_len = voxel_length * res, half = voxel_length * 0.5;
    for (long long w = blockIdx.x; w < total; w += gridDim.x) {
        const int ti = (int)(w / blocks_per_unit), bx = (int)(w - (long long)ti * blocks_per_unit);
        const int u = touched[ti];
        const int v = bx * 256 + threadIdx.x;
        if (v >= nvox) continue;
        const unsigned long long m = unit_mask[u];
        // the frame loop is uniform over the block: the mask goes to scalar registers, a frame's record is read with scalar loads
        unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)m), hi = __builtin_amdgcn_readfirstlane((unsigned)(m >> 32));
        const int x = v / (res * res), y = (v / res) % res, z = v % res;
        const double px = half + voxel_length * x + unit_len * unit_index[3 * u + 0];
        const double py = half + voxel_length * y + unit_len * unit_index[3 * u + 1];
        const double pz = half + voxel_length * z + unit_len * unit_index[3 * u + 2];
        float* vox = ts_block(slab_base, slab_units, (int64_t)nvox * 20, u) + (int64_t)v * 5;
        float s[5];
        bool have = false;
        for (int half_ = 0; half_ < 2; ++half_) {
            unsigned bits = half_ ? hi : lo;
            while (bits) {
                const int f = __builtin_ctz(bits) + 32 * half_;
                bits &= bits - 1;
                const TsdfFrame& F = frames[f];
                TsdfCam cam;
                cam.fx = F.fx; cam.fy = F.fy; cam.cx = F.cx; cam.cy = F.cy; cam.ifx = F.ifx; cam.ify = F.ify;
#pragma unroll
                for (int i = 0; i < 12; ++i) cam.e[i] = F.view[i];
                float tsdf;
                int64_t pix;
                if (!tsdf_observe(px, py, pz, cam, F.depth, H, W, sdf_trunc, tsdf, pix)) continue;
                if (!have) {
                    s[0] = vox[0]; s[1] = vox[1];
                    if (F.color) {
                        s[2] = vox[2]; s[3] = vox[3]; s[4] = vox[4];
                    }
                    have = true;
                }
                tsdf_blend(s, tsdf, F.color ? F.color + pix * 3 : nullptr);
            }
        }
        if (have) {
            vox[0] = s[0];
            vox[1] = s[1];
            if (frames[0].color) {
                vox[2] = s[2]; vox[3] = s[3]; vox[4] = s[4];
            }
        }
    }
}

// ScalableTSDFVolume::GetTSDFAt: trilinear interpolation of the tsdf over the 8 voxel centres around p (a corner in a unit that does
// not exist contributes 0; the first corner's unit missing gives 0 altogether)
__device__ double ts_tsdf_at(const double p[3], const long long* __restrict__ keys, const int32_t* __restrict__ slots, unsigned mask,
                             const int64_t* __restrict__ slab_base, int slab_units, int res, double voxel_length) {
    const double unit_len = voxel_length * res;
    const int64_t unit_bytes = (int64_t)res * res * res * 20;
    int index0[3], idx0[3];
    double r[3];
    for (int a = 0; a < 3; ++a) {
        const double pl = p[a] - 0.5 * voxel_length;
        index0[a] = (int)floor(pl / unit_len);
        const double pg = (pl - (double)index0[a] * unit_len) / voxel_length;
        int i0 = (int)floor(pg);
        i0 = i0 < 0 ? 0 : (i0 >= res ? res - 1 : i0);
        idx0[a] = i0;
        r[a] = pg - (double)i0;
    }
    if (ts_find(keys, slots, mask, ts_pack(index0[0], index0[1], index0[2])) < 0) return 0.0;
    double sum = 0.0;
    for (int c = 0; c < 8; ++c) {
        int index1[3], idx1[3];
        double w = 1.0;
        for (int a = 0; a < 3; ++a) {
            const int sh = (c >> (2 - a)) & 1;
            w *= sh ? r[a] : 1.0 - r[a];
            idx1[a] = idx0[a] + sh;
            index1[a] = index0[a];
            if (idx1[a] >= res) {
                idx1[a] -= res;
                index1[a] += 1;
            }
        }
        const int s = ts_find(keys, slots, mask, ts_pack(index1[0], index1[1], index1[2]));
        if (s >= 0) sum += w * (double)ts_block(slab_base, slab_units, unit_bytes, s)[((int64_t)idx1[0] * res * res + idx1[1] * res + idx1[2]) * 5];
    }
    return sum;
}

// Open3D ScalableTSDFVolume::ExtractPointCloud (normals: GetNormalAt, the normalised central difference of GetTSDFAt at +-0.99 voxel): a voxel with weight != 0 and |tsdf| < 0.98 looks at its +x, +y,
// +z neighbour (in the neighbouring unit, found through the table, when it is the last of its row); a sign change puts a point at the
// linear zero crossing, colour interpolated alike.  WRITE = false counts per unit, WRITE = true writes at unit_offset[u] + a
// per-unit cursor (order inside a unit is arbitrary, as the order of units is in Open3D's hash map).
template <bool WRITE>
__global__ __launch_bounds__(256) void tsdf_extract_kernel(const int32_t* __restrict__ unit_index, const long long* __restrict__ keys,
                                                            const int32_t* __restrict__ slots, unsigned mask, const int64_t* __restrict__ slab_base,
                                                            int slab_units, int res, double voxel_length, int32_t* __restrict__ unit_count,
                                                            const int64_t* __restrict__ unit_offset, float* __restrict__ points,
                                                            float* __restrict__ colors, float* __restrict__ normals) {
    const int nvox = res * res * res;
    const int bpu = (nvox + 255) / 256;                      // the unit is folded into blockIdx.x (no 65 535-unit limit)
    const int u = blockIdx.x / bpu;
    const int v = (blockIdx.x - u * bpu) * 256 + threadIdx.x;
    const int64_t unit_bytes = (int64_t)nvox * 20;
    int found = 0;
    float pts[3][3], cols[3][3];
    double ptd[3][3];
    if (v < nvox) {
        const float* base = ts_block(slab_base, slab_units, unit_bytes, u);
        const float f0 = base[(int64_t)v * 5], w0 = base[(int64_t)v * 5 + 1];
        if (w0 != 0.0f && f0 < 0.98f && f0 >= -0.98f) {
            const int idx[3] = {v / (res * res), (v / res) % res, v % res};
            const int ui3[3] = {unit_index[3 * u], unit_index[3 * u + 1], unit_index[3 * u + 2]};
            const double unit_len = voxel_length * res, half = voxel_length * 0.5;
            double p0[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) p0[a] = half + voxel_length * idx[a] + unit_len * ui3[a];
            const int stride[3] = {res * res, res, 1};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float* nb = base;
                int64_t nv = (int64_t)v + stride[a];
                if (idx[a] + 1 >= res) {             // the neighbour lives in the next unit along this axis
                    const int ns = ts_find(keys, slots, mask, ts_pack(ui3[0] + (a == 0), ui3[1] + (a == 1), ui3[2] + (a == 2)));
                    nb = ns >= 0 ? ts_block(slab_base, slab_units, unit_bytes, ns) : nullptr;
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
                            ptd[found][k] = p[k];
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
            if (normals) {
                const double gap = 0.99 * voxel_length;
                double n[3];
                for (int a = 0; a < 3; ++a) {
                    double pp[3] = {ptd[k][0], ptd[k][1], ptd[k][2]}, pm[3] = {ptd[k][0], ptd[k][1], ptd[k][2]};
                    pp[a] += gap;
                    pm[a] -= gap;
                    n[a] = ts_tsdf_at(pp, keys, slots, mask, slab_base, slab_units, res, voxel_length) -
                           ts_tsdf_at(pm, keys, slots, mask, slab_base, slab_units, res, voxel_length);
                }
                const double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                for (int c = 0; c < 3; ++c) normals[(at + k) * 3 + c] = (float)(len > 0.0 ? n[c] / len : n[c]);      // Eigen's normalized(): a zero vector stays zero
            }
        }
    }
}

// ScalableTSDFVolume::ExtractTriangleMesh: marching cubes over every voxel cube of every unit.  Thread = the cube whose corner 0 is
// voxel v of unit u; corner c sits at voxel + (c & 1, (c >> 1) & 1, (c >> 2) & 1), in a neighbouring unit (found through the table)
// when it runs off this one.  A cube with a corner of weight 0 (or in a unit that does not exist) is skipped, as in Open3D; bit c of
// the case = tsdf(corner c) < 0; mc_tab = [tri table int32 [256][tri_width], -1 terminated, three edge ids per triangle | edge
// corners int32 [12][2]] (bodyslam_amd/marching_cubes.py).  WRITE = false counts triangles per unit; WRITE = true writes, per
// triangle corner, the vertex on its cube edge (linear zero crossing between the edge's two corners, colour alike) and the edge's
// 62-bit identity (global voxel coordinates of its lower corner, 20 bits each, biased, + axis): the host merges equal identities
// into one vertex.
template <bool WRITE>
__global__ __launch_bounds__(256) void tsdf_mesh_kernel(const int32_t* __restrict__ unit_index, const long long* __restrict__ keys,
                                                         const int32_t* __restrict__ slots, unsigned mask, const int64_t* __restrict__ slab_base,
                                                         int slab_units, int res, double voxel_length, const int32_t* __restrict__ mc_tab, int tri_width,
                                                         int32_t* __restrict__ unit_count, const int64_t* __restrict__ unit_offset,
                                                         float* __restrict__ verts, float* __restrict__ cols, long long* __restrict__ vkeys,
                                                         int32_t* __restrict__ err) {
    const int nvox = res * res * res;
    const int bpu = (nvox + 255) / 256;
    const int u = blockIdx.x / bpu;
    const int v = (blockIdx.x - u * bpu) * 256 + threadIdx.x;
    const int64_t unit_bytes = (int64_t)nvox * 20;
    int ntri = 0, cs = 0;
    const float* cp[8];
    if (v < nvox) {
        const int idx[3] = {v / (res * res), (v / res) % res, v % res};
        const int ui3[3] = {unit_index[3 * u], unit_index[3 * u + 1], unit_index[3 * u + 2]};
        const float* blk[8];
        bool have[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) have[o] = false;
        blk[0] = ts_block(slab_base, slab_units, unit_bytes, u);
        have[0] = true;
        bool ok = true;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            int q[3], o = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                q[a] = idx[a] + ((c >> a) & 1);
                if (q[a] >= res) {
                    q[a] -= res;
                    o |= 1 << a;
                }
            }
            if (!have[o]) {
                const int ns = ts_find(keys, slots, mask, ts_pack(ui3[0] + (o & 1), ui3[1] + ((o >> 1) & 1), ui3[2] + ((o >> 2) & 1)));
                blk[o] = ns >= 0 ? ts_block(slab_base, slab_units, unit_bytes, ns) : nullptr;
                have[o] = true;
            }
            cp[c] = blk[o] ? blk[o] + ((int64_t)q[0] * res * res + q[1] * res + q[2]) * 5 : nullptr;
            if (!cp[c] || cp[c][1] == 0.0f) ok = false;
            else if (cp[c][0] < 0.0f) cs |= 1 << c;
        }
        if (ok && cs != 0 && cs != 255) {
            const int32_t* t = mc_tab + cs * tri_width;
            while (t[3 * ntri] >= 0) ++ntri;
        } else {
            cs = 0;
        }
    }
    if (!WRITE) {
        int s = ntri;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(unit_count + u, s);
        return;
    }
    if (!ntri) return;
    const int idx[3] = {v / (res * res), (v / res) % res, v % res};
    const int ui3[3] = {unit_index[3 * u], unit_index[3 * u + 1], unit_index[3 * u + 2]};
    const double unit_len = voxel_length * res, half = voxel_length * 0.5;
    const int64_t at = (unit_offset[u] + atomicAdd(unit_count + u, ntri)) * 3;          // first triangle corner of this cube
    const int32_t* t = mc_tab + cs * tri_width;
    const int32_t* ed = mc_tab + 256 * tri_width;
    for (int k = 0; k < 3 * ntri; ++k) {
        const int e = t[k], a = ed[2 * e], b = ed[2 * e + 1];
        const int axis = (a ^ b) == 1 ? 0 : ((a ^ b) == 2 ? 1 : 2);
        const float f0 = cp[a][0], f1 = cp[b][0];
        const double w = (double)(0.0f - f0) / ((double)f1 - (double)f0);           // Open3D: (0 - f0) / (f1 - f0)
        long long key = (long long)axis;
        double p[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const int g = ui3[ax] * res + idx[ax] + ((a >> ax) & 1);                 // global voxel coordinate of the edge's lower corner
            p[ax] = half + voxel_length * (double)(idx[ax] + ((a >> ax) & 1)) + unit_len * (double)ui3[ax];
            if (g < -(1 << 19) || g >= (1 << 19)) *err = 1;
            key |= (long long)((g + (1 << 19)) & ((1 << 20) - 1)) << (2 + 20 * ax);
        }
        p[axis] += w * voxel_length;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            verts[(at + k) * 3 + c] = (float)p[c];
            cols[(at + k) * 3 + c] = (float)(((double)cp[a][2 + c] + w * ((double)cp[b][2 + c] - (double)cp[a][2 + c])) / 255.0);
        }
        vkeys[at + k] = key;
    }
}

}  // namespace bs

using namespace bs;

static bool det_ok(const double* m) {
    const double det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
    return det > 1e-12 || det < -1e-12;
}

static void tsdf_cam(TsdfCam& cam, const double* K, const double* m12) {
    cam.fx = K[0]; cam.fy = K[1]; cam.cx = K[2]; cam.cy = K[3];
    cam.ifx = 1.0 / K[0]; cam.ify = 1.0 / K[1];
    for (int i = 0; i < 12; ++i) cam.e[i] = m12[i];
}

extern "C" int bs_tsdf_touch(const float* depth, int32_t H, int32_t W, int32_t stride, const double* K, const double* pose, double unit_length,
                             double sdf_trunc, void* table_keys, int32_t* table_slots, int32_t* table_stamp, int32_t table_cap, int32_t frame_id,
                             int32_t* unit_index, int32_t max_units, int32_t* counters, int32_t* touched, void* stream) {
    if (!initialized()) { set_error("bs_tsdf_touch: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(depth && K && pose && table_keys && table_slots && table_stamp && unit_index && counters && touched, "bs_tsdf_touch: null argument");
    BS_REQUIRE(H > 0 && W > 0 && stride > 0 && unit_length > 0.0 && sdf_trunc > 0.0 && max_units > 0, "bs_tsdf_touch: bad geometry");
    BS_REQUIRE(table_cap >= 256 && (table_cap & (table_cap - 1)) == 0, "bs_tsdf_touch: table_cap=%d must be a power of two >= 256", table_cap);
    TsdfCam cam;
    tsdf_cam(cam, K, pose);
    const int span = (int)floor(2.0 * sdf_trunc / unit_length) + 2;          // hi - lo + 1 never exceeds this
    BS_REQUIRE(span <= 64, "bs_tsdf_touch: sdf_trunc / unit_length too large");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    BS_CHECK_HIP(hipMemsetAsync(counters + 1, 0, sizeof(int32_t), st));    // n_touched (n_units persists; the overflow flag is sticky: the caller clears it)
    const int64_t threads = (int64_t)cdiv(W, stride) * cdiv(H, stride) * span * span * span;
    hipLaunchKernelGGL(tsdf_touch_kernel, dim3((unsigned)cdiv64(threads, 256)), dim3(256), 0, st, depth, H, W, stride, cam, unit_length, sdf_trunc, span,
                       reinterpret_cast<long long*>(table_keys), table_stamp, (unsigned)(table_cap - 1), frame_id, counters);
    BS_CHECK_LAUNCH();
    // world -> camera for the frustum test of the units: the inverse of the 3x4 `pose` (general affine inverse)
    TsdfCam view = cam;
    {
        const double* m = pose;
        const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
        const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
        const double id = 1.0 / det;
        const double inv[9] = {(e * i - f * h) * id, (c * h - b * i) * id, (b * f - c * e) * id, (f * g - d * i) * id, (a * i - c * g) * id,
                               (c * d - a * f) * id, (d * h - e * g) * id, (b * g - a * h) * id, (a * e - b * d) * id};
        for (int r = 0; r < 3; ++r) {
            for (int q = 0; q < 3; ++q) view.e[4 * r + q] = inv[3 * r + q];
            view.e[4 * r + 3] = -(inv[3 * r] * m[3] + inv[3 * r + 1] * m[7] + inv[3 * r + 2] * m[11]);
        }
    }
    const int cull = diag_env("BS_TSDF_NO_CULL") == nullptr && det_ok(pose);
    hipLaunchKernelGGL(tsdf_assign_kernel, dim3(cdiv(table_cap, 256)), dim3(256), 0, st, reinterpret_cast<const long long*>(table_keys), table_slots,
                       table_stamp, (unsigned)table_cap, frame_id, unit_index, max_units, counters, touched, view, unit_length, W, H, cull);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_tsdf_integrate(const float* depth, const uint8_t* color, int32_t H, int32_t W, const double* K, const double* extrinsic,
                                 const int32_t* unit_index, const int32_t* touched, int32_t n_touched, const int64_t* slab_base, int32_t slab_units,
                                 int32_t res, double voxel_length, double sdf_trunc, const int32_t* n_touched_dev, void* stream) {
    if (!initialized()) { set_error("bs_tsdf_integrate: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(n_touched >= 0 && H > 0 && W > 0 && res > 0 && res <= 64 && slab_units > 0 && voxel_length > 0.0 && sdf_trunc > 0.0,
               "bs_tsdf_integrate: bad geometry");
    if (n_touched == 0) return BS_OK;
    BS_REQUIRE(depth && K && extrinsic && unit_index && touched && slab_base && n_touched_dev, "bs_tsdf_integrate: null argument");
    TsdfCam cam;
    tsdf_cam(cam, K, extrinsic);
    const int nvox = res * res * res, bpu = cdiv(nvox, 256);
    BS_REQUIRE((long long)bpu * n_touched < 0x7fffffffll, "bs_tsdf_integrate: too many units for one launch");
    // n_touched (host) only sizes the grid; the kernel reads the true count from n_touched_dev, so a caller that has not read it back
    // passes any generous value (a few blocks per CU walk the work items)
    const long long want = (long long)bpu * n_touched;
    const unsigned grid = (unsigned)(want < 16ll * cu_count() ? want : 16ll * cu_count());
    const int cull = 0;
    hipLaunchKernelGGL(tsdf_integrate_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), depth, color, H, W,
                       cam, unit_index, touched, n_touched_dev, bpu, slab_base, slab_units, res, voxel_length, sdf_trunc, cull);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

// ---- the frame-batched path ---------------------------------------------------------------------------------------------------------
static bool affine_inverse(const double* m, double* out12) {      // rows 0..2 of the inverse of the affine 4x4 whose rows 0..2 are m
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (!(det > 1e-12 || det < -1e-12)) return false;
    const double id = 1.0 / det;
    const double inv[9] = {(e * i - f * h) * id, (c * h - b * i) * id, (b * f - c * e) * id, (f * g - d * i) * id, (a * i - c * g) * id,
                           (c * d - a * f) * id, (d * h - e * g) * id, (b * g - a * h) * id, (a * e - b * d) * id};
    for (int r = 0; r < 3; ++r) {
        for (int q = 0; q < 3; ++q) out12[4 * r + q] = inv[3 * r + q];
        out12[4 * r + 3] = -(inv[3 * r] * m[3] + inv[3 * r + 1] * m[7] + inv[3 * r + 2] * m[11]);
    }
    return true;
}

// pinned staging for the frame records: two slots, each guarded by the event of its last copy (the host never waits for the stream
// unless a third batch is uploaded before the first one's copy has run)
namespace {
struct FrameStage {
    TsdfFrame* host[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    int next = 0;
};
FrameStage g_stage;
std::mutex g_stage_mu;
}  // namespace

extern "C" int bs_tsdf_frames_upload(const float* const* depth, const uint8_t* const* color, const double* K, const double* extrinsics, const double* poses,
                                     int32_t n_frames, void* frames_dev, void* stream) {
    if (!initialized()) { set_error("bs_tsdf_frames_upload: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(depth && K && extrinsics && poses && frames_dev, "bs_tsdf_frames_upload: null argument");
    BS_REQUIRE(n_frames > 0 && n_frames <= BS_TSDF_BATCH_MAX, "bs_tsdf_frames_upload: n_frames=%d outside 1..%d", n_frames, BS_TSDF_BATCH_MAX);
    for (int f = 0; f < n_frames; ++f) {
        BS_REQUIRE(depth[f], "bs_tsdf_frames_upload: frame %d has no depth image", f);
        BS_REQUIRE(!color || (color[f] != nullptr) == (color[0] != nullptr), "bs_tsdf_frames_upload: either every frame of a batch has a colour image or none");
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lock(g_stage_mu);
    const int slot = g_stage.next;
    g_stage.next ^= 1;
    if (!g_stage.host[slot]) {
        BS_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&g_stage.host[slot]), sizeof(TsdfFrame) * BS_TSDF_BATCH_MAX, hipHostMallocDefault));
        BS_CHECK_HIP(hipEventCreateWithFlags(&g_stage.done[slot], hipEventDisableTiming));
    } else {
        BS_CHECK_HIP(hipEventSynchronize(g_stage.done[slot]));
    }
    TsdfFrame* h = g_stage.host[slot];
    for (int f = 0; f < n_frames; ++f) {
        h[f].depth = depth[f];
        h[f].color = color ? color[f] : nullptr;
        h[f].fx = K[0]; h[f].fy = K[1]; h[f].cx = K[2]; h[f].cy = K[3];
        h[f].ifx = 1.0 / K[0]; h[f].ify = 1.0 / K[1];
        for (int i = 0; i < 12; ++i) {
            h[f].view[i] = extrinsics[16 * f + i];
            h[f].pose[i] = poses[16 * f + i];
        }
    }
    BS_CHECK_HIP(hipMemcpyAsync(frames_dev, h, sizeof(TsdfFrame) * n_frames, hipMemcpyHostToDevice, st));
    BS_CHECK_HIP(hipEventRecord(g_stage.done[slot], st));
    return BS_OK;
}

extern "C" int bs_tsdf_touch_batch(const void* frames_dev, int32_t n_frames, int32_t H, int32_t W, int32_t stride, double unit_length, double sdf_trunc,
                                   void* table_keys, void* table_fmask, int32_t table_cap, int32_t* counters, void* stream) {
    if (!initialized()) { set_error("bs_tsdf_touch_batch: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(frames_dev && table_keys && table_fmask && counters, "bs_tsdf_touch_batch: null argument");
    BS_REQUIRE(n_frames > 0 && n_frames <= BS_TSDF_BATCH_MAX && H > 0 && W > 0 && stride > 0 && unit_length > 0.0 && sdf_trunc > 0.0,
               "bs_tsdf_touch_batch: bad geometry");
    BS_REQUIRE(table_cap >= 256 && (table_cap & (table_cap - 1)) == 0, "bs_tsdf_touch_batch: table_cap=%d must be a power of two >= 256", table_cap);
    const int span = (int)floor(2.0 * sdf_trunc / unit_length) + 2;
    BS_REQUIRE(span <= 64, "bs_tsdf_touch_batch: sdf_trunc / unit_length too large");
    const int64_t threads = (int64_t)cdiv(W, stride) * cdiv(H, stride) * span * span * span;
    hipLaunchKernelGGL(tsdf_touch_batch_kernel, dim3((unsigned)cdiv64(threads, 256), n_frames), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       static_cast<const TsdfFrame*>(frames_dev), H, W, stride, unit_length, sdf_trunc, span, reinterpret_cast<long long*>(table_keys),
                       reinterpret_cast<unsigned long long*>(table_fmask), (unsigned)(table_cap - 1), counters);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_tsdf_integrate_batch(const void* frames_dev, int32_t n_frames, int32_t H, int32_t W, const void* table_keys, int32_t* table_slots,
                                       void* table_fmask, int32_t table_cap, int32_t* unit_index, int32_t max_units, int32_t* counters, int32_t* touched,
                                       void* unit_mask, const int64_t* slab_base, int32_t slab_units, int32_t res, double voxel_length, double sdf_trunc,
                                       void* stream) {
    if (!initialized()) { set_error("bs_tsdf_integrate_batch: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(frames_dev && table_keys && table_slots && table_fmask && unit_index && counters && touched && unit_mask && slab_base,
               "bs_tsdf_integrate_batch: null argument");
    BS_REQUIRE(n_frames > 0 && n_frames <= BS_TSDF_BATCH_MAX && H > 0 && W > 0 && res > 0 && res <= 64 && slab_units > 0 && voxel_length > 0.0 &&
                   sdf_trunc > 0.0 && max_units > 0,
               "bs_tsdf_integrate_batch: bad geometry");
    BS_REQUIRE(table_cap >= 256 && (table_cap & (table_cap - 1)) == 0, "bs_tsdf_integrate_batch: table_cap=%d must be a power of two >= 256", table_cap);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const TsdfFrame* frames = static_cast<const TsdfFrame*>(frames_dev);
    BS_CHECK_HIP(hipMemsetAsync(counters + 1, 0, sizeof(int32_t), st));
    const int cull = diag_env("BS_TSDF_NO_CULL") == nullptr;
    hipLaunchKernelGGL(tsdf_assign_batch_kernel, dim3(cdiv(table_cap, 256)), dim3(256), 0, st, frames, reinterpret_cast<const long long*>(table_keys),
                       table_slots, reinterpret_cast<unsigned long long*>(table_fmask), (unsigned)table_cap, unit_index, max_units, counters, touched,
                       reinterpret_cast<unsigned long long*>(unit_mask), voxel_length * res, W, H, cull);
    BS_CHECK_LAUNCH();
    const int nvox = res * res * res, bpu = cdiv(nvox, 256);
    const long long want = (long long)bpu * max_units;
    const unsigned grid = (unsigned)(want < 16ll * cu_count() ? want : 16ll * cu_count());
    hipLaunchKernelGGL(tsdf_integrate_batch_kernel, dim3(grid), dim3(256), 0, st, frames, H, W, unit_index, touched, counters + 1,
                       reinterpret_cast<const unsigned long long*>(unit_mask), bpu, slab_base, slab_units, res, voxel_length, sdf_trunc);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_tsdf_extract(const int32_t* unit_index, int32_t units, const void* table_keys, const int32_t* table_slots, int32_t table_cap,
                               const int64_t* slab_base, int32_t slab_units, int32_t res, double voxel_length, int32_t* unit_count,
                               const int64_t* unit_offset, float* points, float* colors, float* normals, void* stream) {
    if (!initialized()) { set_error("bs_tsdf_extract: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(units >= 0 && res > 0 && res <= 64 && slab_units > 0 && voxel_length > 0.0, "bs_tsdf_extract: bad geometry");
    if (units == 0) return BS_OK;
    BS_REQUIRE(unit_index && table_keys && table_slots && slab_base && unit_count, "bs_tsdf_extract: null argument");
    BS_REQUIRE(table_cap >= 256 && (table_cap & (table_cap - 1)) == 0, "bs_tsdf_extract: table_cap must be a power of two >= 256");
    BS_REQUIRE((points == nullptr) == (unit_offset == nullptr) && (points == nullptr) == (colors == nullptr),
               "bs_tsdf_extract: the write pass needs unit_offset, points and colors; the count pass none of them");
    const int nvox = res * res * res;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    BS_CHECK_HIP(hipMemsetAsync(unit_count, 0, sizeof(int32_t) * units, st));
    BS_REQUIRE((long long)cdiv(nvox, 256) * units < 0x7fffffffll, "bs_tsdf_extract: too many units for one launch");
    const dim3 grid((unsigned)(cdiv(nvox, 256) * units));
    const long long* keys = reinterpret_cast<const long long*>(table_keys);
    if (!points)
        hipLaunchKernelGGL(tsdf_extract_kernel<false>, grid, dim3(256), 0, st, unit_index, keys, table_slots, (unsigned)(table_cap - 1), slab_base,
                           slab_units, res, voxel_length, unit_count, unit_offset, points, colors, (float*)nullptr);
    else
        hipLaunchKernelGGL(tsdf_extract_kernel<true>, grid, dim3(256), 0, st, unit_index, keys, table_slots, (unsigned)(table_cap - 1), slab_base,
                           slab_units, res, voxel_length, unit_count, unit_offset, points, colors, normals);
    BS_CHECK_LAUNCH();
    return BS_OK;
}

extern "C" int bs_tsdf_mesh(const int32_t* unit_index, int32_t units, const void* table_keys, const int32_t* table_slots, int32_t table_cap,
                            const int64_t* slab_base, int32_t slab_units, int32_t res, double voxel_length, const int32_t* mc_tab, int32_t tri_width,
                            int32_t* unit_count, const int64_t* unit_offset, float* vertices, float* colors, int64_t* vertex_keys, int32_t* err,
                            void* stream) {
    if (!initialized()) { set_error("bs_tsdf_mesh: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(units >= 0 && res > 0 && res <= 64 && slab_units > 0 && voxel_length > 0.0 && tri_width >= 4, "bs_tsdf_mesh: bad geometry");
    if (units == 0) return BS_OK;
    BS_REQUIRE(unit_index && table_keys && table_slots && slab_base && unit_count && mc_tab && err, "bs_tsdf_mesh: null argument");
    BS_REQUIRE(table_cap >= 256 && (table_cap & (table_cap - 1)) == 0, "bs_tsdf_mesh: table_cap must be a power of two >= 256");
    BS_REQUIRE((vertices == nullptr) == (unit_offset == nullptr) && (vertices == nullptr) == (colors == nullptr) &&
                   (vertices == nullptr) == (vertex_keys == nullptr),
               "bs_tsdf_mesh: the write pass needs unit_offset, vertices, colors and vertex_keys; the count pass none of them");
    const int nvox = res * res * res;
    BS_REQUIRE((long long)cdiv(nvox, 256) * units < 0x7fffffffll, "bs_tsdf_mesh: too many units for one launch");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    BS_CHECK_HIP(hipMemsetAsync(unit_count, 0, sizeof(int32_t) * units, st));
    const dim3 grid((unsigned)(cdiv(nvox, 256) * units));
    const long long* keys = reinterpret_cast<const long long*>(table_keys);
    if (!vertices)
        hipLaunchKernelGGL(tsdf_mesh_kernel<false>, grid, dim3(256), 0, st, unit_index, keys, table_slots, (unsigned)(table_cap - 1), slab_base, slab_units,
                           res, voxel_length, mc_tab, tri_width, unit_count, unit_offset, vertices, colors, (long long*)nullptr, err);
    else
        hipLaunchKernelGGL(tsdf_mesh_kernel<true>, grid, dim3(256), 0, st, unit_index, keys, table_slots, (unsigned)(table_cap - 1), slab_base, slab_units,
                           res, voxel_length, mc_tab, tri_width, unit_count, unit_offset, vertices, colors, reinterpret_cast<long long*>(vertex_keys), err);
    BS_CHECK_LAUNCH();
    return BS_OK;
}
