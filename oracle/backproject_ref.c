/* CPU oracle (plain C) for the integer part of the 3DM back-projection -- TEST INFRASTRUCTURE ONLY.
 *
 * Restates BodySLAM_not_refactored/3DM/scaling_system.py:72-77 (pixel_to_3d) together with the
 * RGBD constants of 3DM/slam_utils.py:173,212-220,232 (z = u16 / depth_scale as float32, values
 * >= depth_trunc zeroed, valid iff z > 0) and emits the row-major order of the valid pixels.
 * Checked against oracle/geom3d_ref.py and tests/golden/geom3d_backproject.npz by
 * tests/test_oracle_geom3d.py.  Built by oracle/Makefile into oracle/_build/libbackproject_ref.so.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 */
#include <stdint.h>

/* returns the number of valid pixels; xyz may be NULL (indices only) */
int64_t bsref_backproject(const uint16_t* depth, int H, int W, const double K[4], double depth_scale,
                          double depth_trunc, const double* pose /* 16, row-major, nullable */,
                          float* xyz, int32_t* idx) {
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const float scale32 = (float)depth_scale, trunc32 = (float)depth_trunc;
    int64_t m = 0;
    for (int v = 0; v < H; ++v) {
        for (int u = 0; u < W; ++u) {
            float z32 = (float)depth[(int64_t)v * W + u] / scale32;
            if (z32 >= trunc32) z32 = 0.0f;
            if (!(z32 > 0.0f)) continue;
            if (idx) idx[m] = (int32_t)((int64_t)v * W + u);
            if (xyz) {
                double z = (double)z32;
                double x = ((double)u - cx) * z / fx;
                double y = ((double)v - cy) * z / fy;
                if (pose) {
                    double wx = pose[0] * x + pose[1] * y + pose[2] * z + pose[3];
                    double wy = pose[4] * x + pose[5] * y + pose[6] * z + pose[7];
                    double wz = pose[8] * x + pose[9] * y + pose[10] * z + pose[11];
                    x = wx; y = wy; z = wz;
                }
                xyz[3 * m + 0] = (float)x;
                xyz[3 * m + 1] = (float)y;
                xyz[3 * m + 2] = (float)z;
            }
            ++m;
        }
    }
    return m;
}
