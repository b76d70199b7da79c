/* A host without Python: runs an exported ZoeDepth engine on one batch of frames through the C ABI alone.
 *   gcc -O2 -Iinclude examples/zoedepth_host.c -o zoedepth_host -Lbodyslam_amd -lbodyslam_hip -Wl,-rpath,$PWD/bodyslam_amd
 *   ./zoedepth_host model.bseng frames.u8 B H W depth_m.f32        (frames.u8: B*H*W*3 bytes, RGB, row-major)
 * What the reference does in Python -- DepthEstimator.infer_depth_map (BodySLAM_Refactored/src/depth_estimation/interface.py:39-45) --
 * a C / C++ / Go / Rust host does with these six calls; the engine file comes from `python -m bodyslam_amd.engine_export`. */
#include <stdio.h>
#include <stdlib.h>

#include "bodyslam_hip.h"

#define CHECK(call)                                                         \
    do {                                                                    \
        if ((call) != 0) {                                                  \
            fprintf(stderr, "%s failed: %s\n", #call, bs_last_error());     \
            return 1;                                                       \
        }                                                                   \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 7) {
        fprintf(stderr, "usage: %s engine frames.u8 B H W depth_out.f32\n", argv[0]);
        return 2;
    }
    const long B = atol(argv[3]), H = atol(argv[4]), W = atol(argv[5]);
    const long n_in = B * H * W * 3, n_out = B * H * W;
    unsigned char* frames = (unsigned char*)malloc((size_t)n_in);
    float* depth = (float*)malloc((size_t)n_out * sizeof(float));
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(frames, 1, (size_t)n_in, f) != (size_t)n_in) {
        fprintf(stderr, "cannot read %ld bytes of frames from %s\n", n_in, argv[2]);
        return 1;
    }
    fclose(f);
    bs_engine* eng = NULL;
    CHECK(bs_init(0));
    CHECK(bs_engine_load(argv[1], &eng));
    CHECK(bs_engine_upload(eng, "frames", frames, n_in));
    CHECK(bs_engine_run(eng, NULL));                                   /* the default stream */
    CHECK(bs_engine_download(eng, "depth_m", depth, n_out * (long)sizeof(float)));
    f = fopen(argv[6], "wb");
    if (!f || fwrite(depth, sizeof(float), (size_t)n_out, f) != (size_t)n_out) {
        fprintf(stderr, "cannot write %s\n", argv[6]);
        return 1;
    }
    fclose(f);
    printf("depth of %ld frames %ldx%ld: first pixel %.6f m, engine holds %lld bytes of device memory\n", B, W, H, depth[0],
           (long long)bs_engine_device_bytes(eng));
    CHECK(bs_engine_destroy(eng));
    free(frames);
    free(depth);
    return 0;
}
