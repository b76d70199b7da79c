// Probe: does an out-of-range `buffer_load_dwordx4 ... lds` write ZEROS into LDS (or leave it untouched)?
// The implicit-GEMM kernel relies on zeros for the padding taps.   hipcc --offload-arch=gfx950 -o /tmp/probe lds_dma_oob.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* a, float* out, int nbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int lane = threadIdx.x & 63;
    float* s = (float*)smem;
    for (int i = 0; i < 4; ++i) s[lane * 4 + i] = -7.0f;   // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, nbytes, 0x00020000);
    unsigned voff = (lane & 1) ? 0xFFFFFFF0u : lane * 16;     // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = s[lane * 4 + i];
}
int main() {
    float h[256], *d, *o, r[256];
    for (int i = 0; i < 256; ++i) h[i] = 100.0f + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, d, o, (int)sizeof(h));
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int zeros = 0, sentinel = 0, good = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        float v = r[l * 4 + i];
        if (l & 1) { zeros += (v == 0.0f); sentinel += (v == -7.0f); } else good += (v == h[l * 4 + i]);
    }
    printf("in-range correct %d/128, OOB lanes: zeros %d/128, sentinel-left %d/128, sample %f\n", good, zeros, sentinel, r[4]);
    return 0;
}
