// Probe: the result of a transcendental instruction (v_rcp_f32 / v_exp_f32: own unit, a quarter of the VALU rate) consumed by a packed-fp32
// instruction a few instructions later -- the distances the compiler emits in logbinom_kernel's GELU.  The destination registers hold a stale
// value (5.0) before the v_rcp; a lane that accumulates 5.0 * x instead of 0.5 * x has read the register before the unit wrote it.
//   hipcc --offload-arch=gfx950 -O2 -o trans_fwd_probe trans_fwd_probe.hip ;  ./trans_fwd_probe [seconds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PRE "v_mov_b32 v20, 5.0\n v_mov_b32 v21, 5.0\n s_nop 7\n v_rcp_f32 v20, %[two]\n v_rcp_f32 v21, %[two]\n"
#define OPS : [acc] "+v"(acc), [t] "+v"(t) : [x] "v"(x), [two] "v"(two) : "v20", "v21", "v22", "v23", "v24", "v25", "s4", "memory"

template <int V>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    f32x2 acc = {0.f, 0.f};
    f32x2 x = {1.0f, 1.0f};
    float t = 0.f, two = 2.0f;
    for (int i = 0; i < iters; ++i) {
        if (V == 0)        // the GELU's sequence: rcp, rcp, pk_mul, s_mov, exp, pk_fma(rcp results)
            asm volatile(PRE "v_pk_mul_f32 v[22:23], %[x], %[x]\n s_mov_b32 s4, 0\n v_exp_f32 v24, %[t]\n v_pk_fma_f32 %[acc], v[20:21], %[x], %[acc]\n" OPS);
        else if (V == 1)   // two instructions between
            asm volatile(PRE "v_pk_mul_f32 v[22:23], %[x], %[x]\n s_mov_b32 s4, 0\n v_pk_fma_f32 %[acc], v[20:21], %[x], %[acc]\n" OPS);
        else if (V == 2)   // one VALU between (the documented minimum: one wait state)
            asm volatile(PRE "v_pk_mul_f32 v[22:23], %[x], %[x]\n v_pk_fma_f32 %[acc], v[20:21], %[x], %[acc]\n" OPS);
        else if (V == 3)   // one s_nop between
            asm volatile(PRE "s_nop 0\n v_pk_fma_f32 %[acc], v[20:21], %[x], %[acc]\n" OPS);
        else if (V == 4)   // nothing between (inline assembly: the compiler's hazard pass does not see inside)
            asm volatile(PRE "v_pk_fma_f32 %[acc], v[20:21], %[x], %[acc]\n" OPS);
        else if (V == 5)   // three transcendentals queued, consumer of the SECOND right behind the third
            asm volatile(PRE "v_exp_f32 v24, %[t]\n v_exp_f32 v25, %[t]\n v_pk_fma_f32 %[acc], v[20:21], %[x], %[acc]\n" OPS);
        else if (V == 6)   // unpacked consumer, one VALU between
            asm volatile(PRE "v_mul_f32 v22, %[t], %[t]\n v_fma_f32 %[t], v21, 0, %[t]\n v_pk_fma_f32 %[acc], v[20:21], %[x], %[acc]\n" OPS);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + t;
}

template <int V>
static void run(float* out, std::vector<float>& h, int blocks, int iters, long* bad, long* q) {
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    const float want = (float)iters;      // 0.5 + 0.5 per iteration
    for (size_t i = 0; i < h.size(); ++i)
        if (h[i] != want) { ++*bad; ++q[(i & 63) / 16]; }
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 20.0;
    const int blocks = 256 * 12, iters = 1024;
    float* out;
    std::vector<float> h((size_t)blocks * 256);
    (void)hipMalloc(&out, h.size() * 4);
    long bad[7] = {0}, q[7][4] = {{0}}, launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        run<0>(out, h, blocks, iters, &bad[0], q[0]);
        run<1>(out, h, blocks, iters, &bad[1], q[1]);
        run<2>(out, h, blocks, iters, &bad[2], q[2]);
        run<3>(out, h, blocks, iters, &bad[3], q[3]);
        run<4>(out, h, blocks, iters, &bad[4], q[4]);
        run<5>(out, h, blocks, iters, &bad[5], q[5]);
        run<6>(out, h, blocks, iters, &bad[6], q[6]);
        ++launches;
    }
    const char* names[7] = {"rcp rcp pk_mul s_mov exp | pk_fma", "rcp rcp pk_mul s_mov | pk_fma", "rcp rcp pk_mul | pk_fma", "rcp rcp s_nop | pk_fma",
                            "rcp rcp | pk_fma", "rcp rcp exp exp | pk_fma", "rcp rcp v_mul v_fma(v21) | pk_fma"};
    printf("%ld launches of each form, %d waves x %d iterations per launch\n", launches, blocks * 4, iters);
    for (int v = 0; v < 7; ++v)
        printf("  %-36s lanes with a wrong sum: %ld   by quarter of the wave: %ld %ld %ld %ld\n", names[v], bad[v], q[v][0], q[v][1], q[v][2], q[v][3]);
    return 0;
}
