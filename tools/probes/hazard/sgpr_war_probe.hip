// Probe: does a scalar register that a packed-fp32 VALU instruction reads keep its value for all four quarter-passes of the wave when the NEXT
// instructions of the same wave overwrite it (SALU write / SMEM return)?  Every lane runs the same arithmetic on the same inputs, so any lane
// whose result differs from lane 0's shows an operand that changed under the instruction.  Run alone and beside tools/probes/gpu_churn.py.
//   hipcc --offload-arch=gfx950 -O2 -o sgpr_war_probe sgpr_war_probe.hip ;  ./sgpr_war_probe [seconds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(256) void probe(const float* w, float* out, int iters) {
    f32x2 acc = {0.f, 0.f};
    f32x2 x = {1.0f, 1.0f};
    float t = 0.f, u = 0.f, u2 = 0.f, one = 1.0f;
    const float* p = w + (blockIdx.x & 7) * 64;       // uniform per block; eight different lines
    for (int i = 0; i < iters; ++i) {
        if (V == 0)        // the compiler's pattern in logbinom_kernel: pk_fma reads s[4:5]; one VALU later the SALU rewrites s4 / s5
            asm volatile("s_load_dwordx2 s[4:5], %[p], 0x0\n s_waitcnt lgkmcnt(0)\n v_pk_fma_f32 %[acc], s[4:5], %[x], %[acc]\n v_mov_b32 %[t], %[t]\n"
                         "s_add_u32 s4, s4, 0x3f000000\n s_addc_u32 s5, s5, 0x3f000000\n"
                         : [acc] "+v"(acc), [t] "+v"(t) : [p] "s"(p), [x] "v"(x) : "s4", "s5", "scc", "memory");
        else if (V == 1)   // SALU write directly behind the read
            asm volatile("s_load_dwordx2 s[4:5], %[p], 0x0\n s_waitcnt lgkmcnt(0)\n v_pk_fma_f32 %[acc], s[4:5], %[x], %[acc]\n"
                         "s_add_u32 s4, s4, 0x3f000000\n s_addc_u32 s5, s5, 0x3f000000\n"
                         : [acc] "+v"(acc), [t] "+v"(t) : [p] "s"(p), [x] "v"(x) : "s4", "s5", "scc", "memory");
        else if (V == 2)   // SMEM return into the registers directly behind the read (other values: w + 32)
            asm volatile("s_load_dwordx2 s[4:5], %[p], 0x0\n s_waitcnt lgkmcnt(0)\n v_pk_fma_f32 %[acc], s[4:5], %[x], %[acc]\n"
                         "s_load_dwordx2 s[4:5], %[p], 0x80\n s_waitcnt lgkmcnt(0)\n"
                         : [acc] "+v"(acc), [t] "+v"(t) : [p] "s"(p), [x] "v"(x) : "s4", "s5", "scc", "memory");
        else if (V == 3)   // as 1 with the plain (unpacked) FMA
            asm volatile("s_load_dwordx2 s[4:5], %[p], 0x0\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %[t], s4, 1.0, %[t]\n"
                         "s_add_u32 s4, s4, 0x3f000000\n"
                         : [acc] "+v"(acc), [t] "+v"(t) : [p] "s"(p), [x] "v"(x) : "s4", "s5", "scc", "memory");
        else if (V == 4)   // as 1 with four idle cycles before the write
            asm volatile("s_load_dwordx2 s[4:5], %[p], 0x0\n s_waitcnt lgkmcnt(0)\n v_pk_fma_f32 %[acc], s[4:5], %[x], %[acc]\n s_nop 3\n"
                         "s_add_u32 s4, s4, 0x3f000000\n s_addc_u32 s5, s5, 0x3f000000\n"
                         : [acc] "+v"(acc), [t] "+v"(t) : [p] "s"(p), [x] "v"(x) : "s4", "s5", "scc", "memory");
        else if (V == 5)   // two transcendental instructions in flight (their own unit, a quarter of the rate) in front of the read, then the SALU rewrites the pair
            asm volatile("s_load_dwordx4 s[4:7], %[p], 0x0\n s_waitcnt lgkmcnt(0)\n v_rcp_f32 %[u], %[one]\n v_rcp_f32 %[u2], %[one]\n v_add_f32 %[t], %[t], %[t]\n"
                         "v_pk_fma_f32 %[acc], s[4:5], %[x], %[acc]\n s_mov_b32 s4, s6\n s_mov_b32 s5, s7\n v_mul_f32 %[t], %[t], %[t]\n"
                         "v_pk_fma_f32 %[acc], s[4:5], %[x], %[acc]\n"
                         : [acc] "+v"(acc), [t] "+v"(t), [u] "=&v"(u), [u2] "=&v"(u2) : [p] "s"(p), [x] "v"(x), [one] "v"(one) : "s4", "s5", "s6", "s7", "scc", "memory");
        else if (V == 6)   // as 5 with v_exp_f32
            asm volatile("s_load_dwordx4 s[4:7], %[p], 0x0\n s_waitcnt lgkmcnt(0)\n v_exp_f32 %[u], %[t]\n v_exp_f32 %[u2], %[t]\n"
                         "v_pk_fma_f32 %[acc], s[4:5], %[x], %[acc]\n s_mov_b32 s4, s6\n s_mov_b32 s5, s7\n v_mul_f32 %[t], %[t], %[t]\n"
                         "v_pk_fma_f32 %[acc], s[4:5], %[x], %[acc]\n"
                         : [acc] "+v"(acc), [t] "+v"(t), [u] "=&v"(u), [u2] "=&v"(u2) : [p] "s"(p), [x] "v"(x), [one] "v"(one) : "s4", "s5", "s6", "s7", "scc", "memory");
        else if (V == 7)   // as 5 with eight idle cycles between the read and the rewrite
            asm volatile("s_load_dwordx4 s[4:7], %[p], 0x0\n s_waitcnt lgkmcnt(0)\n v_rcp_f32 %[u], %[one]\n v_rcp_f32 %[u2], %[one]\n v_add_f32 %[t], %[t], %[t]\n"
                         "v_pk_fma_f32 %[acc], s[4:5], %[x], %[acc]\n s_nop 7\n s_mov_b32 s4, s6\n s_mov_b32 s5, s7\n v_mul_f32 %[t], %[t], %[t]\n"
                         "v_pk_fma_f32 %[acc], s[4:5], %[x], %[acc]\n"
                         : [acc] "+v"(acc), [t] "+v"(t), [u] "=&v"(u), [u2] "=&v"(u2) : [p] "s"(p), [x] "v"(x), [one] "v"(one) : "s4", "s5", "s6", "s7", "scc", "memory");
        else if (V == 8)   // SALU write, then the read one VALU later (the other direction), transcendentals in flight
            asm volatile("s_load_dwordx4 s[4:7], %[p], 0x0\n s_waitcnt lgkmcnt(0)\n v_rcp_f32 %[u], %[one]\n v_rcp_f32 %[u2], %[one]\n"
                         "s_mov_b32 s4, s6\n s_mov_b32 s5, s7\n v_pk_fma_f32 %[acc], s[4:5], %[x], %[acc]\n"
                         : [acc] "+v"(acc), [t] "+v"(t), [u] "=&v"(u), [u2] "=&v"(u2) : [p] "s"(p), [x] "v"(x), [one] "v"(one) : "s4", "s5", "s6", "s7", "scc", "memory");
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + t + (u + u2) * 0.0f;
}

template <int V>
static long run(const float* w, float* out, std::vector<float>& h, int blocks, int iters, long* per_quarter) {
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, w, out, iters);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int wv = 0; wv < blocks * 4; ++wv) {
        const float r = h[wv * 64];
        for (int l = 1; l < 64; ++l)
            if (memcmp(&h[wv * 64 + l], &r, 4)) { ++bad; ++per_quarter[l / 16]; }
    }
    return bad;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 20.0;
    const int blocks = 256 * 12, iters = 1024;
    float *w, *out;
    std::vector<float> hw(8 * 64 + 64), h((size_t)blocks * 256);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.001f * (float)(1 + i % 97);
    hipMalloc(&w, hw.size() * 4);
    hipMalloc(&out, h.size() * 4);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    long bad[9] = {0}, q[9][4] = {{0}}, launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        bad[0] += run<0>(w, out, h, blocks, iters, q[0]);
        bad[1] += run<1>(w, out, h, blocks, iters, q[1]);
        bad[2] += run<2>(w, out, h, blocks, iters, q[2]);
        bad[3] += run<3>(w, out, h, blocks, iters, q[3]);
        bad[4] += run<4>(w, out, h, blocks, iters, q[4]);
        bad[5] += run<5>(w, out, h, blocks, iters, q[5]);
        bad[6] += run<6>(w, out, h, blocks, iters, q[6]);
        bad[7] += run<7>(w, out, h, blocks, iters, q[7]);
        bad[8] += run<8>(w, out, h, blocks, iters, q[8]);
        ++launches;
    }
    const char* names[9] = {"pk_fma, v_mov, SALU write", "pk_fma, SALU write", "pk_fma, SMEM return", "v_fma, SALU write", "pk_fma, s_nop 3, SALU write",
                            "rcp x2, pk_fma, SALU write", "exp x2, pk_fma, SALU write", "rcp x2, pk_fma, s_nop 7, SALU", "rcp x2, SALU write, pk_fma"};
    printf("%ld launches of each form, %d waves x %d iterations per launch\n", launches, blocks * 4, iters);
    for (int v = 0; v < 9; ++v)
        printf("  %-28s lanes that differ from lane 0: %ld   by quarter of the wave: %ld %ld %ld %ld\n", names[v], bad[v], q[v][0], q[v][1], q[v][2], q[v][3]);
    return 0;
}
