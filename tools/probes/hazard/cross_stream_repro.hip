// An ATTEMPT at a standalone form of profiles/r06_reproducibility.txt (8) -- two kernels of ONE process on two streams -- that does NOT reproduce the effect
// (0 of 10 000 relaunches in every row): kept as a record of what is not sufficient.  The effect itself is reproduced with the library's own kernels by
// tools/probes/gather_beside_stream.py.
//   victim  = what is left of bs_logbinom_depth_ex when everything but its LDS traffic is taken out: a block stages the low-resolution window of a
//             16 x 16 output tile (64 + 40 floats per cell) from global memory into dynamic LDS, one barrier, every pixel gathers the four corners of
//             40 values (eight at a time, between lgkmcnt(0) waits), interpolates, sums, stores.  WIDTH 4: 32 ds_read_b32 per eight values; WIDTH 16: 8 ds_read_b128.
//   trigger = bs_rank1_bias as it is in the library: a [G, K] x [K, N] product on v_mfma_f32_16x16x32_bf16, 17 KiB of LDS, 192 short blocks -- small
//             enough to sit on a CU beside three of the victim's 42-KiB blocks.  NOMFMA: the same kernel with a plain add in place of the MFMA.
// Stream A relaunches the victim on untouched inputs and a compare kernel counts the pixels that differ from the first launch's; stream B runs bursts of
// the trigger.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o cross_stream_repro cross_stream_repro.hip ;  ./cross_stream_repro [relaunches]
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int NB = 64, NH = 40, T = 16;

template <int WIDTH>
__global__ __launch_bounds__(256) void victim(const float* Eh, const float* bins, float* out, int H, int W, int He, int We, float sy, float sx, int ncell_max) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_bins = sm + NB;
    float* s_eh = s_bins + ncell_max * NB;
    const int b = blockIdx.z, ty0 = blockIdx.y * T, tx0 = blockIdx.x * T;
    const int ty1 = min(ty0 + T - 1, H - 1), tx1 = min(tx0 + T - 1, W - 1);
    const int sy0 = min((int)(sy * (float)ty0), He - 1), sx0 = min((int)(sx * (float)tx0), We - 1);
    const int sy1 = min((int)(sy * (float)ty1) + 1, He - 1), sx1 = min((int)(sx * (float)tx1) + 1, We - 1);
    const int nr = sy1 - sy0 + 1, nc = sx1 - sx0 + 1;
    for (int i = threadIdx.x; i < nr * nc * (NB / 4); i += 256) {
        const int k4 = i % (NB / 4), cell = i / (NB / 4), rr = cell / nc, cc = cell - rr * nc;
        *reinterpret_cast<f32x4*>(s_bins + cell * NB + k4 * 4) = *reinterpret_cast<const f32x4*>(bins + (((int64_t)b * He + sy0 + rr) * We + sx0 + cc) * NB + k4 * 4);
    }
    for (int i = threadIdx.x; i < nr * nc * (NH / 4); i += 256) {
        const int k4 = i % (NH / 4), cell = i / (NH / 4), rr = cell / nc, cc = cell - rr * nc;
        *reinterpret_cast<f32x4*>(s_eh + cell * NH + k4 * 4) = *reinterpret_cast<const f32x4*>(Eh + (((int64_t)b * He + sy0 + rr) * We + sx0 + cc) * NH + k4 * 4);
    }
    __syncthreads();
    const int oy = ty0 + (threadIdx.x >> 4), ox = tx0 + (threadIdx.x & 15);
    if (oy >= H || ox >= W) return;
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    int y0 = min((int)fy, He - 1), x0 = min((int)fx, We - 1);
    const int y1 = y0 + (y0 < He - 1), x1 = x0 + (x0 < We - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
    const int c00 = (y0 - sy0) * nc + (x0 - sx0), c01 = (y0 - sy0) * nc + (x1 - sx0), c10 = (y1 - sy0) * nc + (x0 - sx0), c11 = (y1 - sy0) * nc + (x1 - sx0);
    float acc = 0.f;
    for (int h0 = 0; h0 < NH; h0 += 8) {
        float e[8];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
        if (WIDTH == 4) {
            float g00[8], g01[8], g10[8], g11[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                g00[k] = *reinterpret_cast<volatile const float*>(s_eh + c00 * NH + h0 + k);
                g01[k] = *reinterpret_cast<volatile const float*>(s_eh + c01 * NH + h0 + k);
                g10[k] = *reinterpret_cast<volatile const float*>(s_eh + c10 * NH + h0 + k);
                g11[k] = *reinterpret_cast<volatile const float*>(s_eh + c11 * NH + h0 + k);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = hy * (hx * g00[k] + lx * g01[k]) + ly * (hx * g10[k] + lx * g11[k]);
        } else {
            f32x4 q00[2], q01[2], q10[2], q11[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                q00[q] = *reinterpret_cast<const f32x4*>(s_eh + c00 * NH + h0 + 4 * q);
                q01[q] = *reinterpret_cast<const f32x4*>(s_eh + c01 * NH + h0 + 4 * q);
                q10[q] = *reinterpret_cast<const f32x4*>(s_eh + c10 * NH + h0 + 4 * q);
                q11[q] = *reinterpret_cast<const f32x4*>(s_eh + c11 * NH + h0 + 4 * q);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) e[4 * q + k] = hy * (hx * q00[q][k] + lx * q01[q][k]) + ly * (hx * q10[q][k] + lx * q11[q][k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += e[k];
    }
    // (the bin centres: read once so that the staging above is not dead)
    acc += s_bins[c00 * NB + (threadIdx.x & 63)] * 1e-3f;
    out[((int64_t)b * H + oy) * W + ox] = acc;
}

template <bool NOMFMA>
__global__ __launch_bounds__(256) void trigger(const __bf16* abar, const __bf16* dW, float* out, int G, int N, int K) {
    __shared__ float red[4][64][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16, g0 = blockIdx.y * 64;
    const int steps = K / 32, per = steps / 4, s0 = wave * per, s1 = wave == 3 ? steps : s0 + per;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __bf16* ap[4];
    for (int i = 0; i < 4; ++i) {
        int g = g0 + i * 16 + frow;
        g = g < G ? g : G - 1;
        ap[i] = abar + (int64_t)g * K + fq * 8;
    }
    int n = n0 + frow;
    n = n < N ? n : N - 1;
    const __bf16* bp = dW + (int64_t)n * K + fq * 8;
    for (int st = s0; st < s1; ++st) {
        const int k = st * 32;
        bf16x8 af[4];
        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(bp + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(ap[i] + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (NOMFMA) acc[i][0] += (float)af[i][0] + (float)bf[0];
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, af[i], acc[i], 0, 0, 0);
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 4; ++e) red[wave][i * 16 + frow][fq * 4 + e] = acc[i][e];
    __syncthreads();
    for (int r = 0; r < 4; ++r) {
        const int idx = r * 256 + threadIdx.x, gl = idx >> 4, c = idx & 15;
        if (g0 + gl < G && n0 + c < N) out[(int64_t)(g0 + gl) * N + n0 + c] += ((red[0][gl][c] + red[1][gl][c]) + red[2][gl][c]) + red[3][gl][c];
    }
}

__global__ void compare(const float* a, const float* b, int n, unsigned long long* cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && __float_as_uint(a[i]) != __float_as_uint(b[i])) atomicAdd(cnt, 1ull);
}

template <int WIDTH, int TRIG>       // TRIG: 0 none, 1 the trigger, 2 the trigger without its MFMA
static void run(const char* label, int relaunches, const float* Eh, const float* bins, float* out, float* base, const __bf16* abar, const __bf16* dW, float* r1out,
                unsigned long long* cnt, hipStream_t sA, hipStream_t sB) {
    const int B = 8, H = 96, W = 128, He = 48, We = 64;
    const float sy = (float)(He - 1) / (float)(H - 1), sx = (float)(We - 1) / (float)(W - 1);
    const int ncell = ((int)(sy * (T - 1)) + 3) * ((int)(sx * (T - 1)) + 3);
    const size_t lds = sizeof(float) * (size_t)(NB + ncell * (NB + NH));
    const dim3 grid(W / T, H / T, B);
    hipLaunchKernelGGL(victim<WIDTH>, grid, dim3(256), lds, sA, Eh, bins, base, H, W, He, We, sy, sx, ncell);
    (void)hipDeviceSynchronize();
    unsigned long long bad_launches = 0, bad_pixels = 0;
    for (int done = 0; done < relaunches; done += 100) {
        if (TRIG)
            for (int r = 0; r < 20 * 24; ++r) {
                if (TRIG == 1) hipLaunchKernelGGL(trigger<false>, dim3(192, 1), dim3(256), 0, sB, abar, dW, r1out, 16, 3072, 1024);
                else hipLaunchKernelGGL(trigger<true>, dim3(192, 1), dim3(256), 0, sB, abar, dW, r1out, 16, 3072, 1024);
            }
        (void)hipMemsetAsync(cnt, 0, 8 * 100, sA);
        for (int r = 0; r < 100; ++r) {       // queued back to back: the host is not in the loop
            hipLaunchKernelGGL(victim<WIDTH>, grid, dim3(256), lds, sA, Eh, bins, out, H, W, He, We, sy, sx, ncell);
            hipLaunchKernelGGL(compare, dim3((B * H * W + 255) / 256), dim3(256), 0, sA, out, base, B * H * W, cnt + r);
        }
        unsigned long long h[100];
        (void)hipMemcpyAsync(h, cnt, 8 * 100, hipMemcpyDeviceToHost, sA);
        (void)hipStreamSynchronize(sA);
        for (int r = 0; r < 100; ++r) { bad_launches += h[r] != 0; bad_pixels += h[r]; }
        (void)hipDeviceSynchronize();
    }
    printf("  %-58s %llu of %d relaunches differ from the first (%llu pixels)\n", label, bad_launches, relaunches, bad_pixels);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 5000;
    const int B = 8, H = 96, W = 128, He = 48, We = 64;
    std::vector<float> hE((size_t)B * He * We * NH), hB((size_t)B * He * We * NB);
    srand(1);
    for (auto& v : hE) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hB) v = (float)rand() / RAND_MAX;
    std::vector<__bf16> hA(16 * 1024), hW((size_t)3072 * 1024);
    for (auto& v : hA) v = (__bf16)((float)rand() / RAND_MAX - 0.5f);
    for (auto& v : hW) v = (__bf16)(((float)rand() / RAND_MAX - 0.5f) * 0.01f);
    float *Eh, *bins, *out, *base, *r1out;
    __bf16 *abar, *dW;
    unsigned long long* cnt;
    (void)hipMalloc(&Eh, hE.size() * 4); (void)hipMalloc(&bins, hB.size() * 4); (void)hipMalloc(&out, (size_t)B * H * W * 4); (void)hipMalloc(&base, (size_t)B * H * W * 4);
    (void)hipMalloc(&abar, hA.size() * 2); (void)hipMalloc(&dW, hW.size() * 2); (void)hipMalloc(&r1out, 16 * 3072 * 4); (void)hipMalloc(&cnt, 8 * 100);
    (void)hipMemcpy(Eh, hE.data(), hE.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(bins, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(abar, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemset(r1out, 0, 16 * 3072 * 4);
    hipStream_t sA, sB;
    (void)hipStreamCreate(&sA); (void)hipStreamCreate(&sB);
    printf("victim: 8 images of 96 x 128 pixels from 48 x 64 cells, 16 x 16 tiles (384 blocks x 4 waves); trigger bursts: 480 launches of 192 blocks on a second stream\n");
    run<4, 0>("4-byte gathers, alone", n, Eh, bins, out, base, abar, dW, r1out, cnt, sA, sB);
    run<4, 1>("4-byte gathers, beside the MFMA kernel", n, Eh, bins, out, base, abar, dW, r1out, cnt, sA, sB);
    run<4, 2>("4-byte gathers, beside that kernel without its MFMA", n, Eh, bins, out, base, abar, dW, r1out, cnt, sA, sB);
    run<16, 1>("16-byte reads, beside the MFMA kernel", n, Eh, bins, out, base, abar, dW, r1out, cnt, sA, sB);
    run<4, 1>("4-byte gathers, beside the MFMA kernel (again)", n, Eh, bins, out, base, abar, dW, r1out, cnt, sA, sB);
    return 0;
}
