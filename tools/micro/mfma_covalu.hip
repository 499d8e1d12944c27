// Microbenchmark: wave A runs a dependent MFMA chain, wave B (same SIMD) runs VALU / LDS work: do they overlap? (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE bit0: MFMA waves active, bit1: VALU waves active.  Block = 512 threads = 8 waves: waves 0-3 MFMA role, 4-7 VALU role
// (wave w and w+4 share a SIMD).
template <int KIND>
__global__ __launch_bounds__(512) void k_co(float *out, unsigned long long *cyc, int iters, int mode, float p0, float p1) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mfma_role = wave < 4;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (mode & 32) { if (mfma_role) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3); }
    if (mode & 64) { if (mfma_role) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float res = 0.f;
    if (mfma_role) {
        if (mode & 1) {
            f32x16 acc;
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            float x = lane * 0.001f, y = 1.f + lane * 0.002f;
            if (mode & 8) {       // 16x16x4 shape (8 passes)
                typedef float f32x4v __attribute__((ext_vector_type(4)));
                f32x4v a4 = {0.f, 0.f, 0.f, 0.f};
                for (int i = 0; i < iters; ++i) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4, 0, 0, 0);
                }
                res += a4[0] + a4[1] + a4[2] + a4[3];
            } else if (mode & 16) {   // 32x32x1 two blocks (16 passes)
                typedef float f32x32v __attribute__((ext_vector_type(32)));
                f32x32v a32;
                for (int r = 0; r < 32; ++r) a32[r] = 0.f;
                for (int i = 0; i < iters; ++i) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) a32 = __builtin_amdgcn_mfma_f32_32x32x1f32(x, y, a32, 0, 0, 0);
                }
                for (int r = 0; r < 32; ++r) res += a32[r];
            } else if (mode & 4) {       // accumulators in AGPRs
                for (int i = 0; i < iters; ++i) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y));
                }
            } else {
                for (int i = 0; i < iters; ++i) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
                }
            }
            for (int r = 0; r < 16; ++r) res += acc[r];
        }
    } else {
        if (mode & 2) {
            if (KIND == 0) {          // 4 independent VALU chains
                float a = lane, b = lane + 1, c = lane + 2, d = lane + 3;
                for (int i = 0; i < iters; ++i) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) {
                        a = fmaxf(a + p0, p1); b = fmaxf(b + p0, p1); c = fmaxf(c + p0, p1); d = fmaxf(d + p0, p1);
                    }
                }
                res = a + b + c + d;
            } else {                  // LDS reads
                float a = 0.f;
                int idx = lane;
                for (int i = 0; i < iters; ++i) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) { a += lds[idx]; idx = (idx + 67) & 4095; }
                }
                res = a;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char *name, int mode) {
    const int blocks = 256, iters = 400;
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, (size_t)blocks * 512 * 4);
    (void)hipMalloc(&cyc, blocks * 8 * 8);
    k_co<KIND><<<blocks, 512>>>(out, cyc, 10, mode, 0.5f, 0.f);
    (void)hipDeviceSynchronize();
    k_co<KIND><<<blocks, 512>>>(out, cyc, iters, mode, 0.5f, 0.f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
    printf("%-34s mfma wave: %.1f ticks/MFMA   other wave: %.2f ticks/op\n", name, (double)h[0] / (iters * 16.0),
           (double)h[4] / (iters * (KIND == 0 ? 256.0 : 32.0)));
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main() {
    run<0>("MFMA alone", 1);
    run<0>("VALU alone (8 ops x 32)", 2);
    run<0>("MFMA + VALU on the same SIMD", 3);
    run<1>("LDS reads alone", 2);
    run<1>("MFMA + LDS reads on the same SIMD", 3);
    run<0>("MFMA prio0 + VALU prio3", 3 + 32);
    run<1>("MFMA prio0 + LDS prio3", 3 + 32);
    run<0>("MFMA prio3 + VALU prio0", 3 + 64);
    run<0>("16x16x4 alone", 9);
    run<0>("16x16x4 + VALU", 11);
    run<1>("16x16x4 + LDS reads", 11);
    run<0>("32x32x1(2 blocks) alone", 17);
    run<0>("32x32x1(2 blocks) + VALU", 19);
    run<0>("MFMA(AGPR acc) alone", 5);
    run<0>("MFMA(AGPR acc) + VALU", 7);
    run<1>("MFMA(AGPR acc) + LDS reads", 7);
    return 0;
}
