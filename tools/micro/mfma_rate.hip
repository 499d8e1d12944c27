// Microbenchmark: issue rate of f32 MFMAs on gfx950, dependent chain vs independent accumulators, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int KIND>
__global__ void k_rate(float *out, unsigned long long *cyc, int iters) {
    f32x16 acc[NACC];
    f32x4 acc4[NACC];
    for (int a = 0; a < NACC; ++a) {
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        for (int r = 0; r < 4; ++r) acc4[a][r] = 0.f;
    }
    float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                if (KIND == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
                else acc4[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc4[a], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) {
        for (int r = 0; r < 16; ++r) s += acc[a][r];
        for (int r = 0; r < 4; ++r) s += acc4[a][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int KIND>
void run(const char *name, int threads, int blocks) {
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipMalloc(&cyc, blocks * 8);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_rate<NACC, KIND><<<blocks, threads>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate<NACC, KIND><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double n_mfma = (double)iters * 8 * NACC;
    double flops = (KIND == 0 ? 4096.0 : 2048.0) * n_mfma * (threads / 64) * blocks;
    printf("%-28s threads %4d blocks %5d: %.3f ms, ticks/MFMA (wave 0) %.1f, ns/MFMA/wave %.2f, %.1f TFLOP/s\n", name, threads, blocks, ms,
           (double)h[0] / n_mfma, ms * 1e6 / n_mfma, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    // one wave per SIMD: 256 threads per block, 256 blocks (one per CU)
    run<1, 0>("32x32x2 dependent", 256, 256);
    run<2, 0>("32x32x2 2 accumulators", 256, 256);
    run<4, 0>("32x32x2 4 accumulators", 256, 256);
    run<1, 0>("32x32x2 dep, 2 waves/SIMD", 512, 256);
    run<1, 0>("32x32x2 dep, 4 waves/SIMD", 1024, 256);
    run<2, 0>("32x32x2 2acc, 2 waves/SIMD", 512, 256);
    run<1, 1>("16x16x4 dependent", 256, 256);
    run<2, 1>("16x16x4 2 accumulators", 256, 256);
    run<4, 1>("16x16x4 4 accumulators", 256, 256);
    run<4, 1>("16x16x4 4acc 2 waves/SIMD", 512, 256);
    return 0;
}
