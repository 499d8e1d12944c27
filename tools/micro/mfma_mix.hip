// Microbenchmark: throughput of "pair tiles" (MFMA chain + the VALU/LDS work of one tile of the fused kernel's sweep) with
// 1 and 2 waves per SIMD, for the 32x32x2 and the 16x16x4 f32 MFMA shapes (same flops per tile).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k_mix(float *out, unsigned long long *cyc, int tiles) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 0.001f;
    __syncthreads();
    float P[16], S[16], w[16];
    for (int s = 0; s < 16; ++s) { P[s] = lane * 0.01f + s; S[s] = 0.f; w[s] = 1.f + 0.001f * s; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int row = lane;
    for (int t = 0; t < tiles; ++t) {
        float r[16], g[16];
        const f32x4 *pr = reinterpret_cast<const f32x4 *>(lds + ((t * 36) & 4095));           // broadcast row
        const f32x4 *pg = reinterpret_cast<const f32x4 *>(lds + 4096 + ((row * 36) & 4095 & ~3));  // per-lane row
        for (int q = 0; q < 4; ++q) {
            f32x4 a = pr[q], b = pg[q];
            r[4 * q] = a[0]; r[4 * q + 1] = a[1]; r[4 * q + 2] = a[2]; r[4 * q + 3] = a[3];
            g[4 * q] = b[0]; g[4 * q + 1] = b[1]; g[4 * q + 2] = b[2]; g[4 * q + 3] = b[3];
        }
        row = (row + 7) & 63;
        float z[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) z[s] = fmaxf((P[s] + r[s]) + g[s], 0.f);
        if (SHAPE == 0) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.5f;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s], z[s], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 16; ++q) S[q] += fmaxf(acc[q], 0.f);
        } else {
            f32x4 acc[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.5f, 0.5f, 0.5f, 0.5f};
#pragma unroll
            for (int s = 0; s < 8; ++s) {          // 8 K-steps of 4, 4 output blocks: 32 MFMAs of 8 passes
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s + 8 * (b >> 1)], z[s + 8 * (b & 1)], acc[b], 0, 0, 0);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) S[4 * b + q] += fmaxf(acc[b][q], 0.f);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float res = 0.f;
    for (int s = 0; s < 16; ++s) res += S[s];
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int SHAPE>
void run(const char *name, int threads) {
    const int blocks = 256, tiles = 2000;
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, (size_t)blocks * 512 * 4);
    (void)hipMalloc(&cyc, blocks * 8 * 8);
    k_mix<SHAPE><<<blocks, threads>>>(out, cyc, 10);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k_mix<SHAPE><<<blocks, threads>>>(out, cyc, tiles);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
    const double waves_per_simd = threads / 256.0;
    printf("%-26s %d wave(s)/SIMD: %.0f ticks per tile per wave, %.0f ticks of SIMD time per tile, %.1f TFLOP/s\n", name,
           (int)waves_per_simd, (double)h[0] / tiles, (double)h[0] / tiles / waves_per_simd,
           65536.0 * tiles * (threads / 64) * blocks / (ms * 1e-3) / 1e12);
}

int main() {
    run<0>("32x32x2 tile", 256);
    run<0>("32x32x2 tile", 512);
    run<1>("16x16x4 tile", 256);
    run<1>("16x16x4 tile", 512);
    return 0;
}
