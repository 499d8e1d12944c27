// Microbenchmark: does VALU work between dependent MFMAs overlap with the matrix pipe? (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// NV = VALU ops per MFMA; FEED = 1: the VALU result is the MFMA's B operand, 0: independent side chain
template <int NV, int FEED, int NACC>
__global__ void k_mix(float *out, unsigned long long *cyc, int iters, float p0, float p1) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float z[16], side = threadIdx.x;
    for (int s = 0; s < 16; ++s) z[s] = threadIdx.x * 0.001f + s;
    float w = 1.0f + threadIdx.x * 0.002f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                float b = z[s];
                if (FEED) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) b = fmaxf(b + p0, p1);     // 2 VALU each
                } else {
#pragma unroll
                    for (int v = 0; v < NV; ++v) side = fmaxf(side + p0, p1);
                }
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, acc[a], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sres = side;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) sres += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sres;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int FEED, int NACC>
void run(const char *name, int threads) {
    const int blocks = 256, iters = 500;
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, (size_t)blocks * threads * 4);
    (void)hipMalloc(&cyc, blocks * 8);
    k_mix<NV, FEED, NACC><<<blocks, threads>>>(out, cyc, 10, 0.5f, 0.f);
    (void)hipDeviceSynchronize();
    k_mix<NV, FEED, NACC><<<blocks, threads>>>(out, cyc, iters, 0.5f, 0.f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    printf("%-40s threads %4d: ticks per MFMA %.1f\n", name, threads, (double)h[0] / (iters * 16.0 * NACC));
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main() {
    run<0, 1, 1>("dependent, no VALU", 256);
    run<1, 1, 1>("dependent, 2 VALU feeding B", 256);
    run<2, 1, 1>("dependent, 4 VALU feeding B", 256);
    run<4, 1, 1>("dependent, 8 VALU feeding B", 256);
    run<8, 1, 1>("dependent, 16 VALU feeding B", 256);
    run<2, 0, 1>("dependent, 4 VALU independent", 256);
    run<4, 0, 1>("dependent, 8 VALU independent", 256);
    run<8, 0, 1>("dependent, 16 VALU independent", 256);
    run<2, 1, 2>("2 accumulators, 4 VALU feeding B", 256);
    run<4, 1, 2>("2 accumulators, 8 VALU feeding B", 256);
    run<2, 1, 1>("dependent, 4 VALU feeding, 2 waves/SIMD", 512);
    run<4, 1, 1>("dependent, 8 VALU feeding, 2 waves/SIMD", 512);
    return 0;
}
