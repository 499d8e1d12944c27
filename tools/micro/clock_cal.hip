// Calibration: what s_memtime and s_memrealtime count against the host's hipEvent clock, on an otherwise idle chip (one
// wavefront spinning) and with every SIMD busy with f32 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_spin(unsigned long long *o, unsigned long long rt_ticks, int mfma) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r = r0;
    f32x4 acc = {0, 0, 0, 0};
    float a = threadIdx.x * 0.001f;
    while (r - r0 < rt_ticks) {
        if (mfma)
            for (int i = 0; i < 64; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, 0.5f, acc, 0, 0, 0);
        r = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { o[0] = t1 - t0; o[1] = r - r0; }
    if (acc[0] == 12345.f) o[2] = 1;
}
int main() {
    unsigned long long *o, h[3];
    (void)hipMalloc(&o, 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) k_spin<<<1, 64>>>(o, 2000000ull, 0);
            else k_spin<<<256, 512>>>(o, rep == 0 ? 2000000ull : 50000000ull, 1);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
            printf("%s: hipEvent %.3f ms; s_memrealtime %llu ticks = %.2f MHz; s_memtime %llu ticks = %.1f MHz\n", mode ? "all SIMDs on f32 MFMAs" : "one idle-chip wavefront",
                   ms, h[1], h[1] / (ms * 1e3), h[0], h[0] / (ms * 1e3));
        }
    return 0;
}
