// Check of the 4-lane column sum used by the fused kernel (gfx950 v_permlane16_swap / v_permlane32_swap): every lane must
// end up with the sum of lanes n16, 16+n16, 32+n16, 48+n16.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *o) {
    const float v = threadIdx.x;
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    o[threadIdx.x] = __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
int main() {
    float *d, h[64];
    (void)hipMalloc(&d, 256);
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += h[i] != (float)((i & 15) * 4 + 96);
    printf("permlane column sum: %s\n", bad ? "WRONG" : "ok");
    return bad != 0;
}
