// f32-grade products on the bf16 matrix pipe: C[16x16] = W[16x32] * Z[32x16] with both operands split EXACTLY into three bf16 pieces
// (truncation: x = x1 + x2 + x3, 8 + 8 + 8 mantissa bits) and the six largest of the nine partial products summed in f32
// (v_mfma_f32_16x16x32_bf16, 16 cycles each on gfx950 against 8 x 32 cycles of v_mfma_f32_16x16x4_f32 for the same K = 32).
// Checks the operand layout (lane 16q + m: row / column m, K slots 8q .. 8q + 7 as four dwords of bf16 pairs) and the accuracy
// against float64 and against the f32 MFMA, then times both forms with the split's VALU work included.
//   hipcc --offload-arch=gfx950 -O3 -o bf16x6 bf16x6.hip && ./bf16x6
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const float (&v)[8], u32x4 &p1, u32x4 &p2, u32x4 &p3) {
    float r1[8], r2[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        r1[s] = v[s] - __uint_as_float(__float_as_uint(v[s]) & 0xffff0000u);
        r2[s] = r1[s] - __uint_as_float(__float_as_uint(r1[s]) & 0xffff0000u);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        p1[j] = __builtin_amdgcn_perm(__float_as_uint(v[2 * j + 1]), __float_as_uint(v[2 * j]), 0x07060302u);
        p2[j] = __builtin_amdgcn_perm(__float_as_uint(r1[2 * j + 1]), __float_as_uint(r1[2 * j]), 0x07060302u);
        p3[j] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * j + 1]), __float_as_uint(r2[2 * j]), 0x07060302u);
    }
}
__device__ __forceinline__ f32x4 mm(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// W [16][32] row-major, Z [32][16] (k, n); out C [16][16]
__global__ void k_check(const float *W, const float *Z, float *C6, float *C32) {
    const int lane = threadIdx.x, q = lane >> 4, m = lane & 15;
    float w[8], z[8];
    for (int s = 0; s < 8; ++s) { w[s] = W[m * 32 + 8 * q + s]; z[s] = Z[(8 * q + s) * 16 + m]; }
    u32x4 w1, w2, w3, z1, z2, z3;
    split3(w, w1, w2, w3);
    split3(z, z1, z2, z3);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = mm(w1, z3, acc); acc = mm(w2, z2, acc); acc = mm(w3, z1, acc);
    acc = mm(w1, z2, acc); acc = mm(w2, z1, acc); acc = mm(w1, z1, acc);
    for (int r = 0; r < 4; ++r) C6[(4 * q + r) * 16 + m] = acc[r];
    f32x4 a32 = {0.f, 0.f, 0.f, 0.f};
    for (int k4 = 0; k4 < 8; ++k4)      // f32 MFMA: K = 4 per instruction, lane q supplies k = 4 k4 + q
        a32 = __builtin_amdgcn_mfma_f32_16x16x4f32(W[m * 32 + 4 * k4 + q], Z[(4 * k4 + q) * 16 + m], a32, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C32[(4 * q + r) * 16 + m] = a32[r];
}
// timing: `tiles` dependent-free tiles per wave; MODE 0 = f32 MFMA (16 per tile: 2 row blocks x 8), 1 = bf16x6 incl. the split of z
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE, int EXTRA = 0>
__global__ __launch_bounds__(64, 2) void k_time(const float *src, float *dst, int tiles) {
    const int lane = threadIdx.x;
    float z[8], w32[2][8];
    u32x4 wb[2][3];
    for (int s = 0; s < 8; ++s) { z[s] = src[lane * 8 + s]; w32[0][s] = src[512 + lane * 8 + s]; w32[1][s] = src[1024 + lane * 8 + s]; }
    for (int rb = 0; rb < 2; ++rb) split3(w32[rb], wb[rb][0], wb[rb][1], wb[rb][2]);
    f32x4 S[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int t = 0; t < tiles; ++t) {
        float zz[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) zz[s] = fmaxf(z[s] + S[s >> 2][s & 3] * 1e-9f, 0.f);     // (a dependence on the last tile, like S += relu(acc))
        if (EXTRA == 1) {
#pragma unroll
            for (int s = 0; s < 8; s += 2) {
                f32x2 a = {zz[s], zz[s + 1]}, b = {z[s], z[s + 1]}, c;
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(c) : "v"(a), "v"(b));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(c), "v"(b));
                zz[s] = a[0]; zz[s + 1] = a[1];
            }
        } else if (EXTRA == 2) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float c;
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(c) : "v"(zz[s]), "v"(z[s]));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(zz[s]) : "v"(c), "v"(z[s]));
            }
        } else if (EXTRA == 3) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float c;
                asm volatile("v_max_f32 %0, %1, %2" : "=v"(c) : "v"(zz[s]), "v"(z[s]));
                asm volatile("v_max_f32 %0, %1, %2" : "=v"(zz[s]) : "v"(c), "v"(z[s]));
            }
        } else if (EXTRA == 4) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float c;
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(c) : "v"(zz[s]), "v"(z[s]));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(zz[s]) : "v"(c), "v"(zz[s]), "v"(0x07060302u));
            }
        }
        f32x4 d[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        if (MODE == 0) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int s = 0; s < 8; ++s) d[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w32[rb][s], zz[s], d[rb], 0, 0, 0);
        } else {
            u32x4 z1, z2, z3;
            split3(zz, z1, z2, z3);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                d[rb] = mm(wb[rb][0], z3, d[rb]); d[rb] = mm(wb[rb][1], z2, d[rb]); d[rb] = mm(wb[rb][2], z1, d[rb]);
                d[rb] = mm(wb[rb][0], z2, d[rb]); d[rb] = mm(wb[rb][1], z1, d[rb]); d[rb] = mm(wb[rb][0], z1, d[rb]);
            }
        }
        S[0] += d[0]; S[1] += d[1];
    }
    for (int r = 0; r < 4; ++r) dst[(blockIdx.x * 64 + lane) * 8 + r] = S[0][r], dst[(blockIdx.x * 64 + lane) * 8 + 4 + r] = S[1][r];
}
int main() {
    std::vector<float> W(16 * 32), Z(32 * 16), C6(256), C32(256);
    srand(1);
    for (auto &v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    for (auto &v : Z) v = (rand() / (float)RAND_MAX) * 3.f * ((rand() & 3) ? 1.f : 0.f);       // relu-like: a quarter zeros
    float *dW, *dZ, *d6, *d32;
    hipMalloc(&dW, 2048); hipMalloc(&dZ, 2048); hipMalloc(&d6, 1024); hipMalloc(&d32, 1024);
    hipMemcpy(dW, W.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dZ, Z.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dW, dZ, d6, d32);
    hipMemcpy(C6.data(), d6, 1024, hipMemcpyDeviceToHost); hipMemcpy(C32.data(), d32, 1024, hipMemcpyDeviceToHost);
    double e6 = 0, e32 = 0, scale = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0;
            for (int k = 0; k < 32; ++k) ref += (double)W[i * 32 + k] * (double)Z[k * 16 + j];
            e6 = fmax(e6, fabs(C6[i * 16 + j] - ref)); e32 = fmax(e32, fabs(C32[i * 16 + j] - ref)); scale = fmax(scale, fabs(ref));
        }
    printf("max |C - float64|: bf16x6 %.3e, f32 MFMA %.3e (largest |C| %.2f)\n", e6, e32, scale);
    const int blocks = 2048, tiles = 4000;
    float *src, *dst;
    hipMalloc(&src, 1536 * 4); hipMalloc(&dst, (size_t)blocks * 512 * 4);
    std::vector<float> s(1536);
    for (auto &v : s) v = rand() / (float)RAND_MAX - 0.3f;
    hipMemcpy(src, s.data(), 1536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[7] = {"f32 MFMA     ", "bf16x6 + split", "  + 8 v_pk_add_f32 (16 adds)", "  + 16 v_add_f32", "  + 16 v_max_f32", "  + 8 v_and_b32 + 8 v_perm_b32", ""};
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_time<0>, dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            else if (mode == 1) hipLaunchKernelGGL(k_time<1>, dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            else if (mode == 2) hipLaunchKernelGGL((k_time<1, 1>), dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            else if (mode == 3) hipLaunchKernelGGL((k_time<1, 2>), dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            else if (mode == 4) hipLaunchKernelGGL((k_time<1, 3>), dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            else hipLaunchKernelGGL((k_time<1, 4>), dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // 2048 wavefronts = 2 per SIMD: cycles per tile and SIMD = ms * clock / (tiles * 2)
            if (rep) printf("%s: %.3f ms, %.0f cycles per tile and wavefront pair at 2.4 GHz (%.0f per wavefront)\n", names[mode], ms, ms * 1e-3 * 2.4e9 / tiles, ms * 1e-3 * 2.4e9 / tiles / 2);
        }
    }
    return 0;
}
