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
#include <cstring>
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
// The same decomposition with round-to-nearest pieces: v_cvt_pk_bf16_f32 rounds and packs two values in one instruction and
// v_dot2c_f32_bf16 takes a remainder straight from the packed pair (x - 1 * piece_lo - 0 * piece_hi: exact, the result is representable),
// so no piece is ever expanded to float32 again: 7 instructions per two values instead of 11.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{lo, hi}, bf16x2));
}
__device__ __forceinline__ float rem_lo(unsigned pk, float x) { return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), __builtin_bit_cast(bf16x2, 0x8000bf80u), x, false); }
__device__ __forceinline__ float rem_hi(unsigned pk, float x) { return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), __builtin_bit_cast(bf16x2, 0xbf800000u), x, false); }
__device__ __forceinline__ void split3r(const float (&v)[8], u32x4 &p1, u32x4 &p2, u32x4 &p3) {
    float a[8], b[8];       // level by level: no instruction reads the result of the one in front of it
#pragma unroll
    for (int j = 0; j < 4; ++j) p1[j] = cvt_pk(v[2 * j], v[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[2 * j] = rem_lo(p1[j], v[2 * j]); a[2 * j + 1] = rem_hi(p1[j], v[2 * j + 1]); }
#pragma unroll
    for (int j = 0; j < 4; ++j) p2[j] = cvt_pk(a[2 * j], a[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) { b[2 * j] = rem_lo(p2[j], a[2 * j]); b[2 * j + 1] = rem_hi(p2[j], a[2 * j + 1]); }
#pragma unroll
    for (int j = 0; j < 4; ++j) p3[j] = cvt_pk(b[2 * j], b[2 * j + 1]);
}
// Third form: the remainders on the MATRIX pipe.  With the pieces packed as a B operand (lane 16 q + m: column m, K slots 8 q + s =
// its values v[s]) and the values themselves as the accumulator (rows 4 q + r of row block rb = v[4 rb + r]), D = C + A B with
// A[i][k] = -1 for k = 8 (i / 4) + 4 rb + i % 4 is x - x1 in place: 2 MFMAs per level and 8 values instead of 8 v_and + 8 v_sub.
// (-1) * piece is exact, the other 31 products are zeros, the result is representable: exact if the pipe adds C unrounded.
__device__ __forceinline__ f32x4 mm(u32x4 a, u32x4 b, f32x4 c);
__device__ __forceinline__ void ident_operand(u32x4 (&A)[2]) {
    const int lane = threadIdx.x & 63, qa = lane >> 4, ma = lane & 15;
    for (int rb = 0; rb < 2; ++rb)
        for (int d = 0; d < 4; ++d)
            A[rb][d] = (qa == (ma >> 2) && d == 2 * rb + ((ma & 3) >> 1)) ? (0xbf80u << (16 * (ma & 1))) : 0u;
}
__device__ __forceinline__ u32x4 pack_hi(const f32x4 &lo, const f32x4 &hi) {
    u32x4 p;
    p[0] = __builtin_amdgcn_perm(__float_as_uint(lo[1]), __float_as_uint(lo[0]), 0x07060302u);
    p[1] = __builtin_amdgcn_perm(__float_as_uint(lo[3]), __float_as_uint(lo[2]), 0x07060302u);
    p[2] = __builtin_amdgcn_perm(__float_as_uint(hi[1]), __float_as_uint(hi[0]), 0x07060302u);
    p[3] = __builtin_amdgcn_perm(__float_as_uint(hi[3]), __float_as_uint(hi[2]), 0x07060302u);
    return p;
}
__device__ __forceinline__ void split3m(const float (&v)[8], const u32x4 (&A)[2], u32x4 &p1, u32x4 &p2, u32x4 &p3) {
    f32x4 x0 = {v[0], v[1], v[2], v[3]}, x1 = {v[4], v[5], v[6], v[7]};
    p1 = pack_hi(x0, x1);
    x0 = mm(A[0], p1, x0); x1 = mm(A[1], p1, x1);
    p2 = pack_hi(x0, x1);
    x0 = mm(A[0], p2, x0); x1 = mm(A[1], p2, x1);
    p3 = pack_hi(x0, x1);
}
// Fourth form: the same remainders with v_mfma_f32_16x16x16_bf16 (K = 16: lane 16 q + m holds K slots 4 q .. 4 q + 3; a row block's
// four values are two dwords of the packed piece, the -I operand is the same for both row blocks)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 mm16(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split3k(const float (&v)[8], u32x4 &p1, u32x4 &p2, u32x4 &p3) {
    const int lane = threadIdx.x & 63, qa = lane >> 4, ma = lane & 15;
    u32x2 I;
    for (int d = 0; d < 2; ++d) I[d] = (qa == (ma >> 2) && d == ((ma & 3) >> 1)) ? (0xbf80u << (16 * (ma & 1))) : 0u;
    f32x4 x0 = {v[0], v[1], v[2], v[3]}, x1 = {v[4], v[5], v[6], v[7]};
    p1 = pack_hi(x0, x1);
    x0 = mm16(I, u32x2{p1[0], p1[1]}, x0); x1 = mm16(I, u32x2{p1[2], p1[3]}, x1);
    p2 = pack_hi(x0, x1);
    x0 = mm16(I, u32x2{p2[0], p2[1]}, x0); x1 = mm16(I, u32x2{p2[2], p2[3]}, x1);
    p3 = pack_hi(x0, x1);
}
template <int SPLIT> __device__ __forceinline__ void split_any(const float (&v)[8], const u32x4 (&A)[2], u32x4 &p1, u32x4 &p2, u32x4 &p3) {
    if (SPLIT == 3) split3k(v, p1, p2, p3); else if (SPLIT == 2) split3m(v, A, p1, p2, p3); else if (SPLIT) split3r(v, p1, p2, p3); else split3(v, p1, p2, p3);
}
// exactness of a split: every lane splits 8 values, the host adds the pieces in float64
template <int SPLIT> __global__ void k_exact(const float *x, unsigned *pieces) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    float v[8];
    for (int s = 0; s < 8; ++s) v[s] = x[t * 8 + s];
    u32x4 p1, p2, p3, A[2];
    ident_operand(A);
    split_any<SPLIT>(v, A, p1, p2, p3);
    for (int j = 0; j < 4; ++j) { pieces[(t * 3 + 0) * 4 + j] = p1[j]; pieces[(t * 3 + 1) * 4 + j] = p2[j]; pieces[(t * 3 + 2) * 4 + j] = p3[j]; }
}
__device__ __forceinline__ f32x4 mm(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// W [16][32] row-major, Z [32][16] (k, n); out C [16][16]
template <int SPLIT> __global__ void k_check(const float *W, const float *Z, float *C6, float *C32) {
    const int lane = threadIdx.x, q = lane >> 4, m = lane & 15;
    float w[8], z[8];
    for (int s = 0; s < 8; ++s) { w[s] = W[m * 32 + 8 * q + s]; z[s] = Z[(8 * q + s) * 16 + m]; }
    u32x4 w1, w2, w3, z1, z2, z3;
    split3(w, w1, w2, w3);                 // the weights' pieces come from the host: truncation in both forms
    u32x4 A[2];
    ident_operand(A);
    split_any<SPLIT>(z, A, z1, z2, z3);
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
template <int MODE, int EXTRA = 0, int SPLIT = 0>
__global__ __launch_bounds__(64, 2) void k_time(const float *src, float *dst, int tiles) {
    const int lane = threadIdx.x;
    float z[8], w32[2][8];
    u32x4 wb[2][3], A[2];
    ident_operand(A);
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
            split_any<SPLIT>(zz, A, z1, z2, z3);
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
    for (int sp = 0; sp < 3; ++sp) {
        if (sp == 2) hipLaunchKernelGGL(k_check<2>, dim3(1), dim3(64), 0, 0, dW, dZ, d6, d32);
        else if (sp) hipLaunchKernelGGL(k_check<1>, dim3(1), dim3(64), 0, 0, dW, dZ, d6, d32);
        else hipLaunchKernelGGL(k_check<0>, dim3(1), dim3(64), 0, 0, dW, dZ, d6, d32);
        hipMemcpy(C6.data(), d6, 1024, hipMemcpyDeviceToHost); hipMemcpy(C32.data(), d32, 1024, hipMemcpyDeviceToHost);
        double e6 = 0, e32 = 0, scale = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double ref = 0;
                for (int k = 0; k < 32; ++k) ref += (double)W[i * 32 + k] * (double)Z[k * 16 + j];
                e6 = fmax(e6, fabs(C6[i * 16 + j] - ref)); e32 = fmax(e32, fabs(C32[i * 16 + j] - ref)); scale = fmax(scale, fabs(ref));
            }
        printf("max |C - float64|: bf16x6 (%s pieces) %.3e, f32 MFMA %.3e (largest |C| %.2f)\n", sp == 2 ? "truncated, remainders by MFMA" : sp ? "nearest, dot2c" : "truncated", e6, e32, scale);
    }
    {   // exactness of both splits on 2^22 values: uniform, tiny, huge, negative, powers of two, bf16 numbers, zeros
        const int nv = 1 << 22;
        std::vector<float> xv(nv);
        for (int i = 0; i < nv; ++i) {
            const int kind = i & 7;
            unsigned bits = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
            float f;
            if (kind < 3) { bits = (bits & 0x807fffffu) | ((unsigned)(100 + rand() % 56) << 23); memcpy(&f, &bits, 4); }       // 2^-27 .. 2^28
            else if (kind == 3) { bits = (bits & 0x807fffffu) | ((unsigned)(1 + rand() % 253) << 23); memcpy(&f, &bits, 4); }   // any normal exponent
            else if (kind == 4) { bits &= 0xffff0000u; bits = (bits & 0x807fffffu) | ((unsigned)(110 + rand() % 30) << 23); memcpy(&f, &bits, 4); }
            else if (kind == 5) { bits |= 0x00007fffu; bits = (bits & 0x807fffffu) | ((unsigned)(110 + rand() % 30) << 23); memcpy(&f, &bits, 4); }   // ties / all ones below
            else if (kind == 6) f = ldexpf(1.f, rand() % 60 - 30) * ((rand() & 1) ? -1.f : 1.f);
            else f = (rand() & 1) ? 0.f : (rand() / (float)RAND_MAX) * 3.f;
            xv[i] = f;
        }
        float *dx; unsigned *dp;
        hipMalloc(&dx, (size_t)nv * 4); hipMalloc(&dp, (size_t)nv / 8 * 12 * 4);
        hipMemcpy(dx, xv.data(), (size_t)nv * 4, hipMemcpyHostToDevice);
        std::vector<unsigned> pc((size_t)nv / 8 * 12), pc0;
        for (int sp = 0; sp < 4; ++sp) {
            if (sp == 3) hipLaunchKernelGGL(k_exact<3>, dim3(nv / 512), dim3(64), 0, 0, dx, dp);
            else if (sp == 2) hipLaunchKernelGGL(k_exact<2>, dim3(nv / 512), dim3(64), 0, 0, dx, dp);
            else if (sp) hipLaunchKernelGGL(k_exact<1>, dim3(nv / 512), dim3(64), 0, 0, dx, dp);
            else hipLaunchKernelGGL(k_exact<0>, dim3(nv / 512), dim3(64), 0, 0, dx, dp);
            hipMemcpy(pc.data(), dp, pc.size() * 4, hipMemcpyDeviceToHost);
            long bad = 0; double worst = 0; int shown = 0;
            for (int t = 0; t < nv / 8; ++t)
                for (int s = 0; s < 8; ++s) {
                    double sum = 0;
                    for (int k = 0; k < 3; ++k) {
                        const unsigned w = pc[((size_t)t * 3 + k) * 4 + s / 2];
                        const unsigned b = (s & 1) ? (w & 0xffff0000u) : (w << 16);
                        float piece; memcpy(&piece, &b, 4);
                        sum += (double)piece;
                    }
                    const double x = xv[(size_t)t * 8 + s];
                    if (sum != x) {
                        ++bad; worst = fmax(worst, fabs(sum - x) / fabs(x));
                        if (shown++ < 4) printf("   x = %.9g (%a): pieces add up to %.9g\n", x, x, sum);
                    }
                }
            if (sp == 0) pc0 = pc;
            if (sp >= 2) {
                long diff = 0;
                for (size_t i = 0; i < pc.size(); ++i) diff += pc[i] != pc0[i];
                printf("truncated / MFMA remainders (K = %d): %ld of %zu piece words differ from the v_and / v_sub form's\n", sp == 2 ? 32 : 16, diff, pc.size());
            }
            printf("%s pieces: %ld of %d values not reproduced exactly by x1 + x2 + x3 (worst relative %.3g)\n", sp == 3 ? "truncated / MFMA remainders, K = 16" : sp == 2 ? "truncated / MFMA remainders" : sp ? "nearest / dot2c" : "truncated", bad, nv, worst);
        }
    }
    const int blocks = 2048, tiles = 4000;
    float *src, *dst;
    hipMalloc(&src, 1536 * 4); hipMalloc(&dst, (size_t)blocks * 512 * 4);
    std::vector<float> s(1536);
    for (auto &v : s) v = rand() / (float)RAND_MAX - 0.3f;
    hipMemcpy(src, s.data(), 1536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[10] = {"f32 MFMA     ", "bf16x6 + split", "  + 8 v_pk_add_f32 (16 adds)", "  + 16 v_add_f32", "  + 16 v_max_f32", "  + 8 v_and_b32 + 8 v_perm_b32", "bf16x6 + split by v_cvt_pk_bf16_f32 / v_dot2c_f32_bf16", "bf16x6 + split with the remainders by MFMA (16 per tile)", "bf16x6 + split with the remainders by v_mfma_f32_16x16x16_bf16", ""};
    for (int mode = 0; mode < 9; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_time<0>, dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            else if (mode == 1) hipLaunchKernelGGL(k_time<1>, dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            else if (mode == 2) hipLaunchKernelGGL((k_time<1, 1>), dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            else if (mode == 3) hipLaunchKernelGGL((k_time<1, 2>), dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            else if (mode == 4) hipLaunchKernelGGL((k_time<1, 3>), dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            else if (mode == 5) hipLaunchKernelGGL((k_time<1, 4>), dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            else if (mode == 6) hipLaunchKernelGGL((k_time<1, 0, 1>), dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            else if (mode == 7) hipLaunchKernelGGL((k_time<1, 0, 2>), dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            else hipLaunchKernelGGL((k_time<1, 0, 3>), dim3(blocks), dim3(64), 0, 0, src, dst, tiles);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // 2048 wavefronts = 2 per SIMD: cycles per tile and SIMD = ms * clock / (tiles * 2)
            if (rep) printf("%s: %.3f ms, %.0f cycles per tile and wavefront pair at 2.4 GHz (%.0f per wavefront)\n", names[mode], ms, ms * 1e-3 * 2.4e9 / tiles, ms * 1e-3 * 2.4e9 / tiles / 2);
        }
    }
    return 0;
}
