// Microbenchmark + semantics check: ReLU through the VOP3P output clamp.
//   v_pk_add_f32 d, a, b clamp   ->  d = min(max(a + b, 0), 1) on both halves: with every operand pre-scaled by a power of two
//   that bounds the sum (exact in binary floating point) this is "add + ReLU" for two elements in ONE VALU instruction.
// Part 1 checks the clamp on a set of values, part 2 times one 16-column pair tile of the fused kernel's sweep
// (16 MFMAs 16x16x4 + its element-wise work) in three forms: max-based (round 1-3), clamp-based, clamp-based without G.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pk_add_clamp(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { return a + b; }
// scalar forms, one instruction per element (inline asm so that the compiler cannot pack them)
__device__ __forceinline__ float s_add(float a, float b) { float d; asm("v_add_f32_e32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float s_addc(float a, float b) { float d; asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float s_max0(float a) { float d; asm("v_max_f32_e32 %0, 0, %1" : "=v"(d) : "v"(a)); return d; }

__global__ void k_sem(const float *a, const float *b, float *o, int n) {
    const int i = threadIdx.x;
    if (2 * i + 1 < n) {
        f32x2 d = pk_add_clamp(f32x2{a[2 * i], a[2 * i + 1]}, f32x2{b[2 * i], b[2 * i + 1]});
        o[2 * i] = d[0];
        o[2 * i + 1] = d[1];
    }
}

template <int FORM>
__global__ __launch_bounds__(512) void k_tile(float *out, unsigned long long *cyc, int tiles, int lds_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];        // >= 4096 floats are used (two regions of 2048)
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) lds[i] = (i % 97) * 0.001f;
    __syncthreads();
    f32x2 P[4], S[4];
    float w[2][8];
    for (int s = 0; s < 4; ++s) { P[s] = f32x2{lane * 0.001f + s * 0.01f, 0.02f}; S[s] = f32x2{0.f, 0.f}; }
    for (int s = 0; s < 8; ++s) { w[0][s] = 0.01f + 0.001f * s; w[1][s] = 0.02f - 0.001f * s; }
    const f32x4 bias = f32x4{0.01f, 0.01f, 0.01f, 0.01f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int row = lane;
    for (int t = 0; t < tiles; ++t) {
        const f32x4 *pr = reinterpret_cast<const f32x4 *>(lds + ((t * 36) & 1023) + 4 * (lane >> 4));
        const f32x4 *pg = reinterpret_cast<const f32x4 *>(lds + 2048 + ((row * 36) & 1023 & ~3) + 4 * (lane >> 4));
        const f32x4 r0 = pr[0], r1 = pr[4], g0 = pg[0], g1 = pg[4];
        row = (row + 7) & 63;
        f32x2 z[4];
        const f32x2 r[4] = {f32x2{r0[0], r0[1]}, f32x2{r0[2], r0[3]}, f32x2{r1[0], r1[1]}, f32x2{r1[2], r1[3]}};
        const f32x2 g[4] = {f32x2{g0[0], g0[1]}, f32x2{g0[2], g0[3]}, f32x2{g1[0], g1[1]}, f32x2{g1[2], g1[3]}};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (FORM == 0) {
                const f32x2 v = (P[s] + r[s]) + g[s];
                z[s] = f32x2{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)};
            } else if (FORM == 1) {
                z[s] = pk_add_clamp(P[s] + r[s], g[s]);
            } else if (FORM == 2) {
                z[s] = pk_add_clamp(P[s], r[s]);
            } else if (FORM == 3) {
                z[s] = f32x2{s_addc(s_add(P[s][0], r[s][0]), g[s][0]), s_addc(s_add(P[s][1], r[s][1]), g[s][1])};
            } else {
                z[s] = f32x2{s_max0(s_add(s_add(P[s][0], r[s][0]), g[s][0])), s_max0(s_add(s_add(P[s][1], r[s][1]), g[s][1]))};
            }
        }
        f32x4 acc[2] = {bias, bias};
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int s = 0; s < 8; ++s) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[rb][s], z[s >> 1][s & 1], acc[rb], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            if (FORM == 0) {
                S[2 * rb] += f32x2{fmaxf(acc[rb][0], 0.f), fmaxf(acc[rb][1], 0.f)};
                S[2 * rb + 1] += f32x2{fmaxf(acc[rb][2], 0.f), fmaxf(acc[rb][3], 0.f)};
            } else if (FORM <= 2) {
                const f32x2 zero = f32x2{0.f, 0.f};
                S[2 * rb] += pk_add_clamp(f32x2{acc[rb][0], acc[rb][1]}, zero);
                S[2 * rb + 1] += pk_add_clamp(f32x2{acc[rb][2], acc[rb][3]}, zero);
            } else if (FORM == 3) {
                S[2 * rb] = f32x2{s_add(S[2 * rb][0], s_addc(acc[rb][0], 0.f)), s_add(S[2 * rb][1], s_addc(acc[rb][1], 0.f))};
                S[2 * rb + 1] = f32x2{s_add(S[2 * rb + 1][0], s_addc(acc[rb][2], 0.f)), s_add(S[2 * rb + 1][1], s_addc(acc[rb][3], 0.f))};
            } else {
                S[2 * rb] = f32x2{s_add(S[2 * rb][0], s_max0(acc[rb][0])), s_add(S[2 * rb][1], s_max0(acc[rb][1]))};
                S[2 * rb + 1] = f32x2{s_add(S[2 * rb + 1][0], s_max0(acc[rb][2])), s_add(S[2 * rb + 1][1], s_max0(acc[rb][3]))};
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float res = 0.f;
    for (int s = 0; s < 4; ++s) res += S[s][0] + S[s][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if (lane == 0) { cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0; if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[gridDim.x * 8] = t1 - t0; cyc[gridDim.x * 8 + 1] = r1 - r0; } }
}

template <int FORM>
void run(const char *name, int threads, int blocks = 256, int lds_bytes = 32768) {
    const int tiles = 4000;
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, (size_t)blocks * 512 * 4);
    (void)hipMalloc(&cyc, blocks * 8 * 8 + 16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile<FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k_tile<FORM><<<blocks, threads, lds_bytes>>>(out, cyc, 10, lds_bytes / 4);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k_tile<FORM><<<blocks, threads, lds_bytes>>>(out, cyc, tiles, lds_bytes / 4);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8 + 2);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    const int waves = threads / 64;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) avg += (double)h[b * 8 + w];
    avg /= (double)blocks * waves * tiles;
    const double per_simd = waves * (blocks / 256.0) / 4.0;
    const double flop = (double)blocks * waves * tiles * 16 * 2048.0;
    printf("%-28s %4d x %3d threads, %5d B LDS, %d wave(s)/SIMD: %7.1f ticks per tile per wave, %7.1f ticks of SIMD time per tile (16 MFMAs = 512), %6.1f TFLOP/s (hipEvent), clock %.3f GHz, %6.1f TFLOP/s by the in-kernel 100 MHz clock\n",
           name, blocks, threads, lds_bytes, (int)per_simd, avg, avg / per_simd, flop / (ms * 1e-3) / 1e12, (double)h[blocks * 8] / (double)h[blocks * 8 + 1] * 0.1,
           flop / ((double)h[blocks * 8 + 1] * 1e-8) / 1e12);
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main() {
    const float av[] = {-2.f, -0.5f, 0.f, 0.3f, 0.999f, 1.0f, 1.5f, 1e30f, -1e30f, 0.25f, 1e-40f, -0.f, NAN, INFINITY, -INFINITY, 0.5f};
    const float bv[] = {0.f, 0.25f, 0.f, 0.3f, 0.0005f, 0.f, 0.f, 0.f, 0.f, 0.75f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.5000001f};
    const int n = 16;
    float *da, *db, *dout, ho[16];
    (void)hipMalloc(&da, 64);
    (void)hipMalloc(&db, 64);
    (void)hipMalloc(&dout, 64);
    (void)hipMemcpy(da, av, 64, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, bv, 64, hipMemcpyHostToDevice);
    k_sem<<<1, 64>>>(da, db, dout, n);
    (void)hipMemcpy(ho, dout, 64, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const float s = av[i] + bv[i];
        const float want = std::isnan(s) ? 0.f : fminf(fmaxf(s, 0.f), 1.f);
        const bool ok = ho[i] == want || (std::isnan(s) && (ho[i] == 0.f || std::isnan(ho[i])));
        printf("clamp(%g + %g) = %.9g   (min(max(.,0),1) = %.9g)%s\n", av[i], bv[i], ho[i], want, ok ? "" : "   <-- differs");
        bad += !ok;
    }
    printf("semantics: %s\n", bad ? "DIFFERENT from min(max(x,0),1)" : "v_pk_add_f32 clamp == min(max(a+b,0),1) on both halves");
    for (int threads : {256, 512}) {
        run<0>("max form (P+r)+g, max, max+add", threads);
        run<1>("clamp form (P+r), +g clamp", threads);
        run<2>("clamp form without g", threads);
        run<3>("scalar clamp form (32 VALU)", threads);
        run<4>("scalar max form (40 VALU)", threads);
    }
    // the fused kernel's launch shape: one wavefront per workgroup, 20 KB of LDS each, 8 per CU asked for
    for (int lds : {16384, 18432, 19456, 20480}) {
        run<1>("clamp form, 1-wave workgroups", 64, 2048, lds);
        run<1>("clamp form, 1-wave workgroups", 64, 8192, lds);
    }
    return bad;
}
