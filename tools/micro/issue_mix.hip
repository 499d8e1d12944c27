// Microbenchmark: what an instruction costs in SIMD time beside f32 MFMAs (round 4).  One tile = 16 v_mfma_f32_16x16x4_f32 (two
// interleaved accumulation chains) + N extra instructions of ONE class, all 256 CUs, two wavefronts per SIMD.  Prints the MFMA
// rate (hipEvent wall time: it carries ~0.4 ms of launch overhead per kernel, so compare rows, not absolute TFLOP/s), the cycles
// of SIMD time per tile (s_memtime inside the kernel: the reliable column) and the shader clock the wavefronts saw
// (s_memtime / s_memrealtime).  Findings: an independent VALU instruction costs 1.4-2.6 cycles beside the matrix pipe, a
// ds_read_b128 whose data is used a tile later ~27, a global load (any width) ~27 per 256 bytes when the window misses the L1;
// loads that are waited for at once cost their latency.  In the fused kernel every vector-memory instruction saved was worth
// more than ten VALU instructions (weights as dwordx4: 1160 -> 348 loads per molecule, +1.5 % atoms/s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { BASE, VALU16, PK16, LDS4B, LDS4G, GL4_L1, GL4_L2, GLX4_L2, VALU32, LDS8G, GL16_L2, P_LDS4G, P_GL4_L2, P_GL16_L2, P_GLX4_L2, P_GLX16_L2, DPP16, P_LDS2B, P_LDS4B, P_U16 };

template <int KIND>
__global__ __launch_bounds__(512) void k_mix(float *out, unsigned long long *cyc, const float *wts, int tiles) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (i % 97) * 0.001f;
    __syncthreads();
    float w[2][8], z[8];
    f32x2 v[8];
    for (int s = 0; s < 8; ++s) { w[0][s] = 0.01f + 0.001f * s; w[1][s] = 0.02f - 0.001f * s; z[s] = 0.1f * s + lane * 0.001f; v[s] = f32x2{0.5f * s, 0.25f * lane}; }
    f32x4 S[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    f32x4 la = f32x4{0, 0, 0, 0};
    float ga = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int row = lane;
    f32x4 pl[4] = {la, la, la, la};       // pipelined variants: loaded in tile t, used in tile t + 1
    float pg[16] = {0};
    f32x4 pq[4] = {la, la, la, la};
    // per-wave weight window: L1-resident (4 KB) or L2-resident (1 MB per wave, 2 GB... no: shared 8 MB buffer walked with a stride)
    const float *wbase = wts + ((size_t)(blockIdx.x * 8 + wave) % 64) * 32768;
    size_t goff = lane;
    for (int t = 0; t < tiles; ++t) {
        f32x4 acc[2] = {f32x4{0.1f, 0.1f, 0.1f, 0.1f}, f32x4{0.2f, 0.2f, 0.2f, 0.2f}};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0][s], z[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1][s], z[s], acc[1], 0, 0, 0);
            if (KIND == VALU16) {
                asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(v[s][0]) : "v"(z[s]));
                asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(v[s][1]) : "v"(z[s]));
            }
            if (KIND == VALU32) {
                asm volatile("v_max_f32_e32 %0, %1, %0\n\tv_add_f32_e32 %0, %1, %0" : "+v"(v[s][0]) : "v"(z[s]));
                asm volatile("v_max_f32_e32 %0, %1, %0\n\tv_add_f32_e32 %0, %1, %0" : "+v"(v[s][1]) : "v"(z[s]));
            }
            if (KIND == DPP16) {      // the partner's row broadcast out of a register of the lane that owns it: folded into the add
                v[s][0] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, z[s]), 0x150 + 5, 0xf, 0xf, true));
                v[s][1] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, z[(s + 1) & 7]), 0x150 + 11, 0xf, 0xf, true));
            }
            if (KIND == PK16) {
                asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1" : "+v"(v[s]) : "v"(v[(s + 1) & 7]));
            }
            if ((KIND == LDS4B) && (s & 1) == 0) la += *reinterpret_cast<const f32x4 *>(lds + ((t * 36 + s * 72) & 4095 & ~3) + 4 * (lane >> 4));
            if ((KIND == LDS4G || KIND == LDS8G) && ((s & 1) == 0 || KIND == LDS8G)) {
                la += *reinterpret_cast<const f32x4 *>(lds + 4096 + ((row * 36) & 4095 & ~3) + 4 * (lane >> 4));
                row = (row * 5 + 7) & 127;
            }
            if ((KIND == GL4_L1) && (s & 1) == 0) ga += wbase[(goff + s * 64) & 1023];
            if ((KIND == GL4_L2 || KIND == GL16_L2) && ((s & 1) == 0 || KIND == GL16_L2)) { ga += wbase[(goff + s * 64) & 32767]; if (KIND == GL16_L2) ga += wbase[(goff + s * 64 + 8192) & 32767]; }
            if ((KIND == GLX4_L2) && s == 0) { const f32x4 q = *reinterpret_cast<const f32x4 *>(wbase + ((4 * goff) & 32767)); ga += q[0] + q[3]; }
        }
        if (KIND == P_LDS4G) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                la += pl[k];
                pl[k] = *reinterpret_cast<const f32x4 *>(lds + 4096 + ((row * 36) & 4095 & ~3) + 4 * (lane >> 4));
                row = (row * 5 + 7) & 127;
            }
        }
        if (KIND == P_LDS2B || KIND == P_LDS4B) {
#pragma unroll
            for (int k = 0; k < (KIND == P_LDS2B ? 2 : 4); ++k) {
                la += pl[k];
                pl[k] = *reinterpret_cast<const f32x4 *>(lds + ((t * 36 + k * 72) & 4095 & ~3) + 4 * (lane >> 4));
            }
        }
        if (KIND == P_U16) {
            la[0] += pl[0][0];
            pl[0][0] = (float)reinterpret_cast<const unsigned short *>(lds)[(row * 29 + t) & 8191];
            row = (row * 5 + 7) & 127;
        }
        if (KIND == P_GL4_L2 || KIND == P_GL16_L2) {
#pragma unroll
            for (int k = 0; k < (KIND == P_GL4_L2 ? 4 : 16); ++k) { ga += pg[k]; pg[k] = wbase[(goff + k * 64) & 32767]; }
        }
        if (KIND == P_GLX4_L2 || KIND == P_GLX16_L2) {
#pragma unroll
            for (int k = 0; k < (KIND == P_GLX4_L2 ? 1 : 4); ++k) { ga += pq[k][0] + pq[k][3]; pq[k] = *reinterpret_cast<const f32x4 *>(wbase + ((4 * goff + 256 * k) & 32767)); }
        }
        if (KIND == GL4_L2 || KIND == GLX4_L2 || KIND == GL16_L2 || KIND >= P_GL4_L2) goff += 577;
        S[0] += acc[0];
        S[1] += acc[1];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float res = ga + la[0] + la[1] + la[2] + la[3];
    for (int k = 0; k < 4; ++k) res += pl[k][0] + pq[k][1];
    for (int k = 0; k < 16; ++k) res += pg[k];
    for (int s = 0; s < 8; ++s) res += v[s][0] + v[s][1];
    res += S[0][0] + S[0][1] + S[0][2] + S[0][3] + S[1][0] + S[1][1] + S[1][2] + S[1][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if (lane == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0; }
}

template <int KIND>
void run(const char *name, const float *wts) {
    const int blocks = 256, threads = 512, tiles = 6000;
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, (size_t)blocks * 512 * 4);
    (void)hipMalloc(&cyc, blocks * 8 * 16);
    k_mix<KIND><<<blocks, threads>>>(out, cyc, wts, 10);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k_mix<KIND><<<blocks, threads>>>(out, cyc, wts, tiles);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 16);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double ticks = 0, rt = 0;
    for (int i = 0; i < blocks * 8; ++i) { ticks += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
    const double flop = (double)blocks * 8 * tiles * 16 * 2048.0;
    printf("%-46s %6.1f MFMA TFLOP/s  %7.1f cycles of SIMD time per tile  clock %.3f GHz\n", name, flop / (ms * 1e-3) / 1e12,
           ticks / (blocks * 8.0 * tiles) / 2.0, ticks / rt * 0.1);
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main() {
    float *wts;
    (void)hipMalloc(&wts, (size_t)64 * 32768 * 4);
    (void)hipMemset(wts, 0, (size_t)64 * 32768 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<BASE>("16 MFMAs alone", wts);
        run<VALU16>("+ 16 v_max_f32", wts);
        run<VALU32>("+ 32 v_max_f32 / v_add_f32", wts);
        run<PK16>("+ 16 v_pk_add_f32", wts);
        run<LDS4B>("+ 4 ds_read_b128, 4 addresses per wave", wts);
        run<LDS4G>("+ 4 ds_read_b128, a row per lane", wts);
        run<LDS8G>("+ 8 ds_read_b128, a row per lane", wts);
        run<GL4_L1>("+ 4 global_load_dword, 4 KB window (L1)", wts);
        run<GL4_L2>("+ 4 global_load_dword, 128 KB window (L2)", wts);
        run<GL16_L2>("+ 16 global_load_dword, 128 KB window (L2)", wts);
        run<GLX4_L2>("+ 1 global_load_dwordx4, 128 KB window (L2)", wts);
        run<DPP16>("+ 16 v_add_f32_dpp row_newbcast", wts);
        run<P_LDS2B>("+ 2 ds_read_b128, one address per 16 lanes, used a tile later", wts);
        run<P_LDS4B>("+ 4 ds_read_b128, one address per 16 lanes, used a tile later", wts);
        run<P_U16>("+ 1 ds_read_u16 per lane, used a tile later", wts);
        run<P_LDS4G>("+ 4 ds_read_b128 a row per lane, used a tile later", wts);
        run<P_GL4_L2>("+ 4 global_load_dword (L2), used a tile later", wts);
        run<P_GL16_L2>("+ 16 global_load_dword (L2), used a tile later", wts);
        run<P_GLX4_L2>("+ 1 global_load_dwordx4 (L2), used a tile later", wts);
        run<P_GLX16_L2>("+ 4 global_load_dwordx4 (L2), used a tile later", wts);
    }
    return 0;
}
