// Microbenchmark: the inner loop of the tiled all-pairs sweep (epnn_large.hip.h, k_lg_sweep) in isolation -- per partner j and
// 32-atom tile: z1 = relu(P_i + R_j), 32 MFMAs 16x16x4 (two column blocks x two row blocks x 8 K steps), S_i += relu(d) --
// in several forms of the R_j fetch, at 1 / 2 / 4 wavefronts per SIMD, with the shader clock read beside the time
// (s_memtime ticks per 100 MHz s_memrealtime tick).  Answers: what bounds the sweep (matrix pipe, VALU issue, LDS, clock)?
//   MODE 0  R_j row broadcast from LDS (two ds_read_b128 per partner): the shipped form
//   MODE 1  R rows of 16 partners in registers (lane n16 of every 16-lane row holds partner n16's features), R_j reaches the
//           add through DPP row_newbcast: no LDS read per partner, no extra VALU
//   MODE 4  the same with the broadcast folded into the add (v_add_f32_dpp, inline asm: hipcc keeps a v_mov_b32_dpp)
//   MODE 2  MFMAs only (constant operands): the matrix pipe's ceiling at this clock
//   MODE 3  the VALU work of MODE 0 without any R fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mf(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }
template <int J>
__device__ __forceinline__ float bcast(float v) {      // value of lane J of this lane's 16-lane row
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + J, 0xf, 0xf, true));
}

template <int J>
__device__ __forceinline__ float add_bcast(float p, float r) {      // relu(p + (r of lane J of this lane's 16-lane row))
    float o;
    asm volatile("v_add_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\tv_max_f32_e32 %0, 0, %0" : "=&v"(o) : "v"(r), "v"(p), "n"(J));
    return o;
}

template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k_sweep(const float *__restrict__ Rg, float *out, unsigned long long *cyc, int nj16) {
    __shared__ __attribute__((aligned(16))) float Rs[64 * 32];
    const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, n16 = lane & 15;
    for (int i = tid; i < 64 * 32; i += 256) Rs[i] = Rg[i];
    __syncthreads();
    float pb[2][8];
    f32x4 P0[2], P1[2], S0[2], S1[2], b2v[2];
    for (int rb = 0; rb < 2; ++rb) {
        for (int s = 0; s < 8; ++s) pb[rb][s] = Rg[2048 + (rb * 8 + s) * 64 + lane];
        P0[rb] = *reinterpret_cast<const f32x4 *>(Rg + 4096 + lane * 8 + 4 * rb);
        P1[rb] = *reinterpret_cast<const f32x4 *>(Rg + 4096 + 512 + lane * 8 + 4 * rb);
        b2v[rb] = *reinterpret_cast<const f32x4 *>(Rg + 4096 + 1024 + 16 * rb + 4 * q);
        S0[rb] = f32x4{0, 0, 0, 0};
        S1[rb] = f32x4{0, 0, 0, 0};
    }
    const int po = 16 * (q & 1) + 4 * (q >> 1);
    auto block = [&](const f32x4 &za, const f32x4 &zb, f32x4(&Sc)[2]) {
        const float z[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
        f32x4 d[2] = {b2v[0], b2v[1]};
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int s = 0; s < 8; ++s) d[rb] = mf(pb[rb][s], z[s], d[rb]);
        if (MODE == 2) { Sc[0] = d[0]; Sc[1] = d[1]; }
        else { Sc[0] += relu4(d[0]); Sc[1] += relu4(d[1]); }
    };
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int jj = 0; jj < nj16; ++jj) {
        const int jb = (jj & 3) * 16;
        if (MODE == 0) {
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const float *rr = Rs + (jb + j) * 32 + po;
                const f32x4 ra = *reinterpret_cast<const f32x4 *>(rr), rb_ = *reinterpret_cast<const f32x4 *>(rr + 8);
                block(relu4(P0[0] + ra), relu4(P0[1] + rb_), S0);
                block(relu4(P1[0] + ra), relu4(P1[1] + rb_), S1);
            }
        } else if (MODE == 1) {
            // lane (q, n16) holds partner jb + n16's features po .. po+3 and po+8 .. po+11
            const float *rr = Rs + (jb + n16) * 32 + po;
            const f32x4 ra = *reinterpret_cast<const f32x4 *>(rr), rb_ = *reinterpret_cast<const f32x4 *>(rr + 8);
#define STEP(J)                                                                                                        \
    {                                                                                                                  \
        const f32x4 a = {bcast<J>(ra[0]), bcast<J>(ra[1]), bcast<J>(ra[2]), bcast<J>(ra[3])};                          \
        const f32x4 b = {bcast<J>(rb_[0]), bcast<J>(rb_[1]), bcast<J>(rb_[2]), bcast<J>(rb_[3])};                      \
        block(relu4(P0[0] + a), relu4(P0[1] + b), S0);                                                                 \
        block(relu4(P1[0] + a), relu4(P1[1] + b), S1);                                                                 \
    }
            STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
            STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#undef STEP
        } else if (MODE == 4) {
            const float *rr = Rs + (jb + n16) * 32 + po;
            const f32x4 ra = *reinterpret_cast<const f32x4 *>(rr), rb_ = *reinterpret_cast<const f32x4 *>(rr + 8);
#define STEP(J)                                                                                                        \
    {                                                                                                                  \
        f32x4 a0, a1, c0, c1;                                                                                          \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                \
            a0[k] = add_bcast<J>(P0[0][k], ra[k]); a1[k] = add_bcast<J>(P0[1][k], rb_[k]);                             \
            c0[k] = add_bcast<J>(P1[0][k], ra[k]); c1[k] = add_bcast<J>(P1[1][k], rb_[k]);                             \
        }                                                                                                              \
        block(a0, a1, S0);                                                                               \
        block(c0, c1, S1);                                                                               \
    }
            STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
            STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#undef STEP
        } else if (MODE == 2) {
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                block(P0[0], P0[1], S0);
                block(P1[0], P1[1], S1);
            }
        } else {
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                block(relu4(P0[0] + S1[0]), relu4(P0[1] + S1[1]), S0);
                block(relu4(P1[0] + S0[0]), relu4(P1[1] + S0[1]), S1);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 res = S0[0] + S0[1] + S1[0] + S1[1];
    out[blockIdx.x * 256 + tid] = res[0] + res[1] + res[2] + res[3];
    if (lane == 0) {
        cyc[(blockIdx.x * 4 + (tid >> 6)) * 2] = t1 - t0;
        cyc[(blockIdx.x * 4 + (tid >> 6)) * 2 + 1] = r1 - r0;
    }
}

template <int MODE, int WPS>
void run(const char *name, const float *Rg, float *out, unsigned long long *cyc) {
    const int blocks = 256 * WPS, nj16 = 4000 / WPS;            // ~64k partners per wave at one wave per SIMD
    k_sweep<MODE, WPS><<<blocks, 256>>>(Rg, out, cyc, nj16);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k_sweep<MODE, WPS><<<blocks, 256>>>(Rg, out, cyc, nj16);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
    std::vector<double> tk, clk;
    for (int w = 0; w < blocks * 4; ++w) {
        tk.push_back((double)h[2 * w] / (nj16 * 16));
        clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1);
    }
    std::sort(tk.begin(), tk.end());
    std::sort(clk.begin(), clk.end());
    const double flop = 65536.0 * nj16 * 16 * 4 * blocks;
    printf("%-34s %d wave(s)/SIMD: %6.0f ticks per partner per wave (median), %6.0f of SIMD time; clock %.2f GHz; %.3f ms; %.1f TFLOP/s = %.3f of 157.3\n",
           name, WPS, tk[tk.size() / 2], tk[tk.size() / 2] / WPS, clk[clk.size() / 2], ms, flop / (ms * 1e-3) / 1e12,
           flop / (ms * 1e-3) / 1e12 / 157.3);
    fflush(stdout);
}

int main() {
    float *Rg, *out;
    unsigned long long *cyc;
    std::vector<float> h(8192);
    unsigned s = 12345u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    (void)hipMalloc(&Rg, 8192 * 4);
    (void)hipMemcpy(Rg, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, (size_t)1024 * 256 * 4);
    (void)hipMalloc(&cyc, 1024 * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 1>("LDS broadcast row (shipped)", Rg, out, cyc);
        run<0, 2>("LDS broadcast row (shipped)", Rg, out, cyc);
        run<0, 4>("LDS broadcast row (shipped)", Rg, out, cyc);
        run<1, 1>("registers + DPP row_newbcast", Rg, out, cyc);
        run<1, 2>("registers + DPP row_newbcast", Rg, out, cyc);
        run<1, 4>("registers + DPP row_newbcast", Rg, out, cyc);
        run<4, 1>("registers + v_add_f32_dpp", Rg, out, cyc);
        run<4, 2>("registers + v_add_f32_dpp", Rg, out, cyc);
        run<4, 4>("registers + v_add_f32_dpp", Rg, out, cyc);
        run<3, 2>("VALU of the tile, no R fetch", Rg, out, cyc);
        run<3, 4>("VALU of the tile, no R fetch", Rg, out, cyc);
        run<2, 1>("MFMAs only", Rg, out, cyc);
        run<2, 2>("MFMAs only", Rg, out, cyc);
        run<2, 4>("MFMAs only", Rg, out, cyc);
    }
    return 0;
}
