// MLP_layer.call (reference charge_gn.py:41-45) as a stand-alone operator: rows x n_in -> relu 32 -> relu 32 -> n_out.
// One wave per 32 rows; the three Dense layers are chained through MFMA accumulators (rows on the MFMA columns).
#pragma once
#include "epnn_common.h"

struct MlpArgs {
    const float *x;                 // [rows][n_in]
    const float *W1, *b1, *W2, *b2, *W3, *b3;   // Keras layout [in][out]
    float *out;                     // [rows][n_out]
    int rows, n_in, n_out;
};

__global__ __launch_bounds__(256) void k_mlp_forward(MlpArgs M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, hh = lane >> 5;
    const int row0 = (blockIdx.x * 4 + wave) * 32;
    if (row0 >= M.rows) return;
    const int row = row0 + c;
    const bool live = row < M.rows;
    const float *xr = M.x + (size_t)(live ? row : row0) * M.n_in;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = M.b1[epnn_kappa(hh, r)];
    for (int s = 0; 2 * s < M.n_in; ++s) {
        const int f = 2 * s + hh;
        const float a = f < M.n_in ? M.W1[(size_t)f * 32 + c] : 0.f;
        const float b = f < M.n_in ? xr[f] : 0.f;
        acc = epnn_mfma(a, b, acc);
    }
    float z[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = fmaxf(acc[r], 0.f);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = M.b2[epnn_kappa(hh, r)];
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = epnn_mfma(M.W2[epnn_kappa(hh, s) * 32 + c], z[s], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = fmaxf(acc[r], 0.f);
    for (int tile = 0; tile * 32 < M.n_out; ++tile) {
        const int oc = tile * 32 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = tile * 32 + epnn_kappa(hh, r);
            acc[r] = o < M.n_out ? M.b3[o] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
            acc = epnn_mfma(oc < M.n_out ? M.W3[(size_t)epnn_kappa(hh, s) * M.n_out + oc] : 0.f, z[s], acc);
        if (live) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = tile * 32 + epnn_kappa(hh, r);
                if (o < M.n_out) M.out[(size_t)row * M.n_out + o] = acc[r];
            }
        }
    }
}
