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

// ---- any widths (GenMlp, epnn_common.h).  EPNN_GMLP_ROWS rows of activations live in LDS (row stride WMAX + 1: the rows of one output
// column sit in different banks); thread (output o, four rows) runs the dot products over the layer's inputs.
#define EPNN_GMLP_ST (EPNN_GMLP_WMAX + 1)
// the stack on the rows held in `a0`; the result is in the returned buffer (a0 or a1).  All threads of the workgroup call it.
__device__ __forceinline__ float *gmlp_rows(const GenMlp &G, float *a0, float *a1) {
    float *src = a0, *dst = a1;
    for (int l = 0; l < G.n; ++l) {
        const int ni = G.dims[l], no = G.dims[l + 1];
        const float *W = G.w + G.offW[l], *b = G.w + G.offB[l];
        const bool act = l + 1 < G.n;
        // thread = (output o, four rows): neighbouring threads read neighbouring kernel columns (one coalesced row of W per input,
        // used for four rows), the activations are LDS broadcasts; every dot product runs over its inputs in ascending order
        for (int idx = threadIdx.x; idx < (EPNN_GMLP_ROWS / 4) * no; idx += blockDim.x) {
            const int o = idx % no, r4 = (idx / no) * 4;
            const float *in = src + r4 * EPNN_GMLP_ST;
            float acc[4] = {b[o], b[o], b[o], b[o]};
#pragma unroll 4
            for (int i = 0; i < ni; ++i) {
                const float wv = W[(size_t)i * no + o];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = fmaf(in[r * EPNN_GMLP_ST + i], wv, acc[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[r];
                if (act) {                                   // Dense(n, activation) of MLP_layer (charge_gn.py:38); Keras' definitions
                    if (G.act == EPNN_ACT_RELU) v = fmaxf(acc[r], 0.f);
                    else if (G.act == EPNN_ACT_TANH) v = tanhf(acc[r]);
                    else if (G.act == EPNN_ACT_SIGMOID) v = 1.f / (1.f + expf(-acc[r]));
                }
                dst[(r4 + r) * EPNN_GMLP_ST + o] = v;
            }
        }
        __syncthreads();
        float *t = src; src = dst; dst = t;
    }
    return src;
}
__global__ __launch_bounds__(256) void k_mlp_generic(GenMlp G, const float *x, float *out, int rows) {
    __shared__ float a0[EPNN_GMLP_ROWS * EPNN_GMLP_ST], a1[EPNN_GMLP_ROWS * EPNN_GMLP_ST];
    const int row0 = blockIdx.x * EPNN_GMLP_ROWS, ni = G.dims[0], no = G.dims[G.n];
    for (int idx = threadIdx.x; idx < EPNN_GMLP_ROWS * ni; idx += blockDim.x) {
        const int r = idx / ni, i = idx - r * ni;
        a0[r * EPNN_GMLP_ST + i] = row0 + r < rows ? x[(size_t)(row0 + r) * ni + i] : 0.f;
    }
    __syncthreads();
    const float *res = gmlp_rows(G, a0, a1);
    for (int idx = threadIdx.x; idx < EPNN_GMLP_ROWS * no; idx += blockDim.x) {
        const int r = idx / no, o = idx - r * no;
        if (row0 + r < rows) out[(size_t)(row0 + r) * no + o] = res[r * EPNN_GMLP_ST + o];
    }
}
