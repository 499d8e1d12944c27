// Dense front-end: the reference's literal tensors -> the flat atom arrays + pair list the kernels run on.
//
//   make_model inputs (charge_gn.py:376-384): h_inp,e_inp (B,N,N,48), x_inp (B,N,N,nx), q_inp,mask_inp (B,N,N,1);
//     per-atom h,x,q = sum over axis 1 / sum of mask over axis 1, 0 where the denominator is 0 (divide_no_nan).
//   GNN_layer.call / EPN_layer.call inputs (charge_gn.py:57,88): h (B,N,48), x (B,N,nx), q (B,N,1) per atom,
//     e (B,N,N,48), mask (B,N,N,1).
//
// e and mask are arbitrary tensors here (not necessarily what get_init_edges produces), so the pair list is built
// without assuming symmetry: an unordered pair {i<j} becomes ONE symmetric entry when e_ij == e_ji bit for bit and
// mask_ij == mask_ji, otherwise up to two one-sided entries (i,j,e_ij) / (j,i,e_ji); a non-zero diagonal e_ii is a
// one-sided entry too.  Entry weight = mask_ij * is_near_ij (charge_gn.py:90-94,116), node_mask_k =
// clip(sum_j mask_jk, 0, 1) (charge_gn.py:59).  Trailing atoms that are zero in every input (the padding that
// gen_padded_init_state adds) are cut off: n_eff = 1 + last non-trivial atom; their effect on the real atoms is
// the closed-form padded-partner term, exactly as on the compact path.
#pragma once
#include "epnn_host.h"
#include "epnn_frontend.hip.h"

struct DenseArgs {
    int B, N, nx, model_level;
    const float *h_in, *e_in, *x_in, *q_in, *mask_in;
    float *xs, *hs, *qs, *nms;      // per slot (b*N + k)
    int *flag;                      // per slot: atom is non-trivial
    int *neff;                      // per molecule
    float tol;
    // flat side
    int A;
    const int *moff, *mol_of;
    float *xf, *hf, *qf, *nmf;
    int *row_cnt;
    const int *row_off;
    int pcap;
    int *pi, *pj, *psym;
    float *pe, *pwi, *pwj;
    int *status;
    float *out;                     // (B,N,C)
    const float *src;               // flat [A][C]
    int C;
};

// one wave per slot (b,k): per-atom features, node mask and the "non-trivial" flag
__global__ __launch_bounds__(256) void k_dn_atoms(DenseArgs D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = blockIdx.x * 4 + wave;
    if (slot >= D.B * D.N) return;
    const int b = slot / D.N, k = slot - b * D.N, N = D.N, nx = D.nx;
    const size_t mb = (size_t)b * N * N;
    // node mask / denominator: sum_j mask[b,j,k]   (sequential-in-lane then tree; order fixed)
    float den = 0.f;
    for (int j = lane; j < N; j += 64) den += D.mask_in[mb + (size_t)j * N + k];
    for (int d = 32; d >= 1; d >>= 1) den += __shfl_xor(den, d, 64);
    int nontriv = den != 0.f;
    // any non-zero in row k / column k of e, or in row k of mask
    const float *erow = D.e_in + (mb + (size_t)k * N) * EPNN_EDIM;
    for (int i = lane; i < N * EPNN_EDIM; i += 64) nontriv |= erow[i] != 0.f;
    for (int i = lane; i < N * EPNN_EDIM; i += 64) {
        const int j = i / EPNN_EDIM, ch = i - j * EPNN_EDIM;
        nontriv |= D.e_in[(mb + (size_t)j * N + k) * EPNN_EDIM + ch] != 0.f;
    }
    // per-atom features
    const int F = nx + EPNN_EDIM + 1;
    for (int f = lane; f < F; f += 64) {
        float v;
        if (D.model_level) {
            float s = 0.f;
            if (f < nx) for (int j = 0; j < N; ++j) s += D.x_in[(mb + (size_t)j * N + k) * nx + f];
            else if (f < nx + EPNN_EDIM) for (int j = 0; j < N; ++j) s += D.h_in[(mb + (size_t)j * N + k) * EPNN_EDIM + (f - nx)];
            else for (int j = 0; j < N; ++j) s += D.q_in[mb + (size_t)j * N + k];
            v = den != 0.f ? s / den : 0.f;                       // tf.math.divide_no_nan
        } else {
            if (f < nx) v = D.x_in[(size_t)slot * nx + f];
            else if (f < nx + EPNN_EDIM) v = D.h_in[(size_t)slot * EPNN_EDIM + (f - nx)];
            else v = D.q_in[slot];
        }
        nontriv |= v != 0.f;
        if (f < nx) D.xs[(size_t)slot * nx + f] = v;
        else if (f < nx + EPNN_EDIM) D.hs[(size_t)slot * EPNN_EDIM + (f - nx)] = v;
        else D.qs[slot] = v;
    }
    const int any = __ballot(nontriv != 0) != 0ull;
    if (lane == 0) {
        D.nms[slot] = fminf(fmaxf(den, 0.f), 1.f);                // charge_gn.py:59
        D.flag[slot] = any;
    }
}

__global__ __launch_bounds__(64) void k_dn_neff(DenseArgs D) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int last = 0;
    for (int k = lane; k < D.N; k += 64)
        if (D.flag[b * D.N + k]) last = max(last, k);
    for (int d = 32; d >= 1; d >>= 1) last = max(last, __shfl_xor(last, d, 64));
    if (lane == 0) D.neff[b] = last + 1;
}

__global__ __launch_bounds__(256) void k_dn_compact(DenseArgs D) {
    const int F = D.nx + EPNN_EDIM + 2;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < D.A * F; idx += gridDim.x * 256) {
        const int at = idx / F, f = idx - at * F;
        const int b = D.mol_of[at];
        const int slot = b * D.N + (at - D.moff[b]);
        if (f < D.nx) D.xf[(size_t)at * D.nx + f] = D.xs[(size_t)slot * D.nx + f];
        else if (f < D.nx + EPNN_EDIM) D.hf[(size_t)at * EPNN_EDIM + (f - D.nx)] = D.hs[(size_t)slot * EPNN_EDIM + (f - D.nx)];
        else if (f == D.nx + EPNN_EDIM) D.qf[at] = D.qs[slot];
        else D.nmf[at] = D.nms[slot];
    }
}

// classification of the ordered pair (i,j) seen from row i: 0 none, 1 symmetric entry (only for j > i), 2 one-sided
__device__ __forceinline__ int dn_classify(const DenseArgs &D, int b, int i, int j, float *wout) {
    const int N = D.N;
    const size_t mb = (size_t)b * N * N;
    const float *eij = D.e_in + (mb + (size_t)i * N + j) * EPNN_EDIM;
    const float *eji = D.e_in + (mb + (size_t)j * N + i) * EPNN_EDIM;
    bool nz = false, same = true;
    float mx = 0.f;
    for (int ch = 0; ch < EPNN_EDIM; ++ch) {
        const float a = eij[ch], c = eji[ch];
        nz |= a != 0.f;
        same &= __float_as_uint(a) == __float_as_uint(c);
        mx = fmaxf(mx, a);
    }
    const float mij = D.mask_in[mb + (size_t)i * N + j], mji = D.mask_in[mb + (size_t)j * N + i];
    same &= __float_as_uint(mij) == __float_as_uint(mji);
    *wout = mx > D.tol ? mij : 0.f;          // mask_ij * is_near_ij, is_near = max_k clip(e) != tol
    if (i == j) return nz ? 2 : 0;
    if (same) return (j > i && nz) ? 1 : 0;
    return nz ? 2 : 0;
}

// one wave per flat atom row: count (fill == 0) or write (fill == 1) its entries, ascending j
template <int FILL>
__global__ __launch_bounds__(256) void k_dn_pairs(DenseArgs D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= D.A) return;
    if (FILL && D.row_off[D.A] > D.pcap) return;
    const int b = D.mol_of[row], a0 = D.moff[b], n = D.moff[b + 1] - a0, i = row - a0;
    int base = FILL ? D.row_off[row] : 0;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        int kind = 0;
        float w = 0.f;
        if (j < n) kind = dn_classify(D, b, i, j, &w);
        const unsigned long long bal = __ballot(kind != 0);
        if (FILL && kind) {
            const int s = base + __popcll(bal & ((1ull << lane) - 1ull));
            D.pi[s] = row;
            D.pj[s] = a0 + j;
            D.psym[s] = kind == 1;
            D.pwi[s] = (i == j) ? 0.f : w;
            D.pwj[s] = kind == 1 ? w : 0.f;
            const float *eij = D.e_in + (((size_t)b * D.N + i) * D.N + j) * EPNN_EDIM;
            for (int ch = 0; ch < EPNN_EDIM; ++ch) D.pe[(size_t)s * EPNN_EDIM + ch] = eij[ch];
        }
        base += __popcll(bal);
    }
    if (!FILL && lane == 0) D.row_cnt[row] = base;
}

// flat [A][C] -> padded (B,N,C), zeros beyond n_eff
__global__ __launch_bounds__(256) void k_dn_scatter(DenseArgs D) {
    const size_t total = (size_t)D.B * D.N * D.C;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int cidx = (int)(idx % D.C);
        const size_t slot = idx / D.C;
        const int b = (int)(slot / D.N), k = (int)(slot - (size_t)b * D.N);
        const int n = D.moff[b + 1] - D.moff[b];
        D.out[idx] = k < n ? D.src[(size_t)(D.moff[b] + k) * D.C + cidx] : 0.f;
    }
}
