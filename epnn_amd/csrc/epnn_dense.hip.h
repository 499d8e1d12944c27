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
    int *host_status;               // pinned host ints: [0] status bits, [1] number of listed pairs (written by the last kernel of the call)
    float *out;                     // (B,N,C)
    const float *src;               // flat [A][C]
    int C;
};

// node mask / denominator per slot (b,k): sum_j mask[b,j,k]  (charge_gn.py:59, 382-384); one thread per slot, the
// 64 threads of a wave read 64 consecutive k of one row j at a time (coalesced)
__global__ __launch_bounds__(256) void k_dn_den(DenseArgs D, float *den) {
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= D.B * D.N) return;
    const int b = slot / D.N, k = slot - b * D.N, N = D.N;
    const size_t mb = (size_t)b * N * N;
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < N; ++j) s += D.mask_in[mb + (size_t)j * N + k];
    den[slot] = s;
    D.nms[slot] = fminf(fmaxf(s, 0.f), 1.f);
    if (s != 0.f) atomicOr(&D.flag[slot], 1);
}

// per-atom features of the model-level entry: x, h, q = sum over axis 1 / den (tf.math.divide_no_nan).  Threads run
// over the contiguous (k, channel) plane of one molecule and loop over j, so every step of the loop is one fully
// coalesced row of the input tensor -- this kernel is a pure HBM stream (the (N,N,.) inputs are read exactly once).
template <int WHICH>   // 0: h_inp (48 channels), 1: x_inp (nx), 2: q_inp (1)
__global__ __launch_bounds__(256) void k_dn_feat(DenseArgs D, const float *den) {
    const int N = D.N, C = WHICH == 0 ? EPNN_EDIM : (WHICH == 1 ? D.nx : 1);
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;          // (k, c) flattened
    if (idx >= N * C) return;
    const float *src = (WHICH == 0 ? D.h_in : (WHICH == 1 ? D.x_in : D.q_in)) + (size_t)b * N * N * C;
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < N; ++j) s += src[(size_t)j * N * C + idx];      // summation order = j order (fixed)
    const int k = idx / C, c = idx - k * C;
    const float dn = den[b * N + k];
    const float v = dn != 0.f ? s / dn : 0.f;
    const size_t slot = (size_t)b * N + k;
    if (WHICH == 0) D.hs[slot * EPNN_EDIM + c] = v;
    else if (WHICH == 1) D.xs[slot * D.nx + c] = v;
    else D.qs[slot] = v;
    if (v != 0.f) atomicOr(&D.flag[slot], 1);
}

// The same four passes (k_dn_den, k_dn_feat<0..2>) in ONE launch for calls on one or a few molecules, where a launch costs more
// than its work: thread = (atom k, channel c of [h | x | q]), every thread adds up the mask column of its atom itself (same
// order over j as k_dn_den: the same bits).
// one (atom k, channel c of [h | x | q]) of molecule b; `mark(k)` records that atom k is non-trivial
template <typename MARK>
__device__ __forceinline__ void dn_feat_all_item(const DenseArgs &D, float *den, int b, int idx, MARK &&mark) {
    const int N = D.N, CT = EPNN_EDIM + D.nx + 1;
    const int k = idx / CT, c = idx - k * CT;
    const int which = c < EPNN_EDIM ? 0 : (c < EPNN_EDIM + D.nx ? 1 : 2);
    const int C = which == 0 ? EPNN_EDIM : (which == 1 ? D.nx : 1), cc = which == 0 ? c : (which == 1 ? c - EPNN_EDIM : 0);
    const float *src = (which == 0 ? D.h_in : (which == 1 ? D.x_in : D.q_in)) + (size_t)b * N * N * C + (size_t)k * C + cc;
    const float *mk = D.mask_in + (size_t)b * N * N + k;
    float s = 0.f, dn = 0.f;
    for (int j0 = 0; j0 < N; j0 += 16) {                       // sixteen rows (32 loads) in flight, added in row order
        float sv[16], dv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int jc = min(j0 + u, N - 1);
            sv[u] = src[(size_t)jc * N * C];
            dv[u] = mk[(size_t)jc * N];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (j0 + u < N) {
                s += sv[u];
                dn += dv[u];
            }
    }
    const float v = dn != 0.f ? s / dn : 0.f;
    const size_t slot = (size_t)b * N + k;
    if (which == 0) D.hs[slot * EPNN_EDIM + cc] = v;
    else if (which == 1) D.xs[slot * D.nx + cc] = v;
    else D.qs[slot] = v;
    if (c == 0) {
        den[slot] = dn;
        D.nms[slot] = fminf(fmaxf(dn, 0.f), 1.f);
        if (dn != 0.f) mark(k);
    }
    if (v != 0.f) mark(k);
}
__global__ __launch_bounds__(256) void k_dn_feat_all(DenseArgs D, float *den) {
    const int N = D.N, CT = EPNN_EDIM + D.nx + 1;
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * CT) return;
    dn_feat_all_item(D, den, b, idx, [&](int k) { atomicOr(&D.flag[(size_t)b * N + k], 1); });
}

// layer-level entry: per-atom tensors are given; copy and flag
__global__ __launch_bounds__(256) void k_dn_copy_atoms(DenseArgs D) {
    const int F = D.nx + EPNN_EDIM + 1;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < D.B * D.N * F; idx += gridDim.x * 256) {
        const int slot = idx / F, f = idx - slot * F;
        float v;
        if (f < D.nx) { v = D.x_in[(size_t)slot * D.nx + f]; D.xs[(size_t)slot * D.nx + f] = v; }
        else if (f < D.nx + EPNN_EDIM) { v = D.h_in[(size_t)slot * EPNN_EDIM + (f - D.nx)]; D.hs[(size_t)slot * EPNN_EDIM + (f - D.nx)] = v; }
        else { v = D.q_in[slot]; D.qs[slot] = v; }
        if (v != 0.f) atomicOr(&D.flag[slot], 1);
    }
}

// e scan: one thread per ordered pair (b,i,j) reads its 48 channels as 12 x 16 B; a non-zero e[i][j] marks both atom i
// and atom j (integer atomics, order-free; only the ~7 % near pairs issue any)
__device__ __forceinline__ bool dn_e_nonzero(const DenseArgs &D, size_t r) {
    const f32x4 *e4 = reinterpret_cast<const f32x4 *>(D.e_in + r * EPNN_EDIM);
    bool nz = false;
#pragma unroll
    for (int q = 0; q < EPNN_EDIM / 4; ++q) {
        const f32x4 v = e4[q];
        nz |= (v[0] != 0.f) | (v[1] != 0.f) | (v[2] != 0.f) | (v[3] != 0.f);
    }
    return nz;
}
__global__ __launch_bounds__(256) void k_dn_escan(DenseArgs D) {
    const size_t pairs = (size_t)D.B * D.N * D.N;
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < pairs; r += (size_t)gridDim.x * 256) {
        if (dn_e_nonzero(D, r)) {
            const int j = (int)(r % D.N), i = (int)((r / D.N) % D.N), b = (int)(r / ((size_t)D.N * D.N));
            atomicOr(&D.flag[b * D.N + i], 1);
            atomicOr(&D.flag[b * D.N + j], 1);
        }
    }
}

// One or a few molecules per call (B N^2 <= 65536): a call is made of latencies, so what the host needs before it can plan is
// TWO launches instead of a memset, three kernels and a download: (1) the per-atom features and the e scan side by side (the
// block index says which), flags written as the call's generation number (no memset: a flag counts when it equals `gen`);
// (2) the effective atom counts, written into page-locked host memory as well.  The same loops in the same order as
// k_dn_feat_all / k_dn_escan / k_dn_neff: the same bits.  (One workgroup per molecule was tried first: 0.116 ms instead of
// 0.127 for a 9-atom call but 0.229 instead of 0.207 at 41 atoms -- the work is small, but it is not one CU's worth of latency.)
__global__ __launch_bounds__(256) void k_dn_front_small(DenseArgs D, float *den, int gen, int feat_blocks) {
    const int b = blockIdx.y, N = D.N, CT = EPNN_EDIM + D.nx + 1;
    if ((int)blockIdx.x < feat_blocks) {
        const int idx = blockIdx.x * 256 + threadIdx.x;
        if (idx < N * CT) dn_feat_all_item(D, den, b, idx, [&](int k) { D.flag[(size_t)b * N + k] = gen; });
    } else {
        const int r = (blockIdx.x - feat_blocks) * 256 + threadIdx.x;
        if (r < N * N && dn_e_nonzero(D, (size_t)b * N * N + r)) {
            D.flag[(size_t)b * N + r / N] = gen;
            D.flag[(size_t)b * N + r % N] = gen;
        }
    }
}
__global__ __launch_bounds__(64) void k_dn_neff_small(DenseArgs D, int gen, int *neff_host) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int last = 0;
    for (int k = lane; k < D.N; k += 64)
        if (D.flag[b * D.N + k] == gen) last = max(last, k);
    for (int d = 32; d >= 1; d >>= 1) last = max(last, __shfl_xor(last, d, 64));
    if (lane == 0) {
        D.neff[b] = last + 1;
        neff_host[b] = last + 1;
        __threadfence_system();
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

// classification of the ordered pair (i,j) seen from row i: 0 none, 1 symmetric entry (only for j > i), 2 one-sided
__device__ __forceinline__ int dn_classify(const DenseArgs &D, int b, int i, int j, float *wout) {
    const int N = D.N;
    const size_t mb = (size_t)b * N * N;
    // both rows as 12 + 12 sixteen-byte loads, all in flight (96 four-byte loads were two windows of a wavefront's 63)
    const f32x4 *eij = reinterpret_cast<const f32x4 *>(D.e_in + (mb + (size_t)i * N + j) * EPNN_EDIM);
    const f32x4 *eji = reinterpret_cast<const f32x4 *>(D.e_in + (mb + (size_t)j * N + i) * EPNN_EDIM);
    f32x4 va[EPNN_EDIM / 4], vc[EPNN_EDIM / 4];
#pragma unroll
    for (int q = 0; q < EPNN_EDIM / 4; ++q) {
        va[q] = eij[q];
        vc[q] = eji[q];
    }
    bool nz = false, same = true;
    float mx = 0.f;
#pragma unroll
    for (int q = 0; q < EPNN_EDIM / 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = va[q][r], c = vc[q][r];
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
__device__ __forceinline__ void dn_pairs_row(const DenseArgs &D, int row, int lane, int base_in = -1) {
    const int b = D.mol_of[row], a0 = D.moff[b], n = D.moff[b + 1] - a0, i = row - a0;
    if (!FILL) {
        // the atom's own features, slot (b, i) -> flat row (what used to be a launch of its own): one lane per feature
        const int F = D.nx + EPNN_EDIM + 2, f = lane;
        const int slot = b * D.N + i;
        if (f < D.nx) D.xf[(size_t)row * D.nx + f] = D.xs[(size_t)slot * D.nx + f];
        else if (f < D.nx + EPNN_EDIM) D.hf[(size_t)row * EPNN_EDIM + (f - D.nx)] = D.hs[(size_t)slot * EPNN_EDIM + (f - D.nx)];
        else if (f == D.nx + EPNN_EDIM) D.qf[row] = D.qs[slot];
        else if (f < F) D.nmf[row] = D.nms[slot];
    }
    int base = FILL ? (base_in >= 0 ? base_in : D.row_off[row]) : 0;
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
            const f32x4 *eij = reinterpret_cast<const f32x4 *>(D.e_in + (((size_t)b * D.N + i) * D.N + j) * EPNN_EDIM);
            f32x4 *dst = reinterpret_cast<f32x4 *>(D.pe + (size_t)s * EPNN_EDIM);
            f32x4 ev[EPNN_EDIM / 4];
#pragma unroll
            for (int q = 0; q < EPNN_EDIM / 4; ++q) ev[q] = eij[q];
#pragma unroll
            for (int q = 0; q < EPNN_EDIM / 4; ++q) dst[q] = ev[q];
        }
        base += __popcll(bal);
    }
    if (!FILL && lane == 0) D.row_cnt[row] = base;
}
template <int FILL>
__global__ __launch_bounds__(256) void k_dn_pairs(DenseArgs D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= D.A) return;
    if (FILL && D.row_off[D.A] > D.pcap) return;
    dn_pairs_row<FILL>(D, row, lane);
}
// The pair list of a small call (at most 1024 flat atoms) in two launches instead of a memset and three kernels: the count
// launch clears the status words, the fill launch scans for itself -- every wavefront adds up the counts of the rows before its
// own (at most 16 loads per lane) and the total, writes its row's offset, and fills.  Same per-row code.
__global__ __launch_bounds__(256) void k_dn_pairs_count_small(DenseArgs D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x < 4) D.status[threadIdx.x] = 0;
    const int row = blockIdx.x * 4 + wave;
    if (row < D.A) dn_pairs_row<0>(D, row, lane);
}
__global__ __launch_bounds__(256) void k_dn_pairs_fill_small(DenseArgs D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= D.A) return;
    int before = 0, total = 0;
    for (int r0 = 0; r0 < D.A; r0 += 64) {
        const int r = r0 + lane;
        const int c = D.row_cnt[min(r, D.A - 1)];
        if (r < D.A) {
            total += c;
            if (r < row) before += c;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        before += __shfl_xor(before, d, 64);
        total += __shfl_xor(total, d, 64);
    }
    int *roff = const_cast<int *>(D.row_off);
    if (lane == 0) {
        roff[row] = before;
        if (row == D.A - 1) {
            roff[D.A] = total;
            if (total > D.pcap) atomicOr(D.status, EPNN_ST_PAIR_OVERFLOW);
        }
    }
    if (total > D.pcap) return;
    dn_pairs_row<1>(D, row, lane, before);
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
    if (blockIdx.x == 0 && threadIdx.x == 0 && D.host_status) {
        // every kernel of the call ran before this one on the stream: hand status + pair count to the host from here (two
        // device-to-host copies less at the end of a call that is made of latencies)
        volatile int *hs = D.host_status;
        hs[0] = *D.status;
        hs[1] = D.row_off[D.A];
        __threadfence_system();
    }
}
