// Tiled path for molecules the fused kernel cannot hold (n > 32: the 2220-atom protein, the 100k-atom box).
// Same arithmetic as the fused kernel (epnn_wave.hip.h), split into per-step kernels with the per-atom state in HBM:
//
//   GNN step (charge_gn.py:60-74)
//     k_lg_proj   P_i = Wi^T a_i + b1, R_j = Wj^T a_j, zp_i = relu(W2^T relu(P_i) + b2)       per 32-atom tile
//     k_lg_sweep  S0[chunk][i] = sum_{j in chunk} relu(W2^T relu(P_i + R_j) + b2)              ALL pairs, G ignored
//     k_lg_corr   for every listed pair: relu(W2^T relu(P_i+R_j+G)+b2) - relu(W2^T relu(P_i+R_j)+b2)   (both sides)
//     k_lg_update S_i = sum_chunk S0 + sum corr + (N-n) zp_i ; h_i = node_mask * update MLP([h_i | M_i])
//   EPN step (charge_gn.py:98-118)
//     k_lg_proj, k_lg_epn (pair tiles: 0.5*(f_ij - f_ji)), k_lg_apply (q_i += sum over the atom's pairs)
//
// The all-pairs sweep is the reference's semantics (charge_gn.py:70 sums over every j, near or not); the near
// pairs are <1% of the pairs of a large system, so adding their G-term as a correction costs ~1%.
// Both z2 evaluations of a correction use bit-identical inputs to the sweep's, so "without G" cancels exactly
// against the sweep's term up to the summation order.  All sums are order-fixed (no float atomics).
#pragma once
#include "epnn_host.h"
#include "epnn_common.h"

// G^T tile: acc[r] = G[pair c][kappa(hh,r)] = sum_ch We[ch][k] e[pair][ch]   (We fragments read from the packed weights)
__device__ __forceinline__ f32x16 lg_gtile(const float *__restrict__ weF, const float *__restrict__ erow, bool valid, int lane) {
    float ev[24];
    if (valid) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(erow + 4 * q);
            ev[4 * q] = v[0]; ev[4 * q + 1] = v[1]; ev[4 * q + 2] = v[2]; ev[4 * q + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int s = 0; s < 24; ++s) ev[s] = 0.f;
    }
    f32x16 acc = epnn_splat16(0.f);
#pragma unroll
    for (int s = 0; s < 24; ++s) acc = epnn_mfma(weF[s * 64 + lane], ev[s], acc);
    return acc;
}

struct LargeArgs {
    const float *wpack;
    WeightIndex wi;
    int nx, T, N, A, B;
    const int *moff, *mol_of, *mflag;       // mflag[b] != 0: molecule runs on this path
    const float *xin, *Q, *h_in, *q_in, *nm_in;
    float *a_eo;                            // [A][AST]
    float *P, *R;                           // [A][32] kappa-permuted
    float *zp;                              // [A][32]
    float *S0;                              // [maxchunk][A][32]
    float *corr;                            // [pcap][2][32]
    float *dl;                              // [pcap]
    const int *row_off, *pi, *pj, *psym;
    const float *pe, *pwi, *pwj;
    const int *dn_off, *dn_ent;             // per atom: pairs in which it is the second index
    const int4 *atiles;                     // (first atom, count, molecule, first chunk-independent S0 row) per 32-atom tile
    int natiles;
    const int4 *stasks;                     // sweep tasks: (first atile, natiles<=4, j_lo, j_hi | chunk<<... ) see host
    const int *stask_chunk;                 // chunk index of each sweep task
    int nstasks;
    int pcap;
    float *q_out, *h_out;
    int *status;
    int *host_status;                       // pinned host ints, or null: the last kernel of the EPN stack hands status + pair count over
    int *dn_cnt;
};

// ------------------------------------------------------------------------------------------------ init
__global__ __launch_bounds__(256) void k_lg_init(LargeArgs L) {
    const int fq = L.nx + EPNN_EDIM;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < L.A * EPNN_AST; idx += gridDim.x * 256) {
        const int at = idx / EPNN_AST, slot = idx - at * EPNN_AST;
        const int b = L.mol_of[at];
        if (!L.mflag[b]) continue;
        float v = 0.f;
        if (slot < 64) {
            const int f = (slot & 31) * 2 + (slot >> 5);
            if (f < L.nx) v = L.xin[(size_t)at * L.nx + f];
            else if (f < fq) v = L.h_in ? L.h_in[(size_t)at * EPNN_EDIM + (f - L.nx)] : 0.f;
            else if (f == fq) v = L.q_in ? L.q_in[at] : L.Q[b] / (float)(L.moff[b + 1] - L.moff[b]);
            else if (f == EPNN_F1) v = 1.f;                      // carries the first Dense's bias (Wi row 59 = b1)
        }
        L.a_eo[idx] = v;
    }
}

// ------------------------------------------------------------------------------------------------ "down" lists
// entry = pair slot; the top bit marks a one-sided entry (an arbitrary dense e / mask may list (i,j) without (j,i)): such an
// entry carries the EPN weight pwj but no GNN correction for its second atom
#define EPNN_DN_ONESIDED ((int)0x80000000)
#define EPNN_DN_SLOT 0x7fffffff
__global__ __launch_bounds__(256) void k_lg_dn_count(LargeArgs L) {
    const int np = L.row_off[L.A];
    if (np > L.pcap) return;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < np; p += gridDim.x * 256)
        if (L.psym[p] || L.pwj[p] != 0.f) atomicAdd(&L.dn_cnt[L.pj[p]], 1);
}
// exclusive scan dn_cnt[0..A) -> dn_off[0..A]; dn_cnt is reused as the fill cursor (reset to 0)
__global__ __launch_bounds__(1024) void k_lg_dn_scan(LargeArgs L, int *dn_off_w) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < L.A; base += 1024) {
        const int idx = base + tid;
        const int v = idx < L.A ? L.dn_cnt[idx] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int c0 = carry;
        if (idx < L.A) {
            dn_off_w[idx] = c0 + woff + incl - v;
            L.dn_cnt[idx] = 0;
        }
        __syncthreads();
        if (tid == 1023) carry = c0 + woff + incl;
        __syncthreads();
    }
    if (tid == 0) dn_off_w[L.A] = carry;
}
__global__ __launch_bounds__(256) void k_lg_dn_fill(LargeArgs L, int *dn_ent_w) {
    const int np = L.row_off[L.A];
    if (np > L.pcap) return;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < np; p += gridDim.x * 256)
        if (L.psym[p] || L.pwj[p] != 0.f) {
            const int j = L.pj[p];
            const int pos = atomicAdd(&L.dn_cnt[j], 1);
            dn_ent_w[L.dn_off[j] + pos] = p | (L.psym[p] ? 0 : EPNN_DN_ONESIDED);     // flag: no GNN correction through this entry
        }
}
// order every atom's list by pair slot so that sums over it have a fixed order: one thread per entry counts the entries
// of its list with a smaller slot and writes itself to that position of the second buffer (lists have ~12 entries; the
// insertion sort per atom this replaces was 17 us of dependent loads on the 2220-atom protein)
__global__ __launch_bounds__(256) void k_lg_dn_rank(LargeArgs L, const int *dn_in, int *dn_out) {
    if (L.row_off[L.A] > L.pcap) return;
    const int total = L.dn_off[L.A];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int v = dn_in[e], slot = v & EPNN_DN_SLOT;
        const int j = L.pj[slot];
        const int lo = L.dn_off[j], hi = L.dn_off[j + 1];
        int rank = 0;
        for (int k = lo; k < hi; ++k) rank += (dn_in[k] & EPNN_DN_SLOT) < slot ? 1 : 0;
        dn_out[lo + rank] = v;
    }
}

// ------------------------------------------------------------------------------------------------ projection
// one wave per 32-atom tile; `arow` = this lane's half (hh) of its atom's even/odd feature row (global a_eo or an LDS image).
// WHAT: 3 = P, R (and zp) by this wave; 1 = P (and zp) only; 2 = R only (the tail kernel gives the halves to two waves).
// `wA` = the wave's weight fragments (Wi for WHAT & 1, else Wj), already in registers when PRE.
template <int WHAT, bool PRE>
__device__ __forceinline__ void lg_proj_wave(const LargeArgs &L, const PairMlpPack &M, int with_zp, const int4 tl, const float *arow, int lane,
                                             const float (&wA)[EPNN_KA]) {
    const int c = lane & 31, hh = lane >> 5;
    const int at = tl.x + (c < tl.y ? c : 0);
    const float *wp = L.wpack;
    float bv[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(arow + 4 * q);
        bv[4 * q] = v[0]; bv[4 * q + 1] = v[1]; bv[4 * q + 2] = v[2]; bv[4 * q + 3] = v[3];
    }
    f32x16 accP = epnn_splat16(0.f), accR = epnn_splat16(0.f);
#pragma unroll
    for (int s = 0; s < EPNN_KA; ++s) {
        if (WHAT & 1) accP = epnn_mfma(PRE ? wA[s] : wp[M.wiF + s * 64 + lane], bv[s], accP);
        if (WHAT & 2) accR = epnn_mfma(PRE && !(WHAT & 1) ? wA[s] : wp[M.wjF + s * 64 + lane], bv[s], accR);
    }
    if (c < tl.y) {
        if (WHAT & 1) epnn_st16(L.P + (size_t)at * 32 + hh * 16, accP);
        if (WHAT & 2) epnn_st16(L.R + (size_t)at * 32 + hh * 16, accR);
    }
    if ((WHAT & 1) && with_zp) {
        // padded partner: R = 0, G = 0  ->  zp_i = relu(W2^T relu(P_i) + b2); rows = atoms, cols = out
        f32x16 acc = epnn_splat16(wp[M.b2 + c]);
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = epnn_mfma(fmaxf(accP[s], 0.f), wp[M.w2F + s * 64 + lane], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = epnn_kappa(hh, r);
            if (row < tl.y) L.zp[(size_t)(tl.x + row) * 32 + c] = fmaxf(acc[r], 0.f);
        }
    }
}
__global__ __launch_bounds__(256) void k_lg_proj(LargeArgs L, PairMlpPack M, int with_zp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, hh = lane >> 5;
    const int it = blockIdx.x * 4 + wave;
    if (it >= L.natiles) return;
    const int4 tl = L.atiles[it];
    const int at = tl.x + (c < tl.y ? c : 0);
    const float none[EPNN_KA] = {};
    lg_proj_wave<3, false>(L, M, with_zp, tl, L.a_eo + (size_t)at * EPNN_AST + hh * 32, lane, none);
}

// ------------------------------------------------------------------------------------------------ all-pairs sweep
// workgroup = up to 4 atom tiles (one per wave) x one j-chunk of the same molecule; R_j staged in LDS.
// v_mfma_f32_16x16x4_f32 in the fused kernel's layout (epnn_wave.hip.h): lane (q, n16) owns atoms n16 and 16 + n16 of
// the tile and features 16 rb + 4 q + r of each; tile j = "partner j of every atom", S accumulates in registers.  The
// co-resident wavefronts' VALU work (two adds and two max per feature and pair) overlaps this MFMA shape better than
// 32x32x2 (tools/micro/mfma_covalu.hip), which is what bounds the sweep.  W2 / b2 come from the fused kernel's pack.
#define EPNN_LG_JC 64
__device__ __forceinline__ void lg_sweep_body(const LargeArgs &L, int w2off, int b2off, float *Rs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, n16 = lane & 15, fo = 4 * q;
    const int4 tk = L.stasks[blockIdx.x];            // first atile, #atiles, j_lo, j_hi (global atom indices)
    const int chunk = L.stask_chunk[blockIdx.x];
    const bool active = wave < tk.y;
    const int4 tl = L.atiles[tk.x + (active ? wave : 0)];
    const float *wp = L.wpack;
    const bool two = tl.y > 16;                      // the tile's second column block holds atoms
    // P and R rows are stored in the operand order of the 32x32x2 kernels (k_lg_proj): position 16 hh + r holds feature
    // kappa(hh, r) = 8 (r >> 2) + 4 hh + (r & 3), so this lane's features 16 rb + 4 q .. + 3 sit together at po + 8 rb
    const int po = 16 * (q & 1) + 4 * (q >> 1);
    const int c0 = n16 < tl.y ? n16 : 0, c1 = 16 + n16 < tl.y ? 16 + n16 : 0;
    f32x4 P0[2], P1[2], S0[2], S1[2], b2v[2];
    float pb[2][8];
    W16_LD(pb, w2off, 2, 8);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        P0[rb] = w16_ld(L.P + (size_t)(tl.x + c0) * 32 + po + 8 * rb);
        P1[rb] = w16_ld(L.P + (size_t)(tl.x + c1) * 32 + po + 8 * rb);
        b2v[rb] = w16_ld(wp + b2off + 16 * rb + fo);
        S0[rb] = w16_splat(0.f);
        S1[rb] = w16_splat(0.f);
    }
    auto block = [&](const f32x4 (&Pc)[2], f32x4 (&Sc)[2], const f32x4 (&r)[2]) {
        const f32x4 za = w16_relu(Pc[0] + r[0]), zb = w16_relu(Pc[1] + r[1]);
        const float z[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
        f32x4 d[2] = {b2v[0], b2v[1]};
        w16_mm<2, 8>(pb, z, d);
        Sc[0] += w16_relu(d[0]);
        Sc[1] += w16_relu(d[1]);
    };
    for (int j0 = tk.z; j0 < tk.w; j0 += EPNN_LG_JC) {
        const int nj = min(EPNN_LG_JC, tk.w - j0);
        __syncthreads();
        for (int i = tid; i < nj * 32; i += 256) Rs[i] = L.R[(size_t)j0 * 32 + i];
        __syncthreads();
        if (active) {
            f32x4 ra[2] = {w16_ld(Rs + po), w16_ld(Rs + po + 8)};
            for (int j = 0; j < nj; ++j) {
                const float *nr = Rs + min(j + 1, nj - 1) * 32 + po;      // the next partner's row while this one is in the pipe
                const f32x4 rn[2] = {w16_ld(nr), w16_ld(nr + 8)};
                block(P0, S0, ra);
                if (two) block(P1, S1, ra);
                ra[0] = rn[0];
                ra[1] = rn[1];
            }
        }
    }
    if (!active) return;
    float *dst = L.S0 + ((size_t)chunk * L.A) * 32;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        if (n16 < tl.y) w16_st(dst + (size_t)(tl.x + n16) * 32 + 16 * rb + fo, S0[rb]);
        if (16 + n16 < tl.y) w16_st(dst + (size_t)(tl.x + 16 + n16) * 32 + 16 * rb + fo, S1[rb]);
    }
}
template <int MODE>
__device__ __forceinline__ void lg_pairs_body(const LargeArgs &L, const PairMlpPack &M, int blk);
// The sweep alone: ~110 registers, four wavefronts per SIMD -- what a large system's thousands of workgroups need.
__global__ __launch_bounds__(256) void k_lg_sweep(LargeArgs L, int w2off, int b2off) {
    __shared__ __attribute__((aligned(16))) float Rs[EPNN_LG_JC * 32];
    lg_sweep_body(L, w2off, b2off, Rs);
}
// The sweep plus, as extra workgroups, the near-pair correction tiles of the same step (both need only this step's P and
// R).  The pair tiles' registers halve the kernel's occupancy (179 registers: two wavefronts per SIMD), so this form is for
// systems whose sweep has at most two workgroups per CU anyway (the 2220-atom protein: 504), where it saves a launch per
// step; larger systems run the two kernels side by side on two streams.
__global__ __launch_bounds__(256) void k_lg_sweep_pairs(LargeArgs L, int w2off, int b2off, PairMlpPack Mpair) {
    __shared__ __attribute__((aligned(16))) float Rs[EPNN_LG_JC * 32];
    if ((int)blockIdx.x >= L.nstasks) {
        lg_pairs_body<0>(L, Mpair, (int)blockIdx.x - L.nstasks);
        return;
    }
    lg_sweep_body(L, w2off, b2off, Rs);
}

// ------------------------------------------------------------------------------------------------ pair tiles
// one wave per 32 listed pairs.  mode 0: GNN correction -> corr[p][side][32]; mode 1: EPN -> dl[p]
template <int MODE>
__device__ __forceinline__ void lg_pairs_body(const LargeArgs &L, const PairMlpPack &M, int blk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, hh = lane >> 5;
    const int np = L.row_off[L.A];
    if (np > L.pcap) return;
    const int slot = (blk * 4 + wave) * 32 + c;
    bool valid = slot < np;
    int gi = 0, gj = 0, sym = 0;
    if (valid) {
        gi = L.pi[slot];
        gj = L.pj[slot];
        sym = L.psym[slot];
        valid = L.mflag[L.mol_of[gi]] != 0;
    }
    if (__ballot(valid) == 0ull) return;
    const float *wp = L.wpack;
    const f32x16 g = lg_gtile(wp + M.weF, L.pe + (size_t)(valid ? slot : 0) * EPNN_EDIM + hh * 24, valid, lane);
    float pi_[16], rj_[16], pj_[16], ri_[16], w2[16];
    epnn_ld16(L.P + (size_t)gi * 32 + hh * 16, pi_);
    epnn_ld16(L.R + (size_t)gj * 32 + hh * 16, rj_);
    epnn_ld16(L.P + (size_t)gj * 32 + hh * 16, pj_);
    epnn_ld16(L.R + (size_t)gi * 32 + hh * 16, ri_);
#pragma unroll
    for (int s = 0; s < 16; ++s) w2[s] = wp[M.w2F + s * 64 + lane];
    if (MODE == 0) {
        // rows = pairs kappa(hh,r), cols = out c
        const f32x16 cb2 = epnn_splat16(wp[M.b2 + c]);
        f32x16 aG = cb2, a0 = cb2, bG = cb2, b0 = cb2;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float u0 = pi_[s] + rj_[s], v0 = pj_[s] + ri_[s];
            a0 = epnn_mfma(fmaxf(u0, 0.f), w2[s], a0);
            aG = epnn_mfma(fmaxf(u0 + g[s], 0.f), w2[s], aG);
            b0 = epnn_mfma(fmaxf(v0, 0.f), w2[s], b0);
            bG = epnn_mfma(fmaxf(v0 + g[s], 0.f), w2[s], bG);
        }
        const int base = (blk * 4 + wave) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = base + epnn_kappa(hh, r);
            if (p < np) {
                L.corr[((size_t)p * 2 + 0) * 32 + c] = fmaxf(aG[r], 0.f) - fmaxf(a0[r], 0.f);
                L.corr[((size_t)p * 2 + 1) * 32 + c] = fmaxf(bG[r], 0.f) - fmaxf(b0[r], 0.f);
            }
        }
    } else {
        float b2v[16], w3[16];
        epnn_ld16(wp + M.b2p + hh * 16, b2v);
        epnn_ld16(wp + M.w3p + hh * 16, w3);
        f32x16 au, av;
#pragma unroll
        for (int r = 0; r < 16; ++r) { au[r] = b2v[r]; av[r] = b2v[r]; }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            au = epnn_mfma(w2[s], fmaxf((g[s] + pi_[s]) + rj_[s], 0.f), au);
            av = epnn_mfma(w2[s], fmaxf((g[s] + pj_[s]) + ri_[s], 0.f), av);
        }
        float fu = 0.f, fv = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            fu = fmaf(w3[r], fmaxf(au[r], 0.f), fu);
            fv = fmaf(w3[r], fmaxf(av[r], 0.f), fv);
        }
        fu += epnn_swap32(fu);
        fv += epnn_swap32(fv);
        if (hh == 0 && valid) L.dl[slot] = 0.5f * (fu - fv);
    }
    (void)sym;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_lg_pairs(LargeArgs L, PairMlpPack M) {
    lg_pairs_body<MODE>(L, M, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------ update
// S_i = sum_chunk S0 + sum of the atom's corrections + (N-n) zp_i, one thread per (atom, out), fixed order.
// Written as its own wide launch: the update kernel has only natiles/4 workgroups, far too few to hide ~50
// dependent global loads per element.
// S of one (atom, output): chunk partials in chunk order, the atom's corrections as first index, as second index, padding
__device__ __forceinline__ float lg_reduce_elem(const LargeArgs &L, int at, int o, int n, int nchunk) {
    float s = 0.f;
    {   // eight loads in flight, added in chunk order (the order of the sum is part of the result)
        const float *src = L.S0 + (size_t)at * 32 + o;
        const size_t step = (size_t)L.A * 32;
        int ch = 0;
        for (; ch + 8 <= nchunk; ch += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(ch + u) * step];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; ch < nchunk; ++ch) s += src[(size_t)ch * step];
    }
    {
        const int p0 = L.row_off[at], p1 = L.row_off[at + 1];
        int p = p0;
        for (; p + 4 <= p1; p += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = L.corr[((size_t)(p + u) * 2 + 0) * 32 + o];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        for (; p < p1; ++p) s += L.corr[((size_t)p * 2 + 0) * 32 + o];
    }
    {   // the pairs in which this atom is the second one: entry -> pair -> correction, four chains in flight
        const int e0 = L.dn_off[at], e1 = L.dn_off[at + 1];
        int e = e0;
        for (; e + 4 <= e1; e += 4) {
            int pp[4];
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) pp[u] = L.dn_ent[e + u];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = L.corr[((size_t)(pp[u] & EPNN_DN_SLOT) * 2 + 1) * 32 + o];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (pp[u] >= 0) s += v[u];
        }
        for (; e < e1; ++e) {
            const int p = L.dn_ent[e];
            if (p >= 0) s += L.corr[((size_t)p * 2 + 1) * 32 + o];
        }
    }
    s += (float)(L.N - n) * L.zp[(size_t)at * 32 + o];
    return s;
}
// The same sums for four atoms of a tile at once (atoms a8, a8+8, a8+16, a8+24 of the tile, one output o): every stage keeps
// the loads of all four in flight together, each element still adds its own terms in the same order as lg_reduce_elem.
__device__ __forceinline__ void lg_reduce4(const LargeArgs &L, const int4 tl, int a8, int o, int n, float (&out)[4]) {
    const int nchunk = tl.w;
    int at[4];
    bool ok[4];
    float s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ok[k] = a8 + 8 * k < tl.y;
        at[k] = tl.x + (ok[k] ? a8 + 8 * k : 0);
        s[k] = 0.f;
    }
    // the list bounds of the later stages are requested now, they arrive under the chunk loads
    int p[4], p1[4], e[4], e1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        p[k] = L.row_off[at[k]];
        p1[k] = L.row_off[at[k] + 1];
        e[k] = L.dn_off[at[k]];
        e1[k] = L.dn_off[at[k] + 1];
    }
    {
        const size_t step = (size_t)L.A * 32;
        for (int ch = 0; ch < nchunk; ch += 16) {             // 64 loads of a thread in flight: a round trip per 16 chunks
            float v[4][16];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int u = 0; u < 16; ++u) v[k][u] = ch + u < nchunk ? L.S0[(size_t)at[k] * 32 + o + (size_t)(ch + u) * step] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (ch + u < nchunk) s[k] += v[k][u];
        }
    }
    {   // corrections of the pairs in which the atom is the first index
        int left = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) left = max(left, p1[k] - p[k]);
        for (; left > 0; left -= 8) {
            float v[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int u = 0; u < 8; ++u) v[k][u] = p[k] + u < p1[k] ? L.corr[((size_t)(p[k] + u) * 2 + 0) * 32 + o] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (p[k] + u < p1[k]) s[k] += v[k][u];
                p[k] += 8;
            }
        }
    }
    {   // ... and the second index: entry -> correction (one-sided entries carry none)
        int left = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) left = max(left, e1[k] - e[k]);
        for (; left > 0; left -= 8) {
            int pp[4][8];
            float v[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int u = 0; u < 8; ++u) pp[k][u] = e[k] + u < e1[k] ? L.dn_ent[e[k] + u] : EPNN_DN_ONESIDED;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int u = 0; u < 8; ++u) v[k][u] = L.corr[((size_t)(pp[k][u] & EPNN_DN_SLOT) * 2 + 1) * 32 + o];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (pp[k][u] >= 0) s[k] += v[k][u];
                e[k] += 8;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = ok[k] ? s[k] + (float)(L.N - n) * L.zp[(size_t)at[k] * 32 + o] : 0.f;
}
__global__ __launch_bounds__(256) void k_lg_reduce(LargeArgs L, float *Sfin) {
    if (L.row_off[L.A] > L.pcap) return;
    const int it = blockIdx.x;                 // one workgroup per 8 atoms of a tile list entry
    const int o = threadIdx.x & 31, a8 = threadIdx.x >> 5;
    const int4 tl = L.atiles[it >> 2];
    const int a = (it & 3) * 8 + a8;
    if (a >= tl.y) return;
    const int at = tl.x + a;
    Sfin[(size_t)at * 32 + o] = lg_reduce_elem(L, at, o, L.moff[tl.z + 1] - L.moff[tl.z], tl.w);
}

// The update MLP (charge_gn.py:71-74) of one 32-atom tile by one wave: weights w1 / w2 / w3 already in registers, the
// atom's h features from `arow` (its half hh of the even/odd feature row, offset to the first h slot), its reduced message
// sum from `srow` ([32] out-major).  New h goes to `dst` (a full even/odd row; global a_eo, and `dst2` when not null).
__device__ __forceinline__ void lg_update_wave(const LargeArgs &L, const UpdPack &U, const float (&w1)[40], const float (&w2)[16],
                                               const float (&w3)[32], const int4 tl, const float *arow, const float *srow,
                                               float *dst, float *dst2, int lane) {
    const int c = lane & 31, hh = lane >> 5;
    const bool live = c < tl.y;
    const int at = tl.x + (live ? c : 0);
    const int nx = L.nx;
    const float *wp = L.wpack;
    float hv[24], sv[16];
#pragma unroll
    for (int s = 0; s < 24; ++s) hv[s] = arow[s];
#pragma unroll
    for (int s = 0; s < 16; ++s) sv[s] = srow[2 * s + hh];
    float cb[16], b1[16];
    epnn_ld16(wp + U.cb3p + hh * 16, cb);
    epnn_ld16(wp + U.bu1p + hh * 16, b1);
    const float Nf = (float)L.N;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = Nf * cb[r];
#pragma unroll
    for (int s = 0; s < 24; ++s) acc = epnn_mfma(w1[s], hv[s], acc);
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = epnn_mfma(w1[24 + s], sv[s], acc);
    float u1[16], b2v[16];
    // node_mask (charge_gn.py:59,72,74): 1 for every real atom unless the dense front-end supplies one
    const float nmc = L.nm_in ? L.nm_in[at] : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) u1[r] = fmaxf(nmc * acc[r] + b1[r], 0.f);
    epnn_ld16(wp + U.bu2p + hh * 16, b2v);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b2v[r];
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = epnn_mfma(w2[s], u1[s], acc);
    float u2[16], b3a[16], b3b[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) u2[r] = fmaxf(acc[r], 0.f);
    epnn_ld16(wp + U.bu3p + hh * 16, b3a);
    epnn_ld16(wp + U.bu3p + 32 + hh * 16, b3b);
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = b3a[r]; o1[r] = b3b[r]; }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        o0 = epnn_mfma(w3[s], u2[s], o0);
        o1 = epnn_mfma(w3[16 + s], u2[s], o1);
    }
    if (live) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = epnn_kappa(hh, r);
            dst[epnn_aeo(nx + f)] = nmc * o0[r];
            if (r < 8) dst[epnn_aeo(nx + 32 + f)] = nmc * o1[r];
            if (dst2) {
                dst2[epnn_aeo(nx + f)] = nmc * o0[r];
                if (r < 8) dst2[epnn_aeo(nx + 32 + f)] = nmc * o1[r];
            }
        }
    }
}
#define LG_LOAD_UPD_WEIGHTS(w1, w2, w3, U)                                            \
    _Pragma("unroll") for (int s = 0; s < 40; ++s) w1[s] = wp[U.u1F + s * 64 + lane]; \
    _Pragma("unroll") for (int s = 0; s < 16; ++s) w2[s] = wp[U.u2F + s * 64 + lane]; \
    _Pragma("unroll") for (int s = 0; s < 32; ++s) w3[s] = wp[U.u3F + s * 64 + lane];

// workgroup = up to 4 atom tiles, one wave per tile runs the update MLP on the reduced S (used when a partition's exchange
// sits between the reduction and the update)
__global__ __launch_bounds__(256) void k_lg_update(LargeArgs L, UpdPack U, int maxchunk, const float *Sfin) {
    __shared__ float Ss[4 * 32 * EPNN_SST];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hh = lane >> 5;
    const int t0 = blockIdx.x * 4;
    const int np = L.row_off[L.A];
    if (np > L.pcap) return;
    // every weight fragment of the three layers is requested before anything else: the kernel has natiles/4 workgroups and
    // is pure latency, so the 88 loads travel while S is staged instead of one by one in front of their MFMAs
    const float *wp = L.wpack;
    float w1[40], w2[16], w3[32];
    LG_LOAD_UPD_WEIGHTS(w1, w2, w3, U)
    {   // stage S of the workgroup's four tiles: the tile descriptors first, then all sixteen loads of a thread in flight
        int4 tls[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) tls[w] = t0 + w < L.natiles ? L.atiles[t0 + w] : make_int4(0, 0, 0, 0);
        const int o = tid & 31, a8 = tid >> 5;
        float v[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int w = it >> 2, a = a8 + 8 * (it & 3);
            v[it] = a < tls[w].y ? Sfin[(size_t)(tls[w].x + a) * 32 + o] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) Ss[((it >> 2) * 32 + a8 + 8 * (it & 3)) * EPNN_SST + o] = v[it];
    }
    (void)maxchunk;
    __syncthreads();
    if (t0 + wave >= L.natiles) return;
    const int4 tl = L.atiles[t0 + wave];
    const int at = tl.x + (c < tl.y ? c : 0);
    const int u0 = (L.nx - hh + 1) >> 1;
    lg_update_wave(L, U, w1, w2, w3, tl, L.a_eo + (size_t)at * EPNN_AST + hh * 32 + u0, Ss + (wave * 32 + c) * EPNN_SST,
                   L.a_eo + (size_t)at * EPNN_AST, nullptr, lane);
}

// ---- one launch for everything between two sweeps (GNN) / two pair passes (EPN): workgroup = one 32-atom tile.
// GNN: all 256 threads reduce the tile's S (chunk partials, corrections, padding) into LDS, wave 0 runs the update MLP and
// then the NEXT step's projections from an LDS image of the tile's feature rows.  The three kernels this replaces
// (k_lg_reduce, k_lg_update, k_lg_proj) are 6-9 us of latency each on a 2220-atom system.  Same device functions, same
// operand values, same order: bit-identical to the separate launches (tests: partition == whole).
struct LgNext {
    PairMlpPack M;       // projections of the next sweep / pair pass
    int run;             // 0: nothing follows
    int with_zp;
};
__global__ __launch_bounds__(256) void k_lg_gnn_tail(LargeArgs L, UpdPack U, LgNext X) {
    __shared__ __attribute__((aligned(16))) float Ss[32 * EPNN_SST];
    __shared__ __attribute__((aligned(16))) float Ai[32 * EPNN_AST];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hh = lane >> 5;
    if (L.row_off[L.A] > L.pcap) return;
    const int4 tl = L.atiles[blockIdx.x];
    const float *wp = L.wpack;
    // every weight fragment is requested before the reduction: wave 0 the update MLP's, waves 1 / 2 the next projections'
    // Wi / Wj -- they travel while the partial sums are collected
    float w1[40], w2[16], w3[32], wA[EPNN_KA];
    if (wave == 0) { LG_LOAD_UPD_WEIGHTS(w1, w2, w3, U) }
    if (X.run && (wave == 1 || wave == 2)) {
        const int off = wave == 1 ? X.M.wiF : X.M.wjF;
#pragma unroll
        for (int s = 0; s < EPNN_KA; ++s) wA[s] = wp[off + s * 64 + lane];
    }
    for (int i = tid; i < tl.y * EPNN_AST; i += 256) Ai[i] = L.a_eo[(size_t)tl.x * EPNN_AST + i];
    {
        const int o = tid & 31, a8 = tid >> 5, n = L.moff[tl.z + 1] - L.moff[tl.z];
        float sv[4];
        lg_reduce4(L, tl, a8, o, n, sv);
#pragma unroll
        for (int k = 0; k < 4; ++k) Ss[(a8 + 8 * k) * EPNN_SST + o] = sv[k];
    }
    __syncthreads();
    const int row = c < tl.y ? c : 0;
    if (wave == 0) {
        const int u0 = (L.nx - hh + 1) >> 1;
        lg_update_wave(L, U, w1, w2, w3, tl, Ai + row * EPNN_AST + hh * 32 + u0, Ss + c * EPNN_SST,
                       L.a_eo + (size_t)(tl.x + row) * EPNN_AST, Ai + row * EPNN_AST, lane);
    }
    if (!X.run) return;
    __syncthreads();                                            // the image now holds the new h
    if (wave == 1) lg_proj_wave<1, true>(L, X.M, X.with_zp, tl, Ai + row * EPNN_AST + hh * 32, lane, wA);
    if (wave == 2) lg_proj_wave<2, true>(L, X.M, X.with_zp, tl, Ai + row * EPNN_AST + hh * 32, lane, wA);
}

// q_i += sum_j antisym_ij (charge_gn.py:118); thread per atom, fixed order (own row first, then the down list)
__device__ __forceinline__ float lg_apply_atom(const LargeArgs &L, int at) {
    float acc = 0.f;
    for (int e = L.dn_off[at]; e < L.dn_off[at + 1]; ++e) {
        const int p = L.dn_ent[e] & EPNN_DN_SLOT;
        acc -= L.pwj[p] * L.dl[p];
    }
    for (int p = L.row_off[at]; p < L.row_off[at + 1]; ++p) acc += L.pwi[p] * L.dl[p];
    return acc;
}
// status bits + number of listed pairs to the host (every other kernel of the forward ran before this one on the stream)
__device__ __forceinline__ void lg_handoff(const LargeArgs &L) {
    volatile int *hs = L.host_status;
    hs[0] = *L.status;
    hs[1] = L.row_off[L.A];
    __threadfence_system();
}
__global__ __launch_bounds__(256) void k_lg_apply(LargeArgs L, int last) {
    if (last && L.host_status && blockIdx.x == 0 && threadIdx.x == 0) lg_handoff(L);
    if (L.row_off[L.A] > L.pcap) return;
    const int fq = L.nx + EPNN_EDIM;
    for (int at = blockIdx.x * 256 + threadIdx.x; at < L.A; at += gridDim.x * 256) {
        if (!L.mflag[L.mol_of[at]]) continue;
        float *qp = L.a_eo + (size_t)at * EPNN_AST + epnn_aeo(fq);
        const float q = *qp + lg_apply_atom(L, at);
        *qp = q;
        if (last && L.q_out) L.q_out[at] = q;
    }
}
// EPN counterpart of k_lg_gnn_tail: workgroup (one wave) = one 32-atom tile: charge update of its atoms, then the next
// step's projections from the LDS image (k_lg_apply + k_lg_proj in one launch).
__global__ __launch_bounds__(64) void k_lg_epn_tail(LargeArgs L, LgNext X, int last) {
    __shared__ __attribute__((aligned(16))) float Ai[32 * EPNN_AST];
    const int lane = threadIdx.x, c = lane & 31, hh = lane >> 5;
    if (last && L.host_status && blockIdx.x == 0 && lane == 0) lg_handoff(L);
    if (L.row_off[L.A] > L.pcap) return;
    const int4 tl = L.atiles[blockIdx.x];
    const int fq = L.nx + EPNN_EDIM;
    for (int i = lane; i < tl.y * EPNN_AST; i += 64) Ai[i] = L.a_eo[(size_t)tl.x * EPNN_AST + i];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (lane < tl.y) {
        const int at = tl.x + lane;
        const float q = Ai[lane * EPNN_AST + epnn_aeo(fq)] + lg_apply_atom(L, at);
        Ai[lane * EPNN_AST + epnn_aeo(fq)] = q;
        L.a_eo[(size_t)at * EPNN_AST + epnn_aeo(fq)] = q;
        if (last && L.q_out) L.q_out[at] = q;
    }
    if (!X.run) return;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int row = c < tl.y ? c : 0;
    const float none[EPNN_KA] = {};
    lg_proj_wave<3, false>(L, X.M, X.with_zp, tl, Ai + row * EPNN_AST + hh * 32, lane, none);
}

__global__ __launch_bounds__(256) void k_lg_export_h(LargeArgs L) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < L.A * EPNN_EDIM; idx += gridDim.x * 256) {
        const int at = idx / EPNN_EDIM, f = idx - at * EPNN_EDIM;
        if (!L.mflag[L.mol_of[at]]) continue;
        L.h_out[idx] = L.a_eo[(size_t)at * EPNN_AST + epnn_aeo(L.nx + f)];
    }
}
__global__ __launch_bounds__(256) void k_lg_export_q(LargeArgs L) {
    const int fq = L.nx + EPNN_EDIM;
    for (int at = blockIdx.x * 256 + threadIdx.x; at < L.A; at += gridDim.x * 256) {
        if (!L.mflag[L.mol_of[at]]) continue;
        L.q_out[at] = L.a_eo[(size_t)at * EPNN_AST + epnn_aeo(fq)];
    }
}

// ================================================================================================ host side
struct LargePlanHost {
    std::vector<int4> atiles, stasks;
    std::vector<int> stask_chunk;
    int maxchunk = 0;
};

static int large_plan(epnn_handle *h) {
    Plan &P = h->plan;
    LargePlanHost lp;
    // row-block partition (epnn_set_partition): the tile groups of all tiled molecules, in atom order, are dealt out in
    // contiguous ranges; everything else of the plan -- tiles, j-chunks -- is the same on every process, so the partial
    // sums of an atom are the same numbers in the same order whoever computes them
    int total_groups = 0;
    for (int b : P.large_list) total_groups += ((P.offsets[b + 1] - P.offsets[b] + 31) / 32 + 3) / 4;
    const int g_own0 = (int)((long long)total_groups * h->part_rank / h->part_world);
    const int g_own1 = (int)((long long)total_groups * (h->part_rank + 1) / h->part_world);
    int gidx = 0;
    h->part_row_lo = P.A;
    h->part_row_hi = 0;
    h->part_lo.assign(h->part_world, P.A);
    h->part_hi.assign(h->part_world, 0);
    for (int b : P.large_list) {
        const int a0 = P.offsets[b], n = P.offsets[b + 1] - a0;
        const int first_tile = (int)lp.atiles.size();
        const int ntile = (n + 31) / 32, ngroup = (ntile + 3) / 4;
        // split the j range so that the sweep has a few thousand workgroups, in pieces of EPNN_LG_JC atoms
        // workgroups of the sweep = ngroup * nchunk.  Measured on the 2220-atom protein: a whole number of rounds of 512
        // workgroups (two per CU) beats 1024 smaller ones (per-workgroup prologue and more partial sums to reduce);
        // big systems get >= 2048 workgroups for a fine-grained tail.
        int want;
        if (ngroup >= 512) want = std::max(1, (4096 + ngroup - 1) / ngroup);
        else want = std::max(1, (int)std::lround(512.0 * std::max(1, (int)std::lround(ngroup / 64.0)) / ngroup));
        int nchunk = std::max(1, std::min(want, (n + 15) / 16));
        int clen = (n + nchunk - 1) / nchunk;
        nchunk = (n + clen - 1) / clen;
        for (int i0 = 0; i0 < n; i0 += 32) lp.atiles.push_back(make_int4(a0 + i0, std::min(32, n - i0), b, nchunk));
        lp.maxchunk = std::max(lp.maxchunk, nchunk);
        for (int tg = 0; tg < ntile; tg += 4, ++gidx) {
            {   // owner of this tile group (same formula on every process)
                int r = 0;
                while (r + 1 < h->part_world && gidx >= (int)((long long)total_groups * (r + 1) / h->part_world)) ++r;
                h->part_lo[r] = std::min(h->part_lo[r], a0 + tg * 32);
                h->part_hi[r] = std::max(h->part_hi[r], a0 + std::min(n, (tg + 4) * 32));
            }
            if (gidx < g_own0 || gidx >= g_own1) continue;
            h->part_row_lo = std::min(h->part_row_lo, a0 + tg * 32);
            h->part_row_hi = std::max(h->part_row_hi, a0 + std::min(n, (tg + 4) * 32));
            for (int ch = 0; ch < nchunk; ++ch) {
                lp.stasks.push_back(make_int4(first_tile + tg, std::min(4, ntile - tg), a0 + ch * clen,
                                              a0 + std::min(n, (ch + 1) * clen)));
                lp.stask_chunk.push_back(ch);
            }
        }
    }
    if (h->part_row_hi < h->part_row_lo) h->part_row_lo = h->part_row_hi = 0;        // no group of its own
    for (int r = 0; r < h->part_world; ++r)
        if (h->part_hi[r] < h->part_lo[r]) h->part_lo[r] = h->part_hi[r] = 0;
    h->l_natiles = (int)lp.atiles.size();
    h->l_nstasks = (int)lp.stasks.size();
    h->l_maxchunk = lp.maxchunk;
    if (P.large_list.empty()) return 0;          // (the molecule flags travel with the plan's other index arrays)
    const size_t A = (size_t)P.A;
    if (h->l_tiles.ensure(lp.atiles.size() * sizeof(int4)) || h->l_stasks.ensure(lp.stasks.size() * sizeof(int4)) ||
        h->l_schunk.ensure(lp.stask_chunk.size() * sizeof(int)) || h->l_a.ensure(A * EPNN_AST * 4) ||
        h->l_P.ensure(A * 32 * 4) || h->l_R.ensure(A * 32 * 4) || h->l_zp.ensure(A * 32 * 4) ||
        h->l_S0.ensure((size_t)lp.maxchunk * A * 32 * 4) || h->l_csr_off.ensure((A + 1) * sizeof(int)) ||
        h->l_cnt.ensure((A + 1) * sizeof(int)) || h->l_sfin.ensure(A * 32 * 4))
        return 1;
    HIPCHK(hipMemcpyAsync(h->l_tiles.p, lp.atiles.data(), lp.atiles.size() * sizeof(int4), hipMemcpyHostToDevice, h->stream));
    if (!lp.stasks.empty()) {
        HIPCHK(hipMemcpyAsync(h->l_stasks.p, lp.stasks.data(), lp.stasks.size() * sizeof(int4), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->l_schunk.p, lp.stask_chunk.data(), lp.stask_chunk.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

struct PairSource;
static int launch_large_impl(epnn_handle *h, const float *d_x, const float *d_Q, const float *d_hin, const float *d_qin,
                             const float *d_nm, float *d_q, float *d_hout, int run_gnn, int run_epn) {
    const Plan &P = h->plan;
    if (P.large_list.empty()) return 0;
    const size_t pc = (size_t)h->pcap;
    if (h->l_corr.ensure(pc * 2 * 32 * 4) || h->l_dl.ensure(pc * 4) || h->l_csr_ent.ensure(pc * sizeof(int)) ||
        h->l_csr_ent2.ensure(pc * sizeof(int)))
        return 1;
    LargeArgs L{};
    L.wpack = h->d_wpack.as<float>();
    L.wi = h->widx;
    L.nx = h->cfg.nx;
    L.T = h->cfg.T;
    L.N = P.N;
    L.A = P.A;
    L.B = P.B;
    L.moff = h->p_moff;
    L.mol_of = h->p_molof;
    L.mflag = h->p_mflag;
    L.xin = d_x;
    L.Q = d_Q;
    L.h_in = d_hin;
    L.q_in = d_qin;
    L.nm_in = d_nm;
    L.a_eo = h->l_a.as<float>();
    L.P = h->l_P.as<float>();
    L.R = h->l_R.as<float>();
    L.zp = h->l_zp.as<float>();
    L.S0 = h->l_S0.as<float>();
    L.corr = h->l_corr.as<float>();
    L.dl = h->l_dl.as<float>();
    L.row_off = h->d_rowoff.as<int>();
    L.pi = h->d_pi.as<int>();
    L.pj = h->d_pj.as<int>();
    L.psym = h->d_psym.as<int>();
    L.pe = h->d_pe.as<float>();
    L.pwi = h->d_pwi.as<float>();
    L.pwj = h->d_pwj.as<float>();
    L.dn_off = h->l_csr_off.as<int>();
    L.dn_ent = h->l_csr_ent.as<int>();
    L.dn_cnt = h->l_cnt.as<int>();
    L.atiles = h->l_tiles.as<int4>();
    L.natiles = h->l_natiles;
    L.stasks = h->l_stasks.as<int4>();
    L.stask_chunk = h->l_schunk.as<int>();
    L.nstasks = h->l_nstasks;
    L.pcap = h->pcap;
    // compact entry: the forward's last kernel hands status + pair count to the host (two device-to-host copies less)
    L.host_status = (h->want_large_handoff && run_epn && h->cfg.T > 0 && h->part_world == 1) ? h->h_status : nullptr;
    h->did_large_handoff = L.host_status != nullptr;
    L.q_out = d_q;
    L.h_out = d_hout;
    L.status = h->d_status.as<int>();
    hipStream_t st = h->stream;
    const unsigned gA = (unsigned)std::min<size_t>(((size_t)P.A * EPNN_AST + 255) / 256, 4096);
    const unsigned gP = (unsigned)std::min<size_t>((pc + 255) / 256, 4096);
    const unsigned gAt = (unsigned)std::min<size_t>(((size_t)P.A + 255) / 256, 4096);
    const unsigned gT = (unsigned)((L.natiles + 3) / 4);
    const unsigned gPT = (unsigned)((pc + 127) / 128);
    hipLaunchKernelGGL(k_lg_init, dim3(gA), dim3(256), 0, st, L);
    HIPCHK(hipMemsetAsync(L.dn_cnt, 0, ((size_t)P.A + 1) * sizeof(int), st));
    hipLaunchKernelGGL(k_lg_dn_count, dim3(gP), dim3(256), 0, st, L);
    hipLaunchKernelGGL(k_lg_dn_scan, dim3(1), dim3(1024), 0, st, L, h->l_csr_off.as<int>());
    hipLaunchKernelGGL(k_lg_dn_fill, dim3(gP), dim3(256), 0, st, L, h->l_csr_ent2.as<int>());
    hipLaunchKernelGGL(k_lg_dn_rank, dim3(gP), dim3(256), 0, st, L, h->l_csr_ent2.as<int>(), h->l_csr_ent.as<int>());
    // Launch sequence (the single-process case): proj(0) | per GNN step: sweep + correction tiles, tail (reduce, update,
    // next projections) | per EPN step: pair tiles, tail (charge update, next projections).  With a partition the other
    // processes' rows of S arrive between the reduction and the update, so those stay separate launches.
    const bool collective = h->part_world > 1 || h->opt_part_collective;
    const bool split = collective || !h->opt_large_fused;
    const int Tg = run_gnn ? L.T : 0, Te = run_epn ? L.T : 0;
    const unsigned gTile = (unsigned)L.natiles;
    auto next_after_gnn = [&](int t) {
        LgNext X{};
        if (t + 1 < Tg) { X.M = h->widx.msg[t + 1]; X.run = 1; X.with_zp = 1; }
        else if (Te > 0) { X.M = h->widx.pas[0]; X.run = 1; X.with_zp = 0; }
        return X;
    };
    if (Tg > 0) hipLaunchKernelGGL(k_lg_proj, dim3(gT), dim3(256), 0, st, L, h->widx.msg[0], 1);
    else if (Te > 0) hipLaunchKernelGGL(k_lg_proj, dim3(gT), dim3(256), 0, st, L, h->widx.pas[0], 0);
    for (int t = 0; t < Tg; ++t) {
        if (split) {
            if (L.nstasks > 0)
                hipLaunchKernelGGL(k_lg_sweep, dim3((unsigned)L.nstasks), dim3(256), 0, st, L, h->wvidx.g[t].w2, h->wvidx.g[t].b2);
            hipLaunchKernelGGL(k_lg_pairs<0>, dim3(gPT), dim3(256), 0, st, L, h->widx.msg[t]);
            hipLaunchKernelGGL(k_lg_reduce, dim3((unsigned)L.natiles * 4), dim3(256), 0, st, L, h->l_sfin.as<float>());
            if (collective) {
                // the other processes' rows of S (this one's all-pairs sums are complete only for its own atoms); everything
                // after this point is computed by every process for every atom
                HIPCHK(hipGetLastError());
                if (h->part_exchange) {                       // the caller's exchange (host-staged: gloo tests, no communicator)
                    HIPCHK(hipStreamSynchronize(st));
                    if (h->part_exchange(h->part_ctx, h->l_sfin.as<float>(), 32, P.A, h->part_row_lo, h->part_row_hi))
                        EPNN_FAIL("forward: the partition's exchange function reported an error");
                } else {
                    // all-gather of unequal row ranges on the handle's stream: every process broadcasts its own rows in
                    // place, grouped into one RCCL operation (xGMI is point to point: `world` concurrent broadcasts use every
                    // link at once); no host synchronisation, the update kernel simply follows on the stream
                    if (!h->comm || h->comm_world != h->part_world) EPNN_FAIL("forward: a partition is set but neither an exchange function nor a communicator");
                    ncclResult_t rc = ncclGroupStart();
                    for (int r = 0; r < h->part_world && rc == ncclSuccess; ++r) {
                        const size_t cnt = (size_t)(h->part_hi[r] - h->part_lo[r]) * 32;
                        if (cnt == 0) continue;
                        float *rows = h->l_sfin.as<float>() + (size_t)h->part_lo[r] * 32;
                        rc = ncclBroadcast(rows, rows, cnt, ncclFloat, r, h->comm, st);
                    }
                    const ncclResult_t rc2 = ncclGroupEnd();
                    if (rc != ncclSuccess || rc2 != ncclSuccess)
                        EPNN_FAIL("forward: RCCL row exchange failed: %s", ncclGetErrorString(rc != ncclSuccess ? rc : rc2));
                }
            }
            hipLaunchKernelGGL(k_lg_update, dim3(gT), dim3(256), 0, st, L, h->widx.upd[t], h->l_maxchunk, h->l_sfin.as<float>());
            const LgNext X = next_after_gnn(t);
            if (X.run) hipLaunchKernelGGL(k_lg_proj, dim3(gT), dim3(256), 0, st, L, X.M, X.with_zp);
        } else {
            if (L.nstasks > 0 && L.nstasks <= 512) {
                hipLaunchKernelGGL(k_lg_sweep_pairs, dim3((unsigned)L.nstasks + gPT), dim3(256), 0, st, L, h->wvidx.g[t].w2, h->wvidx.g[t].b2, h->widx.msg[t]);
                hipLaunchKernelGGL(k_lg_gnn_tail, dim3(gTile), dim3(256), 0, st, L, h->widx.upd[t], next_after_gnn(t));
                continue;
            }
            // correction tiles beside the sweep (both need only this step's P and R): fork to the second stream, join before the tail
            HIPCHK(hipEventRecord(h->ev_fork, st));
            HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
            hipLaunchKernelGGL(k_lg_pairs<0>, dim3(gPT), dim3(256), 0, h->stream2, L, h->widx.msg[t]);
            HIPCHK(hipEventRecord(h->ev_join, h->stream2));
            if (L.nstasks > 0)
                hipLaunchKernelGGL(k_lg_sweep, dim3((unsigned)L.nstasks), dim3(256), 0, st, L, h->wvidx.g[t].w2, h->wvidx.g[t].b2);
            HIPCHK(hipStreamWaitEvent(st, h->ev_join, 0));
            hipLaunchKernelGGL(k_lg_gnn_tail, dim3(gTile), dim3(256), 0, st, L, h->widx.upd[t], next_after_gnn(t));
        }
    }
    if (d_hout) hipLaunchKernelGGL(k_lg_export_h, dim3(gA), dim3(256), 0, st, L);
    for (int t = 0; t < Te; ++t) {
        hipLaunchKernelGGL(k_lg_pairs<1>, dim3(gPT), dim3(256), 0, st, L, h->widx.pas[t]);
        LgNext X{};
        if (t + 1 < Te) { X.M = h->widx.pas[t + 1]; X.run = 1; X.with_zp = 0; }
        if (split) {
            hipLaunchKernelGGL(k_lg_apply, dim3(gAt), dim3(256), 0, st, L, t + 1 == Te);
            if (X.run) hipLaunchKernelGGL(k_lg_proj, dim3(gT), dim3(256), 0, st, L, X.M, 0);
        } else {
            hipLaunchKernelGGL(k_lg_epn_tail, dim3(gTile), dim3(64), 0, st, L, X, t + 1 == Te);
        }
    }
    if (d_q && Te == 0) hipLaunchKernelGGL(k_lg_export_q, dim3(gAt), dim3(256), 0, st, L);
    HIPCHK(hipGetLastError());
    return 0;
}
