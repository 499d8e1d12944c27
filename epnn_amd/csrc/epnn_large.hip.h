// Tiled path for molecules the fused kernels cannot hold (the 2220-atom protein, the 100k-atom box; any size through the
// dense / layer-level entries beyond 32 atoms).  Same arithmetic as the fused kernel (epnn_wave.hip.h), split into per-step
// kernels with the per-atom state in HBM:
//
//   GNN step (charge_gn.py:60-74)
//     projections   P_i = Wi^T a_i + b1, R_j = Wj^T a_j, zp_i = relu(W2^T relu(P_i) + b2); for the sweep also
//                   Nn_j = -R_j and Yb_j = b2 + W2^T R_j                                                per 32-atom tile
//     k_lg_sweep    S0[chunk][i] = sum_{j in chunk} relu(W2^T relu(P_i + R_j) + b2)                     ALL pairs, G ignored
//                   evaluated as relu(W2^T max(P_i, Nn_j) + Yb_j): relu(P + R) = max(P, -R) + R, and the R part of the
//                   product is the same for every i -- one VALU instruction per element where add + max were two (f32 MFMA
//                   and VALU share the SIMD's issue cycles: tools/micro/sweep_mix.hip)
//     k_lg_tsweep   FIRST step of the compact entry (h = 0, q = Q/n): a_i depends on the atom's feature row x_i alone, so
//                   P_i + R_j takes only (distinct rows)^2 values: S_i = sum over TYPES tau of count(tau) z2(P_i + R_tau),
//                   a few hundred pair evaluations instead of n^2 (k_lg_types groups the atoms; exact algebra)
//     k_lg_pairs<0> for every listed pair: relu(W2^T relu(P_i+R_j+G)+b2) - relu(W2^T relu(P_i+R_j)+b2)   (both sides),
//                   deposited in the two atoms' incidence slots (epnn_frontend.hip.h)
//     tail          S_i = sum_chunk S0 + the atom's slot row + (N-n) zp_i ; h_i = node_mask * update MLP([h_i | M_i]);
//                   next projections
//   EPN stack (charge_gn.py:98-118): h is fixed, only q changes between steps, so the projections of all T steps are
//   taken ONCE with q = 0 (Pst_t, Rst_t) and a pair tile adds q_i w_q itself.  One launch per step: a pair's two lanes
//   rebuild q_i, q_j of this step from the previous step's transfers in the atoms' slot rows (the first slot's pair
//   stores the new q), evaluate 0.5 (f_ij - f_ji) and deposit +-w delta; a last launch adds the final transfers.
//
// The all-pairs sweep is the reference's semantics (charge_gn.py:70 sums over every j, near or not); the near
// pairs are <1% of the pairs of a large system, so adding their G-term as a correction costs ~1%.
// All sums are order-fixed (no float atomics): results are bit-reproducible, and a row-block partition of the sweep over
// several processes reproduces the single-process bits.
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

#ifdef EPNN_LG_CLOCKS
// development build (tools/large_clocks.py): 100 MHz wall clock of workgroup 0 at phase boundaries of the tail and EPN-step launches
#define LG_CLK(base, k) do { if (L.clk && blockIdx.x == 0 && threadIdx.x == 0) L.clk[(base) + (k)] = wall_clock64(); } while (0)
// ... of the workgroup for which `cond` holds (the merged launches of the compact entry: one workgroup of every kind of work)
#define LG_CLKB(cond, slot) do { if (L.clk && (cond) && threadIdx.x == 0) L.clk[slot] = wall_clock64(); } while (0)
#else
#define LG_CLK(base, k) do { } while (0)
#define LG_CLKB(cond, slot) do { } while (0)
#endif
struct LargeArgs {
#ifdef EPNN_LG_CLOCKS
    unsigned long long *clk;
#endif
    const float *wpack;
    WeightIndex wi;
    int nx, T, N, A, B;
    const int *moff, *mol_of, *mflag;       // mflag[b] != 0: molecule runs on this path
    const float *xin, *Q, *h_in, *q_in, *nm_in;
    float *a_eo;                            // [A][AST]
    float *P, *R;                           // [A][32] kappa-permuted
    float *Nn;                              // [A][32] -R, kappa-permuted            (GNN projections)
    float *Yb;                              // [A][32] b2 + W2^T R, natural order     (GNN projections)
    float *zp;                              // [A][32]
    float *S0;                              // [maxchunk][A][32]
    float *corrA;                           // [2 pcap][32] GNN corrections by incidence slot
    float *dlA;                             // [2][2 pcap] EPN transfers by incidence slot (two steps alive)
    float *qbuf;                            // [2][A]      charges of the EPN steps (two steps alive)
    float *Pst, *Rst;                       // [T][A][32]  EPN projections with q = 0, kappa-permuted
    const int *row_off, *pi, *pj, *psym;
    const float *pe, *pwi, *pwj;
    const int *inc_off, *dest_i, *dest_j;   // incidence rows and a pair's two slots (dest_j < 0: one-sided entry)
    const int4 *prec;                       // [2 pcap] (i, j, lo_i, hi_i), (lo_j, hi_j, dest_i, dest_j)
    const int4 *atiles;                     // (first atom, count, molecule, chunks of its molecule's sweep) per 32-atom tile
    int natiles;
    const int4 *stasks;                     // sweep tasks: (first atile, natiles<=4, j_lo, j_hi)
    const int *stask_chunk;                 // chunk index of each sweep task
    const float2 *stask_frac;               // share [lo, hi) of the step's correction tiles each sweep task's wavefronts run
    int nstasks;
    // tile-workgroup sweep (k_lg_sweep2: molecules of at most EPNN_LG_TW_TILES tiles)
    const int4 *stasks2;                    // (atile or -1: correction tiles only, index of its partial sum, j_lo, j_hi)
    const float2 *stask2_frac;              // share [lo, hi) of the step's correction tiles of each of these workgroups
    int nstasks2;
    int pcap;
    // first GNN step by atom types (k_lg_types)
    const int *lmol;                        // [nlarge] molecules on this path
    int nlarge;
    int *typ_row;                           // [A]  row of S_type the atom reads: EPNN_TYPE_MAX * (molecule's place in lmol) + its type
    int *typ_rep, *typ_cnt;                 // [nlarge * EPNN_TYPE_MAX] first atom of a type, atoms of it
    int *typ_n;                             // [nlarge] types of the molecule
    unsigned long long *typ_hash;           // [A] hash of the atom's feature row (merged launches: hashed where the rows are built)
    unsigned long long *typ_hsh;            // [nlarge * EPNN_TYPE_MAX] hash of a type's row
    float *S_type;                          // [nlarge * EPNN_TYPE_MAX][32]
    float *q_out, *h_out;
    int *status;
    int *host_status;                       // pinned host ints, or null: the last kernel of the EPN stack hands status + pair count over
    // builders of the incidence rows for pair lists that come from the dense front-end
    int *dn_cnt, *dn_cur, *dn_off, *dn_ent;
    int *w_inc_off, *w_dest_i, *w_dest_j;
    int4 *w_prec;
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
            else if (f == fq) {
                v = L.q_in ? L.q_in[at] : L.Q[b] / (float)(L.moff[b + 1] - L.moff[b]);
                L.qbuf[at] = v;                                  // the EPN stack's charges, both generations
                L.qbuf[(size_t)L.A + at] = v;
            }
            else if (f == EPNN_F1) v = 1.f;                      // carries the first Dense's bias (Wi row 59 = b1)
        }
        L.a_eo[idx] = v;
    }
}

// ------------------------------------------------------------------------------------------------ incidence rows of a dense list
// Pair lists that come from the dense front-end (arbitrary e / mask; an entry may be one-sided: (i,j) listed without a
// matching (j,i)) have no incidence rows yet.  An atom's row = its pairs as first index, in list order, then the symmetric
// pairs in which it is the second index, ordered by pair slot:
//   inc_off[a] = row_off[a] + dn_off[a],  dest_i(p) = dn_off[i] + p,  dest_j(p) = row_off[j+1] + dn_off[j] + rank of p
__global__ __launch_bounds__(256) void k_lg_dn_count(LargeArgs L) {
    const int np = L.row_off[L.A];
    if (np > L.pcap) return;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < np; p += gridDim.x * 256)
        if (L.psym[p]) atomicAdd(&L.dn_cnt[L.pj[p]], 1);
}
// exclusive scan dn_cnt[0..A) -> dn_off[0..A]
__global__ __launch_bounds__(1024) void k_lg_dn_scan(LargeArgs L) {
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
            const int d = c0 + woff + incl - v;
            L.dn_off[idx] = d;
            L.w_inc_off[idx] = L.row_off[idx] + d;
        }
        __syncthreads();
        if (tid == 1023) carry = c0 + woff + incl;
        __syncthreads();
    }
    if (tid == 0) {
        L.dn_off[L.A] = carry;
        L.w_inc_off[L.A] = L.row_off[L.A] + carry;
    }
}
// the symmetric pairs of every second atom, in arrival order (dn_cnt, zeroed again by nobody: the scan read it, this pass
// counts up from dn_cur = a second zeroed array)
__global__ __launch_bounds__(256) void k_lg_dn_fill(LargeArgs L) {
    const int np = L.row_off[L.A];
    if (np > L.pcap) return;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < np; p += gridDim.x * 256)
        if (L.psym[p]) {
            const int j = L.pj[p];
            L.dn_ent[L.dn_off[j] + atomicAdd(&L.dn_cur[j], 1)] = p;
        }
}
// one thread per pair: its two slots and its record.  A symmetric pair's place in its second atom's row = how many of that
// atom's symmetric pairs have a smaller pair slot (its list is short: the arrival order above is made a fixed one here)
__global__ __launch_bounds__(256) void k_lg_dn_link(LargeArgs L) {
    const int np = L.row_off[L.A];
    if (np > L.pcap) return;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < np; p += gridDim.x * 256) {
        const int i = L.pi[p], j = L.pj[p];
        const int di = L.dn_off[i] + p;
        int dj = -1;
        if (L.psym[p]) {
            int rank = 0;
            for (int k = L.dn_off[j]; k < L.dn_off[j + 1]; ++k) rank += L.dn_ent[k] < p ? 1 : 0;
            dj = L.row_off[j + 1] + L.dn_off[j] + rank;
        }
        L.w_dest_i[p] = di;
        L.w_dest_j[p] = dj;
        const int iv = L.mflag[L.mol_of[i]] ? i : -1 - i;         // pairs of molecules on the fused path are skipped by the tiles
        L.w_prec[2 * p] = make_int4(iv, j, L.row_off[i] + L.dn_off[i], L.row_off[i + 1] + L.dn_off[i + 1]);
        L.w_prec[2 * p + 1] = make_int4(L.row_off[j] + L.dn_off[j], L.row_off[j + 1] + L.dn_off[j + 1], di, dj);
    }
}

// ------------------------------------------------------------------------------------------------ atom types of the first step
// One workgroup per molecule: atoms with bit-identical feature rows x share a type.  An open-addressed table in LDS keyed by
// a hash of the row; the row of a slot's first atom is compared bit for bit afterwards, so a hash collision cannot merge two
// types (it raises EPNN_ST_TYPE_OVERFLOW like a full table does: the host repeats the forward with the all-pairs sweep).
// Types are numbered by their first atom, so the order of every sum over types is fixed.
#define EPNN_TYPE_SLOTS 256
struct LgTypeShared {
    unsigned long long key[EPNN_TYPE_SLOTS];
    int first[EPNN_TYPE_SLOTS], cnt[EPNN_TYPE_SLOTS], num[EPNN_TYPE_SLOTS], list[EPNN_TYPE_SLOTS];
    int bad, used;
};
__device__ __forceinline__ unsigned long long lg_hash_step(unsigned long long hsh, float v) {
    return (hsh ^ (unsigned long long)__float_as_uint(v)) * 1099511628211ull;
}
__device__ __forceinline__ unsigned long long lg_hash_row(const LargeArgs &L, int at) {
    unsigned long long hsh = 1469598103934665603ull;
    for (int f = 0; f < L.nx; ++f) hsh = lg_hash_step(hsh, L.xin[(size_t)at * L.nx + f]);
    return hsh | 1ull;                                        // 0 marks an empty slot
}
__device__ __forceinline__ int lg_type_find(LgTypeShared &T, unsigned long long hsh, bool insert) {
    int s = (int)((hsh >> 17) & (EPNN_TYPE_SLOTS - 1));
    for (int probe = 0; probe < EPNN_TYPE_SLOTS; ++probe) {
        unsigned long long cur = T.key[s];
        if (cur == 0ull && insert) {
            cur = atomicCAS(&T.key[s], 0ull, hsh);
            if (cur == 0ull) {                                // this thread opened the slot
                T.list[atomicAdd(&T.used, 1)] = s;
                cur = hsh;
            }
        }
        if (cur == hsh) return s;
        if (cur == 0ull) return -1;
        s = (s + 1) & (EPNN_TYPE_SLOTS - 1);
    }
    return -1;
}
// the table of molecule lm: its distinct row hashes, the first atom and the number of atoms of each, numbered by first atom.
// HASHED: the atoms' hashes are in L.typ_hash already.
template <int NT, bool HASHED>
__device__ __forceinline__ void lg_types_table(const LargeArgs &L, LgTypeShared &T, int lm) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = L.lmol[lm];
    const int a0 = L.moff[b], n = L.moff[b + 1] - a0;
    for (int s = tid; s < EPNN_TYPE_SLOTS; s += NT) { T.key[s] = 0ull; T.first[s] = 0x7fffffff; T.cnt[s] = 0; }
    if (tid == 0) { T.bad = 0; T.used = 0; }
    __syncthreads();
    // A thread holds eight atoms' hashes at a time (one round trip for all of them), and a wavefront's 512 atoms of EQUAL hash are
    // entered by ONE lane in ONE round: a molecule has a handful of types, so a wavefront needs a handful of rounds per 512 atoms
    // (lane-by-lane atomics on five LDS words serialise the workgroup; a round per 64 atoms and type was 14 us for the 2220-atom
    // protein on a 256-thread workgroup)
    for (int kb = 0; kb < n; kb += 8 * NT) {
        unsigned long long hv[8];
        bool act[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = kb + u * NT + tid;
            act[u] = k < n;
            hv[u] = HASHED ? L.typ_hash[a0 + min(k, n - 1)] : (act[u] ? lg_hash_row(L, a0 + k) : 0ull);
        }
        for (;;) {
            // this lane's first hash still to be entered; the lowest lane that has one leads the round
            unsigned long long cand = 0ull;
            bool any = false;
#pragma unroll
            for (int u = 7; u >= 0; --u)
                if (act[u]) { cand = hv[u]; any = true; }
            const unsigned long long havers = __ballot(any);
            if (havers == 0ull) break;
            const int leader = __ffsll((long long)havers) - 1;
            const unsigned long long lh = __shfl(cand, leader, 64);
            int cnt = 0, first = 0x7fffffff;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool m = act[u] && hv[u] == lh;
                const unsigned long long bal = __ballot(m);
                cnt += __popcll(bal);
                if (bal != 0ull && first == 0x7fffffff) first = a0 + kb + u * NT + (tid & ~63) + (__ffsll((long long)bal) - 1);
                if (m) act[u] = false;
            }
            if (lane == leader) {
                const int sl = lg_type_find(T, lh, true);
                if (sl < 0) atomicOr(&T.bad, 1);
                else {
                    atomicMin(&T.first[sl], first);
                    atomicAdd(&T.cnt[sl], cnt);
                }
            }
        }
    }
    __syncthreads();
    const int nu = T.used;
    // number the types by their first atom
    for (int u0 = tid; u0 < nu; u0 += NT) {
        const int s = T.list[u0];
        int rank = 0;
        for (int u = 0; u < nu; ++u) rank += T.first[T.list[u]] < T.first[s] ? 1 : 0;
        T.num[s] = rank;
    }
    if (tid == 0 && nu > EPNN_TYPE_MAX) T.bad = 1;
    __syncthreads();
    for (int u0 = tid; u0 < nu && !T.bad; u0 += NT) {
        const int s = T.list[u0];
        L.typ_rep[lm * EPNN_TYPE_MAX + T.num[s]] = T.first[s];
        L.typ_cnt[lm * EPNN_TYPE_MAX + T.num[s]] = T.cnt[s];
        L.typ_hsh[lm * EPNN_TYPE_MAX + T.num[s]] = T.key[s];
    }
    if (tid == 0) {
        L.typ_n[lm] = T.bad ? 0 : nu;
        if (T.bad) atomicOr(L.status, EPNN_ST_TYPE_OVERFLOW);
    }
}
// an atom's type = the table entry with its hash; its row is compared bit for bit with the entry's first atom's
__device__ __forceinline__ void lg_type_assign(const LargeArgs &L, int at, int lm, unsigned long long hsh) {
    const int nu = L.typ_n[lm];
    int tau = -1;
    for (int u = 0; u < nu; ++u)
        if (L.typ_hsh[lm * EPNN_TYPE_MAX + u] == hsh) tau = u;
    bool ok = tau >= 0;
    if (ok) {
        const int rep = L.typ_rep[lm * EPNN_TYPE_MAX + tau];
        for (int f = 0; f < L.nx; ++f)
            ok &= __float_as_uint(L.xin[(size_t)at * L.nx + f]) == __float_as_uint(L.xin[(size_t)rep * L.nx + f]);
    }
    if (!ok && nu > 0) atomicOr(L.status, EPNN_ST_TYPE_OVERFLOW);
    L.typ_row[at] = lm * EPNN_TYPE_MAX + (ok ? tau : 0);
}
template <int NT>
__device__ __forceinline__ void lg_types_body(const LargeArgs &L, LgTypeShared &T, int lm) {
    lg_types_table<NT, false>(L, T, lm);
    __threadfence_block();
    __syncthreads();                                         // the table is in global memory: this workgroup reads it back
    const int b = L.lmol[lm];
    const int a0 = L.moff[b], n = L.moff[b + 1] - a0;
    for (int k = threadIdx.x; k < n; k += NT) lg_type_assign(L, a0 + k, lm, lg_hash_row(L, a0 + k));
}
__global__ __launch_bounds__(1024) void k_lg_types(LargeArgs L) {
    __shared__ LgTypeShared T;
    lg_types_body<1024>(L, T, (int)blockIdx.x);
}

// S_type[sigma][o] = sum_tau count(tau) relu(W2^T relu(P_sigma + R_tau) + b2)[o]: one wave per 32 types sigma of a molecule.
// rows = types kappa(hh,r), cols = out c (the layout of the correction tiles below)
__device__ __forceinline__ void lg_tsweep_wave(const LargeArgs &L, const PairMlpPack &M, int job) {
    const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
    const int lm = job >> 1, blk = job & 1;
    const int U = L.typ_n[lm];
    if (32 * blk >= U) return;
    const float *wp = L.wpack;
    const int sig = 32 * blk + c;
    const int rep = L.typ_rep[lm * EPNN_TYPE_MAX + (sig < U ? sig : 0)];
    float pr[16], w2[16];
    epnn_ld16(L.P + (size_t)rep * 32 + hh * 16, pr);
#pragma unroll
    for (int s = 0; s < 16; ++s) w2[s] = wp[M.w2F + s * 64 + lane];
    const f32x16 cb2 = epnn_splat16(wp[M.b2 + c]);
    f32x16 S = epnn_splat16(0.f);
    for (int tau = 0; tau < U; ++tau) {
        const int rt = L.typ_rep[lm * EPNN_TYPE_MAX + tau];
        const float cnt = (float)L.typ_cnt[lm * EPNN_TYPE_MAX + tau];
        float rr[16];
        epnn_ld16(L.R + (size_t)rt * 32 + hh * 16, rr);
        f32x16 a = cb2;
#pragma unroll
        for (int s = 0; s < 16; ++s) a = epnn_mfma(fmaxf(pr[s] + rr[s], 0.f), w2[s], a);
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = fmaf(cnt, fmaxf(a[r], 0.f), S[r]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * blk + epnn_kappa(hh, r);
        if (row < U) L.S_type[((size_t)lm * EPNN_TYPE_MAX + row) * 32 + c] = S[r];
    }
}
__global__ __launch_bounds__(64) void k_lg_tsweep(LargeArgs L, PairMlpPack M) { lg_tsweep_wave(L, M, (int)blockIdx.x); }

// ------------------------------------------------------------------------------------------------ projection
// one wave per 32-atom tile; `arow` = this lane's half (hh) of its atom's even/odd feature row (global a_eo or an LDS image).
// WHAT: 3 = P, R (and zp, Nn, Yb) by this wave; 1 = P (and zp) only; 2 = R (and Nn, Yb) only (the tail kernel gives the halves
// to two waves).  `wA` = the wave's weight fragments (Wi for WHAT & 1, else Wj), already in registers when PRE.
// GNNP: projections of a GNN step (zp / Nn / Yb are wanted); ZQ: the charge feature reads as 0 and the results go to
// `oP` / `oR` (the EPN stack's static projections).
// PREW2: `w2pre` = 16 fragments of W2 + b2[c] in element 16, requested by the caller long before: the zp / Yb chains then start
// without a round trip of their own.
__device__ static const float lg_none17[17] = {};
template <int WHAT, bool PRE, bool ZQ, bool PRE2 = false, bool PREW2 = false>
__device__ __forceinline__ void lg_proj_wave(const LargeArgs &L, const PairMlpPack &M, int gnnp, const int4 tl, const float *arow, int lane,
                                             const float (&wA)[EPNN_KA], float *oP, float *oR, const float *wB = nullptr,
                                             const float (&w2pre)[17] = lg_none17) {
    const int c = lane & 31, hh = lane >> 5;
    const int at = tl.x + (c < tl.y ? c : 0);
    const float *wp = L.wpack;
    float bv[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(arow + 4 * q);
        bv[4 * q] = v[0]; bv[4 * q + 1] = v[1]; bv[4 * q + 2] = v[2]; bv[4 * q + 3] = v[3];
    }
    if (ZQ) {
        const int fq = L.nx + EPNN_EDIM;                      // feature f sits in half f & 1 at position f >> 1
#pragma unroll
        for (int s = 0; s < EPNN_KA; ++s)
            if (s == (fq >> 1) && hh == (fq & 1)) bv[s] = 0.f;
    }
    f32x16 accP = epnn_splat16(0.f), accR = epnn_splat16(0.f);
#pragma unroll
    for (int s = 0; s < EPNN_KA; ++s) {
        if (WHAT & 1) accP = epnn_mfma(PRE ? wA[s] : wp[M.wiF + s * 64 + lane], bv[s], accP);
        if (WHAT & 2) accR = epnn_mfma(PRE2 ? wB[s] : (PRE && !(WHAT & 1) ? wA[s] : wp[M.wjF + s * 64 + lane]), bv[s], accR);
    }
    if (c < tl.y) {
        if (WHAT & 1) epnn_st16(oP + (size_t)at * 32 + hh * 16, accP);
        if (WHAT & 2) epnn_st16(oR + (size_t)at * 32 + hh * 16, accR);
        if ((WHAT & 2) && gnnp) {
            f32x16 neg;
#pragma unroll
            for (int r = 0; r < 16; ++r) neg[r] = -accR[r];
            epnn_st16(L.Nn + (size_t)at * 32 + hh * 16, neg);
        }
    }
    if ((WHAT & 1) && gnnp) {
        // padded partner: R = 0, G = 0  ->  zp_i = relu(W2^T relu(P_i) + b2); rows = atoms, cols = out
        f32x16 acc = epnn_splat16(PREW2 ? w2pre[16] : wp[M.b2 + c]);
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = epnn_mfma(fmaxf(accP[s], 0.f), PREW2 ? w2pre[s] : wp[M.w2F + s * 64 + lane], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = epnn_kappa(hh, r);
            if (row < tl.y) L.zp[(size_t)(tl.x + row) * 32 + c] = fmaxf(acc[r], 0.f);
        }
    }
    if ((WHAT & 2) && gnnp) {
        // the partner's share of the sweep's second Dense: Yb_j = b2 + W2^T R_j; rows = atoms, cols = out
        f32x16 acc = epnn_splat16(PREW2 ? w2pre[16] : wp[M.b2 + c]);
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = epnn_mfma(accR[s], PREW2 ? w2pre[s] : wp[M.w2F + s * 64 + lane], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = epnn_kappa(hh, r);
            if (row < tl.y) L.Yb[(size_t)(tl.x + row) * 32 + c] = acc[r];
        }
    }
}
__global__ __launch_bounds__(256) void k_lg_proj(LargeArgs L, PairMlpPack M, int gnnp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, hh = lane >> 5;
    const int it = blockIdx.x * 4 + wave;
    if (it >= L.natiles) return;
    const int4 tl = L.atiles[it];
    const int at = tl.x + (c < tl.y ? c : 0);
    const float none[EPNN_KA] = {};
    lg_proj_wave<3, false, false>(L, M, gnnp, tl, L.a_eo + (size_t)at * EPNN_AST + hh * 32, lane, none, L.P, L.R);
}
// the EPN stack's projections with q = 0, all T steps: one wave per (tile, step)
__global__ __launch_bounds__(256) void k_lg_epn_static(LargeArgs L) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, hh = lane >> 5;
    const int job = blockIdx.x * 4 + wave;
    if (job >= L.natiles * L.T) return;
    const int it = job / L.T, t = job - it * L.T;
    const int4 tl = L.atiles[it];
    const int at = tl.x + (c < tl.y ? c : 0);
    const float none[EPNN_KA] = {};
    lg_proj_wave<3, false, true>(L, L.wi.pas[t], 0, tl, L.a_eo + (size_t)at * EPNN_AST + hh * 32, lane, none,
                                 L.Pst + (size_t)t * L.A * 32, L.Rst + (size_t)t * L.A * 32);
}

// ------------------------------------------------------------------------------------------------ all-pairs sweep
// workgroup = up to 4 atom tiles (one per wave) x one j-chunk of the same molecule; Nn_j and Yb_j staged in LDS.
// v_mfma_f32_16x16x4_f32 in the fused kernel's layout (epnn_wave.hip.h): lane (q, n16) owns atoms n16 and 16 + n16 of
// the tile and features 16 rb + 4 q + r of each; tile j = "partner j of every atom", S accumulates in registers.  Per
// partner and 32 atoms: 32 MFMAs, 16 max (first layer), 16 max + 16 add (second layer's relu and the sum).
#define EPNN_LG_JC 64
// the partners of one staged piece.  TWO: the tile's second column block holds atoms (wave-uniform: all but a molecule's last
// tile).  Two partners per trip: the rows of partner j + 1 are fetched while partner j is in the matrix pipe, into the
// other register set (no copies); the four accumulator chains of a partner (two column blocks x two row blocks) are issued
// interleaved.
// the sweep's kernel W2: three bf16 pieces per weight (f32-grade products on the bf16 matrix pipe, epnn_wave.hip.h: w16_split3), or
// with -DEPNN_SWEEP_F32 (development builds, for comparison) the f32 fragments of v_mfma_f32_16x16x4_f32
#ifdef EPNN_SWEEP_F32
typedef float LgW2[2][8];
#define LG_LDW2(dst, off) W16_LD(dst, off, 2, 8)
#else
typedef u32x4 LgW2[2][3];
#define LG_LDW2(dst, off) W16_LDB(dst, off)
#endif
#ifdef EPNN_SWEEP_F32
template <bool TWO>
__device__ __forceinline__ void lg_sweep_rows(const float *Ns, const float *Ys, int nj, int po, int fo, const LgW2 (&pb),
                                              const f32x4 (&P0)[2], const f32x4 (&P1)[2], f32x4 (&S0)[2], f32x4 (&S1)[2]) {
    auto vmax = [](const f32x4 &a, const f32x4 &b) { return f32x4{fmaxf(a[0], b[0]), fmaxf(a[1], b[1]), fmaxf(a[2], b[2]), fmaxf(a[3], b[3])}; };
    auto fetch = [&](int j, f32x4 (&nn)[2], f32x4 (&yy)[2]) {
        nn[0] = w16_ld(Ns + j * 32 + po);
        nn[1] = w16_ld(Ns + j * 32 + po + 8);
        yy[0] = w16_ld(Ys + j * 32 + fo);
        yy[1] = w16_ld(Ys + j * 32 + 16 + fo);
    };
    // The element-wise work of a partner is ONE block in front of its matrix instructions -- the ReLU + sum of the partner BEFORE
    // it (its accumulators are `dp`), then this partner's first layer --, a scheduling barrier, then its 32 MFMAs (the fused
    // kernel's recipe of round 4).  Left to itself the compiler weaves the v_max / v_add instructions between the MFMAs, one
    // `s_nop` behind each, and un-packs packed adds in their shadow: the same arithmetic in the same order, but 4.5 % more of the
    // SIMD's issue time (protein 0.570 -> 0.545 ms, 10 000-atom box 8.12 -> 7.60 ms).
    f32x4 dp0[2] = {w16_splat(0.f), w16_splat(0.f)}, dp1[2] = {w16_splat(0.f), w16_splat(0.f)};      // nothing pending: relu(0) adds nothing
    auto partner_b = [&](const f32x4 (&nn)[2], const f32x4 (&yy)[2]) {
        S0[0] += w16_relu(dp0[0]);
        S0[1] += w16_relu(dp0[1]);
        if (TWO) {
            S1[0] += w16_relu(dp1[0]);
            S1[1] += w16_relu(dp1[1]);
        }
        const f32x4 za = vmax(P0[0], nn[0]), zb = vmax(P0[1], nn[1]);
        const float z0[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
        const f32x4 zc = vmax(P1[0], nn[0]), zd = vmax(P1[1], nn[1]);
        const float z1[8] = {zc[0], zc[1], zc[2], zc[3], zd[0], zd[1], zd[2], zd[3]};
        dp0[0] = yy[0]; dp0[1] = yy[1];
        dp1[0] = yy[0]; dp1[1] = yy[1];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            dp0[0] = w16_mfma(pb[0][k], z0[k], dp0[0]);
            dp0[1] = w16_mfma(pb[1][k], z0[k], dp0[1]);
            if (TWO) {
                dp1[0] = w16_mfma(pb[0][k], z1[k], dp1[0]);
                dp1[1] = w16_mfma(pb[1][k], z1[k], dp1[1]);
            }
        }
    };
    f32x4 na[2], ya[2], nb[2], yb[2];
    fetch(0, na, ya);
    int j = 0;
    for (; j + 2 <= nj; j += 2) {
        fetch(j + 1, nb, yb);
        __builtin_amdgcn_sched_barrier(0);
        partner_b(na, ya);
        fetch(min(j + 2, nj - 1), na, ya);
        __builtin_amdgcn_sched_barrier(0);
        partner_b(nb, yb);
    }
    if (j < nj) partner_b(na, ya);
    S0[0] += w16_relu(dp0[0]);
    S0[1] += w16_relu(dp0[1]);
    if (TWO) {
        S1[0] += w16_relu(dp1[0]);
        S1[1] += w16_relu(dp1[1]);
    }
}
#else
// The bf16 form (round 5).  Per partner and 32 atoms: 16 v_max (first layer), the three-piece split of the 16 activations per lane with
// its remainders on the matrix pipe (w16_ident / w16_pack_hi, epnn_wave.hip.h: x - piece as D = C - I B, 2 MFMAs per 8 values and
// level instead of 8 v_and + 8 v_sub, the same pieces bit for bit; 24 v_perm + 8 MFMAs), 24 MFMAs of the Dense itself, and the
// ReLU + sum of the partner BEFORE this one (16 v_max + 16 v_add) woven into those 24 MFMAs' gaps: a v_mfma_f32_16x16x32_bf16 holds
// the SIMD's vector issue for 8 of its 16 cycles, so one or two independent vector instructions per gap cost nothing
// (MI355X_MICROARCH.md, cycle constants).  The accumulators of two consecutive partners are two register sets (the loop is two
// partners deep anyway), so the pending sums never wait for the running MFMAs.
template <bool TWO>
__device__ __forceinline__ void lg_sweep_rows(const float *Ns, const float *Ys, int nj, int po, int fo, const LgW2 (&pb),
                                              const f32x4 (&P0)[2], const f32x4 (&P1)[2], f32x4 (&S0)[2], f32x4 (&S1)[2]) {
    const w16_u32x2 idI = w16_ident();
    auto vmax = [](const f32x4 &a, const f32x4 &b) { return f32x4{fmaxf(a[0], b[0]), fmaxf(a[1], b[1]), fmaxf(a[2], b[2]), fmaxf(a[3], b[3])}; };
    auto fetch = [&](int j, f32x4 (&nn)[2], f32x4 (&yy)[2]) {
        nn[0] = w16_ld(Ns + j * 32 + po);
        nn[1] = w16_ld(Ns + j * 32 + po + 8);
        yy[0] = w16_ld(Ys + j * 32 + fo);
        yy[1] = w16_ld(Ys + j * 32 + 16 + fo);
    };
    // c0 / c1: this partner's accumulators (column blocks 0 / 1); p0 / p1: the partner's before it, whose ReLU + sum is pending
    auto partner_b = [&](const f32x4 (&nn)[2], const f32x4 (&yy)[2], f32x4 (&c0)[2], f32x4 (&c1)[2], const f32x4 (&p0)[2], const f32x4 (&p1)[2]) {
        f32x4 ra0 = vmax(P0[0], nn[0]), ra1 = vmax(P0[1], nn[1]), rb0 = vmax(P1[0], nn[0]), rb1 = vmax(P1[1], nn[1]);
        u32x4 a1, a2, a3, b1, b2, b3;
        a1 = w16_pack_hi(ra0, ra1);
        if (TWO) b1 = w16_pack_hi(rb0, rb1);
        ra0 = w16_rem(idI, a1[0], a1[1], ra0); ra1 = w16_rem(idI, a1[2], a1[3], ra1);
        if (TWO) { rb0 = w16_rem(idI, b1[0], b1[1], rb0); rb1 = w16_rem(idI, b1[2], b1[3], rb1); }
        __builtin_amdgcn_sched_barrier(0);
        a2 = w16_pack_hi(ra0, ra1);
        if (TWO) b2 = w16_pack_hi(rb0, rb1);
        ra0 = w16_rem(idI, a2[0], a2[1], ra0); ra1 = w16_rem(idI, a2[2], a2[3], ra1);
        if (TWO) { rb0 = w16_rem(idI, b2[0], b2[1], rb0); rb1 = w16_rem(idI, b2[2], b2[3], rb1); }
        __builtin_amdgcn_sched_barrier(0);
        a3 = w16_pack_hi(ra0, ra1);
        if (TWO) b3 = w16_pack_hi(rb0, rb1);
        c0[0] = yy[0]; c0[1] = yy[1];
        c1[0] = yy[0]; c1[1] = yy[1];
        __builtin_amdgcn_sched_barrier(0);
        w16_mm_bf(pb, a1, a2, a3, c0);
        if (TWO) w16_mm_bf(pb, b1, b2, b3, c1);
        S0[0] += w16_relu(p0[0]);
        S0[1] += w16_relu(p0[1]);
        if (TWO) {
            S1[0] += w16_relu(p1[0]);
            S1[1] += w16_relu(p1[1]);
        }
#ifndef EPNN_LG_NOWEAVE
        // 24 (12) MFMAs, 32 (16) vector instructions: two behind each of the first 8 (4) MFMAs, one behind each of the others
#pragma unroll
        for (int i = 0; i < (TWO ? 8 : 4); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
#pragma unroll
        for (int i = 0; i < (TWO ? 16 : 8); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x4 dA0[2] = {w16_splat(0.f), w16_splat(0.f)}, dA1[2] = {w16_splat(0.f), w16_splat(0.f)};      // nothing pending: relu(0) adds nothing
    f32x4 dB0[2] = {w16_splat(0.f), w16_splat(0.f)}, dB1[2] = {w16_splat(0.f), w16_splat(0.f)};
    f32x4 na[2], ya[2], nb[2], yb[2];
    fetch(0, na, ya);
    int j = 0;
    for (; j + 2 <= nj; j += 2) {
        fetch(j + 1, nb, yb);
        __builtin_amdgcn_sched_barrier(0);
        partner_b(na, ya, dA0, dA1, dB0, dB1);
        fetch(min(j + 2, nj - 1), na, ya);
        __builtin_amdgcn_sched_barrier(0);
        partner_b(nb, yb, dB0, dB1, dA0, dA1);
    }
    if (j < nj) {
        partner_b(na, ya, dA0, dA1, dB0, dB1);
        S0[0] += w16_relu(dA0[0]);
        S0[1] += w16_relu(dA0[1]);
        if (TWO) {
            S1[0] += w16_relu(dA1[0]);
            S1[1] += w16_relu(dA1[1]);
        }
    } else {
        S0[0] += w16_relu(dB0[0]);
        S0[1] += w16_relu(dB0[1]);
        if (TWO) {
            S1[0] += w16_relu(dB1[0]);
            S1[1] += w16_relu(dB1[1]);
        }
    }
}
#endif
#ifdef EPNN_SWEEP_F32
static inline int lg_w2_off(const epnn_handle *h, int t) { return h->wvidx.g[t].w2; }
#else
static inline int lg_w2_off(const epnn_handle *h, int t) { return h->wvidx.g[t].w2b; }
#endif
__device__ __forceinline__ void lg_sweep_body(const LargeArgs &L, int w2off, float *Ns, float *Ys) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, n16 = lane & 15, fo = 4 * q;
    const int4 tk = L.stasks[blockIdx.x];            // first atile, #atiles, j_lo, j_hi (global atom indices)
    const int chunk = L.stask_chunk[blockIdx.x];
    const bool active = wave < tk.y;
    const int4 tl = L.atiles[tk.x + (active ? wave : 0)];
    const float *wp = L.wpack;
    const bool two = __builtin_amdgcn_readfirstlane(tl.y) > 16;      // the tile's second column block holds atoms
    // P and Nn rows are stored in the operand order of the 32x32x2 kernels (k_lg_proj): position 16 hh + r holds feature
    // kappa(hh, r) = 8 (r >> 2) + 4 hh + (r & 3), so this lane's features 16 rb + 4 q .. + 3 sit together at po + 8 rb
    const int po = 16 * (q & 1) + 4 * (q >> 1);
    const int c0 = n16 < tl.y ? n16 : 0, c1 = 16 + n16 < tl.y ? 16 + n16 : 0;
    f32x4 P0[2], P1[2], S0[2], S1[2];
    LgW2 pb;
    LG_LDW2(pb, w2off);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        P0[rb] = w16_ld(L.P + (size_t)(tl.x + c0) * 32 + po + 8 * rb);
        P1[rb] = w16_ld(L.P + (size_t)(tl.x + c1) * 32 + po + 8 * rb);
        S0[rb] = w16_splat(0.f);
        S1[rb] = w16_splat(0.f);
    }
    for (int j0 = tk.z; j0 < tk.w; j0 += EPNN_LG_JC) {
        const int nj = min(EPNN_LG_JC, tk.w - j0);
        __syncthreads();
        for (int i = tid; i < nj * 8; i += 256) {
            reinterpret_cast<f32x4 *>(Ns)[i] = reinterpret_cast<const f32x4 *>(L.Nn + (size_t)j0 * 32)[i];
            reinterpret_cast<f32x4 *>(Ys)[i] = reinterpret_cast<const f32x4 *>(L.Yb + (size_t)j0 * 32)[i];
        }
        __syncthreads();
        if (active) {
            if (two) lg_sweep_rows<true>(Ns, Ys, nj, po, fo, pb, P0, P1, S0, S1);
            else lg_sweep_rows<false>(Ns, Ys, nj, po, fo, pb, P0, P1, S0, S1);
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
// ------------------------------------------------------------------------------------------------ GNN correction tiles
// one wave per 32 listed pairs (tile `pt`): z2(P + R + G) - z2(P + R) for both directions -> the two atoms' incidence slots.
// SEARCH: the pair records do not exist yet (the first step's tiles run in the launch that also links the list): the slot
// of the pair in its second atom's row is looked up here.
template <bool SEARCH>
__device__ __forceinline__ void lg_pair_tile(const LargeArgs &L, const PairMlpPack &M, int pt, int np, const int *nbr, bool all_tiled = false) {
    const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
    const int slot = pt * 32 + c;
    bool valid = slot < np;
    int gi = 0, gj = 0, di = -1, dj = -1;
    if (SEARCH) {
        if (valid) {
            gi = L.pi[slot];
            gj = L.pj[slot];
            di = L.dest_i[slot];
            if (!all_tiled) valid = L.mflag[L.mol_of[gi]] != 0;      // (every molecule of the batch is tiled: nothing to look up)
        }
        if (valid) {
            const int lo = L.inc_off[gj], hi = L.inc_off[gj + 1];
            for (int k0 = lo; k0 < hi && dj < 0; k0 += 16) {
                int v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = k0 + u < hi ? nbr[k0 + u] : -1;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (v[u] == gi) dj = k0 + u;
            }
        }
    } else if (valid) {
        const int4 ra = L.prec[2 * slot], rb = L.prec[2 * slot + 1];
        valid = ra.x >= 0;                                     // (a pair of a molecule that is not on this path: -1 - i)
        gi = valid ? ra.x : 0;
        gj = ra.y;
        di = rb.z;
        dj = rb.w;
    }
    if (__ballot(valid) == 0ull) return;
    if (!valid) { di = -1; dj = -1; gi = 0; gj = 0; }
    const float *wp = L.wpack;
    const f32x16 g = lg_gtile(wp + M.weF, L.pe + (size_t)(valid ? slot : 0) * EPNN_EDIM + hh * 24, valid, lane);
    float pi_[16], rj_[16], pj_[16], ri_[16], w2[16];
    epnn_ld16(L.P + (size_t)gi * 32 + hh * 16, pi_);
    epnn_ld16(L.R + (size_t)gj * 32 + hh * 16, rj_);
    epnn_ld16(L.P + (size_t)gj * 32 + hh * 16, pj_);
    epnn_ld16(L.R + (size_t)gi * 32 + hh * 16, ri_);
#pragma unroll
    for (int s = 0; s < 16; ++s) w2[s] = wp[M.w2F + s * 64 + lane];
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
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = epnn_kappa(hh, r);                     // lane `row` (either half) holds that pair's slots
        const int ti = __shfl(di, row, 64), tj = __shfl(dj, row, 64);
        if (ti >= 0) L.corrA[(size_t)ti * 32 + c] = fmaxf(aG[r], 0.f) - fmaxf(a0[r], 0.f);
        if (tj >= 0) L.corrA[(size_t)tj * 32 + c] = fmaxf(bG[r], 0.f) - fmaxf(b0[r], 0.f);
    }
}
// the correction tiles as a launch of their own (a process of a partition without sweep tasks; `large_fused` = 0)
__global__ __launch_bounds__(256) void k_lg_pairs(LargeArgs L, PairMlpPack M) {
    const int np = L.row_off[L.A];
    if (np > L.pcap) return;
    const int pt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pt * 32 < np) lg_pair_tile<false>(L, M, pt, np, nullptr);
}
// The sweep.  Its wavefronts first share out the step's near-pair correction tiles (both need only this step's P and R;
// ~1 % of the work, and a launch of their own would be 10 us of latencies in front of or behind every sweep), then run their
// sweep task.  The launch asks for as much LDS as keeps the workgroups spread evenly over the CUs (`wpc` per CU, see the
// host side): left to itself the dispatcher put three of a 2220-atom system's 504 workgroups on some CUs and one on others,
// and the launch lasted 103 us where its wavefronts lived 83.
__global__ __launch_bounds__(256) void k_lg_sweep(LargeArgs L, int w2off, PairMlpPack Mpair, int do_pairs) {
    extern __shared__ __attribute__((aligned(16))) float lg_smem[];
    if (do_pairs) {
        const int np = L.row_off[L.A];
        if (np <= L.pcap) {
            // this task's share of the tiles: the host deals them by how much shorter a task's partner range is than its
            // molecule's longest (a molecule's last chunk is the short one: its wavefronts take the tiles and still finish
            // with the others; dealt evenly, every tile was ~5 us at the end of a wavefront that had a full sweep task)
            const int npt = (np + 31) >> 5;
            const float2 fr = L.stask_frac[blockIdx.x];
            const int lo = min(npt, (int)(fr.x * (float)npt)), hi = fr.y >= 1.f ? npt : min(npt, (int)(fr.y * (float)npt));
            for (int pt = lo + (int)(threadIdx.x >> 6); pt < hi; pt += 4) lg_pair_tile<false>(L, Mpair, pt, np, nullptr);
        }
    }
#ifdef EPNN_LG_CLOCKS
    // development build: shader clock (s_memtime) and 100 MHz clock (s_memrealtime) at the start and the end of this workgroup's
    // sweep task -- their ratio is the shader clock the sweep ran at, the starts and ends show the launch's ramp and drain
    unsigned long long c0 = 0, r0 = 0;
    if (threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    lg_sweep_body(L, w2off, lg_smem, lg_smem + EPNN_LG_JC * 32);
#ifdef EPNN_LG_CLOCKS
    if (threadIdx.x == 0 && L.clk && blockIdx.x < 1024) {
        unsigned long long *o = L.clk + 128 + 4 * (size_t)blockIdx.x;
        o[0] = c0; o[1] = r0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ------------------------------------------------------------------------------------------------ tile-workgroup sweep
// The sweep of molecules of up to EPNN_LG_TW_TILES tiles (4096 atoms): workgroup = ONE 32-atom tile x four consecutive pieces of
// the partner range, one piece per wavefront.  Against k_lg_sweep (four tiles x one piece, the staged partner rows shared):
//   * a wavefront stages its own partners (Nn_j, Yb_j rows: 256 bytes each, up to EPNN_LG_JP per trip) in its own quarter of the
//     workgroup's LDS -- no workgroup barrier anywhere in the partner loop (the shared staging had two per 64 partners);
//   * the four wavefronts' sums are added in LDS, ((w0 + w1) + w2) + w3, and ONE partial sum per workgroup goes to memory: the
//     reduction behind the sweep reads 7 partial sums per atom of the 2220-atom protein instead of 28;
// (Round 5 also ran a tile's tail -- reduction, update MLP, next projections -- inside this launch, by the last of the tile's
// workgroups to store its partial sum: SLOWER than the launch it replaces.  All workgroups of a 2220-atom system are resident at once
// and end together, so every tile's tail starts at the launch's end and is a 17.7 us chain on one workgroup against 13.8 us for the
// launch; and the release / acquire fences that order the partial sums across the XCDs' L2s, ~500 per launch, cost 41 us by
// themselves.  HISTORY.md part B has the four timings.)
// It reads 4 x the partner rows of the shared staging from L2 (protein: 40 MB per sweep, L2-resident arrays), which is why systems
// of more than EPNN_LG_TW_TILES tiles keep k_lg_sweep.
#define EPNN_LG_JP 80                       // partners a wavefront stages per trip: 4 x 80 x 256 B = 80 KB per workgroup, two per CU
#define EPNN_LG_TW_TILES 128
__device__ __forceinline__ void lg_wave_sync() {         // LDS written by this wavefront is read by this wavefront: order only
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__global__ __launch_bounds__(256) void k_lg_sweep2(LargeArgs L, int w2off, PairMlpPack Mpair, int do_pairs) {
    extern __shared__ __attribute__((aligned(16))) float lg_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, n16 = lane & 15, fo = 4 * q;
    const int4 tk = L.stasks2[blockIdx.x];           // atile (or -1), partial index, j_lo, j_hi (global atom indices)
#ifdef EPNN_LG_CLOCKS
    // development build: shader clock (s_memtime) and 100 MHz clock (s_memrealtime) at the start and the end of this workgroup
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (threadIdx.x == 0) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    if (do_pairs) {
        const int np = L.row_off[L.A];
        const float2 fr = L.stask2_frac[blockIdx.x];
        if (np <= L.pcap && fr.y > fr.x) {
            const int npt = (np + 31) >> 5;
            const int lo = min(npt, (int)(fr.x * (float)npt)), hi = fr.y >= 1.f ? npt : min(npt, (int)(fr.y * (float)npt));
            for (int pt = lo + wave; pt < hi; pt += 4) lg_pair_tile<false>(L, Mpair, pt, np, nullptr);
        }
    }
    if (tk.x < 0) return;
    const int4 tl = L.atiles[tk.x];
    const float *wp = L.wpack;
    const bool two = __builtin_amdgcn_readfirstlane(tl.y) > 16;      // the tile's second column block holds atoms
    const int po = 16 * (q & 1) + 4 * (q >> 1);                       // (operand order of P / Nn rows: see lg_sweep_body)
    const int c0 = n16 < tl.y ? n16 : 0, c1 = 16 + n16 < tl.y ? 16 + n16 : 0;
    f32x4 P0[2], P1[2], S0[2], S1[2];
    LgW2 pb;
    LG_LDW2(pb, w2off);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        P0[rb] = w16_ld(L.P + (size_t)(tl.x + c0) * 32 + po + 8 * rb);
        P1[rb] = w16_ld(L.P + (size_t)(tl.x + c1) * 32 + po + 8 * rb);
        S0[rb] = w16_splat(0.f);
        S1[rb] = w16_splat(0.f);
    }
    // this wavefront's piece of the workgroup's partner range
    const int sub = (tk.w - tk.z + 3) >> 2;
    const int jlo = min(tk.w, tk.z + wave * sub), jhi = min(tk.w, jlo + sub);
    float *Ns = lg_smem + wave * (EPNN_LG_JP * 64), *Ys = Ns + EPNN_LG_JP * 32;
    for (int j0 = jlo; j0 < jhi; j0 += EPNN_LG_JP) {
        const int nj = min(EPNN_LG_JP, jhi - j0);
        if (j0 > jlo) lg_wave_sync();                     // the rows of the trip before have been read
        {
            const f32x4 *gn = reinterpret_cast<const f32x4 *>(L.Nn + (size_t)j0 * 32), *gy = reinterpret_cast<const f32x4 *>(L.Yb + (size_t)j0 * 32);
            constexpr int PER = EPNN_LG_JP * 8 / 64;      // 16-byte pieces per lane and array
            f32x4 v[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) v[u] = gn[min(u * 64 + lane, nj * 8 - 1)];       // unconditional loads of a clamped index
#pragma unroll
            for (int u = 0; u < PER; ++u)
                if (u * 64 + lane < nj * 8) reinterpret_cast<f32x4 *>(Ns)[u * 64 + lane] = v[u];
#pragma unroll
            for (int u = 0; u < PER; ++u) v[u] = gy[min(u * 64 + lane, nj * 8 - 1)];
#pragma unroll
            for (int u = 0; u < PER; ++u)
                if (u * 64 + lane < nj * 8) reinterpret_cast<f32x4 *>(Ys)[u * 64 + lane] = v[u];
        }
        lg_wave_sync();
        if (two) lg_sweep_rows<true>(Ns, Ys, nj, po, fo, pb, P0, P1, S0, S1);
        else lg_sweep_rows<false>(Ns, Ys, nj, po, fo, pb, P0, P1, S0, S1);
    }
    // the four wavefronts' sums, added in wavefront order; one partial sum per workgroup
    lg_wave_sync();
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        w16_st(Ns + n16 * 32 + 16 * rb + fo, S0[rb]);
        w16_st(Ns + (16 + n16) * 32 + 16 * rb + fo, S1[rb]);
    }
    __syncthreads();
    {
        float *dst = L.S0 + ((size_t)tk.y * L.A) * 32 + (size_t)tl.x * 32;
        const f32x4 a = reinterpret_cast<const f32x4 *>(lg_smem)[tid], b = reinterpret_cast<const f32x4 *>(lg_smem + EPNN_LG_JP * 64)[tid],
                    c = reinterpret_cast<const f32x4 *>(lg_smem + 2 * EPNN_LG_JP * 64)[tid], d = reinterpret_cast<const f32x4 *>(lg_smem + 3 * EPNN_LG_JP * 64)[tid];
        if ((tid >> 3) < tl.y) reinterpret_cast<f32x4 *>(dst)[tid] = ((a + b) + c) + d;      // element 4 tid .. 4 tid + 3 = atom tid >> 3
    }
#ifdef EPNN_LG_CLOCKS
    if (threadIdx.x == 0 && L.clk && blockIdx.x < 1024) {
        unsigned long long *o = L.clk + 128 + 4 * (size_t)blockIdx.x;
        o[0] = clk_c0; o[1] = clk_r0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ------------------------------------------------------------------------------------------------ merged launches of the compact entry
// What a forward of the compact entry can do at once goes into ONE launch, the kinds of work told apart by the block index
// (a second stream would do the same on paper; its fork / join events cost more than these short kernels last):
//   k_lg_first        tiles: feature rows a_eo, their hashes, first projections | near-partner counts
//   k_lg_scan_types   prefix sums of the pair list | type table of every tiled molecule
//   k_lg_fill_assign  fill of the pair list | every atom's type
//   k_lg_second       pair records (link) | the first step's type sums | the first step's correction tiles
struct LgFirst {
    int tile_wgs, count_wgs, hash;
};
// the prefix sums of the pair list | the type table of every tiled molecule (from the hashes k_lg_first left)
__global__ __launch_bounds__(1024) void k_lg_scan_types(LargeArgs L, FrontArgs F) {
    __shared__ LgTypeShared T;
    __shared__ int wsA[16], wsB[16];
    __shared__ int carryA, carryB;
    LG_CLKB(blockIdx.x == 0, 114);
    LG_CLKB(blockIdx.x == 1, 122);
    if (blockIdx.x == 0) front_scan_both_body(F, wsA, wsB, carryA, carryB);
    else lg_types_table<1024, true>(L, T, (int)blockIdx.x - 1);
    LG_CLKB(blockIdx.x == 0, 115);
    LG_CLKB(blockIdx.x == 1, 123);
}
// the fill of the pair list | every tiled atom's type
__global__ __launch_bounds__(256) void k_lg_fill_assign(LargeArgs L, FrontArgs F, int fill_wgs) {
    __shared__ FrontFillShared Sh;
    LG_CLKB(blockIdx.x == 0, 112);
    if ((int)blockIdx.x < fill_wgs) {
        front_fill_body<false>(F, Sh, (int)blockIdx.x);
        LG_CLKB(blockIdx.x == 0, 113);
        return;
    }
    const int at = ((int)blockIdx.x - fill_wgs) * 256 + (int)threadIdx.x;
    if (at >= L.A) return;
    const int lm = L.mflag[L.mol_of[at]] - 1;                  // mflag = 1 + the molecule's place among the tiled ones
    if (lm >= 0) lg_type_assign(L, at, lm, L.typ_hash[at]);
}
__global__ __launch_bounds__(256) void k_lg_first(LargeArgs L, PairMlpPack M, LgFirst W, FrontArgs F) {
    __shared__ __attribute__((aligned(16))) float sm[4 * 32 * EPNN_AST];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hh = lane >> 5;
    int blk = (int)blockIdx.x;
    LG_CLKB(blk == 0, 104);
    LG_CLKB(blk == W.tile_wgs, 109);
    if (blk < W.tile_wgs) {
        // feature rows of the workgroup's (up to) four tiles: built in LDS by all threads (one slot each per trip), stored to
        // a_eo from there, projected from there (k_lg_init + k_lg_proj without the round trip through memory)
        const int t0 = blk * 4;
        const int fq = L.nx + EPNN_EDIM;
        int4 tls[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) tls[w] = t0 + w < L.natiles ? L.atiles[t0 + w] : make_int4(0, 0, 0, 0);
        // both projections' weight fragments are requested now: they travel while the rows are built
        float wI[EPNN_KA], wJ[EPNN_KA], w2n[17];
#pragma unroll
        for (int s = 0; s < EPNN_KA; ++s) {
            wI[s] = L.wpack[M.wiF + s * 64 + lane];
            wJ[s] = L.wpack[M.wjF + s * 64 + lane];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) w2n[s] = L.wpack[M.w2F + s * 64 + lane];
        w2n[16] = L.wpack[M.b2 + c];
        for (int idx = tid; idx < 4 * 32 * EPNN_AST; idx += 256) sm[idx] = 0.f;
        __syncthreads();
        {   // one thread per atom and half of its inputs (x and q | h), every load of a thread in flight together
            const int w = (tid & 127) >> 5, a = tid & 31, part = tid >> 7;
            const int4 tl = w == 0 ? tls[0] : w == 1 ? tls[1] : w == 2 ? tls[2] : tls[3];
            if (a < tl.y) {
                const int at = tl.x + a, b = tl.z;
                float *row = sm + (w * 32 + a) * EPNN_AST;
                if (part == 0) {
                    float xv[16];
#pragma unroll
                    for (int f = 0; f < 16; ++f) {
                        const float t = L.xin[(size_t)at * L.nx + min(f, L.nx - 1)];
                        xv[f] = f < L.nx ? t : 0.f;
                    }
                    const float qv = L.q_in ? L.q_in[at] : L.Q[b] / (float)(L.moff[b + 1] - L.moff[b]);
#pragma unroll
                    for (int f = 0; f < 16; ++f)
                        if (f < L.nx) row[epnn_aeo(f)] = xv[f];
                    row[epnn_aeo(fq)] = qv;
                    row[epnn_aeo(EPNN_F1)] = 1.f;                // carries the first Dense's bias (Wi row 59 = b1)
                    L.qbuf[at] = qv;                             // the EPN stack's charges, both generations
                    L.qbuf[(size_t)L.A + at] = qv;
                } else if (L.h_in) {
                    float hv[EPNN_EDIM];
#pragma unroll
                    for (int f = 0; f < EPNN_EDIM; ++f) hv[f] = L.h_in[(size_t)at * EPNN_EDIM + f];
#pragma unroll
                    for (int f = 0; f < EPNN_EDIM; ++f) row[epnn_aeo(L.nx + f)] = hv[f];
                }
            }
        }
        __syncthreads();
        LG_CLKB(blk == 0, 105);
        // (16-byte stores, each wavefront its own tile's rows: 34 four-byte stores per thread, and a reload of the tile record
        //  behind them -- a load waits for every store before it --, held the projections back 8.8 us of this launch's 19.5)
        {
            const int4 tlw = wave == 0 ? tls[0] : wave == 1 ? tls[1] : wave == 2 ? tls[2] : tls[3];
            const f32x4 *src = reinterpret_cast<const f32x4 *>(sm + wave * 32 * EPNN_AST);
            f32x4 *dst = reinterpret_cast<f32x4 *>(L.a_eo + (size_t)tlw.x * EPNN_AST);
            for (int i = lane; i < tlw.y * (EPNN_AST / 4); i += 64) dst[i] = src[i];
        }
        if (W.hash && tid < 128) {                             // hash of every atom's feature row (the atom types of the first step)
            const int w = tid >> 5, a = tid & 31;
            const int4 tl = w == 0 ? tls[0] : w == 1 ? tls[1] : w == 2 ? tls[2] : tls[3];
            if (a < tl.y) {
                unsigned long long hsh = 1469598103934665603ull;
                for (int f = 0; f < L.nx; ++f) hsh = lg_hash_step(hsh, sm[(w * 32 + a) * EPNN_AST + epnn_aeo(f)]);
                L.typ_hash[tl.x + a] = hsh | 1ull;
            }
        }
        if (t0 + wave >= L.natiles) return;
        const int4 tl = wave == 0 ? tls[0] : wave == 1 ? tls[1] : wave == 2 ? tls[2] : tls[3];
        const int row = c < tl.y ? c : 0;
        LG_CLKB(blk == 0, 106);
        lg_proj_wave<3, true, false, true, true>(L, M, 1, tl, sm + (wave * 32 + row) * EPNN_AST + hh * 32, lane, wI, L.P, L.R, wJ, w2n);
        LG_CLKB(blk == 0, 107);
        return;
    }
    blk -= W.tile_wgs;
    if (blk < W.count_wgs) front_count_body(F, sm, blk);
    LG_CLKB(blk == 0, 110);
}
// systems of up to EPNN_FRONT_INLINE_A atoms: the fill of the pair list with the prefix sums worked out by every workgroup itself
// (front_fill_body<true>) | the type table of every tiled molecule -- no k_lg_scan_types launch; the atoms' types are assigned in
// k_lg_second then
__global__ __launch_bounds__(256) void k_lg_fill_types(LargeArgs L, FrontArgs F, int fill_wgs) {
    __shared__ FrontFillShared Sh;
    LG_CLKB(blockIdx.x == 0, 112);
    LG_CLKB((int)blockIdx.x == fill_wgs, 122);
    if ((int)blockIdx.x < fill_wgs) {
        front_fill_body<true>(F, Sh, (int)blockIdx.x);
        LG_CLKB(blockIdx.x == 0, 113);
        return;
    }
    lg_types_table<256, true>(L, *reinterpret_cast<LgTypeShared *>(&Sh), (int)blockIdx.x - fill_wgs);
    LG_CLKB((int)blockIdx.x == fill_wgs, 123);
}
struct LgSecond {
    int link_wgs, tsweep_wgs, all_tiled, assign_wgs;
};
__global__ __launch_bounds__(256) void k_lg_second(LargeArgs L, PairMlpPack M, LgSecond W, FrontArgs F) {
    int blk = (int)blockIdx.x;
    LG_CLKB(blk == 0, 116);
    LG_CLKB(blk == W.link_wgs, 118);
    LG_CLKB(blk == W.link_wgs + W.tsweep_wgs + W.assign_wgs, 120);
    if (blk < W.link_wgs) {
        front_link_body(F, blk, W.link_wgs);
        LG_CLKB(blk == 0, 117);
        return;
    }
    blk -= W.link_wgs;
    if (blk < W.tsweep_wgs) {
        if (threadIdx.x < 64) lg_tsweep_wave(L, M, blk);
        LG_CLKB(blk == 0, 119);
        return;
    }
    blk -= W.tsweep_wgs;
    if (blk < W.assign_wgs) {                                  // (the type table comes from k_lg_fill_types: every tiled atom's type)
        const int at = blk * 256 + (int)threadIdx.x;
        if (at >= L.A) return;
        const int lm = L.mflag[L.mol_of[at]] - 1;
        if (lm >= 0) lg_type_assign(L, at, lm, L.typ_hash[at]);
        return;
    }
    blk -= W.assign_wgs;
    const int np = L.row_off[L.A];
    if (np > L.pcap) return;
    const int pt = blk * 4 + (int)(threadIdx.x >> 6);
    if (pt * 32 < np) lg_pair_tile<true>(L, M, pt, np, F.nbr, W.all_tiled != 0);
    LG_CLKB(blk == 0, 121);
}

// ------------------------------------------------------------------------------------------------ reduction of S
// S of (atom, output o): the sweep's chunk partials in chunk order (or the atom's type row in the first step), the atom's
// incidence slots in slot order, the padded partners.  NA atoms per thread, the loads of all of them in flight together; every
// element adds its own terms in the same order whatever NA is.
template <int NA>
__device__ __forceinline__ void lg_reduce(const LargeArgs &L, const int4 tl, const int (&a)[NA], int o, int n, int types, float (&out)[NA]) {
    const int nchunk = types ? 0 : tl.w;
    int at[NA], lo[NA], hi[NA];
    bool ok[NA];
    float s[NA], zpv[NA];
    // The loads are issued in as few dependent rounds as the data allows (every round is ~1 us on data the previous launch
    // wrote): (1) row bounds, the padded partners' term, the atom's type, and the first batch of partial sums -- ALL of them for a
    // system on the tile-workgroup sweep, which leaves one partial sum per workgroup (7 for the protein); (2) the type's row and the
    // first sixteen slots of the atom's row (a protein atom has at most 19 partners); (3) the rest.  The order in which an element's
    // terms are ADDED is what it always was: partial sums in piece order, slots in slot order, padding.
    constexpr int CB = 32 / NA;                                // partial sums of a thread in flight (64 spill in the fused tail)
    constexpr int S1 = 32 / NA < 16 ? 32 / NA : 16;
    const size_t step = (size_t)L.A * 32;
    int trow[NA];
    float v0[NA][CB];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        ok[k] = a[k] < tl.y;
        at[k] = tl.x + (ok[k] ? a[k] : 0);
        lo[k] = L.inc_off[at[k]];
        hi[k] = L.inc_off[at[k] + 1];
        zpv[k] = L.zp[(size_t)at[k] * 32 + o];
        trow[k] = types ? L.typ_row[at[k]] : 0;
        s[k] = 0.f;
#pragma unroll
        // (unconditional loads of a clamped index: `cond ? p[i] : 0` is a branch per load, each waiting for the one before)
        for (int u = 0; u < CB; ++u) v0[k][u] = L.S0[(size_t)at[k] * 32 + o + (size_t)max(min(u, nchunk - 1), 0) * step];
    }
#pragma unroll
    for (int k = 0; k < NA; ++k)
#pragma unroll
        for (int u = 0; u < CB; ++u)
            if (u < nchunk) s[k] += v0[k][u];
    for (int ch = CB; ch < nchunk; ch += CB) {                 // (systems on the four-tile sweep: more pieces than one batch)
        float v[NA][CB];
#pragma unroll
        for (int k = 0; k < NA; ++k)
#pragma unroll
            for (int u = 0; u < CB; ++u) v[k][u] = L.S0[(size_t)at[k] * 32 + o + (size_t)min(ch + u, nchunk - 1) * step];
#pragma unroll
        for (int k = 0; k < NA; ++k)
#pragma unroll
            for (int u = 0; u < CB; ++u)
                if (ch + u < nchunk) s[k] += v[k][u];
    }
    {
        float c1[NA][S1], tv[NA];
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            tv[k] = types ? L.S_type[(size_t)trow[k] * 32 + o] : 0.f;
#pragma unroll
            for (int u = 0; u < S1; ++u) c1[k][u] = L.corrA[(size_t)max(min(lo[k] + u, hi[k] - 1), 0) * 32 + o];
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            if (types) s[k] = tv[k];
#pragma unroll
            for (int u = 0; u < S1; ++u)
                if (lo[k] + u < hi[k]) s[k] += c1[k][u];
            lo[k] += S1;
        }
    }
    int left = 0;
#pragma unroll
    for (int k = 0; k < NA; ++k) left = max(left, hi[k] - lo[k]);
    constexpr int SB = 32 / NA;
    for (; left > 0; left -= SB) {
        float v[NA][SB];
#pragma unroll
        for (int k = 0; k < NA; ++k)
#pragma unroll
            for (int u = 0; u < SB; ++u) v[k][u] = L.corrA[(size_t)max(min(lo[k] + u, hi[k] - 1), 0) * 32 + o];
#pragma unroll
        for (int k = 0; k < NA; ++k) {
#pragma unroll
            for (int u = 0; u < SB; ++u)
                if (lo[k] + u < hi[k]) s[k] += v[k][u];
            lo[k] += SB;
        }
    }
#pragma unroll
    for (int k = 0; k < NA; ++k) out[k] = ok[k] ? s[k] + (float)(L.N - n) * zpv[k] : 0.f;
}
__global__ __launch_bounds__(256) void k_lg_reduce(LargeArgs L, float *Sfin, int types) {
    if (L.row_off[L.A] > L.pcap) return;
    const int it = blockIdx.x;                 // one workgroup per 8 atoms of a tile list entry
    const int o = threadIdx.x & 31, a8 = threadIdx.x >> 5;
    const int4 tl = L.atiles[it >> 2];
    const int a[1] = {(it & 3) * 8 + a8};
    if (a[0] >= tl.y) return;
    float sv[1];
    lg_reduce<1>(L, tl, a, o, L.moff[tl.z + 1] - L.moff[tl.z], types, sv);
    Sfin[(size_t)(tl.x + a[0]) * 32 + o] = sv[0];
}

// The update MLP (charge_gn.py:71-74) of one 32-atom tile by one wave: weights w1 / w2 / w3 already in registers, the
// atom's h features from `arow` (its half hh of the even/odd feature row, offset to the first h slot), its reduced message
// sum from `srow` ([32] out-major).  New h goes to `dst` (a full even/odd row; global a_eo, and `dst2` when not null).
// LATE: only w1 is in registers at entry; the other two layers' fragments are requested here and travel under the first layer's
// forty MFMAs (the fused tail keeps its registers for the reduction that runs before this).
// `bs` (optional): the five bias vectors staged by the caller as [cb3p 32 | bu1p 32 | bu2p 32 | bu3p 64] (LDS): read from the
// weight pack they are three round trips in the middle of a chain of dependent MFMAs.
// (in two halves: the h part of the first layer does not depend on the step's message sum, so the tail launch runs it on wavefront 0
// WHILE the other wavefronts reduce S -- the same chain in the same order, 24 of its 40 matrix instructions off the critical path)
__device__ __forceinline__ f32x16 lg_update_h(const LargeArgs &L, const UpdPack &U, const float (&w1)[40], const float *arow, int lane, const float *bs) {
    const int hh = lane >> 5;
    const float *wp = L.wpack;
    float hv[24], cb[16];
#pragma unroll
    for (int s = 0; s < 24; ++s) hv[s] = arow[s];
    epnn_ld16(bs ? bs + hh * 16 : wp + U.cb3p + hh * 16, cb);
    const float Nf = (float)L.N;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = Nf * cb[r];
#pragma unroll
    for (int s = 0; s < 24; ++s) acc = epnn_mfma(w1[s], hv[s], acc);
    return acc;
}
__device__ __forceinline__ void lg_update_rest(const LargeArgs &L, const UpdPack &U, f32x16 acc, const float (&w1)[40], const float (&w2)[16],
                                               const float (&w3)[32], const int4 tl, const float *srow, float *dst, float *dst2, int lane,
                                               const float *bs) {
    const int c = lane & 31, hh = lane >> 5;
    const bool live = c < tl.y;
    const int at = tl.x + (live ? c : 0);
    const int nx = L.nx;
    const float *wp = L.wpack;
    float sv[16], b1[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) sv[s] = srow[2 * s + hh];
    epnn_ld16(bs ? bs + 32 + hh * 16 : wp + U.bu1p + hh * 16, b1);
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = epnn_mfma(w1[24 + s], sv[s], acc);
    float u1[16], b2v[16];
    // node_mask (charge_gn.py:59,72,74): 1 for every real atom unless the dense front-end supplies one
    const float nmc = L.nm_in ? L.nm_in[at] : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) u1[r] = fmaxf(nmc * acc[r] + b1[r], 0.f);
    epnn_ld16(bs ? bs + 64 + hh * 16 : wp + U.bu2p + hh * 16, b2v);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b2v[r];
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = epnn_mfma(w2[s], u1[s], acc);
    float u2[16], b3a[16], b3b[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) u2[r] = fmaxf(acc[r], 0.f);
    epnn_ld16(bs ? bs + 96 + hh * 16 : wp + U.bu3p + hh * 16, b3a);
    epnn_ld16(bs ? bs + 128 + hh * 16 : wp + U.bu3p + 32 + hh * 16, b3b);
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
template <bool LATE>
__device__ __forceinline__ void lg_update_wave(const LargeArgs &L, const UpdPack &U, const float (&w1)[40], float (&w2)[16],
                                               float (&w3)[32], const int4 tl, const float *arow, const float *srow,
                                               float *dst, float *dst2, int lane, const float *bs = nullptr) {
    if (LATE) {
        const float *wq = L.wpack;
#pragma unroll
        for (int s = 0; s < 16; ++s) w2[s] = wq[U.u2F + s * 64 + lane];
#pragma unroll
        for (int s = 0; s < 32; ++s) w3[s] = wq[U.u3F + s * 64 + lane];
    }
    const f32x16 acc = lg_update_h(L, U, w1, arow, lane, bs);
    lg_update_rest(L, U, acc, w1, w2, w3, tl, srow, dst, dst2, lane, bs);
}
#define LG_LOAD_UPD_WEIGHTS(w1, w2, w3, U)                                            \
    _Pragma("unroll") for (int s = 0; s < 40; ++s) w1[s] = wp[U.u1F + s * 64 + lane]; \
    _Pragma("unroll") for (int s = 0; s < 16; ++s) w2[s] = wp[U.u2F + s * 64 + lane]; \
    _Pragma("unroll") for (int s = 0; s < 32; ++s) w3[s] = wp[U.u3F + s * 64 + lane];

// workgroup = up to 4 atom tiles, one wave per tile runs the update MLP on the reduced S (used when a partition's exchange
// sits between the reduction and the update)
__global__ __launch_bounds__(256) void k_lg_update(LargeArgs L, UpdPack U, const float *Sfin) {
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
    __syncthreads();
    if (t0 + wave >= L.natiles) return;
    const int4 tl = L.atiles[t0 + wave];
    const int at = tl.x + (c < tl.y ? c : 0);
    const int u0 = (L.nx - hh + 1) >> 1;
    lg_update_wave<false>(L, U, w1, w2, w3, tl, L.a_eo + (size_t)at * EPNN_AST + hh * 32 + u0, Ss + (wave * 32 + c) * EPNN_SST,
                          L.a_eo + (size_t)at * EPNN_AST, nullptr, lane);
}

// The update stage for an update MLP of other widths than [32, 32] (make_model(layers), charge_gn.py:369-371; always with one
// kernel per stage): workgroup = 16 atoms of a tile.  messages = W3^T S + N b3 (the message MLP's last Dense, summed over all N
// partners, :68-70), masked_input = [h | messages] * node_mask (:71-72), h = update_fn(masked_input) * node_mask (:73-74).
__global__ __launch_bounds__(256) void k_lg_update_generic(LargeArgs L, GenMlp G, int offW3, int offb3, const float *Sfin) {
    __shared__ float a0[EPNN_GMLP_ROWS * EPNN_GMLP_ST], a1[EPNN_GMLP_ROWS * EPNN_GMLP_ST];
    if (L.row_off[L.A] > L.pcap) return;
    const int4 tl = L.atiles[blockIdx.x >> 1];
    const int r0 = (blockIdx.x & 1) * EPNN_GMLP_ROWS, tid = threadIdx.x;
    if (r0 >= tl.y) return;
    const float *W3 = G.w + offW3, *b3 = G.w + offb3;
    const float Nf = (float)L.N;
    // S rows -> a1, then [h | messages] * node_mask -> a0
    for (int idx = tid; idx < EPNN_GMLP_ROWS * 32; idx += 256) {
        const int r = idx >> 5, k = idx & 31;
        a1[r * EPNN_GMLP_ST + k] = r0 + r < tl.y ? Sfin[(size_t)(tl.x + r0 + r) * 32 + k] : 0.f;
    }
    __syncthreads();
    for (int idx = tid; idx < EPNN_GMLP_ROWS * (EPNN_EDIM + 32); idx += 256) {      // (feature fastest: coalesced rows of W3)
        const int r = idx / (EPNN_EDIM + 32), f = idx - r * (EPNN_EDIM + 32);
        const bool live = r0 + r < tl.y;
        const int at = tl.x + (live ? r0 + r : 0);
        const float nmc = L.nm_in ? L.nm_in[at] : 1.f;
        float v;
        if (f < EPNN_EDIM) v = L.a_eo[(size_t)at * EPNN_AST + epnn_aeo(L.nx + f)];
        else {
            const int o = f - EPNN_EDIM;
            float acc = Nf * b3[o];
#pragma unroll 8
            for (int k = 0; k < 32; ++k) acc = fmaf(a1[r * EPNN_GMLP_ST + k], W3[k * 32 + o], acc);
            v = acc;
        }
        a0[r * EPNN_GMLP_ST + f] = live ? nmc * v : 0.f;
    }
    __syncthreads();
    const float *res = gmlp_rows(G, a0, a1);
    for (int idx = tid; idx < EPNN_GMLP_ROWS * EPNN_EDIM; idx += 256) {
        const int r = idx / EPNN_EDIM, f = idx - r * EPNN_EDIM;
        if (r0 + r >= tl.y) continue;
        const int at = tl.x + r0 + r;
        const float nmc = L.nm_in ? L.nm_in[at] : 1.f;
        L.a_eo[(size_t)at * EPNN_AST + epnn_aeo(L.nx + f)] = nmc * res[r * EPNN_GMLP_ST + f];
    }
}

// ---- one launch for everything between two sweeps: workgroup (eight waves) = one 32-atom tile.  All 512 threads reduce the
// tile's S -- one output of two atoms each, every partial sum of a thread in flight at once: three dependent round trips
// (bounds, partials, slot rows) --, wave 0 runs the update MLP, then the NEXT thing's projections from an LDS image of the
// tile's feature rows: waves 1 / 2 the next sweep's P (zp) and R (Nn, Yb), or after the last GNN step waves 1..7 the EPN
// stack's 2T static projections.  Weight fragments are requested before the reduction so that they travel under it.
struct LgNext {
    PairMlpPack M;       // projections of the next sweep
    int run;             // 0: nothing follows; 1: next GNN step; 2: the EPN stack's static projections
};
__global__ __launch_bounds__(512) void k_lg_gnn_tail(LargeArgs L, UpdPack U, LgNext X, int types) {
    __shared__ __attribute__((aligned(16))) float Ss[32 * EPNN_SST];
    __shared__ __attribute__((aligned(16))) float Ai[32 * EPNN_AST];
    __shared__ __attribute__((aligned(16))) float Bs[160];     // the update MLP's biases: [cb3p | bu1p | bu2p | bu3p]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hh = lane >> 5;
    LG_CLK(16 * X.run, 0);
    const int np = L.row_off[L.A];                               // (looked at below: the loads in between do not depend on the pair list)
    const int4 tl = L.atiles[blockIdx.x];
    const float *wp = L.wpack;
    const int row = c < tl.y ? c : 0;
    float w1[40], w2[16], w3[32], wA[EPNN_KA];
    f32x16 acc_h = epnn_splat16(0.f);
    if (wave == 0) {
        // Wavefront 0 takes no part in the reduction: it has the update MLP's weights in registers from the start and runs the h
        // part of the first layer (24 of its 40 matrix instructions, straight from the tile's rows in memory) while the others
        // reduce S -- the chain and its order are those of lg_update_wave, so the bits are the same.
        LG_LOAD_UPD_WEIGHTS(w1, w2, w3, U)
        const int u0 = (L.nx - hh + 1) >> 1;
        acc_h = lg_update_h(L, U, w1, L.a_eo + (size_t)(tl.x + row) * EPNN_AST + hh * 32 + u0, lane, nullptr);
    }
    if (tid >= 352) {                                            // (waves 5..7: nothing else to request)
        const int k = tid - 352;
        Bs[k] = wp[(k < 32 ? U.cb3p : k < 64 ? U.bu1p - 32 : k < 96 ? U.bu2p - 64 : U.bu3p - 96) + k];
    }
    int job = -1;                                                // EPN static projections: job = 2 t + (0: P, 1: R)
    float w2n[17];                                               // next sweep: W2 fragments + b2[c] for the zp / Yb chains of waves 1 / 2
    if (X.run == 1 && (wave == 1 || wave == 2)) {
        const int off = wave == 1 ? X.M.wiF : X.M.wjF;
#pragma unroll
        for (int s = 0; s < EPNN_KA; ++s) wA[s] = wp[off + s * 64 + lane];
#pragma unroll
        for (int s = 0; s < 16; ++s) w2n[s] = wp[X.M.w2F + s * 64 + lane];
        w2n[16] = wp[X.M.b2 + c];
    } else if (X.run == 2 && wave >= 1 && wave - 1 < 2 * L.T) {
        job = wave - 1;
        const PairMlpPack &M = L.wi.pas[job >> 1];
        const int off = (job & 1) ? M.wjF : M.wiF;
#pragma unroll
        for (int s = 0; s < EPNN_KA; ++s) wA[s] = wp[off + s * 64 + lane];
    }
    for (int i = tid; i < tl.y * (EPNN_AST / 4); i += 512)
        reinterpret_cast<f32x4 *>(Ai)[i] = reinterpret_cast<const f32x4 *>(L.a_eo + (size_t)tl.x * EPNN_AST)[i];
    if (np > L.pcap) return;
    if (wave > 0) {                                              // 448 threads: output o of atoms g, g + 14, g + 28
        const int t7 = tid - 64, o = t7 & 31, g = t7 >> 5, n = L.moff[tl.z + 1] - L.moff[tl.z];
        const int a[3] = {g, g + 14, g + 28};
        float sv[3];
        LG_CLK(16 * X.run, 1);
        lg_reduce<3>(L, tl, a, o, n, types, sv);
        LG_CLK(16 * X.run, 2);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (a[k] < 32) Ss[a[k] * EPNN_SST + o] = sv[k];
    }
    __syncthreads();
    LG_CLK(16 * X.run, 3);
    // (the new h goes into the LDS image only; wavefront 0 copies the tile's rows to memory as 16-byte stores AFTER the barrier,
    //  beside the other wavefronts' projections: 24 four-byte stores per lane in front of the barrier were waited for by all)
    if (wave == 0) lg_update_rest(L, U, acc_h, w1, w2, w3, tl, Ss + c * EPNN_SST, Ai + row * EPNN_AST, nullptr, lane, Bs);
    LG_CLK(16 * X.run, 4);
    __syncthreads();                                            // the image now holds the new h
    LG_CLK(16 * X.run, 5);
    if (wave == 0) {
        f32x4 *dst = reinterpret_cast<f32x4 *>(L.a_eo + (size_t)tl.x * EPNN_AST);
        for (int i = lane; i < tl.y * (EPNN_AST / 4); i += 64) dst[i] = reinterpret_cast<const f32x4 *>(Ai)[i];
    }
    if (!X.run) return;
    const float *arow = Ai + row * EPNN_AST + hh * 32;
    if (X.run == 1) {
        if (wave == 1) lg_proj_wave<1, true, false, false, true>(L, X.M, 1, tl, arow, lane, wA, L.P, L.R, nullptr, w2n);
        if (wave == 2) lg_proj_wave<2, true, false, false, true>(L, X.M, 1, tl, arow, lane, wA, L.P, L.R, nullptr, w2n);
        return;
    }
    for (; job >= 0 && job < 2 * L.T; job += 7) {
        const int t = job >> 1;
        float *oP = L.Pst + (size_t)t * L.A * 32, *oR = L.Rst + (size_t)t * L.A * 32;
        if (job != wave - 1) {                                   // a second job of this wave: its fragments were not prefetched
            const PairMlpPack &M = L.wi.pas[t];
            const int off = (job & 1) ? M.wjF : M.wiF;
#pragma unroll
            for (int s = 0; s < EPNN_KA; ++s) wA[s] = wp[off + s * 64 + lane];
        }
        if (job & 1) lg_proj_wave<2, true, true>(L, L.wi.pas[t], 0, tl, arow, lane, wA, oP, oR);
        else lg_proj_wave<1, true, true>(L, L.wi.pas[t], 0, tl, arow, lane, wA, oP, oR);
    }
}

// ------------------------------------------------------------------------------------------------ EPN stack
// status bits + number of listed pairs to the host (every other kernel of the forward ran before this one on the stream)
__device__ __forceinline__ void lg_handoff(const LargeArgs &L) {
    volatile int *hs = L.host_status;
    hs[0] = *L.status;
    hs[1] = L.row_off[L.A];
    *L.status = 0;                                             // ... and leaves the word clear for the next forward (no memset in front of it)
    __threadfence_system();
}
// sum of an atom's incidence row of transfers, slot order
__device__ __forceinline__ float lg_slot_sum(const float *dl, int lo, int hi, float acc = 0.f) {
    for (; lo < hi; lo += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = dl[min(lo + u, hi - 1)];      // unconditional: a load under a condition is a branch, one round trip each
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (lo + u < hi) acc += v[u];
    }
    return acc;
}
// EPN step t (charge_gn.py:101-118): one wave per 32 listed pairs, one pair per column.  Lane (c, hh): half hh = 0 rebuilds
// q of the pair's first atom, hh = 1 of its second (q_t = q_{t-1} + its slot row of step t-1), the halves swap them; the
// pair whose slot opens an atom's row stores the atom's new q for step t+1's readers.  Then
// z1 = relu(G + (Pst_i + q_i wqi) + (Rst_j + q_j wqj)) both ways, delta = 0.5 (f_ij - f_ji), deposits +w_i delta, -w_j delta.
__global__ __launch_bounds__(256) void k_lg_epn_step(LargeArgs L, PairMlpPack M, int t) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, hh = lane >> 5;
    const int slot = (blockIdx.x * 4 + wave) * 32 + c;
    // the pair count and the pair records travel together (the records' array is padded: any slot of the grid may be read)
    LG_CLK(64 + 8 * t, 0);
    // the weights do not depend on the pair record: requested first (56 loads; a wavefront has 63 in flight)
    const float *wp = L.wpack;
    float wg[24], w2[16], wqi[16], wqj[16], b2v[16], w3[16];
#pragma unroll
    for (int s = 0; s < 24; ++s) wg[s] = wp[M.weF + s * 64 + lane];
#pragma unroll
    for (int s = 0; s < 16; ++s) w2[s] = wp[M.w2F + s * 64 + lane];
    epnn_ld16(wp + M.wqi + hh * 16, wqi);
    epnn_ld16(wp + M.wqj + hh * 16, wqj);
    epnn_ld16(wp + M.b2p + hh * 16, b2v);
    epnn_ld16(wp + M.w3p + hh * 16, w3);
    const int np = L.row_off[L.A];
    const int4 ra0 = L.prec[2 * slot], rb0 = L.prec[2 * slot + 1];
    if (np > L.pcap) return;
    bool valid = slot < np && ra0.x >= 0;                      // (a pair of a molecule that is not on this path: -1 - i)
    const int4 ra = valid ? ra0 : make_int4(0, 0, 0, 0), rb = valid ? rb0 : make_int4(0, 0, 0, -1);
    if (__ballot(valid) == 0ull) return;
    const int gi = valid ? ra.x : 0, gj = valid ? ra.y : 0;
    const float *Pst = L.Pst + (size_t)t * L.A * 32, *Rst = L.Rst + (size_t)t * L.A * 32;
    // this half's atom: its charge of the previous step and the transfers it received there
    const int mine = hh ? gj : gi, lo = hh ? rb.x : ra.z, hi = hh ? rb.y : ra.w;
    const float *qprev = L.qbuf + (size_t)((t + 1) & 1) * L.A;          // q_{t-1} lives in generation (t-1) & 1
    float *qnext = L.qbuf + (size_t)(t & 1) * L.A;
    const float *dlp = L.dlA + (size_t)((t + 1) & 1) * 2 * L.pcap;
    // everything that hangs on the pair record is requested TOGETHER, unconditionally (clamped indices): the atom's charge and
    // the first sixteen slots of its row, the pair's e row, the four P / R rows -- one round trip behind the record instead of
    // three (charge and slots, then the e row, then the rows; clock stamps: 2.4 us of a 6 us workgroup)
    const float qload = (t == 0 ? L.qbuf : qprev)[mine];
    float sv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) sv[u] = dlp[min(lo + u, max(hi - 1, lo))];
    float ev[24];
    {
        const float *erow = L.pe + (size_t)(valid ? slot : 0) * EPNN_EDIM + hh * 24;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(erow + 4 * q);
            ev[4 * q] = valid ? v[0] : 0.f; ev[4 * q + 1] = valid ? v[1] : 0.f; ev[4 * q + 2] = valid ? v[2] : 0.f; ev[4 * q + 3] = valid ? v[3] : 0.f;
        }
    }
    float pi_[16], rj_[16], pj_[16], ri_[16];
    epnn_ld16(Pst + (size_t)gi * 32 + hh * 16, pi_);
    epnn_ld16(Rst + (size_t)gj * 32 + hh * 16, rj_);
    epnn_ld16(Pst + (size_t)gj * 32 + hh * 16, pj_);
    epnn_ld16(Rst + (size_t)gi * 32 + hh * 16, ri_);
    // G = We e on the matrix pipe while the rest is still on its way
    f32x16 g = epnn_splat16(0.f);
#pragma unroll
    for (int s = 0; s < 24; ++s) g = epnn_mfma(wg[s], ev[s], g);
    float qa = 0.f;
    if (valid) {
        qa = qload;
        if (t > 0) {
            float acc = 0.f;                                              // the row's sum in slot order, then added to the charge
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (lo + u < hi) acc += sv[u];
            if (hi - lo > 16) acc = lg_slot_sum(dlp, lo + 16, hi, acc);
            qa += acc;
            const int dest = hh ? rb.w : rb.z;
            if (dest == lo) qnext[mine] = qa;                             // the pair that opens the atom's row keeps its charge
        }
    }
    LG_CLK(64 + 8 * t, 1);
    const float qo = epnn_swap32(qa);
    const float qi = hh ? qo : qa, qj = hh ? qa : qo;
    f32x16 au, av;
    LG_CLK(64 + 8 * t, 2);
#pragma unroll
    for (int r = 0; r < 16; ++r) { au[r] = b2v[r]; av[r] = b2v[r]; }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float pu = fmaf(qi, wqi[s], pi_[s]), ru = fmaf(qj, wqj[s], rj_[s]);
        const float pv = fmaf(qj, wqi[s], pj_[s]), rv = fmaf(qi, wqj[s], ri_[s]);
        au = epnn_mfma(w2[s], fmaxf((g[s] + pu) + ru, 0.f), au);
        av = epnn_mfma(w2[s], fmaxf((g[s] + pv) + rv, 0.f), av);
    }
    LG_CLK(64 + 8 * t, 3);
    float fu = 0.f, fv = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        fu = fmaf(w3[r], fmaxf(au[r], 0.f), fu);
        fv = fmaf(w3[r], fmaxf(av[r], 0.f), fv);
    }
    fu += epnn_swap32(fu);
    fv += epnn_swap32(fv);
    if (valid) {
        const float dl = 0.5f * (fu - fv);
        float *out = L.dlA + (size_t)(t & 1) * 2 * L.pcap;
        if (hh == 0) out[rb.z] = L.pwi[slot] * dl;
        else if (rb.w >= 0) out[rb.w] = -(L.pwj[slot] * dl);
    }
    LG_CLK(64 + 8 * t, 4);
}
// after the last step: q_T = q_{T-1} + the atom's last slot row (thread per atom), and the hand-over to the host
__global__ __launch_bounds__(256) void k_lg_epn_final(LargeArgs L) {
    if (L.host_status && blockIdx.x == 0 && threadIdx.x == 0) lg_handoff(L);
    if (L.row_off[L.A] > L.pcap) return;
    const int gen = (L.T - 1) & 1;
    for (int at = blockIdx.x * 256 + threadIdx.x; at < L.A; at += gridDim.x * 256) {
        if (!L.mflag[L.mol_of[at]]) continue;
        const float q = L.qbuf[(size_t)gen * L.A + at] +
                        lg_slot_sum(L.dlA + (size_t)gen * 2 * L.pcap, L.inc_off[at], L.inc_off[at + 1]);
        if (L.q_out) L.q_out[at] = q;
    }
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
    std::vector<int4> atiles, stasks, stasks2;
    std::vector<int> stask_chunk;
    std::vector<float> slack;               // per sweep task: partners its range is shorter than its molecule's longest
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
        if (ntile <= EPNN_LG_TW_TILES && !h->opt_large_sweep_old) {
            // tile-workgroup sweep (k_lg_sweep2): workgroup = one tile x four pieces of 1 / nwg of the partner range.  nwg depends on
            // the molecule alone (its results do not depend on what else is in the batch): as many as keep one molecule's
            // workgroups resident at once (two per CU), pieces of at least ~16 partners
            int nwg = std::max(1, std::min(512 / ntile, (n + 63) / 64));
            if (h->opt_large_chunks > 0) nwg = std::min(h->opt_large_chunks, n);
            const int wlen = (n + nwg - 1) / nwg;
            nwg = (n + wlen - 1) / wlen;
            for (int i0 = 0; i0 < n; i0 += 32) lp.atiles.push_back(make_int4(a0 + i0, std::min(32, n - i0), b, nwg));
            lp.maxchunk = std::max(lp.maxchunk, nwg);
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
                for (int t = tg; t < std::min(ntile, tg + 4); ++t)
                    for (int w = 0; w < nwg; ++w) lp.stasks2.push_back(make_int4(first_tile + t, w, a0 + w * wlen, a0 + std::min(n, (w + 1) * wlen)));
            }
            continue;
        }
        // split the j range so that the sweep has a few thousand workgroups, in pieces of EPNN_LG_JC atoms
        // workgroups of the sweep = ngroup * nchunk.  Measured on the 2220-atom protein: a whole number of rounds of 512
        // workgroups (two per CU) beats 1024 smaller ones (per-workgroup prologue and more partial sums to reduce);
        // big systems get >= 2048 workgroups for a fine-grained tail.
        int want;
        if (ngroup >= 512) want = std::max(1, (4096 + ngroup - 1) / ngroup);
        else want = std::max(1, (int)std::lround(512.0 * std::max(1, (int)std::lround(ngroup / 64.0)) / ngroup));
        if (h->opt_large_chunks > 0) want = h->opt_large_chunks;
        int nchunk = std::max(1, std::min(want, (n + 15) / 16));
        int clen = (n + nchunk - 1) / nchunk;
        // The step's near-pair correction tiles ride on the sweep's wavefronts (k_lg_sweep): a tile costs a wavefront about
        // as long as four partners.  Lengthen the pieces a little so that the LAST piece of the range comes out short enough
        // for its workgroups to take the molecule's tiles (~6 pairs per atom expected) in the time the others sweep.
        {
            const double tiles = 1.2 * (double)n * 6.0 / 32.0;
            while (nchunk > 1 && (double)ngroup * (double)((long long)nchunk * clen - n) < tiles &&
                   (long long)(nchunk - 1) * (clen + 1) < n && clen < n)
                clen += 1;
        }
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
                lp.slack.push_back((float)(clen - (std::min(n, (ch + 1) * clen) - ch * clen)));
            }
        }
    }
    if (h->part_row_hi < h->part_row_lo) h->part_row_lo = h->part_row_hi = 0;        // no group of its own
    for (int r = 0; r < h->part_world; ++r)
        if (h->part_hi[r] < h->part_lo[r]) h->part_lo[r] = h->part_hi[r] = 0;
    h->l_natiles = (int)lp.atiles.size();
    h->l_nstasks = (int)lp.stasks.size();
    h->l_maxchunk = lp.maxchunk;
    // the tile-workgroup launch: its sweep workgroups, then workgroups that run correction tiles only.  While all of them are
    // resident at once (two per CU) the spare slots take the step's correction tiles -- a wavefront with a full sweep task that
    // also ran a tile would end ~5 us after the others --; without (enough) spare slots the tiles are dealt evenly.
    std::vector<float2> frac2;
    {
        const int ns2 = (int)lp.stasks2.size();
        double tiles = 0;
        for (int b : P.large_list) tiles += 1.2 * (double)(P.offsets[b + 1] - P.offsets[b]) * 6.0 / 32.0;
        const int spare = ns2 > 0 && lp.stasks.empty() ? std::max(0, 512 - ns2) : 0;
        int ncorr = std::min(spare, (int)std::ceil(tiles / 12.0));           // ~3 tiles per wavefront
        if (ncorr > 0 && tiles / (4.0 * ncorr) > 10.0) ncorr = 0;               // too few: dealt evenly instead
        for (int k = 0; k < ncorr; ++k) lp.stasks2.push_back(make_int4(-1, 0, 0, 0));
        frac2.resize(lp.stasks2.size());
        if (!lp.stasks.empty()) {
            for (auto &f : frac2) f = make_float2(0.f, 0.f);                    // (the four-tile launch of the same step runs them)
        } else {
            const size_t first = ncorr > 0 ? (size_t)ns2 : 0, cnt = frac2.size() - first;
            for (size_t k = 0; k < frac2.size(); ++k) {
                if (k < first) { frac2[k] = make_float2(0.f, 0.f); continue; }
                frac2[k].x = (float)((double)(k - first) / (double)cnt);
                frac2[k].y = k + 1 == frac2.size() ? 1.f : (float)((double)(k - first + 1) / (double)cnt);
            }
        }
    }
    h->l_nstasks2 = (int)lp.stasks2.size();
    if (P.large_list.empty()) return 0;          // (the molecule flags travel with the plan's other index arrays)
    const size_t A = (size_t)P.A, nl = P.large_list.size(), T = (size_t)std::max(1, h->cfg.T);
    if (h->l_tiles.ensure(lp.atiles.size() * sizeof(int4)) || h->l_stasks.ensure(lp.stasks.size() * sizeof(int4)) ||
        h->l_schunk.ensure(lp.stask_chunk.size() * sizeof(int)) || h->l_stasks2.ensure(lp.stasks2.size() * sizeof(int4)) ||
        h->l_sfrac2.ensure(std::max<size_t>(1, frac2.size()) * sizeof(float2)) ||
        h->l_a.ensure(A * EPNN_AST * 4) ||
        h->l_P.ensure(A * 32 * 4) || h->l_R.ensure(A * 32 * 4) || h->l_zp.ensure(A * 32 * 4) ||
        h->l_Nn.ensure(A * 32 * 4) || h->l_Yb.ensure(A * 32 * 4) || h->l_qbuf.ensure(2 * A * 4) ||
        h->l_Pst.ensure(T * A * 32 * 4) || h->l_Rst.ensure(T * A * 32 * 4) ||
        h->l_S0.ensure((size_t)lp.maxchunk * A * 32 * 4) || h->l_csr_off.ensure((A + 1) * sizeof(int)) ||
        h->l_cnt.ensure(2 * (A + 1) * sizeof(int)) || h->l_sfin.ensure(A * 32 * 4) ||
        h->l_lmol.ensure(nl * sizeof(int)) || h->l_typrow.ensure(A * sizeof(int)) ||
        h->l_typtab.ensure(nl * (2 * EPNN_TYPE_MAX + 1) * sizeof(int)) || h->l_typhash.ensure((A + nl * EPNN_TYPE_MAX) * 8) || h->l_stype.ensure(nl * EPNN_TYPE_MAX * 32 * 4))
        return 1;
    // shares of the correction tiles: by slack where the tasks' slack can take them all, else evenly on top
    std::vector<float2> frac(lp.stasks.size());
    {
        double total = 0, tiles = 0;
        for (float v : lp.slack) total += v;
        for (int b : P.large_list) tiles += (double)(P.offsets[b + 1] - P.offsets[b]) * 6.0 / 32.0;
        const double even = total >= tiles ? 0.0 : std::max(1.0, (tiles - total) / std::max<size_t>(1, lp.slack.size()));
        double sum = 0, run = 0;
        for (float v : lp.slack) sum += v + even;
        for (size_t k = 0; k < frac.size(); ++k) {
            const double w = sum > 0 ? (lp.slack[k] + even) / sum : 1.0 / frac.size();
            frac[k].x = (float)run;
            run += w;
            frac[k].y = k + 1 == frac.size() ? 1.f : (float)run;
        }
    }
    if (h->l_sfrac.ensure(std::max<size_t>(1, frac.size()) * sizeof(float2))) return 1;
    if (!frac.empty()) HIPCHK(hipMemcpyAsync(h->l_sfrac.p, frac.data(), frac.size() * sizeof(float2), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->l_tiles.p, lp.atiles.data(), lp.atiles.size() * sizeof(int4), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->l_lmol.p, P.large_list.data(), nl * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (!lp.stasks2.empty()) {
        HIPCHK(hipMemcpyAsync(h->l_stasks2.p, lp.stasks2.data(), lp.stasks2.size() * sizeof(int4), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->l_sfrac2.p, frac2.data(), frac2.size() * sizeof(float2), hipMemcpyHostToDevice, h->stream));
    }
    if (!lp.stasks.empty()) {
        HIPCHK(hipMemcpyAsync(h->l_stasks.p, lp.stasks.data(), lp.stasks.size() * sizeof(int4), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->l_schunk.p, lp.stask_chunk.data(), lp.stask_chunk.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// capacity of everything that is sized by the pair list
static int ensure_large_pairs(epnn_handle *h) {
    const size_t pc = (size_t)h->pcap;
    return h->l_corr.ensure(pc * 2 * 32 * 4) || h->l_dl.ensure(2 * 2 * pc * 4) || h->l_csr_ent.ensure(pc * sizeof(int));
}

struct PairSource;
// `have_inc`: the pair list came with its incidence rows (compact entry: epnn_frontend.hip.h); otherwise (dense front-end) they
// are built here.  `front`: when not null the pair list has NOT been built yet and this function drives the front-end's
// launches itself, merged with what needs only the atoms (k_lg_first / k_lg_second above).
// does a forward of this plan run the partition's row exchange over the handle's RCCL communicator?  (Every process has the same
// inputs and options, so every process answers alike.)
static bool large_exchanges_over_rccl(const epnn_handle *h, int run_gnn) {
    return (h->part_world > 1 || h->opt_part_collective) && !h->part_exchange && h->comm && !h->plan.large_list.empty() && run_gnn && h->cfg.T > 0;
}
static int launch_large_body(epnn_handle *h, const float *d_x, const float *d_Q, const float *d_hin, const float *d_qin,
                             const float *d_nm, float *d_q, float *d_hout, int run_gnn, int run_epn, bool have_inc, const FrontArgs *front);
static int launch_large_impl(epnn_handle *h, const float *d_x, const float *d_Q, const float *d_hin, const float *d_qin,
                             const float *d_nm, float *d_q, float *d_hout, int run_gnn, int run_epn, bool have_inc, const FrontArgs *front) {
    // a process that fails on its way to a row exchange still joins the exchange's status collective (comm_guard, epnn_host.h)
    if (large_exchanges_over_rccl(h, run_gnn)) h->guard_pending = true;
    return comm_guard_exit(h, launch_large_body(h, d_x, d_Q, d_hin, d_qin, d_nm, d_q, d_hout, run_gnn, run_epn, have_inc, front),
                           "partitioned forward (row exchange)");
}
static int launch_large_body(epnn_handle *h, const float *d_x, const float *d_Q, const float *d_hin, const float *d_qin,
                             const float *d_nm, float *d_q, float *d_hout, int run_gnn, int run_epn, bool have_inc, const FrontArgs *front) {
    const Plan &P = h->plan;
    hipStream_t st = h->stream;
    const unsigned gLink = (unsigned)std::min<size_t>(((size_t)h->pcap + 255) / 256, 1024);
    const unsigned rows4 = (unsigned)((P.A + 3) / 4);
    auto front_alone = [&]() -> int {          // the front-end as four launches of its own
        hipLaunchKernelGGL(k_front_count, dim3(rows4), dim3(256), 0, st, *front);
        hipLaunchKernelGGL(k_front_scan_both, dim3(1), dim3(1024), 0, st, *front);
        hipLaunchKernelGGL(k_front_fill, dim3(rows4), dim3(256), 0, st, *front);
        hipLaunchKernelGGL(k_front_link, dim3(gLink), dim3(256), 0, st, *front);
        HIPCHK(hipGetLastError());
        return 0;
    };
    if (P.large_list.empty()) return front ? front_alone() : 0;
    if (ensure_large_pairs(h) || h->d_incoff.ensure(((size_t)P.A + 1) * sizeof(int))) return 1;
    const size_t pc = (size_t)h->pcap;
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
    L.Nn = h->l_Nn.as<float>();
    L.Yb = h->l_Yb.as<float>();
    L.zp = h->l_zp.as<float>();
    L.S0 = h->l_S0.as<float>();
#ifdef EPNN_LG_CLOCKS
    if (h->lg_clk.ensure((128 + 4 * 1024) * 8)) return 1;
    L.clk = h->lg_clk.as<unsigned long long>();
#endif
    L.corrA = h->l_corr.as<float>();
    L.dlA = h->l_dl.as<float>();
    L.qbuf = h->l_qbuf.as<float>();
    L.Pst = h->l_Pst.as<float>();
    L.Rst = h->l_Rst.as<float>();
    L.row_off = h->d_rowoff.as<int>();
    L.pi = h->d_pi.as<int>();
    L.pj = h->d_pj.as<int>();
    L.psym = h->d_psym.as<int>();
    L.pe = h->d_pe.as<float>();
    L.pwi = h->d_pwi.as<float>();
    L.pwj = h->d_pwj.as<float>();
    L.inc_off = h->d_incoff.as<int>();
    L.dest_i = h->d_desti.as<int>();
    L.dest_j = h->d_destj.as<int>();
    L.prec = h->d_prec.as<int4>();
    L.dn_cnt = h->l_cnt.as<int>();
    L.dn_cur = h->l_cnt.as<int>() + (size_t)P.A + 1;
    L.dn_off = h->l_csr_off.as<int>();
    L.dn_ent = h->l_csr_ent.as<int>();
    L.w_inc_off = h->d_incoff.as<int>();
    L.w_dest_i = h->d_desti.as<int>();
    L.w_dest_j = h->d_destj.as<int>();
    L.w_prec = h->d_prec.as<int4>();
    L.atiles = h->l_tiles.as<int4>();
    L.natiles = h->l_natiles;
    L.stasks = h->l_stasks.as<int4>();
    L.stask_chunk = h->l_schunk.as<int>();
    L.stask_frac = h->l_sfrac.as<float2>();
    L.nstasks = h->l_nstasks;
    L.stasks2 = h->l_stasks2.as<int4>();
    L.stask2_frac = h->l_sfrac2.as<float2>();
    L.nstasks2 = h->l_nstasks2;
    L.pcap = h->pcap;
    L.lmol = h->l_lmol.as<int>();
    L.nlarge = (int)P.large_list.size();
    L.typ_row = h->l_typrow.as<int>();
    L.typ_rep = h->l_typtab.as<int>();
    L.typ_cnt = L.typ_rep + (size_t)L.nlarge * EPNN_TYPE_MAX;
    L.typ_n = L.typ_cnt + (size_t)L.nlarge * EPNN_TYPE_MAX;
    L.S_type = h->l_stype.as<float>();
    L.typ_hash = h->l_typhash.as<unsigned long long>();
    L.typ_hsh = L.typ_hash + (size_t)P.A;
    // compact entry: the forward's last kernel hands status + pair count to the host (two device-to-host copies less)
    L.host_status = (h->want_large_handoff && run_epn && h->cfg.T > 0 && h->part_world == 1) ? h->h_status : nullptr;
    h->did_large_handoff = L.host_status != nullptr;
    L.q_out = d_q;
    L.h_out = d_hout;
    L.status = h->d_status.as<int>();
    const unsigned gA = (unsigned)std::min<size_t>(((size_t)P.A * EPNN_AST + 255) / 256, 4096);
    const unsigned gP = (unsigned)std::min<size_t>((pc + 255) / 256, 4096);
    const unsigned gAt = (unsigned)std::min<size_t>(((size_t)P.A + 255) / 256, 4096);
    const unsigned gT = (unsigned)((L.natiles + 3) / 4);
    const unsigned gPT = (unsigned)((pc + 127) / 128);
    const unsigned gES = (unsigned)((L.natiles * L.T + 3) / 4);
    const int Tg = run_gnn ? L.T : 0, Te = run_epn ? L.T : 0;
    // the first GNN step by atom types: the compact entry only (h = 0 and one q per molecule: a feature row depends on x alone)
    const bool types = Tg > 0 && h->opt_large_dedupe && !h->types_overflowed && !d_hin && !d_qin && !d_nm;
    // Launch sequence (the single-process case): first projections | per GNN step: sweep (its waves also run the step's
    // correction tiles), tail (reduce, update, next projections) | per EPN step: pair tiles; a last launch adds the final
    // transfers.  With a partition the other processes' rows of S arrive between the reduction and the update, so those stay
    // separate launches.
    const bool collective = h->part_world > 1 || h->opt_part_collective;
    const bool split = collective || !h->opt_large_fused || upd_generic_stage(h);
    const bool merged = front && Tg > 0 && !split;           // k_lg_first / k_lg_second
    const unsigned gTile = (unsigned)L.natiles;
    auto next_after_gnn = [&](int t) {
        LgNext X{};
        if (t + 1 < Tg) { X.M = h->widx.msg[t + 1]; X.run = 1; }
        else if (Te > 0) X.run = 2;
        return X;
    };
    bool step0_pairs_done = false;
    if (merged) {
        LgFirst W1{(int)gT, (int)rows4, types ? 1 : 0};
        hipLaunchKernelGGL(k_lg_first, dim3((unsigned)(W1.tile_wgs + W1.count_wgs)), dim3(256), 0, st, L, h->widx.msg[0], W1, *front);
        // (up to EPNN_FRONT_INLINE_A atoms: no scan launch, see k_lg_fill_types)
        const bool inl = P.A <= EPNN_FRONT_INLINE_A && h->opt_front_inline;
        const unsigned gAssign = types ? (unsigned)((P.A + 255) / 256) : 0u;
        if (inl) {
            hipLaunchKernelGGL(k_lg_fill_types, dim3(rows4 + (types ? (unsigned)L.nlarge : 0u)), dim3(256), 0, st, L, *front, (int)rows4);
        } else {
            hipLaunchKernelGGL(k_lg_scan_types, dim3(1u + (types ? (unsigned)L.nlarge : 0u)), dim3(1024), 0, st, L, *front);
            hipLaunchKernelGGL(k_lg_fill_assign, dim3(rows4 + gAssign), dim3(256), 0, st, L, *front, (int)rows4);
        }
        if (types) {
            LgSecond W2{(int)gLink, L.nlarge * 2, P.fused_count() == 0 ? 1 : 0, inl ? (int)gAssign : 0};
            hipLaunchKernelGGL(k_lg_second, dim3((unsigned)(W2.link_wgs + W2.tsweep_wgs + W2.assign_wgs) + gPT), dim3(256), 0, st, L, h->widx.msg[0], W2, *front);
            step0_pairs_done = true;
        } else {
            hipLaunchKernelGGL(k_front_link, dim3(gLink), dim3(256), 0, st, *front);
        }
    } else {
        if (front && front_alone()) return 1;
        hipLaunchKernelGGL(k_lg_init, dim3(gA), dim3(256), 0, st, L);
        if (types) hipLaunchKernelGGL(k_lg_types, dim3((unsigned)L.nlarge), dim3(1024), 0, st, L);
        if (Tg > 0) hipLaunchKernelGGL(k_lg_proj, dim3(gT), dim3(256), 0, st, L, h->widx.msg[0], 1);
        else if (Te > 0) hipLaunchKernelGGL(k_lg_epn_static, dim3(gES), dim3(256), 0, st, L);
        if (!have_inc && !front) {
            HIPCHK(hipMemsetAsync(L.dn_cnt, 0, 2 * ((size_t)P.A + 1) * sizeof(int), st));
            hipLaunchKernelGGL(k_lg_dn_count, dim3(gP), dim3(256), 0, st, L);
            hipLaunchKernelGGL(k_lg_dn_scan, dim3(1), dim3(1024), 0, st, L);
            hipLaunchKernelGGL(k_lg_dn_fill, dim3(gP), dim3(256), 0, st, L);
            hipLaunchKernelGGL(k_lg_dn_link, dim3(gP), dim3(256), 0, st, L);
        }
        if (types) hipLaunchKernelGGL(k_lg_tsweep, dim3((unsigned)L.nlarge * 2), dim3(64), 0, st, L, h->widx.msg[0]);
    }
    // the sweep's workgroups per CU: as many as its task count fills evenly, enforced through the launch's LDS size
    const int wpc = L.nstasks <= 256 ? 1 : L.nstasks <= 512 ? 2 : L.nstasks <= 768 ? 3 : 4;
    const size_t sweep_lds = std::max<size_t>((size_t)2 * EPNN_LG_JC * 32 * 4, ((size_t)163840 / wpc) & ~size_t(255));
    if (L.nstasks + L.nstasks2 > 0 && !h->sweep_attr) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lg_sweep), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lg_sweep2), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
        h->sweep_attr = true;
    }
    // tile-workgroup launch: 80 KB of LDS per workgroup = two per CU
    const size_t sweep2_lds = (size_t)4 * EPNN_LG_JP * 64 * sizeof(float);
    for (int t = 0; t < Tg; ++t) {
        const bool ty = types && t == 0;
        const bool sweep = !ty && L.nstasks + L.nstasks2 > 0;
        if (sweep) {
            if (L.nstasks > 0) hipLaunchKernelGGL(k_lg_sweep, dim3((unsigned)L.nstasks), dim3(256), sweep_lds, st, L, lg_w2_off(h, t), h->widx.msg[t], 1);
            if (L.nstasks2 > 0) hipLaunchKernelGGL(k_lg_sweep2, dim3((unsigned)L.nstasks2), dim3(256), sweep2_lds, st, L, lg_w2_off(h, t), h->widx.msg[t], L.nstasks > 0 ? 0 : 1);
        }
        else if (!(ty && step0_pairs_done)) hipLaunchKernelGGL(k_lg_pairs, dim3(gPT), dim3(256), 0, st, L, h->widx.msg[t]);
        if (!split) {
            hipLaunchKernelGGL(k_lg_gnn_tail, dim3(gTile), dim3(512), 0, st, L, h->widx.upd[t], next_after_gnn(t), ty ? 1 : 0);
            continue;
        }
        hipLaunchKernelGGL(k_lg_reduce, dim3((unsigned)L.natiles * 4), dim3(256), 0, st, L, h->l_sfin.as<float>(), ty ? 1 : 0);
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
                // link at once); the update kernel simply follows on the stream
                if (!h->comm || h->comm_world != h->part_world) EPNN_FAIL("forward: a partition is set but neither an exchange function nor a communicator");
                // every process says "my rows are on their way" before anyone enqueues the broadcasts (comm_guard: one 4-byte
                // all-reduce and a look at its result -- the one host synchronisation of a partitioned GNN step)
                if (comm_guard(h, 0, "partitioned forward (row exchange)")) return 1;
                if (t + 1 < Tg) h->guard_pending = true;          // the next step's exchange is owed from here on
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
        if (upd_generic_stage(h)) {
            GenMlp G = h->gen_upd;
            G.w = h->d_updgen.as<float>();
            hipLaunchKernelGGL(k_lg_update_generic, dim3(2u * (unsigned)L.natiles), dim3(256), 0, st, L, G, h->gen_w3[t], h->gen_b3[t], h->l_sfin.as<float>());
        } else
            hipLaunchKernelGGL(k_lg_update, dim3(gT), dim3(256), 0, st, L, h->widx.upd[t], h->l_sfin.as<float>());
        const LgNext X = next_after_gnn(t);
        if (X.run == 1) hipLaunchKernelGGL(k_lg_proj, dim3(gT), dim3(256), 0, st, L, X.M, 1);
        else if (X.run == 2) hipLaunchKernelGGL(k_lg_epn_static, dim3(gES), dim3(256), 0, st, L);
    }
    if (d_hout) hipLaunchKernelGGL(k_lg_export_h, dim3(gA), dim3(256), 0, st, L);
    for (int t = 0; t < Te; ++t) hipLaunchKernelGGL(k_lg_epn_step, dim3(gPT), dim3(256), 0, st, L, h->widx.pas[t], t);
    if (Te > 0) hipLaunchKernelGGL(k_lg_epn_final, dim3(gAt), dim3(256), 0, st, L);
    if (d_q && Te == 0) hipLaunchKernelGGL(k_lg_export_q, dim3(gAt), dim3(256), 0, st, L);
    HIPCHK(hipGetLastError());
    return 0;
}
