// Fused small-molecule forward: ONE workgroup (4 waves) runs the whole GNN_layer + EPN_layer stack
// (reference charge_gn.py:56-119, 2T steps) for one molecule with n <= 32 real atoms; atom features, the
// per-atom projections P/R, the near-pair terms G and all partial sums stay in LDS for the whole forward.
//
// Pair sweep of the GNN (charge_gn.py:62-70): the n x npad pair rows (npad = 4*ceil((n+1)/4) > n; rows with
// j >= n are "padded partner" rows with R = 0, G = 0, i.e. exactly the rows the reference evaluates for its
// zero-padded atoms) are packed 32 to an MFMA tile.  Because npad is a multiple of 4 and an accumulator quad
// (registers 4q..4q+3 of one half-wave) is 4 consecutive rows, every quad belongs to ONE atom i: its relu-sum is
// added to that atom's partial sum, no per-row masking.  The value of the last row of each i (a padded-partner
// row) is kept as zp_i and added (N - npad) times: together this is the reference's sum over all N partners.
//
// EPN (charge_gn.py:98-118): one tile row per UNORDERED near pair; both directions share G; the transfer
// 0.5*(f_ij - f_ji) is applied as +d to i and -d to j, so total charge is conserved by construction.
#pragma once
#include "epnn_common.h"

struct SmallLds {   // offsets in 4-byte words into dynamic LDS
    int a_eo, P, R, Sw, zp, G, dl, pij, pwi, pwj, pm, glut, nm, total;
};

struct SmallArgs {
    const float *wpack;
    WeightIndex wi;
    const float *xin;      // [A][nx]
    const float *Q;        // [B]
    const int *moff;       // [B+1]
    const int *order;      // molecules of this launch (sorted by n, largest first)
    const int *row_off;    // [A+1]
    const int *pi, *pj, *psym;
    const float *pe, *pwi, *pwj;
    float *q_out;          // [A]
    float *h_out;          // optional [A][48] (GNN_layer output), may be null
    int *status;
    int N, T, nx, gcap, pcap, A;
    int run_gnn, run_epn;  // layer-level entry points run only one of the two stacks
    const float *h_in;     // optional [A][48] initial h (layer-level API), null -> zeros
    const float *q_in;     // optional [A] initial q (layer-level API), null -> Q/n
    const float *nm_in;    // optional [A] node mask (dense front-end), null -> 1 for every real atom
    SmallLds L;
};

// ---- per-atom projection: out[atom c][kappa-permuted 32] = W^T a_c (+ b)   (one 32-atom tile, K = 60)
__device__ __forceinline__ void small_proj(const float *__restrict__ wF, const float *__restrict__ cinit,
                                           const float *a_row, float *out_row, int lane) {
    float bv[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(a_row + 4 * q);
        bv[4 * q] = v[0]; bv[4 * q + 1] = v[1]; bv[4 * q + 2] = v[2]; bv[4 * q + 3] = v[3];
    }
    float wv[EPNN_KA];
#pragma unroll
    for (int s = 0; s < EPNN_KA; ++s) wv[s] = wF[s * 64 + lane];
    f32x16 acc;
    if (cinit) {
        float ci[16];
        epnn_ld16(cinit, ci);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = ci[r];
    } else {
        acc = epnn_splat16(0.f);
    }
#pragma unroll
    for (int s = 0; s < EPNN_KA; ++s) acc = epnn_mfma(wv[s], bv[s], acc);
    epnn_st16(out_row, acc);
}

// ---- G^T tile: acc[r] = G[pair c][kappa(hh,r)] = sum_ch We[ch][k] e[pair][ch]
__device__ __forceinline__ f32x16 small_gtile(const float *__restrict__ weF, const float *__restrict__ erow,
                                              bool valid, int lane) {
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

__global__ __launch_bounds__(256) void k_small_forward(SmallArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hh = lane >> 5;
    const SmallLds &L = A.L;
    float *a_eo = sm + L.a_eo, *Pl = sm + L.P, *Rl = sm + L.R, *Sw = sm + L.Sw, *zp = sm + L.zp;
    float *Gl = sm + L.G, *dl = sm + L.dl, *lwi = sm + L.pwi, *lwj = sm + L.pwj, *nm = sm + L.nm;
    int *lpij = reinterpret_cast<int *>(sm + L.pij);
    unsigned short *pm = reinterpret_cast<unsigned short *>(sm + L.pm);
    unsigned char *glut = reinterpret_cast<unsigned char *>(sm + L.glut);

    if (*A.status & EPNN_ST_PAIR_OVERFLOW) return;
    const int b = A.order[blockIdx.x];
    const int a0 = A.moff[b], n = A.moff[b + 1] - a0;
    const int npad = 4 * ((n + 4) / 4);                   // multiple of 4, >= n+1
    const int npq = npad >> 2;                            // quads per atom row
    const int p0 = A.row_off[a0], np = A.row_off[a0 + n] - p0;
    if (np > A.gcap) {
        if (tid == 0) atomicOr(A.status, EPNN_ST_SMALL_OVERFLOW);
        return;
    }
    const int nx = A.nx, fq = nx + EPNN_EDIM;             // feature index of q
    const int nrows = n * npad, ntile = (nrows + 31) >> 5, ngroups = nrows >> 2;
    const int ngt = (np + 31) >> 5;
    const float inv_npad = 1.0f / (float)npad;

    // ------------------------------------------------------------------ init
    for (int i = tid; i < 32 * EPNN_AST; i += 256) a_eo[i] = 0.f;
    for (int i = tid; i < 40 * EPNN_PST; i += 256) Rl[i] = 0.f;
    for (int i = tid; i < (32 * 36) / 2; i += 256) reinterpret_cast<unsigned *>(pm)[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (int i = tid; i < n * nx; i += 256) {
        const int at = i / nx, f = i - at * nx;
        a_eo[at * EPNN_AST + epnn_aeo(f)] = A.xin[(size_t)(a0 + at) * nx + f];
    }
    if (A.h_in)
        for (int i = tid; i < n * EPNN_EDIM; i += 256) {
            const int at = i / EPNN_EDIM, f = i - at * EPNN_EDIM;
            a_eo[at * EPNN_AST + epnn_aeo(nx + f)] = A.h_in[(size_t)(a0 + at) * EPNN_EDIM + f];
        }
    if (tid < 32) {
        nm[tid] = tid < n ? (A.nm_in ? A.nm_in[a0 + tid] : 1.f) : 0.f;
        if (tid < n)   // charge_gn.py:337-338: q0 = float32(Q) / n
            a_eo[tid * EPNN_AST + epnn_aeo(fq)] = A.q_in ? A.q_in[a0 + tid] : A.Q[b] / (float)n;
    }
    for (int p = tid; p < np; p += 256) {
        const int li = A.pi[p0 + p] - a0, lj = A.pj[p0 + p] - a0;
        lpij[p] = li | (lj << 8);
        lwi[p] = A.pwi[p0 + p];
        lwj[p] = A.pwj[p0 + p];
        pm[li * npad + lj] = (unsigned short)p;
        if (A.psym[p0 + p]) pm[lj * npad + li] = (unsigned short)p;
    }
    for (int g = tid; g < ngroups; g += 256) {
        const int i = g / npq;
        glut[g] = (unsigned char)(i | (((g + 1) % npq == 0) ? 0x80 : 0));
    }
    __syncthreads();

    const float *wp = A.wpack;
    const float Nf = (float)A.N;
    const float padw = (float)(A.N - npad);               // how many more padded-partner rows the reference sums

    // ================================================================== GNN steps (charge_gn.py:60-74)
    for (int t = 0; t < (A.run_gnn ? A.T : 0); ++t) {
        const PairMlpPack &M = A.wi.msg[t];
        const UpdPack &U = A.wi.upd[t];
        // ---- phase A: P, R (waves 0,1) and the G tiles (round robin), zero the partial sums
        for (int i = tid; i < 4 * 32 * EPNN_SST; i += 256) Sw[i] = 0.f;
        for (int task = wave; task < 2 + ngt; task += 4) {
            if (task == 0) {
                small_proj(wp + M.wiF, wp + M.b1p + hh * 16, a_eo + c * EPNN_AST + hh * 32,
                           Pl + c * EPNN_PST + hh * 16, lane);
            } else if (task == 1) {
                small_proj(wp + M.wjF, nullptr, a_eo + c * EPNN_AST + hh * 32, Rl + c * EPNN_PST + hh * 16, lane);
            } else {
                const int slot = (task - 2) * 32 + c;
                const bool valid = slot < np;
                f32x16 g = small_gtile(wp + M.weF, A.pe + (size_t)(p0 + (valid ? slot : 0)) * EPNN_EDIM + hh * 24,
                                       valid, lane);
                if (valid) epnn_st16(Gl + slot * EPNN_PST + hh * 16, g);
            }
        }
        __syncthreads();
        // ---- phase B: dense pair tiles
        {
            float w2[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) w2[s] = wp[M.w2F + s * 64 + lane];
            const f32x16 cb2 = epnn_splat16(wp[M.b2 + c]);
            float *Smine = Sw + wave * 32 * EPNN_SST;
            for (int tau = wave; tau < ntile; tau += 4) {
                const int p = tau * 32 + c;
                const int pc = p < nrows ? p : 0;
                const int i = (int)(((float)pc + 0.5f) * inv_npad);
                const int j = pc - i * npad;
                const unsigned slot = pm[pc];
                float z[16], rj[16];
                epnn_ld16(Pl + i * EPNN_PST + hh * 16, z);
                epnn_ld16(Rl + j * EPNN_PST + hh * 16, rj);
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] += rj[r];
                if (slot != 0xFFFFu) {
                    float g[16];
                    epnn_ld16(Gl + slot * EPNN_PST + hh * 16, g);
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] += g[r];
                }
                f32x16 acc = cb2;
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = epnn_mfma(fmaxf(z[s], 0.f), w2[s], acc);
                // rows = pairs kappa(hh,r), col = out feature c.  Quad q of this half-wave = rows 8q+4hh .. +3.
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int g = tau * 8 + 2 * q + hh;
                    const float last = fmaxf(acc[4 * q + 3], 0.f);
                    const float v = ((fmaxf(acc[4 * q], 0.f) + fmaxf(acc[4 * q + 1], 0.f)) + fmaxf(acc[4 * q + 2], 0.f)) + last;
                    const int info = g < ngroups ? glut[g] : 0x7F;
                    const int gi = info & 0x7F;
                    // two half-waves may target the same (i, out): serialise them in a fixed order
                    if (hh == 0 && gi != 0x7F) atomicAdd(Smine + gi * EPNN_SST + c, v);
                    if (hh == 1 && gi != 0x7F) atomicAdd(Smine + gi * EPNN_SST + c, v);
                    if ((info & 0x80) && gi != 0x7F) zp[gi * EPNN_SST + c] = last;
                }
            }
        }
        __syncthreads();
        // ---- phase C: update MLP (charge_gn.py:71-74) on wave 0, atoms on MFMA columns
        if (wave == 0) {
            const int u0 = (nx - hh + 1) >> 1;
            const float *arow = a_eo + c * EPNN_AST + hh * 32 + u0;
            float hv[24], sv[16];
#pragma unroll
            for (int s = 0; s < 24; ++s) hv[s] = arow[s];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int o = 2 * s + hh;
                const float s4 = ((Sw[(0 * 32 + c) * EPNN_SST + o] + Sw[(1 * 32 + c) * EPNN_SST + o]) +
                                  Sw[(2 * 32 + c) * EPNN_SST + o]) + Sw[(3 * 32 + c) * EPNN_SST + o];
                sv[s] = s4 + padw * zp[c * EPNN_SST + o];
            }
            float cb[16], b1[16];
            epnn_ld16(wp + U.cb3p + hh * 16, cb);
            epnn_ld16(wp + U.bu1p + hh * 16, b1);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = Nf * cb[r];
#pragma unroll
            for (int s = 0; s < 24; ++s) acc = epnn_mfma(wp[U.u1F + s * 64 + lane], hv[s], acc);
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = epnn_mfma(wp[U.u1F + (24 + s) * 64 + lane], sv[s], acc);
            const float nmc = nm[c];
            float u1[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) u1[r] = fmaxf(nmc * acc[r] + b1[r], 0.f);
            float b2v[16];
            epnn_ld16(wp + U.bu2p + hh * 16, b2v);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = b2v[r];
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = epnn_mfma(wp[U.u2F + s * 64 + lane], u1[s], acc);
            float u2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) u2[r] = fmaxf(acc[r], 0.f);
            f32x16 o0, o1;
            float b3a[16], b3b[16];
            epnn_ld16(wp + U.bu3p + hh * 16, b3a);
            epnn_ld16(wp + U.bu3p + 32 + hh * 16, b3b);
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] = b3a[r]; o1[r] = b3b[r]; }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                o0 = epnn_mfma(wp[U.u3F + s * 64 + lane], u2[s], o0);
                o1 = epnn_mfma(wp[U.u3F + (16 + s) * 64 + lane], u2[s], o1);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = epnn_kappa(hh, r);
                a_eo[c * EPNN_AST + epnn_aeo(nx + f)] = nmc * o0[r];
                if (r < 8) a_eo[c * EPNN_AST + epnn_aeo(nx + 32 + f)] = nmc * o1[r];
            }
        }
        __syncthreads();
    }
    if (A.h_out)
        for (int i = tid; i < n * EPNN_EDIM; i += 256) {
            const int at = i / EPNN_EDIM, f = i - at * EPNN_EDIM;
            A.h_out[(size_t)(a0 + at) * EPNN_EDIM + f] = a_eo[at * EPNN_AST + epnn_aeo(nx + f)];
        }

    // ================================================================== EPN steps (charge_gn.py:98-118)
    for (int t = 0; t < (A.run_epn ? A.T : 0); ++t) {
        const PairMlpPack &M = A.wi.pas[t];
        if (wave == 0)
            small_proj(wp + M.wiF, wp + M.b1p + hh * 16, a_eo + c * EPNN_AST + hh * 32, Pl + c * EPNN_PST + hh * 16, lane);
        else if (wave == 1)
            small_proj(wp + M.wjF, nullptr, a_eo + c * EPNN_AST + hh * 32, Rl + c * EPNN_PST + hh * 16, lane);
        __syncthreads();
        {
            float w2[16], b2v[16], w3[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) w2[s] = wp[M.w2F + s * 64 + lane];
            epnn_ld16(wp + M.b2p + hh * 16, b2v);
            epnn_ld16(wp + M.w3p + hh * 16, w3);
            for (int gt = 3 - wave; gt < ngt; gt += 4) {       // waves 3,2 first: waves 0,1 just did P,R
                const int slot = gt * 32 + c;
                const bool valid = slot < np;
                const f32x16 g = small_gtile(wp + M.weF, A.pe + (size_t)(p0 + (valid ? slot : 0)) * EPNN_EDIM + hh * 24,
                                             valid, lane);
                const int ij = valid ? lpij[slot] : 0;
                const int li = ij & 0xFF, lj = ij >> 8;
                float pi_[16], rj_[16], pj_[16], ri_[16];
                epnn_ld16(Pl + li * EPNN_PST + hh * 16, pi_);
                epnn_ld16(Rl + lj * EPNN_PST + hh * 16, rj_);
                epnn_ld16(Pl + lj * EPNN_PST + hh * 16, pj_);
                epnn_ld16(Rl + li * EPNN_PST + hh * 16, ri_);
                f32x16 au, av;
#pragma unroll
                for (int r = 0; r < 16; ++r) { au[r] = b2v[r]; av[r] = b2v[r]; }
                // rows = out feature kappa(hh,r), col = pair c
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
                if (hh == 0 && valid) dl[slot] = 0.5f * (fu - fv);     // charge_gn.py:116
            }
        }
        __syncthreads();
        // q_i += sum_j antisym_ij  (charge_gn.py:118): 8 lanes per atom, fixed combination order
        {
            const int i = tid >> 3, part = tid & 7;
            float acc = 0.f;
            for (int p = part; p < np; p += 8) {
                const int ij = lpij[p];
                const float d = dl[p];
                if ((ij & 0xFF) == i) acc += lwi[p] * d;
                if ((ij >> 8) == i) acc -= lwj[p] * d;
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 4, 64);
            if (part == 0 && i < n) a_eo[i * EPNN_AST + epnn_aeo(fq)] += acc;
        }
        __syncthreads();
    }
    if (tid < n) A.q_out[a0 + tid] = a_eo[tid * EPNN_AST + epnn_aeo(fq)];
}
