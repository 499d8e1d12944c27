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
// Each wave owns a CONTIGUOUS range of tiles, so an atom's rows are touched by at most two (adjacent) waves:
// two partial-sum copies (wave parity) are enough and every copy has a single writer per atom -> fixed sum order.
//
// EPN (charge_gn.py:98-118): one tile row per UNORDERED near pair; both directions share G; the transfer
// 0.5*(f_ij - f_ji) is applied as +d to i and -d to j, so total charge is conserved by construction.
//
// Weights are never waited for inside a phase: every wave fetches the MFMA fragments of its NEXT task into
// registers while it works on the current one (global -> VGPR loads are asynchronous until first use).
//
// Phase plan (barrier between phases; [w] = wave):
//   GNN step t:  A   [0] P  [1] R  [2] h-part of the first update Dense  (+ G tiles of msg[0] before step 0)
//                B   dense pair tiles, contiguous ranges per wave
//                C1  [0] S-part of the first update Dense   [1..3] G tiles of msg[t+1]
//                C2  [1] second update Dense      C3  [2] h features 0..31  [3] h features 32..47
//   EPN step t:  A'  [0] P  [1] R  [2,3] G tiles of pas[t]
//                B'  near-pair tiles, one per wave round robin
//                C'  ordered per-atom charge update
#pragma once
#include "epnn_common.h"

struct SmallLds {   // offsets in 4-byte words into dynamic LDS
    int a_eo, P, R, Sw, zp, G, dl, pij, pwi, pwj, pm, glut, nm, u1h, cst, total;
    int nr, npadmax;
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
    float *gx;             // [pcap][32] overflow rows of G for pairs beyond the LDS slots of the launch
    int glds;              // near-pair slots of G kept in LDS (<= gcap)
    unsigned long long *stamps;   // diagnostic build only (-DEPNN_STAMPS): [block][wave][64] s_memtime values
};

// Diagnostic build: -DEPNN_STAMPS=1 stamps the GNN phases, =2 the EPN phases (tools/dev_stamps.py).
#ifdef EPNN_STAMPS
#define EPNN_STAMP()                                                                                   \
    do {                                                                                               \
        if (lane == 0 && A.stamps && nstamp < 62) {                                                    \
            A.stamps[((size_t)blockIdx.x * 4 + wave) * 64 + nstamp] = __builtin_amdgcn_s_memtime();    \
            ++nstamp;                                                                                  \
        }                                                                                              \
    } while (0)
#else
#define EPNN_STAMP() do { } while (0)
#endif
#if defined(EPNN_STAMPS) && EPNN_STAMPS == 2
#define EPNN_STAMPE() EPNN_STAMP()
#define EPNN_STAMPG() do { } while (0)
#else
#define EPNN_STAMPE() do { } while (0)
#define EPNN_STAMPG() EPNN_STAMP()
#endif
#ifndef EPNN_SMALL_WAVES
#define EPNN_SMALL_WAVES 3   // waves per SIMD the register allocation of the split halves is sized for (both stacks in one launch: 2)
#endif
#ifndef EPNN_ABL
#define EPNN_ABL 0     // diagnostic ablations of the pair-tile loop (1: no partial-sum update, 2: no G lookup)
#endif

// fragment loads: cnt consecutive 256-byte rows of the packed weight buffer -> one register each
#define EPNN_LDW(dst, off, cnt)                                     \
    _Pragma("unroll") for (int s_ = 0; s_ < (cnt); ++s_)(dst)[s_] = wp[(off) + s_ * 64 + lane]

// ---- per-atom projection: out[atom c][kappa-permuted 32] = W^T a_c   (one 32-atom tile, K = 60; weights in w[])
__device__ __forceinline__ void small_proj(const float (&w)[32], const float *a_row, float *out_row, bool store) {
    float bv[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(a_row + 4 * q);
        bv[4 * q] = v[0]; bv[4 * q + 1] = v[1]; bv[4 * q + 2] = v[2]; bv[4 * q + 3] = v[3];
    }
    f32x16 acc = epnn_splat16(0.f);
#pragma unroll
    for (int s = 0; s < EPNN_KA; ++s) acc = epnn_mfma(w[s], bv[s], acc);
    if (store) epnn_st16(out_row, acc);
}

// ---- G^T tile: acc[r] = G[pair c][kappa(hh,r)] = sum_ch We[ch][k] e[pair][ch]   (weights read from global)
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

// G tiles first, first+stride, ... with the We fragments already in registers -> LDS
// (slots >= glds do not fit the LDS budget of the launch and go to the per-pair overflow rows in HBM; a later
// __syncthreads() makes them visible to the other waves of the workgroup)
__device__ __forceinline__ void small_gtiles_reg(const float (&w)[32], const float *pe, int p0, int np, float *Gl,
                                                 int glds, float *Gx, int first, int stride, int lane) {
    const int c = lane & 31, hh = lane >> 5;
    const int ngt = (np + 31) >> 5;
    for (int gt = first; gt < ngt; gt += stride) {
        const int slot = gt * 32 + c;
        const bool valid = slot < np;
        const float *erow = pe + (size_t)(p0 + (valid ? slot : 0)) * EPNN_EDIM + hh * 24;
        float ev[24];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(erow + 4 * q);
            ev[4 * q] = v[0]; ev[4 * q + 1] = v[1]; ev[4 * q + 2] = v[2]; ev[4 * q + 3] = v[3];
        }
        f32x16 acc = epnn_splat16(0.f);
#pragma unroll
        for (int s = 0; s < 24; ++s) acc = epnn_mfma(w[s], valid ? ev[s] : 0.f, acc);
        if (valid) {
            if (slot < glds) epnn_st16(Gl + slot * EPNN_PST + hh * 16, acc);
            else epnn_st16(Gx + (size_t)(p0 + slot) * 32 + hh * 16, acc);
        }
    }
}

// GNN / EPN select which stack is compiled in: <true,true> runs both in one launch, the split pair <true,false> +
// <false,true> hands h over through HBM and gives each half its own (smaller) register allocation.
template <bool GNN, bool EPN>
__global__ __launch_bounds__(256, (GNN && EPN) ? 2 : EPNN_SMALL_WAVES) void k_small_forward(SmallArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hh = lane >> 5;
    const SmallLds &L = A.L;
    float *a_eo = sm + L.a_eo, *Pl = sm + L.P, *Rl = sm + L.R, *Sw = sm + L.Sw, *zp = sm + L.zp;
    float *Gl = sm + L.G, *nm = sm + L.nm, *ubuf = sm + L.u1h, *cst = sm + L.cst;
    f32x4 *prec = reinterpret_cast<f32x4 *>(sm + L.dl);   // per near pair: {li | lj<<8 (bits), w_i, w_j, delta}
    unsigned short *pm = reinterpret_cast<unsigned short *>(sm + L.pm);
    unsigned char *glut = reinterpret_cast<unsigned char *>(sm + L.glut);

    if (*A.status & EPNN_ST_PAIR_OVERFLOW) return;
    int nstamp = 0;
    (void)nstamp;
    EPNN_STAMP();
    const int b = A.order[blockIdx.x];
    const int a0 = A.moff[b], n = A.moff[b + 1] - a0;
    const int npad = 4 * ((n + 4) / 4);                   // multiple of 4, >= n+1
    const int npq = npad >> 2;                            // quads per atom row
    const int p0 = A.row_off[a0], np = A.row_off[a0 + n] - p0;
    if (np > A.gcap) {
        if (tid == 0) atomicOr(A.status, EPNN_ST_SMALL_OVERFLOW);
        return;
    }
    const int nr = L.nr;                                  // LDS rows of the per-atom arrays (largest n of the launch)
    const int nx = A.nx, fq = nx + EPNN_EDIM;             // feature index of q
    const int nrows = n * npad, ntile = (nrows + 31) >> 5, ngroups = nrows >> 2;
    const int ngt = (np + 31) >> 5;
    const int glds = A.glds;
    const float inv_npad = 1.0f / (float)npad;
    const int cr = c < nr ? c : nr - 1;                   // lanes beyond the last LDS row re-read it (results dropped)
    const bool catom = c < n;
    const float *wp = A.wpack;
    const int Tg = (GNN && A.run_gnn) ? A.T : 0, Te = (EPN && A.run_epn) ? A.T : 0;

    // registers that carry weights from the phase in which they are fetched to the phase that uses them
    float pw[32];      // phase A / A' task (Wi | Wj | update h-part | We) and the G tiles of phase C1
    float pb[16];      // W2 fragments of the pair tiles
    float pc[16];      // update-MLP chain: [0] U1 S-part (per step)  [1] U2  [2] U3 tile 0  [3] U3 tile 1
    float pcb[16];     // matching bias vector
    float pb2 = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) pw[s] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) { pb[s] = 0.f; pc[s] = 0.f; pcb[s] = 0.f; }

    // first G tiles need We of the first pair MLP: fetch it before the LDS init so the latency overlaps
    {
        const int weF0 = Tg > 0 ? A.wi.msg[0].weF : A.wi.pas[0].weF;
        if (Tg > 0 || wave >= 2) { EPNN_LDW(pw, weF0, 24); }
    }

    // ------------------------------------------------------------------ init
    for (int i = tid; i < nr * EPNN_AST; i += 256) a_eo[i] = 0.f;
    for (int i = tid; i < L.npadmax * EPNN_PST; i += 256) Rl[i] = 0.f;
    if (GNN)
        for (int i = tid; i < (nr * L.npadmax + 1) / 2; i += 256) reinterpret_cast<unsigned *>(pm)[i] = 0xFFFFFFFFu;
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
        if (tid < n) {  // charge_gn.py:337-338: q0 = float32(Q) / n
            a_eo[tid * EPNN_AST + epnn_aeo(fq)] = A.q_in ? A.q_in[a0 + tid] : A.Q[b] / (float)n;
            a_eo[tid * EPNN_AST + epnn_aeo(EPNN_F1)] = 1.f;      // carries b1 (Wi row 59)
        }
    }
    for (int p = tid; p < np; p += 256) {
        const int li = A.pi[p0 + p] - a0, lj = A.pj[p0 + p] - a0;
        if (EPN) {          // pair records are an EPN structure (the GNN half of a split launch has no room for them)
            f32x4 rec;
            rec[0] = __int_as_float(li | (lj << 8));
            rec[1] = A.pwi[p0 + p];
            rec[2] = A.pwj[p0 + p];
            rec[3] = 0.f;
            prec[p] = rec;
        }
        if (GNN) {          // pair map of the dense tiles
            pm[li * npad + lj] = (unsigned short)p;
            if (A.psym[p0 + p]) pm[lj * npad + li] = (unsigned short)p;
        }
    }
    if (GNN)
        for (int g = tid; g < ngroups; g += 256) {
            const int i = g / npq;
            glut[g] = (unsigned char)(i | (((g + 1) % npq == 0) ? 0x80 : 0));
        }
    __syncthreads();

    const float Nf = (float)A.N;
    const float padw = (float)(A.N - npad);               // how many more padded-partner rows the reference sums
    // contiguous tile range of this wave
    const int tlo = (ntile * wave) >> 2, thi = (ntile * (wave + 1)) >> 2;
    EPNN_STAMP();   // 1: init done

    if (Tg > 0) {
        // G tiles of msg[0] (all waves), then the phase-A weights of step 0
        small_gtiles_reg(pw, A.pe, p0, np, Gl, glds, A.gx, 3 - wave, 4, lane);
        const PairMlpPack &M0 = A.wi.msg[0];
        if (wave == 0) { EPNN_LDW(pw, M0.wiF, EPNN_KA); }
        else if (wave == 1) { EPNN_LDW(pw, M0.wjF, EPNN_KA); }
        else if (wave == 2) { EPNN_LDW(pw, A.wi.upd[0].u1F, 24); }
    } else if (Te > 0) {
        const PairMlpPack &M0 = A.wi.pas[0];
        if (wave == 0) { EPNN_LDW(pw, M0.wiF, EPNN_KA); }
        else if (wave == 1) { EPNN_LDW(pw, M0.wjF, EPNN_KA); }
    }

    // ================================================================== GNN steps (charge_gn.py:60-74)
    for (int t = 0; t < Tg; ++t) {
        const PairMlpPack &M = A.wi.msg[t];
        const UpdPack &U = A.wi.upd[t];
        const bool lastg = t + 1 == Tg;
        // ---- phase A: [0] P  [1] R  [2] h-part of the first update Dense.  W2/b2 of the tiles are fetched now.
        EPNN_LDW(pb, M.w2F, 16);
        pb2 = wp[M.b2 + c];
        for (int i = tid; i < 2 * nr * EPNN_SST; i += 256) Sw[i] = 0.f;
        if (wave == 0) {
            small_proj(pw, a_eo + cr * EPNN_AST + hh * 32, Pl + cr * EPNN_PST + hh * 16, catom);
        } else if (wave == 1) {
            small_proj(pw, a_eo + cr * EPNN_AST + hh * 32, Rl + cr * EPNN_PST + hh * 16, catom);
        } else if (wave == 2) {
            const int u0 = (nx - hh + 1) >> 1;
            const float *arow = a_eo + cr * EPNN_AST + hh * 32 + u0;
            float hv[24], pinit[16];
            epnn_ld16(wp + U.cb3p + hh * 16, pinit);
#pragma unroll
            for (int s = 0; s < 24; ++s) hv[s] = arow[s];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = Nf * pinit[r];
#pragma unroll
            for (int s = 0; s < 24; ++s) acc = epnn_mfma(pw[s], hv[s], acc);
            epnn_st16(ubuf + lane * 16, acc);
        }
        EPNN_STAMPG();   // end of own phase-A work
        __syncthreads();
        EPNN_STAMPG();   // phase B starts
        // ---- phase B: dense pair tiles
        {
            const f32x16 cb2 = epnn_splat16(pb2);
            float *Smine = Sw + (wave & 1) * nr * EPNN_SST;
            for (int tau = tlo; tau < thi; ++tau) {
                const int p = tau * 32 + c;
                const int pc_ = p < nrows ? p : 0;
                const int i = (int)(((float)pc_ + 0.5f) * inv_npad);
                const int j = pc_ - i * npad;
                const unsigned slot = pm[pc_];
                int info[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int g = tau * 8 + 2 * q + hh;
                    info[q] = g < ngroups ? glut[g] : 0x7F;
                }
                float z[16], rj[16];
                epnn_ld16(Pl + i * EPNN_PST + hh * 16, z);
                epnn_ld16(Rl + j * EPNN_PST + hh * 16, rj);
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] += rj[r];
#if EPNN_ABL != 2
                if (slot != 0xFFFFu) {
                    float g[16];
                    if ((int)slot < glds) epnn_ld16(Gl + slot * EPNN_PST + hh * 16, g);
                    else epnn_ld16(A.gx + (size_t)(p0 + slot) * 32 + hh * 16, g);
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] += g[r];
                }
#endif
                f32x16 acc = cb2;
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = epnn_mfma(fmaxf(z[s], 0.f), pb[s], acc);
                // rows = pairs kappa(hh,r), col = out feature c.  Quad q of this half-wave = rows 8q+4hh .. +3.
                // Quad sums -> this wave's partial sums.  LDS float atomics cost ~400 cycles per wave-instruction on
                // gfx950, so this is a plain read-modify-write: consecutive quads of one half-wave that belong to
                // the same atom are merged in registers first (the surviving quads then have distinct addresses),
                // and the two half-waves, which may share an atom, update one after the other.
                float v[4];
                int gi[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float last = fmaxf(acc[4 * q + 3], 0.f);
                    v[q] = ((fmaxf(acc[4 * q], 0.f) + fmaxf(acc[4 * q + 1], 0.f)) + fmaxf(acc[4 * q + 2], 0.f)) + last;
                    gi[q] = info[q] & 0x7F;
                    if ((info[q] & 0x80) && gi[q] != 0x7F) zp[gi[q] * EPNN_SST + c] = last;
                }
#if EPNN_ABL == 1
                asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
#else
                bool wr[4];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const bool same = gi[q] == gi[q + 1];
                    if (same) v[q + 1] += v[q];
                    wr[q] = !same && gi[q] != 0x7F;
                }
                wr[3] = gi[3] != 0x7F;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if (hh == half) {
                        float old[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) old[q] = wr[q] ? Smine[gi[q] * EPNN_SST + c] : 0.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (wr[q]) Smine[gi[q] * EPNN_SST + c] = old[q] + v[q];
                    }
                    // the upper half-wave reads what the lower one just wrote: a cross-lane dependence the compiler
                    // cannot see (for one thread the two branches are exclusive), so pin the order explicitly
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
#endif
            }
        }
        // fetch what phase C1 needs while the other waves finish their tiles
        if (wave == 0) {
            EPNN_LDW(pc, U.u1F + 24 * 64, 16);
            epnn_ld16(wp + U.bu1p + hh * 16, pcb);
        } else {                                                    // We of the next pair MLP for phase C1 / A'
            if (!lastg) { EPNN_LDW(pw, A.wi.msg[t + 1].weF, 24); }
            else if (Te > 0 && wave >= 2) { EPNN_LDW(pw, A.wi.pas[0].weF, 24); }
            // this wave's layer of the update chain (used in C2 / C3, i.e. after C1's matrix work)
            if (wave == 1) { EPNN_LDW(pc, U.u2F, 16); epnn_ld16(wp + U.bu2p + hh * 16, pcb); }
            else if (wave == 2) { EPNN_LDW(pc, U.u3F, 16); epnn_ld16(wp + U.bu3p + hh * 16, pcb); }
            else { EPNN_LDW(pc, U.u3F + 16 * 64, 16); epnn_ld16(wp + U.bu3p + 32 + hh * 16, pcb); }
        }
        EPNN_STAMPG();   // end of own phase-B work
        __syncthreads();
        EPNN_STAMPG();   // phase C starts
        // ---- phase C1: [0] first update Dense (S-part on top of the stashed h-part)   [1..3] next G tiles
        if (wave == 0) {
            float sv[16], a1[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int o = 2 * s + hh;
                sv[s] = (Sw[cr * EPNN_SST + o] + Sw[(nr + cr) * EPNN_SST + o]) + padw * zp[cr * EPNN_SST + o];
            }
            epnn_ld16(ubuf + lane * 16, a1);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = a1[r];
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = epnn_mfma(pc[s], sv[s], acc);
            const float nmc = nm[c];
            f32x16 u1;
#pragma unroll
            for (int r = 0; r < 16; ++r) u1[r] = fmaxf(nmc * acc[r] + pcb[r], 0.f);
            epnn_st16(ubuf + lane * 16, u1);
            // phase-A weights of the next step
            if (!lastg) { EPNN_LDW(pw, A.wi.msg[t + 1].wiF, EPNN_KA); }
            else if (Te > 0) { EPNN_LDW(pw, A.wi.pas[0].wiF, EPNN_KA); }
        } else {
            if (!lastg) small_gtiles_reg(pw, A.pe, p0, np, Gl, glds, A.gx, wave - 1, 3, lane);
            if (wave == 1) {
                if (!lastg) { EPNN_LDW(pw, A.wi.msg[t + 1].wjF, EPNN_KA); }
                else if (Te > 0) { EPNN_LDW(pw, A.wi.pas[0].wjF, EPNN_KA); }
            } else if (wave == 2 && !lastg) {
                EPNN_LDW(pw, A.wi.upd[t + 1].u1F, 24);
            }
        }
        EPNN_STAMPG();
        __syncthreads();
        EPNN_STAMPG();
        // ---- phase C2: [1] second update Dense
        if (wave == 1) {
            float u1[16];
            epnn_ld16(ubuf + lane * 16, u1);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = pcb[r];
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = epnn_mfma(pc[s], u1[s], acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
            epnn_st16(ubuf + lane * 16, acc);
        }
        EPNN_STAMPG();
        __syncthreads();
        EPNN_STAMPG();
        // ---- phase C3: [2] h features 0..31   [3] h features 32..47   (charge_gn.py:73-74)
        if (wave >= 2) {
            float u2[16];
            epnn_ld16(ubuf + lane * 16, u2);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = pcb[r];
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = epnn_mfma(pc[s], u2[s], acc);
            if (catom) {
                const float nmc = nm[c];
                const int fb = wave == 2 ? 0 : 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = fb + epnn_kappa(hh, r);
                    if (f < EPNN_EDIM) a_eo[c * EPNN_AST + epnn_aeo(nx + f)] = nmc * acc[r];
                }
            }
        }
        EPNN_STAMPG();   // end of own phase-C work
        __syncthreads();
        EPNN_STAMPG();   // next step starts
    }
    if (A.h_out)
        for (int i = tid; i < n * EPNN_EDIM; i += 256) {
            const int at = i / EPNN_EDIM, f = i - at * EPNN_EDIM;
            A.h_out[(size_t)(a0 + at) * EPNN_EDIM + f] = a_eo[at * EPNN_AST + epnn_aeo(nx + f)];
        }

    // ================================================================== EPN steps (charge_gn.py:98-118)
    for (int t = 0; t < Te; ++t) {
        const PairMlpPack &M = A.wi.pas[t];
        const bool laste = t + 1 == Te;
        // ---- phase A': [0] P  [1] R  [2,3] G tiles;  W2 for the pair tiles and the step's small vectors fetched now
        EPNN_LDW(pb, M.w2F, 16);
        float cv = 0.f;
        if (wave == 0) cv = hh == 0 ? wp[M.b2p + c] : wp[M.w3p + c];
        if (wave == 0) {
            small_proj(pw, a_eo + cr * EPNN_AST + hh * 32, Pl + cr * EPNN_PST + hh * 16, catom);
            cst[lane] = cv;                                          // [0..31] b2 (kappa order)  [32..63] w3
        } else if (wave == 1) {
            small_proj(pw, a_eo + cr * EPNN_AST + hh * 32, Rl + cr * EPNN_PST + hh * 16, catom);
        } else {
            small_gtiles_reg(pw, A.pe, p0, np, Gl, glds, A.gx, wave - 2, 2, lane);
        }
        EPNN_STAMPE();
        __syncthreads();
        EPNN_STAMPE();
        for (int gt = wave; gt < ngt; gt += 4) {
            const int slot = gt * 32 + c;
            const bool valid = slot < np;
            const int sl = valid ? slot : 0;
            const int ij = __float_as_int(reinterpret_cast<const float *>(prec + sl)[0]);
            const int li = ij & 0xFF, lj = ij >> 8;
            // the two directions one after the other: a dependent chain of this MFMA already runs at the issue
            // rate, and keeping only one direction live saves 48 registers
            float g[16], b2v[16], w3[16];
            if (sl < glds) epnn_ld16(Gl + sl * EPNN_PST + hh * 16, g);
            else epnn_ld16(A.gx + (size_t)(p0 + sl) * 32 + hh * 16, g);
            epnn_ld16(cst + hh * 16, b2v);
            epnn_ld16(cst + 32 + hh * 16, w3);
            float fu = 0.f, fv = 0.f;
#pragma unroll
            for (int dir = 0; dir < 2; ++dir) {
                const int ai = dir == 0 ? li : lj, aj = dir == 0 ? lj : li;
                float ta[16], tb[16];
                epnn_ld16(Pl + ai * EPNN_PST + hh * 16, ta);
                epnn_ld16(Rl + aj * EPNN_PST + hh * 16, tb);
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = b2v[r];
                // rows = out feature kappa(hh,r), col = pair c
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) acc = epnn_mfma(pb[s2], fmaxf((g[s2] + ta[s2]) + tb[s2], 0.f), acc);
                float f = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) f = fmaf(w3[r], fmaxf(acc[r], 0.f), f);
                if (dir == 0) fu = f; else fv = f;
            }
            fu += epnn_swap32(fu);
            fv += epnn_swap32(fv);
            if (hh == 0 && valid) reinterpret_cast<float *>(prec + slot)[3] = 0.5f * (fu - fv);     // charge_gn.py:116
        }
        if (!laste) {                                                // weights of the next step's phase A'
            if (wave == 0) { EPNN_LDW(pw, A.wi.pas[t + 1].wiF, EPNN_KA); }
            else if (wave == 1) { EPNN_LDW(pw, A.wi.pas[t + 1].wjF, EPNN_KA); }
            else { EPNN_LDW(pw, A.wi.pas[t + 1].weF, 24); }
        }
        EPNN_STAMPE();
        __syncthreads();
        EPNN_STAMPE();
        // q_i += sum_j antisym_ij  (charge_gn.py:118): 8 lanes per atom, fixed combination order
        {
            const int i = tid >> 3, part = tid & 7;
            float acc = 0.f;
#pragma unroll 4
            for (int p = part; p < np; p += 8) {
                const f32x4 rec = prec[p];
                const int ij = __float_as_int(rec[0]);
                if ((ij & 0xFF) == i) acc += rec[1] * rec[3];
                if ((ij >> 8) == i) acc -= rec[2] * rec[3];
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 4, 64);
            if (part == 0 && i < n) a_eo[i * EPNN_AST + epnn_aeo(fq)] += acc;
        }
        EPNN_STAMPE();
        __syncthreads();
        EPNN_STAMPE();
    }
    EPNN_STAMP();
    if (tid < n) A.q_out[a0 + tid] = a_eo[tid * EPNN_AST + epnn_aeo(fq)];
#ifdef EPNN_STAMPS
    if (lane == 0 && A.stamps) {
        A.stamps[((size_t)blockIdx.x * 4 + wave) * 64 + 62] = (unsigned long long)nstamp;
        A.stamps[((size_t)blockIdx.x * 4 + wave) * 64 + 63] = ((unsigned long long)n << 32) | (unsigned)np;
    }
#endif
}
