// One column block per wavefront: the fused forward of epnn_wave.hip.h (both stacks, with the in-kernel front-end of the
// compact entry or the pair lists of the dense one) as a workgroup of NW = 2 or 3 wavefronts which either
//   * SPLIT a molecule of 16 (NW - 1) + 1 .. 16 NW atoms: column block b (atoms 16 b .. 16 b + 15) on wavefront b, the last
//     block (the remaining atoms, in partner copies) on the last wavefront, or
//   * run one molecule of at most 16 atoms each, side by side, without ever meeting.
// Everything per atom is per column in k_wave_forward, so each wavefront simply runs the one-block code for its own atoms
// (215 registers).  In a split what the columns of one block read from the others -- the R_j rows, the G rows, the pair
// map, the EPN's P rows and transfer matrix -- is in the workgroup's LDS, and a workgroup barrier stands wherever the
// one-wavefront kernel relies on program order between a table's writers and readers (two per step); the G tiles, the EPN's
// pair blocks and the per-pair part of the front-end are dealt out in turn, the slot assignment of the front-end is done by
// all (same values to the same words), and the last wavefront, whose block has few partner tiles, also runs the last tiles of
// the full blocks.  Same operands as k_wave_forward: an unsplit molecule runs the very same instruction sequence
// (bit-identical), a split one agrees to float32 rounding (tests/test_gpu_parity.py::test_fused_kernel_every_size_vs_oracle,
// test_block_per_wave_kernel_batch_shapes, test_three_block_kernel_every_size_vs_oracle; tests/test_gpu_api.py::
// test_make_model_every_size_class_on_the_block_per_wave_kernel).
//
// Why: a launch lasts as long as its largest molecule, and a 25..32-atom molecule spends half its time in each block
// (186 us alone on a SIMD at 29 atoms): that is the latency of every blocking call and the tail of every pipelined run.  And
// three blocks on one wavefront need ~280 registers (one wavefront per SIMD), on three wavefronts 215.
#pragma once
#include "epnn_wave.hip.h"

// floats at the end of the workgroup's LDS (a split's): scratch of the last block's copies' reduction [512] | the last
// wavefront's share of the full blocks' message sums [NW - 1][512] | P rows of the full blocks [16 (NW - 1)][PST]
#define EPNN_W2_SCR(NW) (512 + ((NW) - 1) * (512 + 16 * EPNN_PST))
#define EPNN_W2_SINGLE 0      // wblk mode: the wavefront has a molecule (n <= 16) to itself
#define EPNN_W2_SPLIT 1       // the workgroup's wavefronts share a molecule (17 <= n <= 32 on two, 33 <= n <= 48 on three, 49 <= n <= 64 on four)
#define EPNN_W2_IDLE 2
#define EPNN_W2_NMAX3 48       // largest molecule of the three-wavefront form
#define EPNN_W2_NMAX4 64       // ... of the four-wavefront form (one lane per partner in the front-end: 64 is the end of this design)
#define EPNN_W2_AUTO_MAX 1024  // option "wave2" = -1: batches of at most this many molecules take this kernel

// both wavefronts: what either wrote (LDS or global memory) before the barrier is read by the other after it
__device__ __forceinline__ void wg2_sync() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

// FRONT: compact entry, the pair lists are built here from coordinates (edge features in the 16-dimensional basis, K = 16);
// !FRONT: the literal make_model entry, pair lists and 48-channel e rows from the dense front-end (epnn_dense.hip.h), h, q and
// a possibly fractional node mask given per atom.  Both stacks in either case.
template <int NW, bool FRONT>
__global__ __launch_bounds__(64 * NW, EPNN_WAVES_PER_SIMD) void k_wave_forward2(WaveArgs A, WaveIndex X) {
    constexpr int PMS = NW == 2 ? 32 : 16 * NW;            // row stride of the pair map (u16 entries): one per atom of the molecule
    constexpr int DSTW = NW == 2 ? EPNN_DST : 16 * NW + 1;  // row stride of the transfer matrix
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, n16 = lane & 15;
    const int c = lane & 31, hh = lane >> 5;               // lane naming of the front-end (row pairs x 32 partners)
    if (!FRONT && (*A.status & EPNN_ST_PAIR_OVERFLOW)) return;      // (every wavefront sees the same word)
    const int4 wb = A.wblk[NW * blockIdx.x + w];            // one entry per wavefront: molecule, first atom, atoms | mode << 8, first pair slot
    const int b = wb.x, a0 = wb.y, n = wb.z & 0xFF, mode = wb.z >> 8;
    if (mode == EPNN_W2_IDLE) return;                       // odd number of unsplit molecules: the last workgroup's second wavefront
    const bool split = mode == EPNN_W2_SPLIT;               // the same for both wavefronts of a workgroup (host)
    if (A.prio_n > 0 && n >= A.prio_n) __builtin_amdgcn_s_setprio(3);
    const int p0 = FRONT ? wb.w : A.row_off[a0];
    int np = FRONT ? 0 : A.row_off[a0 + n] - p0;
    const int nx = A.nx;
    const bool xs3 = nx + 3 <= 4 * (EPNN_XS - 1);
    const float *wp = A.wpack;
    // this wavefront's column block
    const bool blk1 = split && w == NW - 1;                 // the last block: the atoms beyond the full blocks, m1 of them, C1 copies each
    const int m1 = blk1 ? n - 16 * (NW - 1) : 16, C1 = 16 / m1;
    const int Cw = blk1 ? C1 : 1;
    const int copy = blk1 ? n16 / m1 : 0;
    const int col = split ? 16 * w + (blk1 ? n16 % m1 : n16) : n16;     // the column's atom
    const bool cat = split ? (blk1 ? copy < Cw : true) : n16 < n;
    const bool own = cat && copy == 0;                      // the copy that stores the atom's rows / results
    // The sweep of a split is balanced: a full block has n + 1 partner tiles, the last block (n + C1) / C1 -- three for 18
    // atoms --, so the last wavefront also takes the tiles J0 .. n of every full block (it reads their P rows from LDS and
    // hands its share of the message sums back through LDS); all wavefronts then run about the same number of tiles.
    const int ms = split ? n - 16 * (NW - 1) : 16, nt0 = n + 1, nt1s = (n + 16 / ms) / (16 / ms);
    const int nxt = split ? max(0, ((NW - 1) * nt0 + nt1s) / NW - nt1s) / (NW - 1) : 0;    // tiles of every full block done by the last wavefront
    const int J0 = nt0 - nxt;
    // what the wavefronts of a split deal out in turn, an unsplit wavefront does alone
    const int dstep = split ? NW : 1, doff = split ? w : 0;
    const int tid = split ? (int)threadIdx.x : lane, nthr = split ? 64 * NW : 64;
    auto sync = [&]() {                                     // order the molecule's LDS / global traffic among all its lanes
        if (split) wg2_sync();
        else wave_sync_all();
    };

    // ---- LDS layout: the one of k_wave_forward inside the workgroup's budget, the copies' scratch behind it
    const int lds_all = A.lds_words - EPNN_W2_SCR(NW);
    const int lds_words = split ? lds_all : (A.lds_words / NW) & ~3;    // (the scratch is a split's)
    float *scr = sm + lds_all, *scrx = scr + 512, *P0t = scr + 512 + (NW - 1) * 512;
    float *smw = split ? sm : sm + w * lds_words;           // an unsplit wavefront has its half of the workgroup's LDS
    unsigned short *eij = reinterpret_cast<unsigned short *>(smw);
    const int eij_n = FRONT ? n * (n - 1) / 2 : np;        // in-kernel front-end: np is not known yet, every i<j pair has a place
    const int o_r = (((eij_n + 1) >> 1) + 3) & ~3;
    float *Rl = smw + o_r;
    const int o_x = o_r + n * EPNN_PST;
    unsigned short *pm = reinterpret_cast<unsigned short *>(smw + o_x);
    float *Pl = smw + o_x;
    float *Dm = smw + o_x + n * EPNN_PST;
    const int o_gg = o_x + ((n * (PMS / 2) + 3) & ~3);
    const int grows_g = (lds_words - o_gg) / EPNN_PST - 1;
    float *Gl = smw + o_gg;
    int glds = min(np, grows_g);
    bool gover = np > glds;
    int ngt = (np + 31) >> 5;

    // ---- front-end: coordinates -> LDS (both wavefronts, same values)
    double *xs = reinterpret_cast<double *>(Rl);
    if (FRONT && lane < n) {
        xs[3 * lane + 0] = (double)A.xyz[3 * (size_t)(a0 + lane) + 0];
        xs[3 * lane + 1] = (double)A.xyz[3 * (size_t)(a0 + lane) + 1];
        xs[3 * lane + 2] = (double)A.xyz[3 * (size_t)(a0 + lane) + 2];
    }
    // ---- per-column registers: every lane loads from a clamped row and selects afterwards (a load under a condition is a branch
    //      with its own wait: see k_wave_forward)
    const int ia = a0 + min(col, n - 1);
    float nmv = 1.f, qa;
    if (A.nm_in) nmv = A.nm_in[ia];
    if (A.q_in) qa = A.q_in[ia];
    else qa = A.Q[b] / (float)n;                                                          // charge_gn.py:337-338
    float xv[EPNN_XS];
#pragma unroll
    for (int s = 0; s < EPNN_XS; ++s) xv[s] = A.xin[(size_t)ia * nx + min(max(4 * s + q - 1, 0), nx - 1)];
    const float nm = cat ? nmv : 0.f;
    float xq[EPNN_XS];
    {
        const float qv = cat ? qa : 0.f;
#pragma unroll
        for (int s = 0; s < EPNN_XS; ++s) {
            const int phi = 4 * s + q;
            float v = 0.f;
            if (phi == 0) v = nm;
            else if (phi <= nx) v = cat ? xv[s] : 0.f;
            else if (phi == nx + 1) v = qv;
            else if (phi == nx + 2) v = cat ? 1.f : 0.f;
            xq[s] = v;
        }
    }
    // the per-atom chains on the bf16 matrix pipe, exactly as in k_wave_forward (its CHB: same operands, same order of the products)
#ifdef EPNN_CHAIN_F32
    constexpr bool CHB = false;
#else
    constexpr bool CHB = true;
#endif
    u32x4 Bp[3], xqb[3];
    float qc = cat ? qa : 0.f;                              // the column's charge, the same bits in every lane
    auto xq_charge = [&]() {                                // (k_wave_forward: the charge's pieces into the low half of dword 1)
        if (q != 0) return;
        const float a_ = __uint_as_float(__float_as_uint(qc) & 0xffff0000u), r_ = qc - a_;
        const float b_ = __uint_as_float(__float_as_uint(r_) & 0xffff0000u), c_ = r_ - b_;
        const float pc[3] = {a_, b_, c_};
#pragma unroll
        for (int k = 0; k < 3; ++k) xqb[k][1] = (xqb[k][1] & 0xffff0000u) | (__float_as_uint(pc[k]) >> 16);
    };
    if constexpr (CHB) {
        float xs_[8];
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            const int k = q == 0 ? s_ - 3 : 5 + s_, kc = min(max(k, 0), nx - 1);
            const float u = A.xin[(size_t)ia * nx + kc];
            float v = q < 2 && k >= 0 && k < nx && cat ? u : 0.f;
            if (q == 0 && s_ == 0) v = nm;
            if (q == 0 && s_ == 1) v = cat ? 1.f : 0.f;
            if (q == 0 && s_ == 2) v = cat ? qa : 0.f;
            xs_[s_] = v;
        }
        w16_split3(xs_, xqb[0], xqb[1], xqb[2]);
    }
    const bool have_h = !FRONT && A.h_in != nullptr;        // h given by the caller (make_model's h_inp); zeros with the compact entry
    f32x4 hk[3] = {w16_splat(0.f), w16_splat(0.f), w16_splat(0.f)};
    if (have_h && cat) {
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) hk[rb] = w16_ld(A.h_in + (size_t)(a0 + col) * EPNN_EDIM + 16 * rb + 4 * q);
    }
    // edge operand of the G products: the 16 basis coordinates of the pair (FRONT, K = 16: lane (q, n16) takes 4q..4q+3) or
    // its 48-channel e row (K = 48: channels 12q..12q+11), see k_wave_forward
    constexpr int KE = FRONT ? EPNN_ER / 4 : 12;
    auto load_e1 = [&](int slot, float (&e)[KE]) {
        const int sl = slot < np ? slot : 0;
        if (FRONT) {
            const f32x4 v = w16_ld(A.pt + (size_t)(p0 + sl) * EPNN_ER + 4 * q);
            e[0] = v[0]; e[1] = v[1]; e[2] = v[2]; e[3] = v[3];
        } else {
            const float *r = A.pe + (size_t)(p0 + sl) * EPNN_EDIM + 12 * q;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const f32x4 v = w16_ld(r + 4 * k);
                e[(4 * k) % KE] = v[0]; e[(4 * k + 1) % KE] = v[1]; e[(4 * k + 2) % KE] = v[2]; e[(4 * k + 3) % KE] = v[3];
            }
        }
    };
    auto load_e = [&](int gt, float (&e0)[KE], float (&e1)[KE]) {
        load_e1(gt * 32 + n16, e0);
        load_e1(gt * 32 + 16 + n16, e1);
    };
    float gw[2][KE], ge0[KE], ge1[KE];
    W16_G_DECL;
    GW_LD(FRONT ? X.g[0].we16 : X.g[0].we, X.g[0].we16b);
    WAVE_FENCE();

    for (int i = tid; i < n * (PMS / 2); i += nthr) reinterpret_cast<unsigned *>(pm)[i] = 0xFFFFFFFFu;
    sync();
    if (!FRONT) {
        // pair map and pair records from the lists of the dense front-end
        for (int p = tid; p < np; p += nthr) {
            const int li = A.pi[p0 + p] - a0, lj = A.pj[p0 + p] - a0;
            pm[lj * PMS + li] = (unsigned short)p;                      // message into i = li from j = lj
            if (A.psym[p0 + p]) pm[li * PMS + lj] = (unsigned short)p;
            eij[p] = (unsigned short)(li | (lj << 8));
        }
    } else {
        // ---- slots in row-major order: every wavefront of the molecule (same values to the same words)
        int base = 0;
        if (NW == 2) {                                      // up to 32 atoms: rows i0, i0+1 per step, the lower row's pairs first
            for (int i0 = 0; i0 + 1 < n; i0 += 2) {
                const int i = i0 + hh;
                const bool near = c > i && c < n && wave_dist2(xs, i, c) < A.cut2;
                const unsigned long long bal = __ballot(near);
                const unsigned lo = (unsigned)bal, hi = (unsigned)(bal >> 32);
                if (near) {
                    const int slot = base + (hh ? __popc(lo) : 0) + __popc((hh ? hi : lo) & ((1u << c) - 1u));
                    eij[slot] = (unsigned short)(i | (c << 8));
                    pm[c * PMS + i] = (unsigned short)slot;
                    pm[i * PMS + c] = (unsigned short)slot;
                }
                base += __popc(lo) + __popc(hi);
            }
        } else {                                            // up to 64: one row per step, lane = partner
            for (int i = 0; i + 1 < n; ++i) {
                const bool near = lane > i && lane < n && wave_dist2(xs, i, lane) < A.cut2;
                const unsigned long long bal = __ballot(near);
                if (near) {
                    const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
                    eij[slot] = (unsigned short)(i | (lane << 8));
                    pm[lane * PMS + i] = (unsigned short)slot;
                    pm[i * PMS + lane] = (unsigned short)slot;
                }
                base += __popcll(bal);
            }
        }
        np = base;
        glds = min(np, grows_g);
        gover = np > glds;
        ngt = (np + 31) >> 5;
        wave_sync_lds();
        // ---- edge coefficients of every pair, one lane per pair (see k_wave_forward); the two wavefronts take alternate groups of 64
        for (int s0 = 64 * doff; s0 < np; s0 += 64 * dstep) {
            if (s0 + lane < np) {
                const int ij = eij[s0 + lane];
                const double D = wave_dist(xs, ij & 0xFF, ij >> 8);
                const float wgt = wave_near(A.flip, A.nflip, D) ? 1.0f : 0.0f;
                A.pwi[p0 + s0 + lane] = wgt;
                A.pwj[p0 + s0 + lane] = wgt;
                const double tt = D * A.tab_inv_h;
                const int i0 = min(max((int)tt - 1, 0), A.tab_n - 4);
                const float u = (float)(tt - (double)i0);
                const float um1 = u - 1.f, um2 = u - 2.f, um3 = u - 3.f;
                const float w0 = -(um1 * um2 * um3) * (1.f / 6.f), w1 = (u * um2 * um3) * 0.5f;
                const float w2 = -(u * um1 * um3) * 0.5f, w3 = (u * um1 * um2) * (1.f / 6.f);
                const float *trow = A.etab + (size_t)i0 * EPNN_ER;
                float *prow = A.pt + (size_t)(p0 + s0 + lane) * EPNN_ER;
#pragma unroll
                for (int g = 0; g < EPNN_ER / 4; ++g) {
                    const f32x4 v = w0 * w16_ld(trow + 4 * g) + w1 * w16_ld(trow + EPNN_ER + 4 * g) +
                                    w2 * w16_ld(trow + 2 * EPNN_ER + 4 * g) + w3 * w16_ld(trow + 3 * EPNN_ER + 4 * g);
                    w16_st(prow + 4 * g, v);
                }
            }
        }
        sync();                                             // every pair's coefficients are written; the coordinates are dead
        if (ngt > 0) load_e(doff, ge0, ge1);
    }
    for (int i = tid; i < EPNN_PST; i += nthr) Gl[glds * EPNN_PST + i] = 0.f;        // the sweep's zero row
    sync();

    const float Nf = (float)A.N, padw = (float)(A.N - n);
    const int Tg = A.T, Te = A.T;
    const int fo = 4 * q;

    // G rows of the tiles gt = doff, doff + dstep, ...: rows >= glds go to HBM
    auto gtile = [&](int gt, const float (&e0)[KE], const float (&e1)[KE]) {
        const int s0 = gt * 32 + n16, s1 = s0 + 16;
        f32x4 d0[2] = {w16_splat(0.f), w16_splat(0.f)};
        g_mm(e0, d0);
        if (s0 < min(np, glds)) { w16_st(Gl + s0 * EPNN_PST + fo, d0[0]); w16_st(Gl + s0 * EPNN_PST + 16 + fo, d0[1]); }
        if (gover) {
            asm volatile("" ::: "memory");
            if (s0 >= glds && s0 < np) { w16_st(A.gx + (size_t)(p0 + s0) * 32 + fo, d0[0]); w16_st(A.gx + (size_t)(p0 + s0) * 32 + 16 + fo, d0[1]); }
        }
        if (gt * 32 + 16 < np) {
            f32x4 d1[2] = {w16_splat(0.f), w16_splat(0.f)};
            g_mm(e1, d1);
            if (s1 < min(np, glds)) { w16_st(Gl + s1 * EPNN_PST + fo, d1[0]); w16_st(Gl + s1 * EPNN_PST + 16 + fo, d1[1]); }
            if (gover) {
                asm volatile("" ::: "memory");
                if (s1 >= glds && s1 < np) { w16_st(A.gx + (size_t)(p0 + s1) * 32 + fo, d1[0]); w16_st(A.gx + (size_t)(p0 + s1) * 32 + 16 + fo, d1[1]); }
            }
        }
    };
    auto gtiles = [&]() {
        float en0[KE], en1[KE];
        int gt = doff;
        if (!FRONT && gt < ngt) load_e(gt, ge0, ge1);       // (48-channel rows: 24 registers, not held across the other phases)
#pragma unroll 1
        for (; gt + dstep < ngt; gt += 2 * dstep) {
            load_e(gt + dstep, en0, en1);
            WAVE_FENCE();
            gtile(gt, ge0, ge1);
            load_e(min(gt + 2 * dstep, ngt - 1), ge0, ge1);
            WAVE_FENCE();
            gtile(gt + dstep, en0, en1);
        }
        if (gt < ngt) gtile(gt, ge0, ge1);
    };
    auto gprefetch = [&](int weoff, int weoffb) {
        GW_LD(weoff, weoffb);
        if (FRONT && ngt > 0) load_e(doff, ge0, ge1);
    };
    auto vec2 = [&](int off, f32x4 (&v)[2]) {
        v[0] = w16_ld(wp + off + fo);
        v[1] = w16_ld(wp + off + 16 + fo);
    };

    f32x4 Bv[2] = {w16_splat(0.f), w16_splat(0.f)};         // nm*u2 of the column
    // ================================================================== GNN steps (charge_gn.py:60-74)
    {
        f32x4 P[2], U[2];
#ifdef EPNN_SWEEP_F32
        float pb[2][8];
#else
        u32x4 pb[2][3];                                     // W2 as three bf16 pieces (k_wave_forward: w16_split3)
#endif
        f32x4 b2v[2];
        {
            float wa[2][EPNN_XS], wc[2][EPNN_XS];
            W16_LDX(wa, X.wi0, 2, EPNN_XS, EPNN_XS + 12, 0);
            W16_LDX(wc, X.wj0, 2, EPNN_XS, EPNN_XS + 12, 0);
            WAVE_FENCE();
            gtiles();
#ifdef EPNN_SWEEP_F32
            W16_LD(pb, X.g[0].w2, 2, 8);
#else
            W16_LDB(pb, X.g[0].w2b);
#endif
            vec2(X.g[0].b2, b2v);
            WAVE_FENCE();
            f32x4 r[2] = {w16_splat(0.f), w16_splat(0.f)};
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) { P[rb] = w16_splat(0.f); U[rb] = w16_splat(0.f); }
            w16_mm_skip<2, EPNN_XS, EPNN_XS - 1>(wa, xq, P, xs3);
            w16_mm_skip<2, EPNN_XS, EPNN_XS - 1>(wc, xq, r, xs3);
            if (have_h) {                                   // the h steps of Wi / Wj and the h block of the update MLP's first layer
                float wh[2][12], hin[12];
#pragma unroll
                for (int s = 0; s < 12; ++s) hin[s] = hk[s >> 2][s & 3];
                W16_LDX(wh, X.wi0, 2, 12, EPNN_XS + 12, EPNN_XS);
                w16_mm<2, 12>(wh, hin, P);
                W16_LDX(wh, X.wj0, 2, 12, EPNN_XS + 12, EPNN_XS);
                w16_mm<2, 12>(wh, hin, r);
                W16_LD(wh, X.u1h0, 2, 12);
                w16_mm<2, 12>(wh, hin, U);
            }
            if (own) { w16_st(Rl + col * EPNN_PST + fo, r[0]); w16_st(Rl + col * EPNN_PST + 16 + fo, r[1]); }
            if (nxt > 0 && !blk1) { w16_st(P0t + col * EPNN_PST + fo, P[0]); w16_st(P0t + col * EPNN_PST + 16 + fo, P[1]); }
        }
        sync();

#pragma unroll 1
        for (int t = 0; t < Tg; ++t) {
            const WaveGnnPack &M = X.g[t];
            const bool lastg = t + 1 == Tg;
            f32x4 S[2] = {w16_splat(0.f), w16_splat(0.f)};
            float u1s[2][8];
            {
                // partner tiles of this block: tile tt gives copy k of an atom partner tt * Cw + k (see k_wave_forward); wavefront 0
                // of a split stops at tile J0 of its block, wavefront 1 runs the rest of them after its own
                const float *zrow = Gl + glds * EPNN_PST;
                const int nt = blk1 ? nt1s : J0;
                auto sweep = [&](auto over_tag) {
                    constexpr bool OVER = decltype(over_tag)::value;
                    struct Ops { f32x4 r[2], g[2]; float w; };
                    auto grow = [&](int sl, f32x4 (&g)[2]) {
                        const float *gp = Gl + min(sl, glds) * EPNN_PST;
                        g[0] = w16_ld(gp + fo);
                        g[1] = w16_ld(gp + 16 + fo);
                        if (OVER && sl >= glds && sl != 0xFFFF) {
                            g[0] = w16_ld(A.gx + (size_t)(p0 + sl) * 32 + fo);
                            g[1] = w16_ld(A.gx + (size_t)(p0 + sl) * 32 + 16 + fo);
                        }
                    };
                    // tiles tt = 0 .. ntile-1 of the column block whose column is atom `catom` (real: `creal`): partner jb + tt * jm
                    auto run = [&](int ntile, int jb, int jm, int catom, bool creal, const f32x4 (&Pc)[2], f32x4 (&Sc)[2], bool last_phase) {
                        auto load_ops = [&](int tt, Ops &o_) {
                            const int jp = jb + tt * jm;
                            const bool real = jp < n && creal;
                            const float *rrow = real ? Rl + jp * EPNN_PST : zrow;
                            o_.r[0] = w16_ld(rrow + fo);
                            o_.r[1] = w16_ld(rrow + 16 + fo);
                            grow(real ? (int)pm[jp * PMS + catom] : 0xFFFF, o_.g);
                            o_.w = jp < n ? 1.f : (jp == n ? padw : 0.f);
                        };
                        auto tile = [&](const Ops &o_) {
                            const f32x4 za = w16_relu((Pc[0] + o_.r[0]) + o_.g[0]), zb = w16_relu((Pc[1] + o_.r[1]) + o_.g[1]);
                            const float z[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
                            f32x4 d[2] = {b2v[0], b2v[1]};
#ifdef EPNN_SWEEP_F32
                            WAVE_FENCE();                              // (the element-wise work in front of the MFMAs: see k_wave_forward)
                            w16_mm<2, 8>(pb, z, d);
#else
                            u32x4 z1, z2, z3;
                            w16_split3(z, z1, z2, z3);
                            WAVE_FENCE();                              // (the element-wise work in front of the MFMAs: see k_wave_forward)
                            w16_mm_bf(pb, z1, z2, z3, d);
#endif
#pragma unroll
                            for (int rb = 0; rb < 2; ++rb) Sc[rb] += o_.w * w16_relu(d[rb]);
                        };
                        Ops oa, ob;
                        load_ops(0, oa);
                        int tt = 0;
#pragma unroll 1
                        for (; tt + 2 < ntile; tt += 2) {
                            load_ops(tt + 1, ob);
                            WAVE_FENCE();
                            tile(oa);
                            load_ops(tt + 2, oa);
                            WAVE_FENCE();
                            tile(ob);
                        }
                        if constexpr (!CHB) { if (last_phase) { W16_LD(u1s, M.u1s, 2, 8); } }         // first operand of the update MLP
                        if (tt + 1 < ntile) {
                            load_ops(tt + 1, ob);
                            WAVE_FENCE();
                            tile(oa);
                            tile(ob);
                        } else {
                            WAVE_FENCE();
                            tile(oa);
                        }
                    };
                    const bool extra = blk1 && nxt > 0;
                    run(nt, copy, Cw, col, cat, P, S, !extra);
                    if (extra) {
                        // the tiles J0 .. n of every full block: its P rows come from LDS, the sums go back through LDS
#pragma unroll 1
                        for (int k = 0; k < NW - 1; ++k) {
                            const f32x4 Px[2] = {w16_ld(P0t + (16 * k + n16) * EPNN_PST + fo), w16_ld(P0t + (16 * k + n16) * EPNN_PST + 16 + fo)};
                            f32x4 Sx[2] = {w16_splat(0.f), w16_splat(0.f)};
                            run(nxt, J0, 1, 16 * k + n16, true, Px, Sx, k == NW - 2);
                            w16_st(scrx + k * 512 + (n16 * 4 + q) * 8, Sx[0]);
                            w16_st(scrx + k * 512 + (n16 * 4 + q) * 8 + 4, Sx[1]);
                        }
                    }
                };
                if (gover) sweep(std::true_type{});
                else sweep(std::false_type{});
                if (Cw > 1) {
                    // wavefront 1: add the copies in a fixed order (all of them end up with the same bits); its own scratch
                    // (wavefront 0 may still be reading G rows)
                    wave_sync_lds();
                    w16_st(scr + (n16 * 4 + q) * 8, S[0]);
                    w16_st(scr + (n16 * 4 + q) * 8 + 4, S[1]);
                    wave_sync_lds();
                    f32x4 t0_ = w16_splat(0.f), t1_ = w16_splat(0.f);
                    for (int k = 0; k < Cw; ++k) {
                        const int src = (n16 % m1) + m1 * k;
                        t0_ += w16_ld(scr + (src * 4 + q) * 8);
                        t1_ += w16_ld(scr + (src * 4 + q) * 8 + 4);
                    }
                    S[0] = t0_;
                    S[1] = t1_;
                    wave_sync_lds();
                }
                if (split) {
                    // both sweeps are over: the G rows and the R rows may be replaced, and block 0 gets the sums over its
                    // partners J0 .. n from wavefront 1 (added last: a fixed order)
                    wg2_sync();
                    if (!blk1 && nxt > 0) { S[0] += w16_ld(scrx + w * 512 + (n16 * 4 + q) * 8); S[1] += w16_ld(scrx + w * 512 + (n16 * 4 + q) * 8 + 4); }
                }
            }
            // ---- update MLP (charge_gn.py:71-74); the last message Dense is folded into u1s
            if constexpr (CHB) {
                u32x4 w1[2][3], w2b[2][3];
                f32x4 cv[2], bv[2];
                W16_LDB(w1, M.u1sb);
                W16_LDB(w2b, M.u2b);
                vec2(M.cb3, cv);
                vec2(M.bu1, bv);
                WAVE_FENCE();
                f32x4 d[2] = {U[0], U[1]};
                float in[8];
                u32x4 s1, s2, s3;
                w16_feed(S, in);
                w16_split3(in, s1, s2, s3);
                w16_mm_bf(w1, s1, s2, s3, d);
                f32x4 a_[2];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) a_[rb] = w16_relu(nm * (d[rb] + Nf * cv[rb]) + bv[rb]);
                vec2(M.bu2, bv);
                if (!lastg) gprefetch(FRONT ? X.g[t + 1].we16 : X.g[t + 1].we, X.g[t + 1].we16b);
                else { GW_LD(FRONT ? X.e[0].we16 : X.e[0].we, X.e[0].we16b); }
                WAVE_FENCE();
                d[0] = bv[0]; d[1] = bv[1];
                w16_feed(a_, in);
                w16_split3(in, s1, s2, s3);
                w16_mm_bf(w2b, s1, s2, s3, d);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) Bv[rb] = nm * w16_relu(d[rb]);
                w16_feed(Bv, in);
                w16_split3(in, Bp[0], Bp[1], Bp[2]);
            } else
            {
                float w2[2][8], in[8];
                f32x4 cv[2], bv[2];
                W16_LD(w2, M.u2, 2, 8);
                vec2(M.cb3, cv);
                vec2(M.bu1, bv);
                WAVE_FENCE();
                f32x4 d[2] = {U[0], U[1]};
                w16_feed(S, in);
                w16_mm<2, 8>(u1s, in, d);
                f32x4 a_[2];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) a_[rb] = w16_relu(nm * (d[rb] + Nf * cv[rb]) + bv[rb]);
                vec2(M.bu2, bv);
                if (!lastg) gprefetch(FRONT ? X.g[t + 1].we16 : X.g[t + 1].we, X.g[t + 1].we16b);
                else { GW_LD(FRONT ? X.e[0].we16 : X.e[0].we, X.e[0].we16b); }
                WAVE_FENCE();
                d[0] = bv[0]; d[1] = bv[1];
                w16_feed(a_, in);
                w16_mm<2, 8>(w2, in, d);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) Bv[rb] = nm * w16_relu(d[rb]);
            }
            if constexpr (CHB) {
              if (!lastg) {
                u32x4 wh[2][3], wx[2][3];
                W16_LDB(wh, M.pwihb);
                W16_LDB(wx, M.pwixb);
                WAVE_FENCE();
                gtiles();
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) P[rb] = w16_splat(0.f);
                w16_mm_bf(wh, Bp[0], Bp[1], Bp[2], P);
                w16_mm_bf(wx, xqb[0], xqb[1], xqb[2], P);
                W16_LDB(wh, M.pwjhb);
                W16_LDB(wx, M.pwjxb);
                f32x4 cu[2];
                vec2(M.cu3, cu);
                f32x4 r[2] = {w16_splat(0.f), w16_splat(0.f)};
                w16_mm_bf(wh, Bp[0], Bp[1], Bp[2], r);
                w16_mm_bf(wx, xqb[0], xqb[1], xqb[2], r);
                if (own) { w16_st(Rl + col * EPNN_PST + fo, r[0]); w16_st(Rl + col * EPNN_PST + 16 + fo, r[1]); }
                if (nxt > 0 && !blk1) { w16_st(P0t + col * EPNN_PST + fo, P[0]); w16_st(P0t + col * EPNN_PST + 16 + fo, P[1]); }
                W16_LDB(wh, M.pu1b);
#ifdef EPNN_SWEEP_F32
                W16_LD(pb, X.g[t + 1].w2, 2, 8);
#else
                W16_LDB(pb, X.g[t + 1].w2b);
#endif
                vec2(X.g[t + 1].b2, b2v);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) U[rb] = nm * cu[rb];
                w16_mm_bf(wh, Bp[0], Bp[1], Bp[2], U);
                sync();
              }
            } else
            if (!lastg) {
                float wa[2][8 + EPNN_XS], wbm[2][8 + EPNN_XS], in[8 + EPNN_XS];
                W16_LD(wa, M.pwi, 2, 8 + EPNN_XS);
                WAVE_FENCE();
                gtiles();
#pragma unroll
                for (int s = 0; s < 8; ++s) in[s] = Bv[s >> 2][s & 3];
#pragma unroll
                for (int s = 0; s < EPNN_XS; ++s) in[8 + s] = xq[s];
                W16_LD(wbm, M.pwj, 2, 8 + EPNN_XS);
                float wu[2][8];
                f32x4 cu[2];
                W16_LD(wu, M.pu1, 2, 8);
                vec2(M.cu3, cu);
                WAVE_FENCE();
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) P[rb] = w16_splat(0.f);
                w16_mm_skip<2, 8 + EPNN_XS, 7 + EPNN_XS>(wa, in, P, xs3);
                f32x4 r[2] = {w16_splat(0.f), w16_splat(0.f)};
                w16_mm_skip<2, 8 + EPNN_XS, 7 + EPNN_XS>(wbm, in, r, xs3);
                if (own) { w16_st(Rl + col * EPNN_PST + fo, r[0]); w16_st(Rl + col * EPNN_PST + 16 + fo, r[1]); }
                if (nxt > 0 && !blk1) { w16_st(P0t + col * EPNN_PST + fo, P[0]); w16_st(P0t + col * EPNN_PST + 16 + fo, P[1]); }
#ifdef EPNN_SWEEP_F32
                W16_LD(pb, X.g[t + 1].w2, 2, 8);
#else
                W16_LDB(pb, X.g[t + 1].w2b);
#endif
                vec2(X.g[t + 1].b2, b2v);
                WAVE_FENCE();
                float bin[8];
                w16_feed(Bv, bin);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) U[rb] = nm * cu[rb];
                w16_mm<2, 8>(wu, bin, U);
                sync();
            }
        }
    }

    // ================================================================== EPN steps (charge_gn.py:98-118)
    {
        sync();                                             // the GNN's tables are dead: switch to the EPN layout
        for (int i = tid; i < n * DSTW; i += nthr) Dm[i] = 0.f;
        const int qs = (nx + 1) >> 2, ql = (nx + 1) & 3;
#pragma unroll 1
        for (int t = 0; t < Te; ++t) {
            const WaveEpnPack &M = X.e[t];
            if constexpr (CHB) {
                u32x4 wh[2][3], wx[2][3];
                W16_LDB(wh, M.wifhb);
                W16_LDB(wx, M.wifxb);
                WAVE_FENCE();
                f32x4 d[2] = {w16_splat(0.f), w16_splat(0.f)};
                w16_mm_bf(wh, Bp[0], Bp[1], Bp[2], d);
                w16_mm_bf(wx, xqb[0], xqb[1], xqb[2], d);
                W16_LDB(wh, M.wjfhb);
                W16_LDB(wx, M.wjfxb);
                if (own) { w16_st(Pl + col * EPNN_PST + fo, d[0]); w16_st(Pl + col * EPNN_PST + 16 + fo, d[1]); }
                d[0] = w16_splat(0.f); d[1] = w16_splat(0.f);
                w16_mm_bf(wh, Bp[0], Bp[1], Bp[2], d);
                w16_mm_bf(wx, xqb[0], xqb[1], xqb[2], d);
                if (own) { w16_st(Rl + col * EPNN_PST + fo, d[0]); w16_st(Rl + col * EPNN_PST + 16 + fo, d[1]); }
            } else {
                constexpr int KS = 8 + EPNN_XS, SK = 7 + EPNN_XS;
                float wa[2][KS], wbm[2][KS], in[KS];
#pragma unroll
                for (int s = 0; s < 8; ++s) in[s] = Bv[s >> 2][s & 3];
#pragma unroll
                for (int s = 0; s < EPNN_XS; ++s) in[8 + s] = xq[s];
                W16_LD(wa, M.wif, 2, KS);
                W16_LD(wbm, M.wjf, 2, KS);
                WAVE_FENCE();
                f32x4 d[2] = {w16_splat(0.f), w16_splat(0.f)};
                w16_mm_skip<2, KS, SK>(wa, in, d, xs3);
                if (own) { w16_st(Pl + col * EPNN_PST + fo, d[0]); w16_st(Pl + col * EPNN_PST + 16 + fo, d[1]); }
                d[0] = w16_splat(0.f); d[1] = w16_splat(0.f);
                w16_mm_skip<2, KS, SK>(wbm, in, d, xs3);
                if (own) { w16_st(Rl + col * EPNN_PST + fo, d[0]); w16_st(Rl + col * EPNN_PST + 16 + fo, d[1]); }
            }
#ifdef EPNN_SWEEP_F32
            float pb[2][8];
            W16_LD(pb, M.w2, 2, 8);
#else
            u32x4 pb[2][3];
            W16_LDB(pb, M.w2b);
#endif
            f32x4 b2v[2], w3[2];
            vec2(M.b2, b2v);
            vec2(M.w3, w3);
            sync();                                         // P / R rows (and, in step 0, the cleared transfer matrix) are in place
            {
                // blocks of 16 unordered near pairs; in a split dealt out alternately: this wavefront's k-th block is 2k + w
                const int nblk = (np + 15) >> 4;
                const int nown = split ? (nblk - w + NW - 1) / NW : nblk;
                auto bidx = [&](int k) { return dstep * min(k, nown - 1) + doff; };
                struct Rec { int ij; float wi, wj; };
                struct Rows { float e[KE]; f32x4 pi_[2], rj_[2], pj_[2], ri_[2]; };
                auto load_rec = [&](int blk, Rec &r_) {
                    const int sl = blk * 16 + n16 < np ? blk * 16 + n16 : 0;
                    r_.ij = eij[sl];
                    r_.wi = A.pwi[p0 + sl];
                    r_.wj = A.pwj[p0 + sl];
                };
                auto load_rows = [&](int blk, const Rec &r_, Rows &w_) {
                    const int sl = blk * 16 + n16 < np ? blk * 16 + n16 : 0;
                    const int li = r_.ij & 0xFF, lj = r_.ij >> 8;
                    load_e1(sl, w_.e);
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        w_.pi_[rb] = w16_ld(Pl + li * EPNN_PST + 16 * rb + fo);
                        w_.rj_[rb] = w16_ld(Rl + lj * EPNN_PST + 16 * rb + fo);
                        w_.pj_[rb] = w16_ld(Pl + lj * EPNN_PST + 16 * rb + fo);
                        w_.ri_[rb] = w16_ld(Rl + li * EPNN_PST + 16 * rb + fo);
                    }
                };
                auto block = [&](int blk, const Rec &r_, const Rows &w_) {
                    const bool valid = blk * 16 + n16 < np;
                    const int li = r_.ij & 0xFF, lj = r_.ij >> 8;
                    f32x4 g[2] = {w16_splat(0.f), w16_splat(0.f)};
                    g_mm(w_.e, g);
                    const f32x4 ua = w16_relu((g[0] + w_.pi_[0]) + w_.rj_[0]), ub = w16_relu((g[1] + w_.pi_[1]) + w_.rj_[1]);
                    const f32x4 va = w16_relu((g[0] + w_.pj_[0]) + w_.ri_[0]), vb = w16_relu((g[1] + w_.pj_[1]) + w_.ri_[1]);
                    const float zu[8] = {ua[0], ua[1], ua[2], ua[3], ub[0], ub[1], ub[2], ub[3]};
                    const float zv[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                    f32x4 au[2] = {b2v[0], b2v[1]}, av[2] = {b2v[0], b2v[1]};
#ifdef EPNN_SWEEP_F32
                    WAVE_FENCE();
                    w16_mm<2, 8>(pb, zu, au);
                    w16_mm<2, 8>(pb, zv, av);
#else
                    u32x4 zu1, zu2, zu3, zv1, zv2, zv3;
                    w16_split3(zu, zu1, zu2, zu3);
                    w16_split3(zv, zv1, zv2, zv3);
                    WAVE_FENCE();
                    w16_mm_bf(pb, zu1, zu2, zu3, au);
                    w16_mm_bf(pb, zv1, zv2, zv3, av);
#endif
                    WAVE_FENCE();
                    float fd = 0.f;
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        const f32x4 tt = w16_relu(au[rb]) - w16_relu(av[rb]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) fd = fmaf(w3[rb][r], tt[r], fd);
                    }
                    const float d = 0.5f * w16_sumq(fd);               // charge_gn.py:116
                    if (q == 0 && valid && r_.wi != 0.f) Dm[li * DSTW + lj] = r_.wi * d;
                    if (q == 1 && valid && r_.wj != 0.f) Dm[lj * DSTW + li] = -(r_.wj * d);
                };
                if (nown > 0) {
                    Rec r0, r1;
                    Rows w0, w1;
                    load_rec(bidx(0), r0);
                    load_rec(bidx(1), r1);
                    load_rows(bidx(0), r0, w0);
                    int k = 0;
#pragma unroll 1
                    for (; k + 1 < nown; k += 2) {
                        Rec r2, r3;
                        load_rows(bidx(k + 1), r1, w1);
                        load_rec(bidx(k + 2), r2);
                        WAVE_FENCE();
                        block(bidx(k), r0, w0);
                        load_rows(bidx(k + 2), r2, w0);
                        load_rec(bidx(k + 3), r3);
                        WAVE_FENCE();
                        block(bidx(k + 1), r1, w1);
                        r0 = r2;
                        r1 = r3;
                    }
                    if (k < nown) block(bidx(k), r0, w0);
                }
            }
            sync();                                         // every transfer of this step is in the matrix
            if (t + 1 < Te) { GW_LD(FRONT ? X.e[t + 1].we16 : X.e[t + 1].we, X.e[t + 1].we16b); }
            WAVE_FENCE();
            // q_i += sum_j antisym_ij (charge_gn.py:118): lane (q, n16) adds columns j = q mod 4 of its atom's row.  (The next
            // step writes the matrix only behind its own barrier, which this wavefront reaches after these reads.)
            {
                float dq = 0.f;
                const float *row = Dm + (cat ? col : 0) * DSTW;
                for (int j = q; j < n; j += 4) dq += row[j];
                dq = w16_sumq(dq);
#pragma unroll
                for (int s = 0; s < EPNN_XS; ++s)
                    if (s == qs && q == ql) xq[s] += cat ? dq : 0.f;
                if constexpr (CHB) {
                    qc += cat ? dq : 0.f;
                    xq_charge();
                }
            }
        }
#pragma unroll
        for (int s = 0; s < EPNN_XS; ++s)
            if (s == qs && q == ql && own) A.q_out[a0 + col] = xq[s];
    }
    if (A.handoff) {
        sync();
        if (split ? threadIdx.x == 0 : lane == 0) {
            // one report per molecule; the last one to finish hands status + pair count to the host and re-zeroes the control words
            wave_handoff(A, np);
        }
    }
}
