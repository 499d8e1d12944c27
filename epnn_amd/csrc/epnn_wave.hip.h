// Wave-autonomous fused forward: ONE wavefront runs the whole GNN_layer + EPN_layer stack (reference
// charge_gn.py:56-119, 2T steps) of one molecule with n <= 32 real atoms.  No workgroup barriers, no idle waves:
// latency is hidden by the other wavefronts of the SIMD (independent molecules), the matrix pipe of every SIMD is fed
// by whole-molecule dependent chains.
//
// Lane l = 32*hh + c owns ATOM c (atom-level quantities are MFMA accumulators: register r = feature kappa(hh,r) of
// atom c).  Everything that is per atom lives in registers for the whole forward:
//   xq[KX]   the "small" inputs [node_mask, x_0..x_{nx-1}, q, 1] in even/odd K order (feature 2s+hh in register s)
//   hk[24]   h (48 features) in accumulator order: s<16 feature kappa(hh,s), s>=16 feature 32+kappa(hh,s-16)
//   P[16]    P_i = Wi^T a_i + b1 of the current pair MLP;  S[16] the message sum;  u1pre[16] the h-part of the update
// Only what other lanes must read goes through LDS: R_j rows (broadcast reads), the near-pair terms G, the pair map.
//
// GNN pair sweep (charge_gn.py:62-70): tile j = "partner j of every atom": column c of the tile is the pair (i=c, j),
//   z1 = relu(P_i + R_j + G_ij), acc = W2^T z1 + b2 (16 MFMAs), S_i += relu(acc): plain register accumulation, the
//   sum over partners never leaves the lane.  One more tile with R = G = 0 is the reference's zero-padded partner; it
//   is added (N - n) times.
// Update chain (charge_gn.py:71-74) with h never materialised between steps:  the next step needs h only through
//   Wi_h^T h, Wj_h^T h and Wu1_h^T h, and h = nm (Wu3^T u2 + bu3), so the host folds Wu3 into those three matrices
//   (float64 products, epnn_api.hip pack_wave): K = 32 (nm*u2) + KX instead of 48 + KX, and no Wu3 GEMM per step.
//   h itself is produced once, after the last step (for the EPN stack / GNN_layer output).
// EPN (charge_gn.py:98-118): one tile column per UNORDERED near pair, both directions share G; +d to i, -d to j.
#pragma once
#include "epnn_common.h"

struct WaveArgs {
    const float *wpack;
    WaveIndex wx;
    const float *xin;      // [A][nx]
    const float *Q;        // [B]
    const int *moff;       // [B+1]
    const int *order;      // molecules of this launch (largest first)
    const int *row_off;    // [A+1]
    const int *pi, *pj, *psym;
    const float *pe, *pwi, *pwj;
    float *q_out;          // [A]
    float *h_out;          // optional [A][48]
    int *status;
    int N, T, nx, A;
    const float *h_in;     // optional [A][48]
    const float *q_in;     // optional [A]
    const float *nm_in;    // optional [A]
    float *gx;             // [pcap][32] rows of G that do not fit the wave's LDS budget
    int lds_words;         // LDS budget of one wave (floats)
    unsigned long long *stamps;   // diagnostic build only (-DEPNN_STAMPS): [block][64] s_memtime values
};

#ifdef EPNN_STAMPS
#define WAVE_STAMP()                                                                          \
    do {                                                                                      \
        if (lane == 0 && A.stamps && nstamp < 62) {                                           \
            A.stamps[(size_t)blockIdx.x * 64 + nstamp] = __builtin_amdgcn_s_memtime();        \
            ++nstamp;                                                                         \
        }                                                                                     \
    } while (0)
#else
#define WAVE_STAMP() do { } while (0)
#endif

#define EPNN_WLD(dst, off, cnt)                                      \
    _Pragma("unroll") for (int s_ = 0; s_ < (cnt); ++s_)(dst)[s_] = wp[(off) + s_ * 64 + lane]

// order LDS / global traffic between lanes of the wave (the compiler sees no dependence between different lanes)
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void wave_sync_all() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// first feature of the g-th group of four of the hk order
__device__ __forceinline__ int wave_hk_f0(int hh, int g) { return g < 4 ? 4 * hh + 8 * g : 32 + 4 * hh + 8 * (g - 4); }

template <int K>
__device__ __forceinline__ f32x16 wave_chain(const float (&w)[K], const float (&b)[K], f32x16 acc) {
#pragma unroll
    for (int s = 0; s < K; ++s) acc = epnn_mfma(w[s], b[s], acc);
    return acc;
}

// e rows of G tile gt: lane (c,hh) takes channels 24hh..24hh+23 of pair gt*32+c
__device__ __forceinline__ void wave_load_e(const float *pe, int p0, int np, int gt, int c, int hh, float (&ev)[24]) {
    const int slot = gt * 32 + c;
    const float *erow = pe + (size_t)(p0 + (slot < np ? slot : 0)) * EPNN_EDIM + hh * 24;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(erow + 4 * q);
        ev[4 * q] = v[0]; ev[4 * q + 1] = v[1]; ev[4 * q + 2] = v[2]; ev[4 * q + 3] = v[3];
    }
}

// G rows of all near pairs of the molecule for the pair MLP whose We fragments are in w[]: LDS rows [0, glds),
// the rest to the overflow rows in HBM
__device__ __forceinline__ void wave_gtiles(const float (&w)[24], const float *pe, int p0, int np, float *Gl, int glds,
                                            float *Gx, int c, int hh) {
    const int ngt = (np + 31) >> 5;
    float ev[24];
    if (ngt > 0) wave_load_e(pe, p0, np, 0, c, hh, ev);
#pragma unroll 1
    for (int gt = 0; gt < ngt; ++gt) {
        float en[24];
        if (gt + 1 < ngt) wave_load_e(pe, p0, np, gt + 1, c, hh, en);
        else {
#pragma unroll
            for (int s = 0; s < 24; ++s) en[s] = 0.f;
        }
        f32x16 acc = wave_chain<24>(w, ev, epnn_splat16(0.f));
        const int slot = gt * 32 + c;
        if (slot < np) {
            if (slot < glds) epnn_st16(Gl + slot * EPNN_PST + hh * 16, acc);
            else epnn_st16(Gx + (size_t)(p0 + slot) * 32 + hh * 16, acc);
        }
#pragma unroll
        for (int s = 0; s < 24; ++s) ev[s] = en[s];
    }
}

template <bool GNN, bool EPN>
__global__ __launch_bounds__(64, 2) void k_wave_forward(WaveArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x, c = lane & 31, hh = lane >> 5;
    if (*A.status & EPNN_ST_PAIR_OVERFLOW) return;
    const int b = A.order[blockIdx.x];
    const int a0 = A.moff[b], n = A.moff[b + 1] - a0;
    const int p0 = A.row_off[a0], np = A.row_off[a0 + n] - p0;
    const int nx = A.nx;
    const float *wp = A.wpack;
    const WaveIndex &X = A.wx;
    int nstamp = 0;
    (void)nstamp;
    WAVE_STAMP();
    const bool catom = c < n;
    const int cr = catom ? c : n - 1;                      // lanes without an atom re-read the last row (results dropped)

    // ---- LDS layout of THIS molecule inside the wave's fixed budget
    float *Rl = sm;                                        // [n][PST]   R_j rows
    float *Pl = sm + n * EPNN_PST;                         // [n][PST]   P_i rows (EPN); the GNN keeps its pair map here
    unsigned short *pm = reinterpret_cast<unsigned short *>(Pl);      // [j][32]  near-pair slot of (i = lane, j), 0xFFFF = none
    int o = 2 * n * EPNN_PST;
    unsigned short *eij = reinterpret_cast<unsigned short *>(sm + o); // [np]  li | lj << 8
    o += EPN ? ((((np + 1) >> 1) + 3) & ~3) : 0;
    float *edi = sm + o;                                   // [np]  w_i * delta
    o += EPN ? ((np + 3) & ~3) : 0;
    float *edj = sm + o;                                   // [np]  w_j * delta
    o += EPN ? ((np + 3) & ~3) : 0;
    float *Gl = sm + o;                                    // [glds + 1][PST]; row glds is all zeros
    const int glds = min(np, (A.lds_words - o) / EPNN_PST - 1);
    const bool gover = np > glds;                          // some G rows live in HBM

    // ---- per-atom registers
    const float nmv = catom ? (A.nm_in ? A.nm_in[a0 + c] : 1.f) : 0.f;
    float xq[EPNN_KX];
    {
        const float qv = catom ? (A.q_in ? A.q_in[a0 + c] : A.Q[b] / (float)n) : 0.f;     // charge_gn.py:337-338
#pragma unroll
        for (int s = 0; s < EPNN_KX; ++s) {
            const int phi = 2 * s + hh;
            float v = 0.f;
            if (catom) {
                if (phi == 0) v = nmv;
                else if (phi <= nx) v = A.xin[(size_t)(a0 + c) * nx + phi - 1];
                else if (phi == nx + 1) v = qv;
                else if (phi == nx + 2) v = 1.f;
            }
            xq[s] = v;
        }
    }
    float hk[24];
#pragma unroll
    for (int s = 0; s < 24; ++s) hk[s] = 0.f;
    const bool have_h = A.h_in != nullptr;
    if (have_h && catom) {
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(A.h_in + (size_t)(a0 + c) * EPNN_EDIM + wave_hk_f0(hh, g));
            hk[4 * g] = v[0]; hk[4 * g + 1] = v[1]; hk[4 * g + 2] = v[2]; hk[4 * g + 3] = v[3];
        }
    }

    // ---- LDS init
    for (int i = lane; i < EPNN_PST; i += 64) Gl[glds * EPNN_PST + i] = 0.f;
    if (GNN) {
        for (int i = lane; i < n * 16; i += 64) reinterpret_cast<unsigned *>(pm)[i] = 0xFFFFFFFFu;
        wave_sync_lds();
        for (int p = lane; p < np; p += 64) {
            const int li = A.pi[p0 + p] - a0, lj = A.pj[p0 + p] - a0;
            pm[lj * 32 + li] = (unsigned short)p;                       // message into i = li from j = lj
            if (A.psym[p0 + p]) pm[li * 32 + lj] = (unsigned short)p;
        }
    }
    wave_sync_lds();

    WAVE_STAMP();   // init done
    const float Nf = (float)A.N, padw = (float)(A.N - n);
    const int Tg = GNN ? A.T : 0, Te = EPN ? A.T : 0;

    // ================================================================== GNN steps (charge_gn.py:60-74)
    if (GNN) {
        float P[16], u1pre[16];
        // ---- step 0 projections from (xq | hk)
        {
            float w[24];
            EPNN_WLD(w, X.g[0].we, 24);
            wave_gtiles(w, A.pe, p0, np, Gl, glds, A.gx, c, hh);
        }
        {
            float wa[EPNN_KX], wb[24];
            f32x16 acc = epnn_splat16(0.f);
            EPNN_WLD(wa, X.wi0, EPNN_KX);
            acc = wave_chain<EPNN_KX>(wa, xq, acc);
            if (have_h) { EPNN_WLD(wb, X.wi0 + EPNN_KX * 64, 24); acc = wave_chain<24>(wb, hk, acc); }
#pragma unroll
            for (int r = 0; r < 16; ++r) P[r] = acc[r];
            acc = epnn_splat16(0.f);
            EPNN_WLD(wa, X.wj0, EPNN_KX);
            acc = wave_chain<EPNN_KX>(wa, xq, acc);
            if (have_h) { EPNN_WLD(wb, X.wj0 + EPNN_KX * 64, 24); acc = wave_chain<24>(wb, hk, acc); }
            if (catom) epnn_st16(Rl + c * EPNN_PST + hh * 16, acc);
            acc = epnn_splat16(0.f);
            if (have_h) {
                float hm[24];
#pragma unroll
                for (int s = 0; s < 24; ++s) hm[s] = nmv * hk[s];       // masked_input = [h, m] * node_mask (charge_gn.py:72)
                EPNN_WLD(wb, X.u1h0, 24);
                acc = wave_chain<24>(wb, hm, acc);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) u1pre[r] = acc[r];
        }
        wave_sync_all();
        WAVE_STAMP();   // step-0 G tiles + projections

#pragma unroll 1
        for (int t = 0; t < Tg; ++t) {
            const WaveGnnPack &M = X.g[t];
            const bool lastg = t + 1 == Tg;
            float S[16];
            {
                float pb[16], b2k[16];
                EPNN_WLD(pb, M.w2, 16);
                epnn_ld16(wp + M.b2k + hh * 16, b2k);
                f32x16 cb2;
#pragma unroll
                for (int r = 0; r < 16; ++r) { cb2[r] = b2k[r]; S[r] = 0.f; }
                // ---- partner tiles j = 0..n-1
#pragma unroll 2
                for (int j = 0; j < n; ++j) {
                    float rj[16], g[16];
                    epnn_ld16(Rl + j * EPNN_PST + hh * 16, rj);
                    const int slot = pm[j * 32 + c];
                    if (!gover) {
                        epnn_ld16(Gl + min(slot, glds) * EPNN_PST + hh * 16, g);      // 0xFFFF -> the zero row
                    } else {
                        if (slot >= glds && slot != 0xFFFF) epnn_ld16(A.gx + (size_t)(p0 + slot) * 32 + hh * 16, g);
                        else epnn_ld16(Gl + min(slot, glds) * EPNN_PST + hh * 16, g);
                    }
                    f32x16 acc = cb2;
#pragma unroll
                    for (int s = 0; s < 16; ++s) acc = epnn_mfma(pb[s], fmaxf((P[s] + rj[s]) + g[s], 0.f), acc);
#pragma unroll
                    for (int r = 0; r < 16; ++r) S[r] += fmaxf(acc[r], 0.f);
                }
                // ---- the zero-padded partners of the reference (R = 0, G = 0), N - n of them (charge_gn.py:70)
                {
                    f32x16 acc = cb2;
#pragma unroll
                    for (int s = 0; s < 16; ++s) acc = epnn_mfma(pb[s], fmaxf(P[s], 0.f), acc);
#pragma unroll
                    for (int r = 0; r < 16; ++r) S[r] = fmaf(padw, fmaxf(acc[r], 0.f), S[r]);
                }
            }
            if (t < 2) WAVE_STAMP();   // pair tiles
            // ---- update MLP (charge_gn.py:71-74); the last message Dense is folded into u1s
            float bn[16];
            {
                float w[16], cv[16], bv[16];
                EPNN_WLD(w, M.u1s, 16);
                epnn_ld16(wp + M.cb3k + hh * 16, cv);
                epnn_ld16(wp + M.bu1k + hh * 16, bv);
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaf(Nf, cv[r], u1pre[r]);
                acc = wave_chain<16>(w, S, acc);
                float u1[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) u1[r] = fmaxf(fmaf(nmv, acc[r], bv[r]), 0.f);
                EPNN_WLD(w, M.u2, 16);
                epnn_ld16(wp + M.bu2k + hh * 16, bv);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = bv[r];
                acc = wave_chain<16>(w, u1, acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) bn[r] = fmaxf(acc[r], 0.f);
            }
            if (t < 2) WAVE_STAMP();   // U1, U2
            if (!lastg) {
                // next step: G rows, then P / R / u1pre from (nm*u2 | xq) through the folded matrices
                {
                    float w[24];
                    EPNN_WLD(w, X.g[t + 1].we, 24);
                    wave_gtiles(w, A.pe, p0, np, Gl, glds, A.gx, c, hh);
                }
                if (t < 2) WAVE_STAMP();   // G tiles
#pragma unroll
                for (int r = 0; r < 16; ++r) bn[r] *= nmv;
                float w[16], wa[EPNN_KX];
                f32x16 acc = epnn_splat16(0.f);
                EPNN_WLD(w, M.pwi, 16);
                EPNN_WLD(wa, M.pwi + 16 * 64, EPNN_KX);
                acc = wave_chain<16>(w, bn, acc);
                acc = wave_chain<EPNN_KX>(wa, xq, acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) P[r] = acc[r];
                acc = epnn_splat16(0.f);
                EPNN_WLD(w, M.pwj, 16);
                EPNN_WLD(wa, M.pwj + 16 * 64, EPNN_KX);
                acc = wave_chain<16>(w, bn, acc);
                acc = wave_chain<EPNN_KX>(wa, xq, acc);
                if (catom) epnn_st16(Rl + c * EPNN_PST + hh * 16, acc);
                float cu[16];
                epnn_ld16(wp + M.cu3k + hh * 16, cu);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = nmv * cu[r];
                EPNN_WLD(w, M.pu1, 16);
                acc = wave_chain<16>(w, bn, acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) u1pre[r] = acc[r];
                wave_sync_all();
            } else {
                // h = node_mask * (Wu3^T u2 + bu3)  (charge_gn.py:73-74), straight into the hk registers
                float w[16], bv[16];
                f32x16 acc;
                EPNN_WLD(w, X.u3, 16);
                epnn_ld16(wp + X.bu3k + hh * 16, bv);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = bv[r];
                acc = wave_chain<16>(w, bn, acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) hk[r] = nmv * acc[r];
                EPNN_WLD(w, X.u3 + 16 * 64, 16);
                epnn_ld16(wp + X.bu3k + 32 + hh * 16, bv);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = bv[r];
                acc = wave_chain<16>(w, bn, acc);
#pragma unroll
                for (int r = 0; r < 8; ++r) hk[16 + r] = nmv * acc[r];
            }
        }
        if (A.h_out && catom) {
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                f32x4 v;
                v[0] = hk[4 * g]; v[1] = hk[4 * g + 1]; v[2] = hk[4 * g + 2]; v[3] = hk[4 * g + 3];
                *reinterpret_cast<f32x4 *>(A.h_out + (size_t)(a0 + c) * EPNN_EDIM + wave_hk_f0(hh, g)) = v;
            }
        }
    }

    WAVE_STAMP();   // GNN done
    // ================================================================== EPN steps (charge_gn.py:98-118)
    if (EPN) {
        wave_sync_lds();                                    // the pair map is dead: its rows become P rows
        for (int p = lane; p < np; p += 64) {
            const int li = A.pi[p0 + p] - a0, lj = A.pj[p0 + p] - a0;
            eij[p] = (unsigned short)(li | (lj << 8));
        }
        const int qs = (nx + 1) >> 1, qh = (nx + 1) & 1;    // register / half-wave of xq that holds q
        const int ngt = (np + 31) >> 5;
#pragma unroll 1
        for (int t = 0; t < Te; ++t) {
            const WaveEpnPack &M = X.e[t];
            {
                float w[24];
                EPNN_WLD(w, M.we, 24);
                wave_gtiles(w, A.pe, p0, np, Gl, glds, A.gx, c, hh);
            }
            if (t < 2) WAVE_STAMP();   // EPN G tiles
            {
                float wa[EPNN_KX], wb[24];
                f32x16 acc = epnn_splat16(0.f);
                EPNN_WLD(wa, M.wi, EPNN_KX);
                EPNN_WLD(wb, M.wi + EPNN_KX * 64, 24);
                acc = wave_chain<EPNN_KX>(wa, xq, acc);
                acc = wave_chain<24>(wb, hk, acc);
                if (catom) epnn_st16(Pl + c * EPNN_PST + hh * 16, acc);
                acc = epnn_splat16(0.f);
                EPNN_WLD(wa, M.wj, EPNN_KX);
                EPNN_WLD(wb, M.wj + EPNN_KX * 64, 24);
                acc = wave_chain<EPNN_KX>(wa, xq, acc);
                acc = wave_chain<24>(wb, hk, acc);
                if (catom) epnn_st16(Rl + c * EPNN_PST + hh * 16, acc);
            }
            wave_sync_all();
            if (t < 2) WAVE_STAMP();   // EPN P, R
            {
                float pb[16], b2v[16], w3[16];
                EPNN_WLD(pb, M.w2, 16);
                epnn_ld16(wp + M.b2k + hh * 16, b2v);
                epnn_ld16(wp + M.w3k + hh * 16, w3);
#pragma unroll 1
                for (int gt = 0; gt < ngt; ++gt) {
                    const int slot = gt * 32 + c;
                    const bool valid = slot < np;
                    const int sl = valid ? slot : 0;
                    const int ij = eij[sl];
                    const int li = ij & 0xFF, lj = ij >> 8;
                    const float wi = A.pwi[p0 + sl], wj = A.pwj[p0 + sl];
                    float g[16];
                    if (sl < glds) epnn_ld16(Gl + sl * EPNN_PST + hh * 16, g);
                    else epnn_ld16(A.gx + (size_t)(p0 + sl) * 32 + hh * 16, g);
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
#pragma unroll
                        for (int s = 0; s < 16; ++s) acc = epnn_mfma(pb[s], fmaxf((g[s] + ta[s]) + tb[s], 0.f), acc);
                        float f = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) f = fmaf(w3[r], fmaxf(acc[r], 0.f), f);
                        if (dir == 0) fu = f; else fv = f;
                    }
                    fu += epnn_swap32(fu);
                    fv += epnn_swap32(fv);
                    const float d = 0.5f * (fu - fv);                  // charge_gn.py:116
                    if (hh == 0 && valid) { edi[slot] = wi * d; edj[slot] = wj * d; }
                }
            }
            wave_sync_lds();
            if (t < 2) WAVE_STAMP();   // EPN pair tiles
            // q_i += sum_j antisym_ij (charge_gn.py:118): each half-wave scans every other pair, fixed order
            {
                float dq = 0.f;
                for (int p = hh; p < np; p += 2) {
                    const int ij = eij[p];
                    const float di = edi[p], dj = edj[p];
                    dq += ((ij & 0xFF) == c) ? di : 0.f;
                    dq -= ((ij >> 8) == c) ? dj : 0.f;
                }
                dq += epnn_swap32(dq);
#pragma unroll
                for (int s = 0; s < EPNN_KX; ++s)
                    if (s == qs && hh == qh) xq[s] += dq;
            }
            wave_sync_lds();
            if (t < 2) WAVE_STAMP();   // charge update
        }
        // q sits in half-wave qh; hand it to the lower half for the store
        float qout = 0.f;
#pragma unroll
        for (int s = 0; s < EPNN_KX; ++s)
            if (s == qs) qout = xq[s];
        const float qo = epnn_swap32(qout);
        if (qh == 1) qout = qo;
        if (hh == 0 && catom) A.q_out[a0 + c] = qout;
    }
    WAVE_STAMP();
#ifdef EPNN_STAMPS
    if (lane == 0 && A.stamps) {
        A.stamps[(size_t)blockIdx.x * 64 + 62] = (unsigned long long)nstamp;
        A.stamps[(size_t)blockIdx.x * 64 + 63] = ((unsigned long long)n << 32) | (unsigned)np;
    }
#endif
    (void)cr; (void)Tg;
}
