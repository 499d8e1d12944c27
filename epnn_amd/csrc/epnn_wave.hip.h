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
#include <type_traits>

#include "epnn_common.h"
#include "epnn_frontend.hip.h"

struct WaveArgs {
    const float *wpack;
    WaveIndex wx;
    const float *xin;      // [A][nx]
    const float *Q;        // [B]
    const int *moff;       // [B+1]
    const int *order;      // molecules of this launch (largest first)
    const int *row_off;    // [A+1]
    const int *pi, *pj, *psym;
    float *pe, *pwi, *pwj;  // written by the kernel itself with the in-kernel front-end, read-only otherwise
    float *q_out;          // [A]
    float *h_out;          // optional [A][48]
    int *status;
    int N, T, nx, A;
    const float *h_in;     // optional [A][48]
    const float *q_in;     // optional [A]
    const float *nm_in;    // optional [A]
    float *gx;             // [pcap][32] rows of G that do not fit the wave's LDS budget
    int lds_words;         // LDS budget of one wave (floats)
    // in-kernel front-end (FRONT): the wave builds its molecule's near-pair list itself (get_init_edges, charge_gn.py:122-163)
    const float *xyz;      // [A][3]
    const int *pbase;      // [B] first pair slot of molecule b: sum of n(n-1)/2 over the molecules before it
    const double *mu;      // [48]
    double cutoff, eta;
    double cut2;           // smallest float64 t with sqrt(t) >= cutoff: D < cutoff  <=>  D*D (before the sqrt) < cut2
    float tol;
    int *host_status;      // pinned host ints [0] status bits [1] near pairs of the batch: written by the last wave to finish
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

#ifdef EPNN_ABL_W      // diagnostic ablation: every fragment load hits the same few cache lines (results are garbage)
#define EPNN_WLD(dst, off, cnt)                                      \
    _Pragma("unroll") for (int s_ = 0; s_ < (cnt); ++s_)(dst)[s_] = wp[((off) & 1023) + (s_ & 3) * 64 + lane]
#else
#define EPNN_WLD(dst, off, cnt)                                      \
    _Pragma("unroll") for (int s_ = 0; s_ < (cnt); ++s_)(dst)[s_] = wp[(off) + s_ * 64 + lane]
#endif

// order LDS / global traffic between lanes of the wave (the compiler sees no dependence between different lanes)
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void wave_sync_all() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// distance exactly as scipy.spatial.distance_matrix on float32 coordinates promoted to float64 (charge_gn.py:124),
// same operation order as epnn_dist (epnn_frontend.hip.h)
__device__ __forceinline__ double wave_dist(const double *xs, int i, int j) {
    const double dx = xs[3 * j + 0] - xs[3 * i + 0], dy = xs[3 * j + 1] - xs[3 * i + 1], dz = xs[3 * j + 2] - xs[3 * i + 2];
    return sqrt(__dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz)));
}

// its argument: D < cutoff is decided on the squared distance (exactly, see WaveArgs::cut2), the sqrt is taken per near pair only
__device__ __forceinline__ double wave_dist2(const double *xs, int i, int j) {
    const double dx = xs[3 * j + 0] - xs[3 * i + 0], dy = xs[3 * j + 1] - xs[3 * i + 1], dz = xs[3 * j + 2] - xs[3 * i + 2];
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}

// first feature of the g-th group of four of the hk order
__device__ __forceinline__ int wave_hk_f0(int hh, int g) { return g < 4 ? 4 * hh + 8 * g : 32 + 4 * hh + 8 * (g - 4); }

// A dependent chain of v_mfma_f32_32x32x2_f32 issues back to back (64 cycles per MFMA, tools/micro/mfma_rate.hip): one
// accumulator per chain is enough.  What does NOT overlap is a wave's own VALU / LDS work with its own MFMAs
// (tools/micro/mfma_valu.hip), and a co-resident wave's VALU work runs at half rate while the matrix pipe is busy
// (tools/micro/mfma_covalu.hip): the non-MFMA instruction count is what this kernel is tuned for.
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

// keep the loads issued above this point above it: the next chain's operands are fetched while the current chain runs
#define WAVE_FENCE() __builtin_amdgcn_sched_barrier(0)

template <bool GNN, bool EPN, bool FRONT>
__global__ __launch_bounds__(64, 2) void k_wave_forward(WaveArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x, c = lane & 31, hh = lane >> 5;
    if (!FRONT && (*A.status & EPNN_ST_PAIR_OVERFLOW)) return;
    const int b = A.order[blockIdx.x];
    const int a0 = A.moff[b], n = A.moff[b + 1] - a0;
    const int p0 = FRONT ? A.pbase[b] : A.row_off[a0];
    int np = FRONT ? 0 : A.row_off[a0 + n] - p0;
    const int nx = A.nx;
    const float *wp = A.wpack;
    const WaveIndex &X = A.wx;
    int nstamp = 0;
    (void)nstamp;
    WAVE_STAMP();
    // molecules with n <= 16 use the two 16-column halves of every MFMA tile for two copies of their atoms (lane c and
    // lane c + 16 hold the same atom): all per-atom chains replicate for free and the pair sweep handles TWO partners
    // per tile (even partners in the lower copy, odd ones in the upper), half the sweep's MFMA and VALU work
    const bool dual = n <= 16;
    const int ai = dual ? (c & 15) : c;                     // atom of this lane
    const bool hi = dual && c >= 16;                        // upper copy
    const bool catom = ai < n;
    const bool owner = catom && !hi;                        // the lane that stores the atom's rows / results

    // ---- in-kernel front-end: coordinates -> LDS (the pair slots are assigned once the LDS tables exist)
    double *xs = reinterpret_cast<double *>(sm);           // [n][3] float32 coordinates promoted like SciPy does
    if (FRONT) {
        if (hh == 0 && c < n) {
            xs[3 * c + 0] = (double)A.xyz[3 * (size_t)(a0 + c) + 0];
            xs[3 * c + 1] = (double)A.xyz[3 * (size_t)(a0 + c) + 1];
            xs[3 * c + 2] = (double)A.xyz[3 * (size_t)(a0 + c) + 2];
        }
        wave_sync_lds();
    }
    // ---- LDS layout of THIS molecule inside the wave's fixed budget
    float *Rl = sm;                                        // [n][PST]   R_j rows
    float *Pl = sm + n * EPNN_PST;                         // [n][PST]   P_i rows (EPN); the GNN keeps its pair map here
    unsigned short *pm = reinterpret_cast<unsigned short *>(Pl);      // [j][32]  near-pair slot of (i = lane, j), 0xFFFF = none
    int o = 2 * n * EPNN_PST;
    unsigned short *eij = reinterpret_cast<unsigned short *>(sm + o); // [np]  li | lj << 8
    const int eij_n = FRONT ? n * (n - 1) / 2 : np;        // in-kernel front-end: np is not known yet, reserve every i<j pair
    o += EPN ? ((((eij_n + 1) >> 1) + 3) & ~3) : 0;
    float *Dm = sm + o;                                    // [n][DST]  weighted transfers: Dm[i][j] = what i receives from j
    o += EPN ? ((n * EPNN_DST + 3) & ~3) : 0;
    float *Gl = sm + o;                                    // [glds + 1][PST]; row glds is all zeros
    const int grows = (A.lds_words - o) / EPNN_PST - 1;   // G rows the budget leaves room for
    int glds = min(np, grows);
    bool gover = np > glds;                                // some G rows live in HBM
    int ngt = (np + 31) >> 5;

    // ---- per-atom registers
    const float nmv = catom ? (A.nm_in ? A.nm_in[a0 + ai] : 1.f) : 0.f;
    float xq[EPNN_KX];
    {
        const float qv = catom ? (A.q_in ? A.q_in[a0 + ai] : A.Q[b] / (float)n) : 0.f;     // charge_gn.py:337-338
#pragma unroll
        for (int s = 0; s < EPNN_KX; ++s) {
            const int phi = 2 * s + hh;
            float v = 0.f;
            if (catom) {
                if (phi == 0) v = nmv;
                else if (phi <= nx) v = A.xin[(size_t)(a0 + ai) * nx + phi - 1];
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
            const f32x4 v = *reinterpret_cast<const f32x4 *>(A.h_in + (size_t)(a0 + ai) * EPNN_EDIM + wave_hk_f0(hh, g));
            hk[4 * g] = v[0]; hk[4 * g + 1] = v[1]; hk[4 * g + 2] = v[2]; hk[4 * g + 3] = v[3];
        }
    }

    // first G tiles: We and the first e rows are on their way while the LDS tables are built
    float gw[24], ge[24];
    EPNN_WLD(gw, GNN ? X.g[0].we : X.e[0].we, 24);
    if (!FRONT && ngt > 0) wave_load_e(A.pe, p0, np, 0, c, hh, ge);
    WAVE_FENCE();

    // ---- LDS init
    if (EPN)
        for (int i = lane; i < n * EPNN_DST; i += 64) Dm[i] = 0.f;
    if (GNN)
        for (int i = lane; i < n * 16; i += 64) reinterpret_cast<unsigned *>(pm)[i] = 0xFFFFFFFFu;
    wave_sync_lds();
    if (FRONT) {
        // ---- slots in row-major order (rows i0, i0+1 per step; the lower row's pairs first)
        int base = 0;
        for (int i0 = 0; i0 + 1 < n; i0 += 2) {
            const int i = i0 + hh;
            const bool near = c > i && c < n && wave_dist2(xs, i, c) < A.cut2;
            const unsigned long long bal = __ballot(near);
            const unsigned lo = (unsigned)bal, hi = (unsigned)(bal >> 32);
            if (near) {
                const int slot = base + (hh ? __popc(lo) : 0) + __popc((hh ? hi : lo) & ((1u << c) - 1u));
                eij[slot] = (unsigned short)(i | (c << 8));
                pm[c * 32 + i] = (unsigned short)slot;                 // e is symmetric: both directions share the entry
                pm[i * 32 + c] = (unsigned short)slot;
            }
            base += __popc(lo) + __popc(hi);
        }
        np = base;
        glds = min(np, grows);
        gover = np > glds;
        ngt = (np + 31) >> 5;
        wave_sync_lds();
        // ---- Gaussian edge features, one lane per pair (charge_gn.py:148-161: float64, then cast to float32).
        //      e_k = C exp(-eta (D - mu_k)^2) over the evenly spaced mu_k is a geometric-like sequence:
        //      e_{k+1} = e_k rho_k, rho_{k+1} = rho_k exp(-2 eta dmu^2): two exp per pair instead of 48 (float64 products;
        //      the accumulated rounding stays below 1e-12 relative, far inside the float32 cast).
        //      near flag = max_k e_k > tol (charge_gn.py:90-94), evaluated exactly at the mu closest to D.
        const double pi_d = 3.141592653589793;
        const double mu0 = A.mu[0], dmu = (A.mu[EPNN_EDIM - 1] - A.mu[0]) / (double)(EPNN_EDIM - 1);
        const double qq = exp(-2.0 * A.eta * dmu * dmu);
        for (int s0 = 0; s0 < np; s0 += 64) {
            if (s0 + lane < np) {
                const int ij = eij[s0 + lane];
                const double D = wave_dist(xs, ij & 0xFF, ij >> 8);
                double C = (cos(pi_d * (D - 0.0) / A.cutoff) + 1.0) / 2.0;
                if (D <= 0.0) C = 1.0;
                int kb = min(EPNN_EDIM - 1, max(0, (int)((D - mu0) / dmu + 0.5)));
                double best = 1e300;
                int kbest = kb;
                for (int k = max(0, kb - 1); k <= min(EPNN_EDIM - 1, kb + 1); ++k) {
                    const double d = D - A.mu[k];
                    if (d * d < best) { best = d * d; kbest = k; }
                }
                const double db = D - A.mu[kbest];
                const float emax = (float)(C * exp(-A.eta * (db * db)));
                const float w = emax > A.tol ? 1.0f : 0.0f;
                A.pwi[p0 + s0 + lane] = w;
                A.pwj[p0 + s0 + lane] = w;
                const double t0 = D - mu0;
                double e = C * exp(-A.eta * (t0 * t0));
                double rho = exp(A.eta * dmu * (2.0 * t0 - dmu));
                float *erow = A.pe + (size_t)(p0 + s0 + lane) * EPNN_EDIM;
#pragma unroll 1
                for (int k4 = 0; k4 < EPNN_EDIM; k4 += 4) {
                    f32x4 v;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        v[u] = (float)e;
                        e *= rho;
                        rho *= qq;
                    }
                    *reinterpret_cast<f32x4 *>(erow + k4) = v;
                }
            }
        }
        wave_sync_all();
        if (ngt > 0) wave_load_e(A.pe, p0, np, 0, c, hh, ge);
    } else {
        if (GNN)
            for (int p = lane; p < np; p += 64) {
                const int li = A.pi[p0 + p] - a0, lj = A.pj[p0 + p] - a0;
                pm[lj * 32 + li] = (unsigned short)p;                   // message into i = li from j = lj
                if (A.psym[p0 + p]) pm[li * 32 + lj] = (unsigned short)p;
            }
        if (EPN)
            for (int p = lane; p < np; p += 64) {
                const int li = A.pi[p0 + p] - a0, lj = A.pj[p0 + p] - a0;
                eij[p] = (unsigned short)(li | (lj << 8));
            }
    }
    // the G zero row and the transfer matrix last: the front-end used those words as scratch
    for (int i = lane; i < EPNN_PST; i += 64) Gl[glds * EPNN_PST + i] = 0.f;
    wave_sync_lds();

    WAVE_STAMP();   // init done
    const float Nf = (float)A.N, padw = (float)(A.N - n);
    const int Tg = GNN ? A.T : 0, Te = EPN ? A.T : 0;

    // G rows of every near pair for the pair MLP whose We is in gw[] (first e rows in ge[]); rows >= glds go to HBM
    auto gtiles = [&]() {
#pragma unroll 1
        for (int gt = 0; gt < ngt; ++gt) {
            float en[24];
            wave_load_e(A.pe, p0, np, min(gt + 1, ngt - 1), c, hh, en);
            WAVE_FENCE();
            f32x16 acc = wave_chain<24>(gw, ge, epnn_splat16(0.f));
            const int slot = gt * 32 + c;
            // two separate predicated stores: merged into one `cond ? lds : global` pointer they become flat stores,
            // whose completion wait also waits for the e rows fetched ahead
            if (slot < min(np, glds)) epnn_st16(Gl + slot * EPNN_PST + hh * 16, acc);
            if (gover) {
                asm volatile("" ::: "memory");
                if (slot >= glds && slot < np) epnn_st16(A.gx + (size_t)(p0 + slot) * 32 + hh * 16, acc);
            }
#pragma unroll
            for (int s = 0; s < 24; ++s) ge[s] = en[s];
        }
    };
    // start fetching what the NEXT gtiles() needs
    auto gprefetch = [&](int weoff) {
        EPNN_WLD(gw, weoff, 24);
        if (ngt > 0) wave_load_e(A.pe, p0, np, 0, c, hh, ge);
    };

    // ================================================================== GNN steps (charge_gn.py:60-74)
    if (GNN) {
        float P[16], u1pre[16], pb[16], b2k[16], bn[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bn[r] = 0.f;
        // ---- step 0: G rows, then P / R / u1pre from (xq | hk)
        {
            float wa[EPNN_KX], wb[24], wc[EPNN_KX], wd[24];
            EPNN_WLD(wa, X.wi0, EPNN_KX);
            if (have_h) { EPNN_WLD(wb, X.wi0 + EPNN_KX * 64, 24); }
            WAVE_FENCE();
            gtiles();
            EPNN_WLD(wc, X.wj0, EPNN_KX);
            if (have_h) { EPNN_WLD(wd, X.wj0 + EPNN_KX * 64, 24); }
            WAVE_FENCE();
            f32x16 acc = wave_chain<EPNN_KX>(wa, xq, epnn_splat16(0.f));
            if (have_h) acc = wave_chain<24>(wb, hk, acc);
            f32x16 acr = wave_chain<EPNN_KX>(wc, xq, epnn_splat16(0.f));
            if (have_h) acr = wave_chain<24>(wd, hk, acr);
#pragma unroll
            for (int r = 0; r < 16; ++r) P[r] = acc[r];
            if (owner) epnn_st16(Rl + c * EPNN_PST + hh * 16, acr);
            EPNN_WLD(pb, X.g[0].w2, 16);
            epnn_ld16(wp + X.g[0].b2k + hh * 16, b2k);
            if (have_h) { EPNN_WLD(wb, X.u1h0, 24); }
            WAVE_FENCE();
            acc = epnn_splat16(0.f);
            if (have_h) {
                float hm[24];
#pragma unroll
                for (int s = 0; s < 24; ++s) hm[s] = nmv * hk[s];       // masked_input = [h, m] * node_mask (charge_gn.py:72)
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
            float S[16], u1s[16], w2[16], cv[16], bv[16];
            {
                f32x16 cb2;
#pragma unroll
                for (int r = 0; r < 16; ++r) { cb2[r] = b2k[r]; S[r] = 0.f; }
                // ---- partner tiles j = 0..n-1, operands of tile j+1 (and the slot of j+2) fetched during tile j.
                //      Two instances: every G row in LDS (the usual case) / some rows in HBM (both sources read, one is 0)
                auto sweep = [&](auto over_tag) {
                    constexpr bool OVER = decltype(over_tag)::value;
                    auto gload = [&](int slot, float (&g)[16], float (&gh)[16]) {
                        epnn_ld16(Gl + min(slot, glds) * EPNN_PST + hh * 16, g);          // 0xFFFF / overflow -> the zero row
                        if (OVER) {
                            if (slot >= glds && slot != 0xFFFF) epnn_ld16(A.gx + (size_t)(p0 + slot) * 32 + hh * 16, gh);
                            else {
#pragma unroll
                                for (int s = 0; s < 16; ++s) gh[s] = 0.f;
                            }
                        }
                    };
                    // partner index jp in [0, n]: atom jp, or (jp == n) the reference's zero-padded partner (R = 0, G = 0,
                    // charge_gn.py:70) which counts N - n times; beyond n: nothing (weight 0).  Tile t holds partner t of
                    // every atom, or (dual) partners 2t / 2t+1 in the lower / upper copy.  Operands of tile t+1 (and the
                    // slot of t+2) are fetched while tile t is in the matrix pipe.
                    const float *zrow = Gl + glds * EPNN_PST + hh * 16;
                    const int nt = dual ? (n + 2) >> 1 : n + 1;
                    auto partner = [&](int t) -> int { return dual ? 2 * t + (hi ? 1 : 0) : t; };
                    auto rload = [&](int t, float (&r)[16]) {
                        const int jp = partner(t);
                        epnn_ld16(jp < n ? Rl + jp * EPNN_PST + hh * 16 : zrow, r);
                    };
                    auto slot_of = [&](int t) -> int {
                        const int jp = partner(t);
                        return jp < n ? (int)pm[jp * 32 + ai] : 0xFFFF;
                    };
                    auto weight = [&](int t) -> float {
                        const int jp = partner(t);
                        return jp < n ? 1.f : (jp == n ? padw : 0.f);
                    };
                    auto tile = [&](const float (&rj)[16], const float (&g)[16], const float (&gh)[16], float wt) {
                        f32x16 acc = cb2;
#pragma unroll
                        for (int s = 0; s < 16; ++s) {
                            float z = (P[s] + rj[s]) + g[s];
                            if (OVER) z += gh[s];
                            acc = epnn_mfma(pb[s], fmaxf(z, 0.f), acc);
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) S[r] = fmaf(wt, fmaxf(acc[r], 0.f), S[r]);
                    };
                    if (OVER) {          // rare (large or very dense molecules): no fetch-ahead, fewer registers
#pragma unroll 1
                        for (int t = 0; t < nt; ++t) {
                            if (t == nt - 1) { EPNN_WLD(u1s, M.u1s, 16); }    // first operand of the update MLP
                            float rj[16], g[16], gh[16];
                            rload(t, rj);
                            gload(slot_of(t), g, gh);
                            tile(rj, g, gh, weight(t));
                        }
                    } else {
                        float rA[16], gA[16], rB[16], gB[16];
                        rload(0, rA);
                        gload(slot_of(0), gA, gA);
                        int snext = slot_of(1);
                        int t = 0;
#pragma unroll 1
                        for (; t + 2 < nt; t += 2) {                          // tiles t, t+1; neither is the last one
                            rload(t + 1, rB);
                            gload(snext, gB, gB);
                            snext = slot_of(t + 2);
                            WAVE_FENCE();
                            tile(rA, gA, gA, weight(t));
                            rload(t + 2, rA);
                            gload(snext, gA, gA);
                            snext = slot_of(t + 3);
                            WAVE_FENCE();
                            tile(rB, gB, gB, weight(t + 1));
                        }
                        EPNN_WLD(u1s, M.u1s, 16);                             // first operand of the update MLP
                        if (t + 1 < nt) {                                      // two tiles left
                            rload(t + 1, rB);
                            gload(snext, gB, gB);
                            WAVE_FENCE();
                            tile(rA, gA, gA, weight(t));
                            tile(rB, gB, gB, weight(t + 1));
                        } else {
                            WAVE_FENCE();
                            tile(rA, gA, gA, weight(t));
                        }
                    }
                    if (dual) {          // the two copies of an atom hold the even / odd partners: add them, both copies end up complete
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float other = __shfl_xor(S[r], 16, 64);       // all lanes take part in the exchange
                            S[r] = hi ? other + S[r] : S[r] + other;           // lower + upper: same order in both copies
                        }
                    }
                };
                if (gover) sweep(std::true_type{});
                else sweep(std::false_type{});
            }
            if (t < 2) WAVE_STAMP();   // pair tiles
            // ---- update MLP (charge_gn.py:71-74); the last message Dense is folded into u1s
            {
                EPNN_WLD(w2, M.u2, 16);
                epnn_ld16(wp + M.cb3k + hh * 16, cv);
                epnn_ld16(wp + M.bu1k + hh * 16, bv);
                WAVE_FENCE();
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = u1pre[r];
                acc = wave_chain<16>(u1s, S, acc);
                float u1[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) u1[r] = fmaxf(fmaf(nmv, fmaf(Nf, cv[r], acc[r]), bv[r]), 0.f);
                epnn_ld16(wp + M.bu2k + hh * 16, bv);
                if (!lastg) gprefetch(X.g[t + 1].we);
                else if (Te > 0) gprefetch(X.e[0].we);
                WAVE_FENCE();
                acc = wave_chain<16>(w2, u1, epnn_splat16(0.f));
#pragma unroll
                for (int r = 0; r < 16; ++r) bn[r] = fmaxf(acc[r] + bv[r], 0.f);
            }
            if (t < 2) WAVE_STAMP();   // U1, U2
            if (!lastg) {
                // next step: G rows, then P / R / u1pre from (nm*u2 | xq) through the folded matrices
                float wa[16 + EPNN_KX], wb[16 + EPNN_KX];
                EPNN_WLD(wa, M.pwi, 16 + EPNN_KX);
                WAVE_FENCE();
                gtiles();
                if (t < 2) WAVE_STAMP();   // G tiles
#pragma unroll
                for (int r = 0; r < 16; ++r) bn[r] *= nmv;
                float in[16 + EPNN_KX];
#pragma unroll
                for (int s = 0; s < 16; ++s) in[s] = bn[s];
#pragma unroll
                for (int s = 0; s < EPNN_KX; ++s) in[16 + s] = xq[s];
                EPNN_WLD(wb, M.pwj, 16 + EPNN_KX);
                float cu[16], wu[16];
                EPNN_WLD(wu, M.pu1, 16);
                epnn_ld16(wp + M.cu3k + hh * 16, cu);
                WAVE_FENCE();
                f32x16 acc = wave_chain<16 + EPNN_KX>(wa, in, epnn_splat16(0.f));
                f32x16 acr = wave_chain<16 + EPNN_KX>(wb, in, epnn_splat16(0.f));
#pragma unroll
                for (int r = 0; r < 16; ++r) P[r] = acc[r];
                if (owner) epnn_st16(Rl + c * EPNN_PST + hh * 16, acr);
                EPNN_WLD(pb, X.g[t + 1].w2, 16);
                epnn_ld16(wp + X.g[t + 1].b2k + hh * 16, b2k);
                WAVE_FENCE();
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = nmv * cu[r];
                acc = wave_chain<16>(wu, bn, acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) u1pre[r] = acc[r];
                wave_sync_all();
                if (t < 2) WAVE_STAMP();   // projections
            }
        }
        {
            // h = node_mask * (Wu3^T u2 + bu3)  (charge_gn.py:73-74) after the last step, straight into the hk registers
            float w[16], w2[16], bv[16], bw[16];
            EPNN_WLD(w, X.u3, 16);
            EPNN_WLD(w2, X.u3 + 16 * 64, 16);
            epnn_ld16(wp + X.bu3k + hh * 16, bv);
            epnn_ld16(wp + X.bu3k + 32 + hh * 16, bw);
            WAVE_FENCE();
            f32x16 acc = wave_chain<16>(w, bn, epnn_splat16(0.f));
            f32x16 ac2 = wave_chain<16>(w2, bn, epnn_splat16(0.f));
#pragma unroll
            for (int r = 0; r < 16; ++r) hk[r] = nmv * (acc[r] + bv[r]);
#pragma unroll
            for (int r = 0; r < 8; ++r) hk[16 + r] = nmv * (ac2[r] + bw[r]);
        }
        if (A.h_out && owner) {
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
        const int qs = (nx + 1) >> 1, qh = (nx + 1) & 1;    // register / half-wave of xq that holds q
#pragma unroll 1
        for (int t = 0; t < Te; ++t) {
            const WaveEpnPack &M = X.e[t];
            float in[EPNN_KX + 24];
#pragma unroll
            for (int s = 0; s < EPNN_KX; ++s) in[s] = xq[s];
#pragma unroll
            for (int s = 0; s < 24; ++s) in[EPNN_KX + s] = hk[s];
            {
                float wa[EPNN_KX + 24], wb[EPNN_KX + 24];
                EPNN_WLD(wa, M.wi, EPNN_KX + 24);
                WAVE_FENCE();
                gtiles();
                if (t < 2) WAVE_STAMP();   // EPN G tiles
                EPNN_WLD(wb, M.wj, EPNN_KX + 24);
                WAVE_FENCE();
                f32x16 acc = wave_chain<EPNN_KX + 24>(wa, in, epnn_splat16(0.f));
                f32x16 acr = wave_chain<EPNN_KX + 24>(wb, in, epnn_splat16(0.f));
                if (owner) epnn_st16(Pl + c * EPNN_PST + hh * 16, acc);
                if (owner) epnn_st16(Rl + c * EPNN_PST + hh * 16, acr);
            }
            float pb[16], b2v[16], w3[16];
            EPNN_WLD(pb, M.w2, 16);
            epnn_ld16(wp + M.b2k + hh * 16, b2v);
            epnn_ld16(wp + M.w3k + hh * 16, w3);
            wave_sync_all();
            if (t < 2) WAVE_STAMP();   // EPN P, R
            {
                // pair record of the next tile (indices in LDS, weights in HBM) fetched one tile ahead
                int ij_n = eij[c < np ? c : 0];
                float wi_n = A.pwi[p0 + (c < np ? c : 0)], wj_n = A.pwj[p0 + (c < np ? c : 0)];
#pragma unroll 1
                for (int gt = 0; gt < ngt; ++gt) {
                    const int slot = gt * 32 + c;
                    const bool valid = slot < np;
                    const int sl = valid ? slot : 0;
                    const int ij = ij_n;
                    const float wi = wi_n, wj = wj_n;
                    {
                        const int sn = slot + 32 < np ? slot + 32 : 0;
                        ij_n = eij[sn];
                        wi_n = A.pwi[p0 + sn];
                        wj_n = A.pwj[p0 + sn];
                    }
                    const int li = ij & 0xFF, lj = ij >> 8;
                    float g[16], pi_[16], rj_[16], pj_[16], ri_[16];
                    if (sl < glds) epnn_ld16(Gl + sl * EPNN_PST + hh * 16, g);
                    else epnn_ld16(A.gx + (size_t)(p0 + sl) * 32 + hh * 16, g);
                    epnn_ld16(Pl + li * EPNN_PST + hh * 16, pi_);
                    epnn_ld16(Rl + lj * EPNN_PST + hh * 16, rj_);
                    epnn_ld16(Pl + lj * EPNN_PST + hh * 16, pj_);
                    epnn_ld16(Rl + li * EPNN_PST + hh * 16, ri_);
                    // the two directions of the pair are independent chains: interleaved (rows = out feature, col = pair)
                    f32x16 au, av;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { au[r] = b2v[r]; av[r] = b2v[r]; }
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        au = epnn_mfma(pb[s], fmaxf((g[s] + pi_[s]) + rj_[s], 0.f), au);
                        av = epnn_mfma(pb[s], fmaxf((g[s] + pj_[s]) + ri_[s], 0.f), av);
                    }
                    float fu = 0.f, fv = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        fu = fmaf(w3[r], fmaxf(au[r], 0.f), fu);
                        fv = fmaf(w3[r], fmaxf(av[r], 0.f), fv);
                    }
                    fu += epnn_swap32(fu);
                    fv += epnn_swap32(fv);
                    const float d = 0.5f * (fu - fv);                  // charge_gn.py:116
                    // entries with weight 0 are never written (they stay 0): a one-sided entry (j,i) of the dense
                    // front-end must not clear what the entry (i,j) wrote
                    if (hh == 0 && valid && wi != 0.f) Dm[li * EPNN_DST + lj] = wi * d;
                    if (hh == 1 && valid && wj != 0.f) Dm[lj * EPNN_DST + li] = -(wj * d);
                }
            }
            wave_sync_lds();
            if (t + 1 < Te) gprefetch(X.e[t + 1].we);     // on its way during the charge update
            WAVE_FENCE();
            if (t < 2) WAVE_STAMP();   // EPN pair tiles
            // q_i += sum_j antisym_ij (charge_gn.py:118): lane (i, hh) adds row i of the transfer matrix, columns j = hh mod 2
            {
                float dq0 = 0.f, dq1 = 0.f;
                const float *drow = Dm + (catom ? ai : 0) * EPNN_DST + hh;
#pragma unroll 4
                for (int j = 0; j + hh < n; j += 4) {
                    dq0 += drow[j];
                    dq1 += (j + 2 + hh < n) ? drow[j + 2] : 0.f;
                }
                float dq = dq0 + dq1;
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
        if (hh == 0 && owner) A.q_out[a0 + c] = qout;
    }
    if (FRONT && lane == 0) {
        // No memset before and no copy after the launch: the last wave to finish hands status + pair count to the host
        // and leaves the three control words zeroed for the next forward of this handle.
        atomicAdd(A.status + 1, np);
        __threadfence();
        if (atomicAdd(A.status + 2, 1) == (int)gridDim.x - 1) {
            __threadfence();
            const int st = atomicExch(A.status + 0, 0), cnt = atomicExch(A.status + 1, 0);
            atomicExch(A.status + 2, 0);
            volatile int *hs = A.host_status;
            hs[0] = st;
            hs[1] = cnt;
            __threadfence_system();
        }
    }
    WAVE_STAMP();
#ifdef EPNN_STAMPS
    if (lane == 0 && A.stamps) {
        A.stamps[(size_t)blockIdx.x * 64 + 62] = (unsigned long long)nstamp;
        A.stamps[(size_t)blockIdx.x * 64 + 63] = ((unsigned long long)n << 32) | (unsigned)np;
    }
#endif
    (void)Tg;
}
