// Three-column-block variant of the wave-autonomous fused forward (epnn_wave.hip.h) for molecules of 33..48 atoms.
//
// The reference's own data set (`mixed`) has systems of up to 41 atoms; the two-block kernel stops at 32 and the tiled
// kernels (epnn_large.hip.h) spend ~40 launches on a batch's handful of 33..41-atom systems -- 4 % of the atoms, two thirds
// of the forward's time.  Same arithmetic, same operand layouts and the same packed weights as the
// two-block kernel; what differs:
//   * three column blocks per lane: blocks 0 and 1 hold atoms 0..31 (always full here), block 2 the m = n - 32 atoms
//     beyond them in C = 16 / m copies (the partner-copy trick of the two-block kernel's block 1);
//   * ~280 registers: one wavefront per SIMD (it is launched behind the main kernel for the few molecules that need it);
//   * only the compact entry's configuration: in-kernel front-end, both stacks, h = 0 on entry (charge_gn.py:334), the
//     EPN through the folded matrices.  Pair lists from outside and layer-level calls keep the tiled path for n > 32;
//   * tables for up to 48 atoms: pair map row stride 48, transfer matrix row stride 49, the front-end assigns the pair
//     slots one row of the molecule per step (64 lanes = partners 0..63); a 40 KB LDS budget, G rows beyond it in HBM.
// Results agree with the two-block kernel's arithmetic to float32 rounding (tests: sizes 33..48 vs the float64 oracle; the
// reference's stored outputs for its 33..38-atom validation systems).
#pragma once
#include "epnn_wave.hip.h"

#define EPNN_W3_NMAX 48
#define EPNN_W3_PMS 48       // pair map: row stride (u16 entries)
#define EPNN_W3_DST 49       // transfer matrix: row stride (floats)

__global__ __launch_bounds__(64, 1) void k_wave_forward3(WaveArgs A, WaveIndex X) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int KE = EPNN_ER / 4;
    const int lane = threadIdx.x, q = lane >> 4, n16 = lane & 15;
    const int4 wb = A.wblk[blockIdx.x];
    const int b = wb.x, a0 = wb.y, n = wb.z, p0 = wb.w;
    const int nx = A.nx;
    const bool xs3 = nx + 3 <= 4 * (EPNN_XS - 1);
    const float *wp = A.wpack;
    const int fo = 4 * q;
    // block 2: m atoms, C copies of each (column n16 = atom 32 + n16 % m, copy n16 / m)
    const int m2 = n - 32, C2 = 16 / m2;
    const int copy2 = n16 / m2, col2 = 32 + n16 % m2;
    const bool cat2 = copy2 < C2, own2 = cat2 && copy2 == 0;
    const int ai[3] = {n16, 16 + n16, col2};                 // atom of this lane's column in each block
    const bool cat[3] = {true, true, cat2}, own[3] = {true, true, own2};

    // ---- LDS layout (floats): eij | R [n][PST] | GNN: pair map [n][48] u16, G rows, zero row / EPN: P [n][PST], Dm [n][49]
    unsigned short *eij = reinterpret_cast<unsigned short *>(sm);
    const int eij_n = n * (n - 1) / 2;
    const int o_r = (((eij_n + 1) >> 1) + 3) & ~3;
    float *Rl = sm + o_r;
    const int o_x = o_r + n * EPNN_PST;
    unsigned short *pm = reinterpret_cast<unsigned short *>(sm + o_x);
    float *Pl = sm + o_x;
    float *Dm = sm + o_x + n * EPNN_PST;
    const int o_gg = o_x + ((n * (EPNN_W3_PMS / 2) + 3) & ~3);
    const int grows_g = (A.lds_words - o_gg) / EPNN_PST - 1;
    float *Gl = sm + o_gg;

    // ---- front-end: coordinates -> LDS, pair slots in row-major order (one row per step), edge coordinates of every pair
    double *xs = reinterpret_cast<double *>(Rl);
    if (lane < n) {
        xs[3 * lane + 0] = (double)A.xyz[3 * (size_t)(a0 + lane) + 0];
        xs[3 * lane + 1] = (double)A.xyz[3 * (size_t)(a0 + lane) + 1];
        xs[3 * lane + 2] = (double)A.xyz[3 * (size_t)(a0 + lane) + 2];
    }
    float xq[3][EPNN_XS];
    {
        const float qv = A.Q[b] / (float)n;                 // charge_gn.py:337-338
#pragma unroll
        for (int cb = 0; cb < 3; ++cb)
#pragma unroll
            for (int s = 0; s < EPNN_XS; ++s) {
                const int phi = 4 * s + q;
                float v = 0.f;
                if (cat[cb]) {
                    if (phi == 0) v = 1.f;                   // node mask
                    else if (phi <= nx) v = A.xin[(size_t)(a0 + ai[cb]) * nx + phi - 1];
                    else if (phi == nx + 1) v = qv;
                    else if (phi == nx + 2) v = 1.f;
                }
                xq[cb][s] = v;
            }
    }
    const float nm[3] = {1.f, 1.f, cat2 ? 1.f : 0.f};
    float gw[2][KE], ge0[KE], ge1[KE];
    W16_LD(gw, X.g[0].we16, 2, KE);
    for (int i = lane; i < n * (EPNN_W3_PMS / 2); i += 64) reinterpret_cast<unsigned *>(pm)[i] = 0xFFFFFFFFu;
    wave_sync_lds();
    int np = 0;
    for (int i = 0; i + 1 < n; ++i) {
        const bool near = lane > i && lane < n && wave_dist2(xs, i, lane) < A.cut2;
        const unsigned long long bal = __ballot(near);
        if (near) {
            const int slot = np + __popcll(bal & ((1ull << lane) - 1ull));
            eij[slot] = (unsigned short)(i | (lane << 8));
            pm[lane * EPNN_W3_PMS + i] = (unsigned short)slot;       // e is symmetric: both directions share the entry
            pm[i * EPNN_W3_PMS + lane] = (unsigned short)slot;
        }
        np += __popcll(bal);
    }
    const int glds = min(np, grows_g);
    const bool gover = np > glds;
    const int ngt = (np + 31) >> 5;
    wave_sync_lds();
    {
        const double pi_d = 3.141592653589793;
        const double mu0 = A.mu[0], dmu = (A.mu[EPNN_EDIM - 1] - A.mu[0]) / (double)(EPNN_EDIM - 1);
        for (int s0 = 0; s0 < np; s0 += 64) {
            if (s0 + lane < np) {
                const int ij = eij[s0 + lane];
                const double D = wave_dist(xs, ij & 0xFF, ij >> 8);
                float w = 1.0f;
                if (D > A.dsafe) {                              // is_near (charge_gn.py:90-94) evaluated exactly near the cutoff
                    double Cc = (cos(pi_d * (D - 0.0) / A.cutoff) + 1.0) / 2.0;
                    const int kb = min(EPNN_EDIM - 1, max(0, (int)((D - mu0) / dmu + 0.5)));
                    double best = 1e300;
                    for (int k = max(0, kb - 1); k <= min(EPNN_EDIM - 1, kb + 1); ++k) {
                        const double d = D - A.mu[k];
                        best = d * d < best ? d * d : best;
                    }
                    w = (float)(Cc * exp(-A.eta * best)) > A.tol ? 1.0f : 0.0f;
                }
                A.pwi[p0 + s0 + lane] = w;
                A.pwj[p0 + s0 + lane] = w;
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
    }
    wave_sync_all();
    auto load_e1 = [&](int slot, float (&e)[KE]) {
        const int sl = slot < np ? slot : 0;
        const f32x4 v = w16_ld(A.pt + (size_t)(p0 + sl) * EPNN_ER + 4 * q);
        e[0] = v[0]; e[1] = v[1]; e[2] = v[2]; e[3] = v[3];
    };
    auto load_e = [&](int gt, float (&e0)[KE], float (&e1)[KE]) {
        load_e1(gt * 32 + n16, e0);
        load_e1(gt * 32 + 16 + n16, e1);
    };
    if (ngt > 0) load_e(0, ge0, ge1);
    for (int i = lane; i < EPNN_PST; i += 64) Gl[glds * EPNN_PST + i] = 0.f;        // the sweep's zero row
    wave_sync_lds();

    const float Nf = (float)A.N, padw = (float)(A.N - n);
    const int T = A.T;
    // G rows of every pair for the pair MLP whose We is in gw; rows >= glds go to HBM
    auto gtile = [&](int gt, const float (&e0)[KE], const float (&e1)[KE]) {
        const int s0 = gt * 32 + n16, s1 = s0 + 16;
        f32x4 d0[2] = {w16_splat(0.f), w16_splat(0.f)};
        w16_mm<2, KE>(gw, e0, d0);
        if (s0 < glds) { w16_st(Gl + s0 * EPNN_PST + fo, d0[0]); w16_st(Gl + s0 * EPNN_PST + 16 + fo, d0[1]); }
        if (gover) {
            asm volatile("" ::: "memory");
            if (s0 >= glds && s0 < np) { w16_st(A.gx + (size_t)(p0 + s0) * 32 + fo, d0[0]); w16_st(A.gx + (size_t)(p0 + s0) * 32 + 16 + fo, d0[1]); }
        }
        if (gt * 32 + 16 < np) {
            f32x4 d1[2] = {w16_splat(0.f), w16_splat(0.f)};
            w16_mm<2, KE>(gw, e1, d1);
            if (s1 < glds) { w16_st(Gl + s1 * EPNN_PST + fo, d1[0]); w16_st(Gl + s1 * EPNN_PST + 16 + fo, d1[1]); }
            if (gover) {
                asm volatile("" ::: "memory");
                if (s1 >= glds && s1 < np) { w16_st(A.gx + (size_t)(p0 + s1) * 32 + fo, d1[0]); w16_st(A.gx + (size_t)(p0 + s1) * 32 + 16 + fo, d1[1]); }
            }
        }
    };
    auto gtiles = [&]() {
        float en0[KE], en1[KE];
        int gt = 0;
#pragma unroll 1
        for (; gt + 1 < ngt; gt += 2) {
            load_e(gt + 1, en0, en1);
            WAVE_FENCE();
            gtile(gt, ge0, ge1);
            load_e(min(gt + 2, ngt - 1), ge0, ge1);
            WAVE_FENCE();
            gtile(gt + 1, en0, en1);
        }
        if (gt < ngt) gtile(gt, ge0, ge1);
    };
    auto vec2 = [&](int off, f32x4 (&v)[2]) {
        v[0] = w16_ld(wp + off + fo);
        v[1] = w16_ld(wp + off + 16 + fo);
    };
    auto store_rows = [&](float *rows, const f32x4 (&v)[3][2]) {
#pragma unroll
        for (int cb = 0; cb < 3; ++cb)
            if (own[cb]) { w16_st(rows + ai[cb] * EPNN_PST + fo, v[cb][0]); w16_st(rows + ai[cb] * EPNN_PST + 16 + fo, v[cb][1]); }
    };

    f32x4 B[3][2], P[3][2], U[3][2];
    float pb[2][8];
    f32x4 b2v[2];
    // ================================================================== GNN (charge_gn.py:60-74)
    {   // step 0: G rows, then P / R from xq (h = 0: no h terms, u1pre = 0)
        float wa[2][EPNN_XS], wc[2][EPNN_XS];
        W16_LDX(wa, X.wi0, 2, EPNN_XS, EPNN_XS + 12, 0);
        W16_LDX(wc, X.wj0, 2, EPNN_XS, EPNN_XS + 12, 0);
        WAVE_FENCE();
        gtiles();
        W16_LD(pb, X.g[0].w2, 2, 8);
        vec2(X.g[0].b2, b2v);
        WAVE_FENCE();
        f32x4 r[3][2];
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) { P[cb][rb] = w16_splat(0.f); r[cb][rb] = w16_splat(0.f); U[cb][rb] = w16_splat(0.f); }
            w16_mm_skip<2, EPNN_XS, EPNN_XS - 1>(wa, xq[cb], P[cb], xs3);
            w16_mm_skip<2, EPNN_XS, EPNN_XS - 1>(wc, xq[cb], r[cb], xs3);
        }
        store_rows(Rl, r);
    }
    wave_sync_all();
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const WaveGnnPack &M = X.g[t];
        const bool lastg = t + 1 == T;
        f32x4 S[3][2];
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) { S[cb][0] = w16_splat(0.f); S[cb][1] = w16_splat(0.f); }
        float u1s[2][8];
        {
            const float *zrow = Gl + glds * EPNN_PST;
            auto grow = [&](int sl, f32x4 (&g)[2]) {
                const float *gp = Gl + min(sl, glds) * EPNN_PST;             // 0xFFFF / overflow -> the zero row
                g[0] = w16_ld(gp + fo);
                g[1] = w16_ld(gp + 16 + fo);
                if (gover && sl >= glds && sl != 0xFFFF) {
                    g[0] = w16_ld(A.gx + (size_t)(p0 + sl) * 32 + fo);
                    g[1] = w16_ld(A.gx + (size_t)(p0 + sl) * 32 + 16 + fo);
                }
            };
            auto tile = [&](const f32x4 (&Pc)[2], f32x4 (&Sc)[2], const f32x4 (&r)[2], const f32x4 (&g)[2], float w) {
                const f32x4 za = w16_relu((Pc[0] + r[0]) + g[0]), zb = w16_relu((Pc[1] + r[1]) + g[1]);
                const float z[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
                f32x4 d[2] = {b2v[0], b2v[1]};
                w16_mm<2, 8>(pb, z, d);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) Sc[rb] += w * w16_relu(d[rb]);
            };
            // blocks 0 and 1: partner jp of every atom, jp = 0 .. n (jp == n: the reference's zero-padded partner, N - n times)
            {
                struct Ops2 { f32x4 r[2], g0[2], g1[2]; };
                auto load2 = [&](int jp, Ops2 &o_) {
                    const bool real = jp < n;
                    const float *rrow = real ? Rl + jp * EPNN_PST : zrow;
                    o_.r[0] = w16_ld(rrow + fo);
                    o_.r[1] = w16_ld(rrow + 16 + fo);
                    grow(real ? (int)pm[jp * EPNN_W3_PMS + n16] : 0xFFFF, o_.g0);
                    grow(real ? (int)pm[jp * EPNN_W3_PMS + 16 + n16] : 0xFFFF, o_.g1);
                };
                Ops2 oa, ob;
                load2(0, oa);
                int jp = 0;
#pragma unroll 1
                for (; jp + 2 <= n; jp += 2) {                                 // tiles jp, jp + 1; the last one (jp == n) follows
                    load2(jp + 1, ob);
                    WAVE_FENCE();
                    tile(P[0], S[0], oa.r, oa.g0, 1.f);
                    tile(P[1], S[1], oa.r, oa.g1, 1.f);
                    load2(jp + 2, oa);
                    WAVE_FENCE();
                    tile(P[0], S[0], ob.r, ob.g0, 1.f);
                    tile(P[1], S[1], ob.r, ob.g1, 1.f);
                }
                for (; jp <= n; ++jp) {                                        // one or two tiles left (oa holds tile jp)
                    const float w = jp < n ? 1.f : padw;
                    if (jp < n) load2(jp + 1, ob);
                    WAVE_FENCE();
                    tile(P[0], S[0], oa.r, oa.g0, w);
                    tile(P[1], S[1], oa.r, oa.g1, w);
                    oa = ob;
                }
            }
            // block 2: copy k of an atom takes partners k, k + C, ...: done after (n + C) / C tiles
            {
                struct Ops { f32x4 r[2], g[2]; float w; };
                auto load1 = [&](int t2, Ops &o_) {
                    const int jp = t2 * C2 + copy2;
                    const bool real = jp < n && cat2;
                    const float *rrow = real ? Rl + jp * EPNN_PST : zrow;
                    o_.r[0] = w16_ld(rrow + fo);
                    o_.r[1] = w16_ld(rrow + 16 + fo);
                    grow(real ? (int)pm[jp * EPNN_W3_PMS + col2] : 0xFFFF, o_.g);
                    o_.w = !cat2 ? 0.f : (jp < n ? 1.f : (jp == n ? padw : 0.f));
                };
                const int nt2 = (n + C2) / C2;
                Ops oa, ob;
                load1(0, oa);
#pragma unroll 1
                for (int t2 = 0; t2 < nt2; ++t2) {
                    if (t2 + 1 < nt2) load1(t2 + 1, ob);
                    WAVE_FENCE();
                    tile(P[2], S[2], oa.r, oa.g, oa.w);
                    oa = ob;
                }
            }
            W16_LD(u1s, M.u1s, 2, 8);
            if (C2 > 1) {                                                      // add the copies' sums in a fixed order
                wave_sync_lds();
                float *scr = Gl;
                w16_st(scr + (n16 * 4 + q) * 8, S[2][0]);
                w16_st(scr + (n16 * 4 + q) * 8 + 4, S[2][1]);
                wave_sync_lds();
                f32x4 t0_ = w16_splat(0.f), t1_ = w16_splat(0.f);
                for (int k = 0; k < C2; ++k) {
                    const int src = (n16 % m2) + m2 * k;
                    t0_ += w16_ld(scr + (src * 4 + q) * 8);
                    t1_ += w16_ld(scr + (src * 4 + q) * 8 + 4);
                }
                S[2][0] = t0_;
                S[2][1] = t1_;
                wave_sync_lds();
                if (lane < EPNN_PST) Gl[glds * EPNN_PST + lane] = 0.f;          // the scratch may have covered the zero row
            }
        }
        // ---- update MLP (charge_gn.py:71-74); the last message Dense is folded into u1s
        {
            float w2[2][8];
            f32x4 cv[2], bv[2], bv2[2];
            W16_LD(w2, M.u2, 2, 8);
            vec2(M.cb3, cv);
            vec2(M.bu1, bv);
            vec2(M.bu2, bv2);
            if (!lastg) { W16_LD(gw, X.g[t + 1].we16, 2, KE); if (ngt > 0) load_e(0, ge0, ge1); }
            else { W16_LD(gw, X.e[0].we16, 2, KE); }
            WAVE_FENCE();
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                float in[8];
                f32x4 d[2] = {U[cb][0], U[cb][1]};
                w16_feed(S[cb], in);
                w16_mm<2, 8>(u1s, in, d);
                f32x4 a_[2];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) a_[rb] = w16_relu(nm[cb] * (d[rb] + Nf * cv[rb]) + bv[rb]);
                d[0] = bv2[0];
                d[1] = bv2[1];
                w16_feed(a_, in);
                w16_mm<2, 8>(w2, in, d);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) B[cb][rb] = nm[cb] * w16_relu(d[rb]);
            }
        }
        if (!lastg) {
            // next step: G rows, then P / R / u1pre from (nm*u2 | xq) through the folded matrices
            float wa[2][8 + EPNN_XS], wbm[2][8 + EPNN_XS], wu[2][8];
            f32x4 cu[2];
            W16_LD(wa, M.pwi, 2, 8 + EPNN_XS);
            WAVE_FENCE();
            gtiles();
            W16_LD(wbm, M.pwj, 2, 8 + EPNN_XS);
            W16_LD(wu, M.pu1, 2, 8);
            vec2(M.cu3, cu);
            W16_LD(pb, X.g[t + 1].w2, 2, 8);
            vec2(X.g[t + 1].b2, b2v);
            WAVE_FENCE();
            f32x4 r[3][2];
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                float in[8 + EPNN_XS], bin[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) in[s] = B[cb][s >> 2][s & 3];
#pragma unroll
                for (int s = 0; s < EPNN_XS; ++s) in[8 + s] = xq[cb][s];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) { P[cb][rb] = w16_splat(0.f); r[cb][rb] = w16_splat(0.f); U[cb][rb] = nm[cb] * cu[rb]; }
                w16_mm_skip<2, 8 + EPNN_XS, 7 + EPNN_XS>(wa, in, P[cb], xs3);
                w16_mm_skip<2, 8 + EPNN_XS, 7 + EPNN_XS>(wbm, in, r[cb], xs3);
                w16_feed(B[cb], bin);
                w16_mm<2, 8>(wu, bin, U[cb]);
            }
            store_rows(Rl, r);
            wave_sync_all();
        }
    }

    // ================================================================== EPN (charge_gn.py:98-118), h through the folded matrices
    wave_sync_lds();
    for (int i = lane; i < n * EPNN_W3_DST; i += 64) Dm[i] = 0.f;
    wave_sync_lds();
    const int qs = (nx + 1) >> 2, ql = (nx + 1) & 3;        // step / lane group of xq that holds q
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const WaveEpnPack &M = X.e[t];
        {
            constexpr int KS = 8 + EPNN_XS, SK = 7 + EPNN_XS;
            float wa[2][KS], wbm[2][KS];
            W16_LD(wa, M.wif, 2, KS);
            W16_LD(wbm, M.wjf, 2, KS);
            WAVE_FENCE();
            f32x4 pr[3][2], rr[3][2];
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                float in[KS];
#pragma unroll
                for (int s = 0; s < 8; ++s) in[s] = B[cb][s >> 2][s & 3];
#pragma unroll
                for (int s = 0; s < EPNN_XS; ++s) in[8 + s] = xq[cb][s];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) { pr[cb][rb] = w16_splat(0.f); rr[cb][rb] = w16_splat(0.f); }
                w16_mm_skip<2, KS, SK>(wa, in, pr[cb], xs3);
                w16_mm_skip<2, KS, SK>(wbm, in, rr[cb], xs3);
            }
            store_rows(Pl, pr);
            store_rows(Rl, rr);
        }
        f32x4 w3[2];
        W16_LD(pb, M.w2, 2, 8);
        vec2(M.b2, b2v);
        vec2(M.w3, w3);
        wave_sync_all();
        {
            // one column per UNORDERED pair, 16 pairs per block; records two blocks ahead, gathered rows one block ahead
            const int nblk = (np + 15) >> 4;
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
                w16_mm<2, KE>(gw, w_.e, g);
                const f32x4 ua = w16_relu((g[0] + w_.pi_[0]) + w_.rj_[0]), ub = w16_relu((g[1] + w_.pi_[1]) + w_.rj_[1]);
                const f32x4 va = w16_relu((g[0] + w_.pj_[0]) + w_.ri_[0]), vb = w16_relu((g[1] + w_.pj_[1]) + w_.ri_[1]);
                const float zu[8] = {ua[0], ua[1], ua[2], ua[3], ub[0], ub[1], ub[2], ub[3]};
                const float zv[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                f32x4 au[2] = {b2v[0], b2v[1]}, av[2] = {b2v[0], b2v[1]};
                w16_mm<2, 8>(pb, zu, au);
                w16_mm<2, 8>(pb, zv, av);
                float fd = 0.f;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const f32x4 tt = w16_relu(au[rb]) - w16_relu(av[rb]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) fd = fmaf(w3[rb][r], tt[r], fd);
                }
                const float d = 0.5f * w16_sumq(fd);                  // charge_gn.py:116
                if (q == 0 && valid && r_.wi != 0.f) Dm[li * EPNN_W3_DST + lj] = r_.wi * d;
                if (q == 1 && valid && r_.wj != 0.f) Dm[lj * EPNN_W3_DST + li] = -(r_.wj * d);
            };
            if (nblk > 0) {
                Rec r0, r1;
                Rows w0, w1;
                load_rec(0, r0);
                load_rec(min(1, nblk - 1), r1);
                load_rows(0, r0, w0);
                int blk = 0;
#pragma unroll 1
                for (; blk + 1 < nblk; blk += 2) {
                    Rec r2, r3;
                    load_rows(blk + 1, r1, w1);
                    load_rec(min(blk + 2, nblk - 1), r2);
                    WAVE_FENCE();
                    block(blk, r0, w0);
                    load_rows(min(blk + 2, nblk - 1), r2, w0);
                    load_rec(min(blk + 3, nblk - 1), r3);
                    WAVE_FENCE();
                    block(blk + 1, r1, w1);
                    r0 = r2;
                    r1 = r3;
                }
                if (blk < nblk) block(blk, r0, w0);
            }
        }
        wave_sync_lds();
        if (t + 1 < T) { W16_LD(gw, X.e[t + 1].we16, 2, KE); }
        WAVE_FENCE();
        // q_i += sum_j antisym_ij (charge_gn.py:118): lane (q, n16) adds columns j = q mod 4 of the rows of its atoms
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) {
            float dq = 0.f;
            const float *row = Dm + (cat[cb] ? ai[cb] : 0) * EPNN_W3_DST;
            for (int j = q; j < n; j += 4) dq += row[j];
            dq = w16_sumq(dq);
#pragma unroll
            for (int s = 0; s < EPNN_XS; ++s)
                if (s == qs && q == ql) xq[cb][s] += cat[cb] ? dq : 0.f;
        }
        wave_sync_lds();
    }
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
        for (int s = 0; s < EPNN_XS; ++s)
            if (s == qs && q == ql && own[cb]) A.q_out[a0 + ai[cb]] = xq[cb][s];
    if (A.handoff && lane == 0) {
        // the last wave of the forward (over all its launches) hands status + pair count to the host and re-zeroes the control words
        atomicAdd(A.status + 1, np);
        __threadfence();
        if (atomicAdd(A.status + 2, 1) == A.total_waves - 1) {
            __threadfence();
            const int st = atomicExch(A.status + 0, 0), cnt = atomicExch(A.status + 1, 0);
            atomicExch(A.status + 2, 0);
            volatile int *hs = A.host_status;
            hs[0] = st;
            hs[1] = cnt;
            __threadfence_system();
        }
    }
}
