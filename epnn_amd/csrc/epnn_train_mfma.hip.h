// Training step on the matrix pipe (reference charge_gn.py:393-402; forward :56-75 and :87-119), round 3.
//
// The row-fused kernels (epnn_train_fused.hip.h) give every atom i of a molecule a workgroup that runs the pair MLP over
// its N partner rows with scalar FMAs: 41 workgroups each staging the whole molecule and the 164 x 32 first Dense into
// LDS, ~10 us per launch whatever it computes.  Here the pair MLP is evaluated in the factorised form the inference
// kernels use,
//     z1_ij = relu(P_i + R_j + G_ij),  P_i = Wi^T a_i + b1,  R_j = Wj^T a_j,  G_ij = We^T e_ij,
// on v_mfma_f32_16x16x4_f32 with a COLUMN = an atom i and a loop over its partners j: a workgroup owns 16 atoms of a
// molecule (three workgroups at N = 41), its four wavefronts share out the partners, sums over partners stay in registers
// and meet once in LDS.  Weight fragments are read straight from the flat parameter vector (the weights change every
// step: no packed copy to keep in sync).  Same stored activations as the row-fused forward (H1, H2 per pair row, M, U0,
// U1, U2, hn, qn), so either backward can follow.
//
// Layout (epnn_common.h, "wave-autonomous fused kernel"): lane l = 16 q + m.  A operand lane (q, m) = A[m][k = q], B operand
// lane (q, n) = B[k = q][n], accumulator register r of lane (q, n) = D[4 q + r][n].  A 32-feature vector of a column is
// 2 row blocks rb x 4 registers: feature 16 rb + 4 q + r.  K steps in "acc" order (step s = 4 rb' + r' pairs lane q with
// input feature 16 rb' + 4 q + r') take an accumulator set as the next product's B operand as it stands.
#pragma once
#include "epnn_host.h"
#include "epnn_train_fused.hip.h"

#define EPNN_TM_NT 1024          // sixteen wavefronts per workgroup (a CU's four SIMDs, four deep: the chains of dependent MFMAs overlap)
#define EPNN_TM_NW (EPNN_TM_NT / 64)
#define EPNN_TM_FS 60            // atom-feature row in LDS: F = nx + 49 <= 60, zero padded (15 K steps of 4)

// sum over the four lanes q = 0..3 that share a column (v_permlane16/32_swap: no LDS), every lane gets the same bits
__device__ __forceinline__ float tm_sumq(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// D[rb] += W^T in (K = 32, acc order): W row-major [32][ldw] at theta + off, output features 16 rb + m of column block `ob`
template <int NRB>
__device__ __forceinline__ void tm_mm32(const float *theta, int off, int ldw, int ob, const f32x4 (&in)[2], f32x4 (&d)[NRB], int q, int m) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int k = 16 * (s >> 2) + 4 * q + (s & 3);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) d[rb] = tm_mfma(theta[off + k * ldw + ob + 16 * rb + m], in[s >> 2][s & 3], d[rb]);
    }
}

// MODE 0: message network of GNN step t + the update MLP of the workgroup's atoms; MODE 1: pass network, both orders of
// the pair, q_i += sum_j 0.5 (f_ij - f_ji) wgt_ij.  grid = B * ceil(N / 16).
template <int MODE>
__global__ __launch_bounds__(EPNN_TM_NT) void k_tm_fwd(TfPair A, TfUpd U, int nblk) {
    extern __shared__ __attribute__((aligned(16))) float tm_sm[];
    const int N = A.N, nx = A.nx, F = nx + 49;
    const int b = blockIdx.x / nblk, cb = blockIdx.x - b * nblk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, m = lane & 15;
    float *As = tm_sm;                           // [N][FS]  a_j = [x | h | q] of every atom of the molecule, zero padded
    float *Ps = As + N * EPNN_TM_FS;             // [N][32]  P_j
    float *Rs = Ps + N * 32;                     // [N][32]  R_j
    float *red = Rs + N * 32;                    // [NW][16][33]  the wavefronts' partial sums
    const size_t a0 = (size_t)b * N;
    const float *theta = A.theta;
    // ---- the molecule's atom rows
    for (int idx = tid; idx < N * EPNN_TM_FS; idx += EPNN_TM_NT) {
        const int j = idx / EPNN_TM_FS, k = idx - j * EPNN_TM_FS;
        const size_t at = a0 + j;
        As[idx] = k < nx ? A.x[at * nx + k] : (k < nx + 48 ? A.h[at * 48 + (k - nx)] : (k == nx + 48 ? A.q[at] : 0.f));
    }
    __syncthreads();
    // ---- projections of every atom: jobs (column block, P | R) dealt to the wavefronts
    for (int job = wave; job < 2 * nblk; job += EPNN_TM_NW) {
        const int jb = job >> 1, isR = job & 1;
        const int n = jb * 16 + m;
        f32x4 acc[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
            acc[rb] = isR ? f32x4{0.f, 0.f, 0.f, 0.f} : tm_ld4(theta + A.ob1 + 16 * rb + 4 * q);
        const int wrow = isR ? F : 0;
#pragma unroll
        for (int s = 0; s < EPNN_TM_FS / 4; ++s) {
            const int k = 4 * s + q;
            const float bv = n < N ? As[n * EPNN_TM_FS + k] : 0.f;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const float wv = k < F ? theta[A.oW1 + (wrow + k) * 32 + 16 * rb + m] : 0.f;
                acc[rb] = tm_mfma(wv, bv, acc[rb]);
            }
        }
        if (n < N) {
            float *dst = (isR ? Rs : Ps) + n * 32;
            tm_st4(dst + 4 * q, acc[0]);
            tm_st4(dst + 16 + 4 * q, acc[1]);
        }
    }
    __syncthreads();
    // ---- this workgroup's columns
    const int i = cb * 16 + m;                   // column atom of this lane
    const bool live = i < N;
    const int ic = live ? i : 0;
    const size_t bi = a0 + ic;                   // flat atom index
    f32x4 Pi[2], Ri[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        Pi[rb] = tm_ld4(Ps + ic * 32 + 16 * rb + 4 * q);
        Ri[rb] = tm_ld4(Rs + ic * 32 + 16 * rb + 4 * q);
    }
    f32x4 b2v[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) b2v[rb] = tm_ld4(theta + A.ob2 + 16 * rb + 4 * q);
    float we[2][12], w2[2][8];                   // first Dense's edge block (channel 12 q + s), second Dense (acc order)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int s = 0; s < 12; ++s) we[rb][s] = theta[A.oW1 + (2 * F + 12 * q + s) * 32 + 16 * rb + m];
#pragma unroll
        for (int s = 0; s < 8; ++s) w2[rb][s] = theta[A.oW2 + (16 * (s >> 2) + 4 * q + (s & 3)) * 32 + 16 * rb + m];
    }
    f32x4 w3v[2];                                // pass network: w3 in accumulator layout
    float b3 = 0.f;
    if (MODE) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) w3v[rb] = tm_ld4(theta + A.oW3 + 16 * rb + 4 * q);
        b3 = theta[A.ob3];
    }
    const size_t dstride = (size_t)(gridDim.x / nblk) * N * N * 32;      // second direction of the stored activations
    f32x4 S[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float qacc = 0.f;
    auto hidden = [&](const f32x4 (&z1)[2], f32x4 (&z2)[2]) {
        f32x4 d[2] = {b2v[0], b2v[1]};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            d[0] = tm_mfma(w2[0][s], z1[s >> 2][s & 3], d[0]);
            d[1] = tm_mfma(w2[1][s], z1[s >> 2][s & 3], d[1]);
        }
        z2[0] = tm_relu(d[0]);
        z2[1] = tm_relu(d[1]);
    };
    // the edge rows of a wavefront's partners are fetched one partner ahead (a trip's loads would otherwise sit in front of
    // its MFMAs with nothing to hide behind: one wavefront per SIMD and partner)
    auto fetch_e = [&](int j, f32x4 (&ee)[3], float &wg) {
        const size_t row = bi * N + (j < N ? j : 0);
        const float *er = A.e + row * 48 + 12 * q;
        const bool ok = live && j < N;
#pragma unroll
        for (int u = 0; u < 3; ++u) ee[u] = ok ? tm_ld4(er + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
        wg = (MODE && ok) ? A.wgt[row] : 0.f;
    };
    f32x4 en[3];
    float wn;
    fetch_e(wave, en, wn);
    for (int j = wave; j < N; j += EPNN_TM_NW) {
        const size_t row = bi * N + j;           // pair row (i, j)
        // G_ij = We^T e_ij for the sixteen columns
        const float ev[12] = {en[0][0], en[0][1], en[0][2], en[0][3], en[1][0], en[1][1], en[1][2], en[1][3], en[2][0], en[2][1], en[2][2], en[2][3]};
        const float wg = wn;
        fetch_e(j + EPNN_TM_NW, en, wn);
        f32x4 g[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            g[0] = tm_mfma(we[0][s], ev[s], g[0]);
            g[1] = tm_mfma(we[1][s], ev[s], g[1]);
        }
        f32x4 z1[2], z2[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) z1[rb] = tm_relu((Pi[rb] + tm_ld4(Rs + j * 32 + 16 * rb + 4 * q)) + g[rb]);
        hidden(z1, z2);
        if (live) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                tm_st4(A.H1 + row * 32 + 16 * rb + 4 * q, z1[rb]);
                tm_st4(A.H2 + row * 32 + 16 * rb + 4 * q, z2[rb]);
            }
        }
        if (MODE == 0) {
            S[0] += z2[0];
            S[1] += z2[1];
        } else {
            // the swapped row [a_j | a_i | e_ij]
            f32x4 y1[2], y2[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) y1[rb] = tm_relu((tm_ld4(Ps + j * 32 + 16 * rb + 4 * q) + Ri[rb]) + g[rb]);
            hidden(y1, y2);
            if (live) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    tm_st4(A.H1 + dstride + row * 32 + 16 * rb + 4 * q, y1[rb]);
                    tm_st4(A.H2 + dstride + row * 32 + 16 * rb + 4 * q, y2[rb]);
                }
            }
            float fa = 0.f, fb = 0.f;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    fa = fmaf(w3v[rb][r], z2[rb][r], fa);
                    fb = fmaf(w3v[rb][r], y2[rb][r], fb);
                }
            fa = tm_sumq(fa) + b3;
            fb = tm_sumq(fb) + b3;
            if (live) qacc += 0.5f * (fa - fb) * wg;
        }
    }
    // ---- the wavefronts' sums meet (fixed order)
    float *mine = red + (wave * 16 + m) * 33;
    if (MODE == 0) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[16 * rb + 4 * q + r] = S[rb][r];
    } else if (q == 0) {
        mine[0] = qacc;
    }
    __syncthreads();
    if (wave != 0) return;
    if (MODE == 1) {
        if (q == 0 && live) {
            float s = 0.f;
            for (int w = 0; w < EPNN_TM_NW; ++w) s += red[(w * 16 + m) * 33];
            A.qn[bi] = A.q[bi] + s;
        }
        return;
    }
    f32x4 Ssum[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = 0.f;
            for (int w = 0; w < EPNN_TM_NW; ++w) s += red[(w * 16 + m) * 33 + 16 * rb + 4 * q + r];
            Ssum[rb][r] = s;
        }
    // M_i = W3^T (sum_j z2_ij) + N b3                                     (charge_gn.py:68-70)
    f32x4 Mi[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) Mi[rb] = (float)N * tm_ld4(theta + A.ob3 + 16 * rb + 4 * q);
    tm_mm32<2>(theta, A.oW3, 32, 0, Ssum, Mi, q, m);
    if (live) {
        tm_st4(A.M + bi * 32 + 4 * q, Mi[0]);
        tm_st4(A.M + bi * 32 + 16 + 4 * q, Mi[1]);
    }
    if (U.theta == nullptr) return;
    // ---- update MLP of the sixteen atoms: h <- nm * Upd(nm * [h | M])   (charge_gn.py:71-74)
    const float nm = live ? U.nm[bi] : 0.f;
    f32x4 u1[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) u1[rb] = tm_ld4(theta + U.ob0 + 16 * rb + 4 * q);
#pragma unroll
    for (int s = 0; s < 12; ++s) {               // the h block: natural K order, h feature 4 s + q
        const int k = 4 * s + q;
        const float hv = As[ic * EPNN_TM_FS + nx + k] * nm;
        if (live) U.U0[bi * 80 + k] = hv;
        u1[0] = tm_mfma(theta[U.oW0 + k * 32 + m], hv, u1[0]);
        u1[1] = tm_mfma(theta[U.oW0 + k * 32 + 16 + m], hv, u1[1]);
    }
    f32x4 Mn[2] = {Mi[0] * nm, Mi[1] * nm};
    if (live) {
        tm_st4(U.U0 + bi * 80 + 48 + 4 * q, Mn[0]);
        tm_st4(U.U0 + bi * 80 + 48 + 16 + 4 * q, Mn[1]);
    }
    tm_mm32<2>(theta, U.oW0 + 48 * 32, 32, 0, Mn, u1, q, m);
    u1[0] = tm_relu(u1[0]);
    u1[1] = tm_relu(u1[1]);
    f32x4 u2[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) u2[rb] = tm_ld4(theta + U.ob1 + 16 * rb + 4 * q);
    tm_mm32<2>(theta, U.oW1, 32, 0, u1, u2, q, m);
    u2[0] = tm_relu(u2[0]);
    u2[1] = tm_relu(u2[1]);
    f32x4 hn[3];
#pragma unroll
    for (int rb = 0; rb < 3; ++rb) hn[rb] = tm_ld4(theta + U.ob2 + 16 * rb + 4 * q);
    tm_mm32<3>(theta, U.oW2, 48, 0, u2, hn, q, m);
    if (live) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            tm_st4(U.U1 + bi * 32 + 16 * rb + 4 * q, u1[rb]);
            tm_st4(U.U2 + bi * 32 + 16 * rb + 4 * q, u2[rb]);
        }
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) tm_st4(U.hn + bi * 48 + 16 * rb + 4 * q, hn[rb] * nm);
    }
}
