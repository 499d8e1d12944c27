// Shared definitions of the EPNN HIP library (gfx950 / MI355X only).
//
// Arithmetic model (exact re-association of reference charge_gn.py:56-119):
//   first Dense of every pair MLP is split by input block  W1^T[a_i|a_j|e_ij] = Wi^T a_i + Wj^T a_j + We^T e_ij,
//   so   P_i = Wi^T a_i + b1,  R_j = Wj^T a_j  (per atom)   and   G_ij = We^T e_ij  (per near pair only),
//   z1_ij = relu(P_i + R_j + G_ij),  z2_ij = relu(W2^T z1_ij + b2).
//   GNN:  sum_j m_ij = W3^T (sum_j z2_ij) + N b3 ;  EPN:  f_ij = w3 . z2_ij (+ b3, cancels in f_ij - f_ji).
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact f32 fma chain).  Lane l = 32*hh + c.  Accumulator register r of lane
// (c,hh) is element (row kappa(hh,r), col c) with kappa(hh,r) = 4*hh + (r&3) + 8*(r>>2).  All 32-wide feature
// vectors that feed an MFMA as the K dimension are kept "kappa-permuted": element r of half hh is feature
// kappa(hh,r), so that an accumulator can be fed straight back as the next A/B operand (no LDS round trip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EPNN_HID 32          // hidden width of every MLP
#define EPNN_EDIM 48         // edge channels == h_dim
#define EPNN_KA 30           // K-steps of the per-atom projection: features f = 2*s + hh, f < 60
#define EPNN_F1 59           // atom-feature slot holding the constant 1: the first Dense's bias b1 rides in Wi row 59
#define EPNN_AST 68          // LDS/HBM row stride (floats) of the even/odd atom-feature image a_eo
#define EPNN_PST 36          // LDS row stride (floats) of kappa-permuted 32-vectors (P, R, G)
#define EPNN_SST 33          // LDS row stride (floats) of out-major 32-vectors (S partial sums)
#define EPNN_MAXT 8
#define EPNN_SMALL_NMAX 32   // fused kernel handles molecules with n <= 32 real atoms

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// Offsets (in floats) into the packed weight buffer.
struct PairMlpPack {      // one message_fns[t] / pass_fns[t]
    int wiF;              // [KA][64]  Wi[2s+hh][c]      (A operand of P^T = Wi^T a^T)
    int wjF;              // [KA][64]  Wj[2s+hh][c]
    int b1p;              // [2][16]   b1[kappa(hh,r)]
    int weF;              // [24][64]  We[24hh+s][c]     (A operand of G^T = We^T e^T)
    int w2F;              // [16][64]  W2[kappa(hh,s)][c]
    int b2;               // [32]      b2[c]
    int b2p;              // [2][16]   b2[kappa(hh,r)]
    int w3p;              // [2][16]   w3[kappa(hh,r)]   (pass MLP only)
    int wqi, wqj;         // [2][16]   the q rows of Wi / Wj, kappa-permuted: P(q) = P(q = 0) + q wqi (tiled EPN stack: only q changes between steps)
};
struct UpdPack {          // update_fn with message_fns[t]'s last Dense folded in
    int u1F;              // [40][64]  s<24: Wu1[h feature][c]; s>=24: (W3_t Wu1_M)[2(s-24)+hh][c]
    int cb3p;             // [2][16]   (Wu1_M^T b3_t)[kappa(hh,r)]   (times N at run time)
    int bu1p;             // [2][16]
    int u2F;              // [16][64]  Wu2[kappa(hh,s)][c]
    int bu2p;             // [2][16]
    int u3F;              // [2][16][64] Wu3[kappa(hh,s)][32*tile+c]
    int bu3p;             // [2][2][16]
};
struct WeightIndex {
    PairMlpPack msg[EPNN_MAXT];
    PairMlpPack pas[EPNN_MAXT];
    UpdPack upd[EPNN_MAXT];
};

// ---- wave-autonomous fused kernel (epnn_wave.hip.h).  v_mfma_f32_16x16x4_f32: lane l = 16*q + m.
// A operand lane (q,m) = A[m][k=q], B operand lane (q,n) = B[k=q][n], accumulator register r of lane (q,n) =
// D[4q + r][n].  A 32-feature x 32-column product is 2 row blocks (rb) x 2 column blocks (cb) of such tiles; lane
// (q,n) then owns columns n and 16+n and, of each, the 8 features 16rb + 4q + r.  K steps are ordered so that an
// accumulator set feeds the next product directly: step s = 4rb' + r' pairs lane q with input feature
// 16rb' + 4q + r' ("acc" order).  Other K orders: "xq" feature 4s + q;  "e" channel 12q + s.
// Fragment = [rb][step / 4][64 lanes][4]: lane (q,m) holds W[in(step,q)][16rb + m], its four consecutive steps in 16 contiguous
// bytes (one global_load_dwordx4, W16_LDX).  Vectors are in natural feature order.
#define EPNN_XS 4            // K-steps of the xq block: nx + 3 <= 16
#define EPNN_ER 16           // dimension of the edge-feature subspace used by the fused kernel's own front-end
#define EPNN_ETAB_N 2049     // grid points of the table of B^T e(D) over [0, cutoff]
#define EPNN_NFLIP_MAX 128   // changes of the near flag as a function of the distance beyond dsafe that the fused front-end can hold
#define EPNN_DST 33          // LDS row stride (floats) of the per-molecule charge-transfer matrix
// (HU = units of the update MLP's hidden layers as the fused kernel sees them: 32, or 64 for k_wave_forward<.., NRU = 4>; the
//  shapes below are written for 32 -- with 64, "[2][8]" of u2 / pu1 reads [4][16], of u1s [4][8], of pwi / pwj [2][16+XS], of u3
//  [3][16], the [32] vectors of the update MLP [64]: pack_weights, epnn_api_weights.hip.h)
struct WaveGnnPack {           // GNN step t
    int we;               // [2][12][64]  e order       We_t
    int we16;             // [2][4][64]   k = 4q + s    B^T We_t   (edge features in the 16-dimensional basis)
    int we16b;            // [2][3][64][2] dwords       the same as three bf16 pieces per weight (v_mfma_f32_16x16x16_bf16: K slot s of lane q = coefficient 4q + s)
    int w2;               // [2][8][64]   acc order     W2_t
    int w2b;              // [2][3][64][4] dwords       W2_t as three bf16 pieces per weight (w16_split3 / pack_bf16x3), K slot s of lane q = acc order
    int b2;               // [32]
    int u1s;              // [2][8][64]   acc order     W3_t Wu1_M
    int cb3, bu1;         // [32]         Wu1_M^T b3_t (times N at run time), bu1
    int u2;               // [2][8][64]   acc order     Wu2
    int bu2;              // [32]
    int pwi, pwj;         // [2][8+XS][64]  acc rows: Wu3 M_h;  xq rows: [M_h^T bu3, M_x, M_q, b1]   (M = Wi / Wj of step t+1)
    int pu1;              // [2][8][64]   acc order     Wu3 Wu1_H
    int cu3;              // [32]         Wu1_H^T bu3
    // the same kernels as three bf16 pieces per weight, [2][3][64][4] dwords each (w2b's layout; 32-unit update MLPs only): the
    // per-atom chains on the bf16 matrix pipe.  `..hb` = the K = 32 block that multiplies nm u2 (acc order), `..xb` = the xq block in
    // the slot order of wave_xq_slot
    int u1sb, u2b, pu1b, pwihb, pwixb, pwjhb, pwjxb;
};
// K slot (lane group q, slot s of the lane) of the xq block's operand -> index into xq (0 node mask, 1..nx x, nx + 1 q, nx + 2 one),
// -1 = empty.  Lane group 0: mask, one, charge, x[0..4]; lane group 1: x[5..9].  (Every value is a float32 -- x is whatever the
// caller's feature columns hold -- and is split into three pieces like any other operand.)
__host__ __device__ static inline int wave_xq_slot(int q, int s, int nx) {
    if (q == 0) return s == 0 ? 0 : (s == 1 ? nx + 2 : (s == 2 ? nx + 1 : (s - 3 < nx ? s - 2 : -1)));
    if (q == 1) return 5 + s < nx ? 6 + s : -1;
    return -1;
}
struct WaveEpnPack {           // EPN step t
    int we, w2, b2, w3;   // w3: [32]
    int w2b;              // [2][3][64][4] dwords: W2_t as three bf16 pieces per weight (see WaveGnnPack)
    int we16;             // [2][4][64]   B^T We_t
    int we16b;            // [2][3][64][2] dwords: B^T We_t as three bf16 pieces per weight (see WaveGnnPack)
    int wi, wj;           // [2][XS+12][64]  xq rows, then h rows (acc order over 48 features): h given by the caller
    int wif, wjf;         // [2][8+XS][64]   acc rows: Wu3 M_h;  xq rows: [M_h^T bu3, M_x, M_q, b1]: h = nm (Wu3^T u2 + bu3) of the GNN stack
    int wifhb, wifxb, wjfhb, wjfxb;      // wif / wjf as bf16 pieces (see WaveGnnPack)
};
struct WaveIndex {
    WaveGnnPack g[EPNN_MAXT];
    WaveEpnPack e[EPNN_MAXT];
    int wi0, wj0;         // [2][XS+12][64]  first GNN step (h given by the caller, usually zeros)
    int u1h0;             // [2][12][64]     acc order over 48 features  Wu1_H
    int u3;               // [3][8][64]      acc order  Wu3 (48 outputs = 3 row blocks)
    int bu3;              // [48]
};

// A Dense stack of any widths (MLP_layer with other `nodes` than [32, 32], charge_gn.py:30-45): rows x dims[0] -> relu dims[1] ->
// ... -> dims[n] (the last Dense linear).  Kernels in Keras layout [in][out].  Used by epnn_mlp_forward_layers and by the update
// stage of the tiled path when make_model was given other `layers` (epnn_set_update_layers); plain f32 FMA loops, not the hot path.
#define EPNN_GMLP_ROWS 16     // rows per workgroup
#define EPNN_GMLP_WMAX 256    // widest layer
#define EPNN_GMLP_LMAX 8      // Dense layers
enum { EPNN_ACT_RELU = 0, EPNN_ACT_LINEAR = 1, EPNN_ACT_TANH = 2, EPNN_ACT_SIGMOID = 3 };     // (include/epnn.h: epnn_mlp_forward_layers)
struct GenMlp {
    int n;
    int act;                                             // activation of every layer but the last (EPNN_ACT_*; 0 = ReLU)
    int dims[EPNN_GMLP_LMAX + 1];
    int offW[EPNN_GMLP_LMAX], offB[EPNN_GMLP_LMAX];     // floats from `w`
    const float *w;
};

__host__ __device__ static inline int epnn_kappa(int hh, int r) { return 4 * hh + (r & 3) + 8 * (r >> 2); }

// position of atom feature f inside the even/odd image row: half (f&1), slot f>>1
__host__ __device__ static inline int epnn_aeo(int f) { return (f & 1) * 32 + (f >> 1); }

// status bits written by kernels
#define EPNN_ST_PAIR_OVERFLOW 1   // near-pair list capacity exceeded
#define EPNN_ST_TYPE_OVERFLOW 2   // tiled path, first GNN step by atom types: a molecule has more distinct feature rows than the table holds
#define EPNN_TYPE_MAX 64          // distinct atom-feature rows per molecule the first GNN step groups by (more: the all-pairs sweep runs)

#if defined(__HIPCC__)
__device__ __forceinline__ f32x16 epnn_mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 epnn_splat16(float v) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = v;
    return o;
}
// 16 contiguous floats (16-byte aligned) -> registers
__device__ __forceinline__ void epnn_ld16(const float *p, float (&o)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(p + 4 * q);
        o[4 * q + 0] = v[0]; o[4 * q + 1] = v[1]; o[4 * q + 2] = v[2]; o[4 * q + 3] = v[3];
    }
}
__device__ __forceinline__ void epnn_st16(float *p, const f32x16 &v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 o;
        o[0] = v[4 * q + 0]; o[1] = v[4 * q + 1]; o[2] = v[4 * q + 2]; o[3] = v[4 * q + 3];
        *reinterpret_cast<f32x4 *>(p + 4 * q) = o;
    }
}
// value held by the other half-wave's lane with the same c (lane ^ 32)
__device__ __forceinline__ float epnn_swap32(float v) {
    return __shfl_xor(v, 32, 64);
}
#endif
