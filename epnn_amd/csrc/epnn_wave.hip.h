// Wave-autonomous fused forward: ONE wavefront runs the whole GNN_layer + EPN_layer stack (reference
// charge_gn.py:56-119, 2T steps) of one molecule with n <= 32 real atoms -- and, through the compact entry, the
// front-end too (near pairs + Gaussian edge features, charge_gn.py:122-163).  No workgroup barriers, no idle waves:
// latency is hidden by the other wavefront of the SIMD (an independent molecule).
//
// MFMA shape: v_mfma_f32_16x16x4_f32.  What the hardware does (tools/micro/): a dependent MFMA chain issues back to
// back; a wave's own VALU / LDS instructions never overlap its own MFMAs; and while one wave keeps the matrix pipe busy
// the co-resident wave's VALU instructions issue every 6.1 cycles with this shape (8.1 with 32x32x2, 4.1 alone).  Work
// also comes in 16-column units: molecules with n <= 16 atoms and the last 16 pairs of a pair tile skip their second
// column block.
//
// Lane l = 16*q + n16 owns COLUMNS n16 and 16 + n16 (atoms, or near pairs in the pair tiles) and, of each column, the 8
// features 16*rb + 4*q + r (rb = 0,1; r = 0..3): that is the accumulator layout of the four 16x16 blocks [rb][cb] of a
// 32 x 32 product, and with the K steps ordered s = 4*rb' + r' (lane q <-> input feature 16*rb' + 4*q + r') an
// accumulator set is the next product's B operand as it stands.  Everything that is per atom lives in registers for
// the whole forward:  xq (node mask, x, q, 1), P_i = Wi^T a_i + b1, the message sum S, the update's h-part, and h
// (12 registers per column) during the EPN stack.  LDS holds only what other lanes read, rows in natural feature
// order (one 16-byte access per lane and row block): the R_j rows, the near-pair terms G, the pair map, and for the
// EPN the P rows and a per-molecule transfer matrix.
//
// GNN pair sweep (charge_gn.py:62-70): tile j = "partner j of every atom": column i of the tile is the pair (i, j),
//   z1 = relu(P_i + R_j + G_ij), acc = W2^T z1 + b2, S_i += relu(acc): plain register accumulation, the sum over
//   partners never leaves the lane.  One more tile with R = G = 0 is the reference's zero-padded partner; it is added
//   (N - n) times.  Operands of tile j+1 are fetched while tile j is in the matrix pipe.
// Update chain (charge_gn.py:71-74) with h never materialised between steps: the next step needs h only through
//   Wi_h^T h, Wj_h^T h and Wu1_h^T h, and h = nm (Wu3^T u2 + bu3), so the host folds Wu3 into those three matrices
//   (float64 products, pack_weights): K = 32 (nm*u2) + xq instead of 48 + xq, and no Wu3 GEMM per step.  h itself is
//   produced once, after the last step (for the EPN stack / GNN_layer output).
// EPN (charge_gn.py:98-118): one column per UNORDERED near pair, both directions share G; the weighted transfers go to
//   Dm[i][j] / Dm[j][i] and every atom adds its row in a fixed order.
#pragma once
#include <type_traits>

#include "epnn_common.h"
#include "epnn_frontend.hip.h"

struct WaveArgs {
    const float *wpack;
    const float *xin;      // [A][nx]
    const float *Q;        // [B]
    const int4 *wblk;      // per wavefront of this launch (largest molecule first): molecule b, first atom, atoms, first pair slot of the
                           // in-kernel front-end (sum of n(n-1)/2 over the molecules before b)
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
    double cut2;           // smallest float64 t with sqrt(t) >= cutoff: D < cutoff  <=>  D*D (before the sqrt) < cut2
    float *pt;             // [pcap][16] in-kernel front-end: the pairs' edge features in the 16-dimensional basis (B^T e)
    const float *etab;     // [tab_n][16] B^T e(D) on the grid D = i / tab_inv_h, i = 0 .. tab_n - 1 (last point = cutoff)
    double tab_inv_h;
    const double *flip;    // [nflip] distances beyond dsafe at which the near flag changes (epnn_create): near = even number of them below D
    int nflip;
    int tab_n;
    int total_waves;       // wavefronts of this forward over all its launches of this kernel (the last one to finish hands off)
    int prio_n;            // molecules with at least this many atoms run at raised wave priority (0: off), see k_wave_forward
    int handoff;           // in-kernel front-end and nothing else runs: the last wave reports to host_status and re-zeroes `status`
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
#ifdef EPNN_STAMPS_INIT            // (with EPNN_STAMPS) four more stamps inside the front-end: tools/wave_clocks.py --init-detail
#define WAVE_STAMP_I() WAVE_STAMP()
#else
#define WAVE_STAMP_I() do { } while (0)
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

// the same with the partner's coordinates already in registers (cx, cy, cz = atom j)
__device__ __forceinline__ double wave_dist2c(const double *xs, int i, double cx, double cy, double cz) {
    const double dx = cx - xs[3 * i + 0], dy = cy - xs[3 * i + 1], dz = cz - xs[3 * i + 2];
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}
// Near flag of a pair at distance D < cutoff (charge_gn.py:90-94 applied to get_init_edges' rows: max_k e_k > tol).  As a function
// of D it is 1 up to dsafe and changes `nflip` times beyond (normally once, 6e-3 below the cutoff): epnn_create found the changes
// with the reference's float64 expression, float32 cast included, so no cos / exp is evaluated per pair here.
__device__ __forceinline__ bool wave_near(const double *flip, int nflip, double D) {
    int cnt = 0;
    for (int k = 0; k < nflip; ++k) cnt += D > flip[k] ? 1 : 0;
    return !(cnt & 1);
}

// The last wavefront of a forward to finish hands status + pair count to the host and re-zeroes the control words (no other kernel
// runs behind the fused ones of a batch of small molecules).  ONE 64-bit atomic carries the pair count (low word) and the number of
// finished reports (high word): its return value tells the last reporter so and gives it the total, so nothing has to be ordered
// between two atomics.  (Rounds 1-4 used two 32-bit atomics with a __threadfence() between them in EVERY wavefront: an
// agent-scope fence writes the XCD's L2 back and invalidates it -- 1024 of them per launch of the bench batch.)  The status bits
// (word 0) are not written by the fused kernels themselves.
__device__ __forceinline__ void wave_handoff(const WaveArgs &A, int np) {
    unsigned long long *ctr = reinterpret_cast<unsigned long long *>(A.status + 2);       // 8-byte aligned: words 2 | 3 of the control block
    const unsigned long long old = atomicAdd(ctr, (unsigned long long)(unsigned)np | (1ull << 32));
    if ((int)(old >> 32) == A.total_waves - 1) {
        const int cnt = (int)(unsigned)(old & 0xffffffffull) + np;
        const int st = atomicExch(A.status + 0, 0);
        atomicExch(ctr, 0ull);
        volatile int *hs = A.host_status;
        hs[0] = st;
        hs[1] = cnt;
        __threadfence_system();
    }
}

#ifndef EPNN_WAVES_PER_SIMD
#define EPNN_WAVES_PER_SIMD 2     // register budget of the fused kernel: 2 -> 256 VGPRs (3 -> 168: measured slower)
#endif
// keep the loads issued above this point above it: the next chain's operands are fetched while the current chain runs
#define WAVE_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ f32x4 w16_mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 w16_splat(float v) { return f32x4{v, v, v, v}; }
// ---- f32-grade products on the bf16 matrix pipe.  x = x1 + x2 + x3 EXACTLY, each piece the upper 16 bits of what is left (8 + 8 + 8
// mantissa bits: truncation, so the remainders are exact and the third piece fits a bf16); of the nine partial products of two such
// operands the six largest are summed in f32 -- what is left out is below 2^-24 of the product, the f32 MFMA's own rounding.  One
// v_mfma_f32_16x16x32_bf16 takes a whole K = 32 in 16 cycles: six of them 96 cycles against 8 x 32 of v_mfma_f32_16x16x4_f32
// (tools/micro/bf16x6.hip: layout, accuracy against float64 -- 1.3e-6 against the f32 MFMA's 2.4e-6 on |C| ~ 13 -- and timing).
// Operand layout: lane 16 q + m holds row / column m and the K slots 8 q .. 8 q + 7 as four dwords of bf16 pairs (even slot low).
typedef __bf16 w16_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 w16_mfma_bf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(w16_bf16x8, a), __builtin_bit_cast(w16_bf16x8, b), c, 0, 0, 0);
}
// The remainders of a split on the MATRIX pipe (tools/micro/bf16x6.hip: exact, the pipe adds C unrounded).  With four values packed
// as the B operand of v_mfma_f32_16x16x16_bf16 (lane 16 q + m: column m, K slots 4 q + s = its values v[s]) and the values themselves
// as the accumulator (rows 4 q + r = v[r]), D = C + A B with A = -I (lane 16 qa + ma holds -1 in slot ma % 4 when qa = ma / 4) is
// "x - piece" in place: 2 MFMAs per 8 values and level instead of 8 v_and + 8 v_sub -- 12 vector instructions per split instead
// of 44, the SAME pieces bit for bit (round 5: bench 321 -> 327 M atoms/s; protein 0.382 -> 0.372 ms with the tiled sweep's own
// arrangement, epnn_large.hip.h).  The K = 16 instruction takes half the pipe time of the K = 32 one (a row block's four values
// are two dwords of the packed piece) and its operand is two registers, the same for both row blocks.
typedef short w16_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned w16_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ w16_u32x2 w16_ident() {
    const int lane = threadIdx.x & 63, qa = lane >> 4, ma = lane & 15;
    w16_u32x2 I;
#pragma unroll
    for (int d = 0; d < 2; ++d) I[d] = (qa == (ma >> 2) && d == ((ma & 3) >> 1)) ? (0xbf80u << (16 * (ma & 1))) : 0u;
    return I;
}
// x - piece for one row block: p_lo | p_hi = the two dwords of the packed piece that hold the block's four values
__device__ __forceinline__ f32x4 w16_rem(w16_u32x2 I, unsigned p_lo, unsigned p_hi, f32x4 x) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(w16_s16x4, I), __builtin_bit_cast(w16_s16x4, w16_u32x2{p_lo, p_hi}), x, 0, 0, 0);
}
__device__ __forceinline__ u32x4 w16_pack_hi(const f32x4 &lo, const f32x4 &hi) {      // v_perm_b32: the upper halves of two floats side by side
    u32x4 p;
    p[0] = __builtin_amdgcn_perm(__float_as_uint(lo[1]), __float_as_uint(lo[0]), 0x07060302u);
    p[1] = __builtin_amdgcn_perm(__float_as_uint(lo[3]), __float_as_uint(lo[2]), 0x07060302u);
    p[2] = __builtin_amdgcn_perm(__float_as_uint(hi[1]), __float_as_uint(hi[0]), 0x07060302u);
    p[3] = __builtin_amdgcn_perm(__float_as_uint(hi[3]), __float_as_uint(hi[2]), 0x07060302u);
    return p;
}
__device__ __forceinline__ w16_u32x2 w16_pack_hi2(const f32x4 &v) {                    // the same for four values
    w16_u32x2 p;
    p[0] = __builtin_amdgcn_perm(__float_as_uint(v[1]), __float_as_uint(v[0]), 0x07060302u);
    p[1] = __builtin_amdgcn_perm(__float_as_uint(v[3]), __float_as_uint(v[2]), 0x07060302u);
    return p;
}
#ifndef EPNN_SPLIT_VALU
__device__ __forceinline__ void w16_split3(const float (&v)[8], u32x4 &p1, u32x4 &p2, u32x4 &p3) {
    const w16_u32x2 I = w16_ident();       // (a function of the lane alone: the compiler keeps or rebuilds it as registers allow)
    f32x4 x0 = {v[0], v[1], v[2], v[3]}, x1 = {v[4], v[5], v[6], v[7]};
    p1 = w16_pack_hi(x0, x1);
    x0 = w16_rem(I, p1[0], p1[1], x0); x1 = w16_rem(I, p1[2], p1[3], x1);
    p2 = w16_pack_hi(x0, x1);
    x0 = w16_rem(I, p2[0], p2[1], x0); x1 = w16_rem(I, p2[2], p2[3], x1);
    p3 = w16_pack_hi(x0, x1);
}
#else
// (development builds, for comparison: the remainders as v_and_b32 + v_sub_f32, 44 vector instructions per split)
__device__ __forceinline__ void w16_split3(const float (&v)[8], u32x4 &p1, u32x4 &p2, u32x4 &p3) {
    float r1[8], r2[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        r1[s] = v[s] - __uint_as_float(__float_as_uint(v[s]) & 0xffff0000u);
        r2[s] = r1[s] - __uint_as_float(__float_as_uint(r1[s]) & 0xffff0000u);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        p1[j] = __builtin_amdgcn_perm(__float_as_uint(v[2 * j + 1]), __float_as_uint(v[2 * j]), 0x07060302u);
        p2[j] = __builtin_amdgcn_perm(__float_as_uint(r1[2 * j + 1]), __float_as_uint(r1[2 * j]), 0x07060302u);
        p3[j] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * j + 1]), __float_as_uint(r2[2 * j]), 0x07060302u);
    }
}
#endif
// D[rb] += W[rb] z for one column block: w = the kernel's three pieces per row block, z1..z3 the activations' (smallest terms first)
__device__ __forceinline__ void w16_mm_bf(const u32x4 (&w)[2][3], u32x4 z1, u32x4 z2, u32x4 z3, f32x4 (&d)[2]) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        d[rb] = w16_mfma_bf(w[rb][0], z3, d[rb]);
        d[rb] = w16_mfma_bf(w[rb][1], z2, d[rb]);
        d[rb] = w16_mfma_bf(w[rb][2], z1, d[rb]);
        d[rb] = w16_mfma_bf(w[rb][0], z2, d[rb]);
        d[rb] = w16_mfma_bf(w[rb][1], z1, d[rb]);
        d[rb] = w16_mfma_bf(w[rb][0], z1, d[rb]);
    }
}
// the three-piece kernel of a product: [2 row blocks][3 pieces][64 lanes][4 dwords], one 16-byte load each
#define W16_LDB(dst, off)                                                                                       \
    _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_)                                                         \
        _Pragma("unroll") for (int pc_ = 0; pc_ < 3; ++pc_)                                                     \
            (dst)[rb_][pc_] = reinterpret_cast<const u32x4 *>(wp + (size_t)(unsigned)(off))[(rb_ * 3 + pc_) * 64 + lane];
__device__ __forceinline__ f32x4 w16_relu(f32x4 v) {
    return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
}
// sum over the four lanes q = 0..3 that share a column: two VALU lane swaps (gfx950 v_permlane16/32_swap), no LDS
// round trip; every lane ends up with the same bits ((q0 + q1) + (q2 + q3))
__device__ __forceinline__ float w16_sumq(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ f32x4 w16_ld(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void w16_st(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }

// LDS through explicit 32-bit addresses: a running address costs one v_add per tile, constant offsets ride in the instruction
typedef __attribute__((address_space(3))) const f32x4 w16_lds_cf4;
typedef __attribute__((address_space(3))) const unsigned short w16_lds_cu16;
__device__ __forceinline__ unsigned w16_lds_addr(const void *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p; }
__device__ __forceinline__ f32x4 w16_lds_ld4(unsigned a) { return *(w16_lds_cf4 *)(size_t)a; }
__device__ __forceinline__ unsigned w16_lds_ldu16(unsigned a) { return *(w16_lds_cu16 *)(size_t)a; }
typedef __attribute__((address_space(3))) const unsigned w16_lds_cu32;
__device__ __forceinline__ unsigned w16_lds_ldu32(unsigned a) { return *(w16_lds_cu32 *)(size_t)a; }
__device__ __forceinline__ f32x2 w16_lo(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 w16_hi(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }
__device__ __forceinline__ f32x4 w16_cat(f32x2 a, f32x2 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3); }

// fragment loads: [nrb][stride / 4][64 lanes][4], steps s0 .. s0+cnt-1 of every row block (s0, cnt, stride multiples of 4).
// A lane's four consecutive K steps are 16 contiguous bytes: ONE global_load_dwordx4 per four steps (1 KB per wavefront
// instruction, fully coalesced) instead of four global_load_dword -- the kernel issued 1160 vector loads per molecule, nearly
// all of them weights, and every vector-memory instruction costs issue slots beside the matrix pipe (round 4).  One address
// per row block, the step groups as immediate offsets of the load (1 KB apart: inside the 4 KB immediate range).
#define W16_LDX(dst, off, nrb, cnt, stride, s0)                                                                   \
    _Pragma("unroll") for (int rb_ = 0; rb_ < (nrb); ++rb_) {                                                     \
        const f32x4 *fp_ = reinterpret_cast<const f32x4 *>(wp + (size_t)(unsigned)(off)) +                        \
                           (size_t)(unsigned)((rb_ * ((stride) / 4) + (s0) / 4) * 64 + lane);                     \
        _Pragma("unroll") for (int g_ = 0; g_ < (cnt) / 4; ++g_) {                                                \
            const f32x4 v_ = fp_[g_ * 64];                                                                        \
            (dst)[rb_][4 * g_] = v_[0]; (dst)[rb_][4 * g_ + 1] = v_[1]; (dst)[rb_][4 * g_ + 2] = v_[2]; (dst)[rb_][4 * g_ + 3] = v_[3]; \
        }                                                                                                         \
    }
#define W16_LD(dst, off, nrb, steps) W16_LDX(dst, off, nrb, steps, steps, 0)

// D[rb] += sum_s W[rb][s] * in[s]  for one column block (dependent chain of S MFMAs per row block).  Interleaving the row
// blocks' chains (a dependent v_mfma_f32_16x16x4_f32 can issue 40 cycles after its predecessor, the pipe takes one every 32)
// was measured in round 4: no gain with two wavefronts per SIMD, and it costs registers (spills in the block-per-wavefront kernels).
template <int NRB, int S>
__device__ __forceinline__ void w16_mm(const float (&w)[NRB][S], const float (&in)[S], f32x4 (&d)[NRB]) {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int s = 0; s < S; ++s) d[rb] = w16_mfma(w[rb][s], in[s], d[rb]);
}
// same, but K step SKIP is left out when `skip` is set (the last xq step holds only zeros when nx + 3 <= 12)
template <int NRB, int S, int SKIP>
__device__ __forceinline__ void w16_mm_skip(const float (&w)[NRB][S], const float (&in)[S], f32x4 (&d)[NRB], bool skip) {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s == SKIP) {
                if (!skip) d[rb] = w16_mfma(w[rb][s], in[s], d[rb]);
            } else {
                d[rb] = w16_mfma(w[rb][s], in[s], d[rb]);
            }
        }
}
// an accumulator set [NRB row blocks] of one column block as the next product's 4 NRB input steps
template <int NRB>
__device__ __forceinline__ void w16_feed(const f32x4 (&a)[NRB], float (&in)[4 * NRB]) {
#pragma unroll
    for (int s = 0; s < 4 * NRB; ++s) in[s] = a[s >> 2][s & 3];
}

// NRU: row blocks (16 units each) of the update MLP's two hidden layers -- 2 for the reference's [32, 32] (and anything embedded
// in it), 4 for make_model(layers) of up to [64, 64] (charge_gn.py:369-371; epnn_set_update_layers pads to 64 units).  The state
// between steps is nm * u2 (4 NRU K steps), the pair sweep and the EPN stack are the same for every NRU.
// ---- the edge products G = We^T pt of the in-kernel front-end (K = 16 coordinates per pair, §2 of DESIGN.md) on the bf16 pipe as well:
// pt split into three pieces (remainders on the matrix pipe, like every activation), We as three bf16 pieces from the host
// (WaveGnnPack::we16b), six v_mfma_f32_16x16x16_bf16 per row block (48 cycles against 4 x 32 of v_mfma_f32_16x16x4_f32): bench
// 331.6 -> 337.8 M atoms/s.  Pair lists from outside (K = 48, !FRONT) keep the f32 MFMAs; -DEPNN_G_F32 builds them everywhere.
// Both macros are used inside the kernels (k_wave_forward, k_wave_forward2), where FRONT, KE, gw, wp, lane exist.
#ifdef EPNN_G_F32
#define W16_G_BF 0
#else
#define W16_G_BF 1
#endif
#define W16_G_DECL                                                                                                        \
    constexpr bool GB = FRONT && W16_G_BF;                                                                                \
    w16_u32x2 gwb[2][3];                                                                                                  \
    auto g_mm = [&](const float (&e)[KE], f32x4 (&d)[2]) {      /* d[rb] += We[rb] e for 16 pairs (columns) */            \
        if constexpr (GB) {                                                                                               \
            const w16_u32x2 I = w16_ident();                                                                              \
            f32x4 x = {e[0], e[1], e[2], e[3]};                                                                           \
            w16_u32x2 z1, z2, z3;                                                                                         \
            z1 = w16_pack_hi2(x);                                                                                         \
            x = w16_rem(I, z1[0], z1[1], x);                                                                              \
            z2 = w16_pack_hi2(x);                                                                                         \
            x = w16_rem(I, z2[0], z2[1], x);                                                                              \
            z3 = w16_pack_hi2(x);                                                                                         \
            _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) {                                                            \
                d[rb] = w16_rem(gwb[rb][0], z3[0], z3[1], d[rb]);                                                         \
                d[rb] = w16_rem(gwb[rb][1], z2[0], z2[1], d[rb]);                                                         \
                d[rb] = w16_rem(gwb[rb][2], z1[0], z1[1], d[rb]);                                                         \
                d[rb] = w16_rem(gwb[rb][0], z2[0], z2[1], d[rb]);                                                         \
                d[rb] = w16_rem(gwb[rb][1], z1[0], z1[1], d[rb]);                                                         \
                d[rb] = w16_rem(gwb[rb][0], z1[0], z1[1], d[rb]);                                                         \
            }                                                                                                             \
        } else {                                                                                                          \
            w16_mm<2, KE>(gw, e, d);                                                                                      \
        }                                                                                                                 \
    }
// the kernel We of a G product: f32 fragments (offset `off`), or with GB its three bf16 pieces (offset `offb`)
#define GW_LD(off, offb)                                                                                                  \
    do {                                                                                                                  \
        if constexpr (GB) {                                                                                               \
            _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_)                                                           \
                _Pragma("unroll") for (int pc_ = 0; pc_ < 3; ++pc_)                                                       \
                    gwb[rb_][pc_] = reinterpret_cast<const w16_u32x2 *>(wp + (size_t)(unsigned)(offb))[(rb_ * 3 + pc_) * 64 + lane]; \
        } else { W16_LD(gw, off, 2, KE); }                                                                                \
    } while (0)
template <bool GNN, bool EPN, bool FRONT, int NRU = 2>
__global__ __launch_bounds__(64, EPNN_WAVES_PER_SIMD) void k_wave_forward(WaveArgs A, WaveIndex X) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x, q = lane >> 4, n16 = lane & 15;
    const int c = lane & 31, hh = lane >> 5;               // lane naming of the front-end (row pairs x 32 partners)
    if (!FRONT && (*A.status & EPNN_ST_PAIR_OVERFLOW)) return;
    const int4 wb = A.wblk[blockIdx.x];                   // molecule, first atom, atoms, first pair slot: ONE load, not a chain of three
    const int b = wb.x, a0 = wb.y, n = wb.z;
    // A launch lasts as long as its largest molecules (a 29-atom one takes 3.5x the mean), and the end of a run is their
    // latency.  The largest ones therefore run at raised priority: beside a smaller molecule on the same SIMD they win the
    // issue arbitration and run at nearly the speed they would have alone; the total work of the SIMD is unchanged.
    if (A.prio_n > 0 && n >= A.prio_n) __builtin_amdgcn_s_setprio(3);
    const int p0 = FRONT ? wb.w : A.row_off[a0];
    int np = FRONT ? 0 : A.row_off[a0 + n] - p0;
    const int nx = A.nx;
    const bool xs3 = nx + 3 <= 4 * (EPNN_XS - 1);         // the last xq K step is empty
    const float *wp = A.wpack;
    int nstamp = 0;
    (void)nstamp;
#ifdef EPNN_STAMPS
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz: with s_memtime it gives the shader clock this wavefront saw
#endif
    WAVE_STAMP();
    // Column block 0 holds atoms 0..15.  Block 1 holds the m = n - 16 atoms beyond them, C = 16 / m COPIES of each (column
    // n16 = atom 16 + n16 % m, copy n16 / m): every per-atom chain computes all 16 columns anyway, so the copies come for
    // free, and the pair sweep gives each copy a different partner -- block 1 is done after (n + 1) / C tiles instead of n + 1.
    const bool two = n > 16;
    const int m1 = two ? n - 16 : 16, C1 = 16 / m1;
    const int copy1 = n16 / m1;
    const int col1 = 16 + n16 % m1;
    const bool cat0 = n16 < n, cat1 = two && copy1 < C1;
    const bool own1 = cat1 && copy1 == 0;                   // the copy that stores the atom's rows / results

    // ---- the molecule's inputs, requested in ONE round trip behind the record (coordinates, features, charge, mask, h): every lane
    //      loads from clamped rows and selects afterwards -- a load under a condition is a branch with its own wait, and with the
    //      consumers right behind their loads this part of the front-end was four dependent round trips.  The per-column
    //      registers are assembled from these values where the front-end ends.
    const int ia0 = a0 + min(n16, n - 1), ia1 = a0 + (two ? col1 : 0);
    float cxyz[3] = {0.f, 0.f, 0.f};
    if (FRONT) {
        const float *xr = A.xyz + 3 * (size_t)(a0 + min(c, n - 1));
        cxyz[0] = xr[0]; cxyz[1] = xr[1]; cxyz[2] = xr[2];
    }
    float nmv0 = 1.f, nmv1 = 1.f, qa0, qa1, Qb = 0.f;
    if (A.nm_in) { nmv0 = A.nm_in[ia0]; nmv1 = A.nm_in[ia1]; }
    if (A.q_in) { qa0 = A.q_in[ia0]; qa1 = A.q_in[ia1]; }
    else Qb = A.Q[b];
    float xv0[EPNN_XS], xv1[EPNN_XS];
#pragma unroll
    for (int s = 0; s < EPNN_XS; ++s) {
        const int xi = min(max(4 * s + q - 1, 0), A.nx - 1);
        xv0[s] = A.xin[(size_t)ia0 * A.nx + xi];
        xv1[s] = A.xin[(size_t)ia1 * A.nx + xi];
    }
    f32x4 hk0[3], hk1[3];                                  // h: features 16*rb + 4*q + r of the two columns
#pragma unroll
    for (int rb = 0; rb < 3; ++rb) { hk0[rb] = w16_splat(0.f); hk1[rb] = w16_splat(0.f); }
    const bool have_h = A.h_in != nullptr;
    if (have_h) {
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const f32x4 v0 = w16_ld(A.h_in + (size_t)ia0 * EPNN_EDIM + 16 * rb + 4 * q), v1 = w16_ld(A.h_in + (size_t)ia1 * EPNN_EDIM + 16 * rb + 4 * q);
            if (cat0) hk0[rb] = v0;
            if (cat1) hk1[rb] = v1;
        }
    }
    WAVE_FENCE();

    // ---- LDS layout of THIS molecule inside the wave's fixed budget.  The two stacks need different tables, and the G
    //      rows are recomputed by every step anyway, so each stack has its own layout behind the common part:
    //        common  eij [pairs] | R [n][PST]
    //        GNN     pair map [n][NPM] u16 | zero row | G rows ...       (the sweep reads every G row n times: all in LDS
    //                                                                     for molecules up to ~24 atoms)
    //        EPN     P [n][PST] | transfer matrix [n][DST]              (its G term never leaves the registers: computed per
    //                                                                     block of 16 pairs right where it is used)
    unsigned short *eij = reinterpret_cast<unsigned short *>(sm);     // [np]  li | lj << 8
    const int eij_n = FRONT ? n * (n - 1) / 2 : np;        // in-kernel front-end: np is not known yet, reserve every i<j pair
    const int o_r = EPN ? ((((eij_n + 1) >> 1) + 3) & ~3) : 0;
    float *Rl = sm + o_r;                                  // [n][PST]   R_j rows (natural feature order)
    const int o_x = o_r + n * EPNN_PST;
    // Pair map [column][NPM] u16: where the pair tile "partner jp of every atom" finds the G row of column i's pair (i, jp): the
    // row's byte offset from `sm` (the zero row's for a far pair and for jp >= n: the padded partner), or 1 | slot << 1 for a
    // row that did not fit the LDS budget (gx).  NPM >= n + 8 (entries a tile behind the last partner may be read), = 2 mod 4
    // (odd dword stride: the 16 columns of a tile read 16 banks).
    unsigned short *pm = reinterpret_cast<unsigned short *>(sm + o_x);
    const int NPM = ((n + 9) & ~3) + 2;
    float *Pl = sm + o_x;                                  // [n][PST]   P_i rows
    // [n][DMS] weighted transfers: what atom i receives from atom j sits at Dm[i * DMS + 8 (j & 3) + (j >> 2)] -- the columns a lane
    // group q adds up (j = q, q + 4, ...: the charge update) are 8 consecutive floats, one or two 16-byte reads instead of up to eight
    constexpr int DMS = 36;
    float *Dm = sm + o_x + n * EPNN_PST;
    const int o_gg = o_x + ((((n * NPM + 1) >> 1) + 3) & ~3);
    const int grows_g = (A.lds_words - o_gg) / EPNN_PST - 1;
    float *Gl = sm + o_gg;                                 // GNN: row 0 all zeros, row 1 + s = G row of pair slot s < glds
    const unsigned zent = 4u * (unsigned)o_gg;             // the zero row's entry
    auto pm_ent = [&](int slot) -> unsigned short {
        return (unsigned short)(slot < grows_g ? 4u * (unsigned)(o_gg + (slot + 1) * EPNN_PST) : (1u | (unsigned)slot << 1));
    };
    int glds = min(np, grows_g);
    bool gover = np > glds;                                // some G rows live in HBM
    int ngt = (np + 31) >> 5;

    // ---- in-kernel front-end: coordinates -> LDS (the pair slots are assigned once the LDS tables exist)
    double *xs = reinterpret_cast<double *>(Rl);           // [n][3] float32 coordinates promoted like SciPy does (R rows come later)
    if (FRONT) {
        if (hh == 0 && c < n) {
            xs[3 * c + 0] = (double)cxyz[0];
            xs[3 * c + 1] = (double)cxyz[1];
            xs[3 * c + 2] = (double)cxyz[2];
        }
        wave_sync_lds();
        WAVE_STAMP_I();   // coordinates in LDS
    }
    // Edge operand of the G products.  Pair lists from outside carry arbitrary e rows: K = 48, lane (q, n16) takes channels
    // 12q..12q+11 of its pair.  The in-kernel front-end knows its e rows are Gaussians of a distance, which live in a
    // 16-dimensional subspace to 5e-10 (epnn_api.hip edge_basis): it projects every pair once (pt = B^T e) and all 2T
    // G products run with K = 16, lane (q, n16) taking coefficients 4q..4q+3.
    constexpr int KE = FRONT ? EPNN_ER / 4 : 12;
    // In-kernel front-end: the pairs' edge coordinates pt[np][16] stay in the wavefront's LDS as far as they fit: row r sits
    // 64 (r + 1) bytes below the END of the wavefront's budget, the first `ptl` rows are there.  During the GNN stack that is what
    // the G rows leave free (all rows up to ~18 atoms: no trip through HBM at all), the others go to A.pt; the EPN stack's tables
    // are smaller, so most of those come into LDS for its T steps (round 4: the 15.9 MB of pair scratch per launch were mostly
    // these rows, written once and read 2T times).
    int ptl = 0;
    const unsigned pt_top = w16_lds_addr(sm + A.lds_words) + 16u * (unsigned)q;     // (this lane's 16-byte piece of a row)
    auto load_e1 = [&](int slot, float (&e)[KE]) {
        const int sl = slot < np ? slot : 0;
        if (FRONT) {
            f32x4 v;
            if (sl < ptl) v = w16_lds_ld4(pt_top - 64u * (unsigned)(sl + 1));  // the pair's row in the wavefront's own LDS
            else v = w16_ld(A.pt + (size_t)(p0 + sl) * EPNN_ER + 4 * q);
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
    // e operands of G tile gt: pairs gt*32 + n16 and gt*32 + 16 + n16
    auto load_e = [&](int gt, float (&e0)[KE], float (&e1)[KE]) {
        load_e1(gt * 32 + n16, e0);
        load_e1(gt * 32 + 16 + n16, e1);
    };
    // first G tiles: We and the first e rows are on their way while the LDS tables are built
    float gw[2][KE], ge0[KE], ge1[KE];
    W16_G_DECL;
    GW_LD(FRONT ? X.g[0].we16 : (GNN ? X.g[0].we : X.e[0].we), X.g[0].we16b);
    if (GNN && !FRONT && ngt > 0) load_e(0, ge0, ge1);
    WAVE_FENCE();

    // ---- LDS init
    if (GNN)
        for (int i = lane; i < (n * NPM + 1) >> 1; i += 64) reinterpret_cast<unsigned *>(pm)[i] = zent | zent << 16;
    wave_sync_lds();
    if (FRONT) {
        WAVE_STAMP_I();   // per-atom inputs requested, pair map cleared
        // ---- slots in row-major order (rows i0, i0+1 per step; the lower row's pairs first)
        int base = 0;
        const double cx = (double)cxyz[0], cy = (double)cxyz[1], cz = (double)cxyz[2];   // this lane's partner c
        // (four rows per trip, two per half-wave: the two distance chains overlap; rows beyond the last pair of rows give no pairs)
        for (int i0 = 0; i0 + 1 < n; i0 += 4) {
            const int iA = i0 + hh, iB = min(i0 + 2 + hh, n - 1);
            const double dA = wave_dist2c(xs, iA, cx, cy, cz), dB = wave_dist2c(xs, iB, cx, cy, cz);
            const bool nearA = c > iA && c < n && dA < A.cut2;
            const bool nearB = i0 + 3 < n && c > iB && c < n && dB < A.cut2;
            const unsigned long long balA = __ballot(nearA), balB = __ballot(nearB);
            const unsigned loA = (unsigned)balA, hiA = (unsigned)(balA >> 32), loB = (unsigned)balB, hiB = (unsigned)(balB >> 32);
            const int baseB = base + __popc(loA) + __popc(hiA);
            if (nearA) {
                const int slot = base + (hh ? __popc(loA) : 0) + __popc((hh ? hiA : loA) & ((1u << c) - 1u));
                eij[slot] = (unsigned short)(iA | (c << 8));
                pm[c * NPM + iA] = pm_ent(slot);                       // e is symmetric: both directions share the row
                pm[iA * NPM + c] = pm_ent(slot);
            }
            if (nearB) {
                const int slot = baseB + (hh ? __popc(loB) : 0) + __popc((hh ? hiB : loB) & ((1u << c) - 1u));
                eij[slot] = (unsigned short)(iB | (c << 8));
                pm[c * NPM + iB] = pm_ent(slot);
                pm[iB * NPM + c] = pm_ent(slot);
            }
            base = baseB + __popc(loB) + __popc(hiB);
        }
        np = base;
        glds = min(np, grows_g);                            // the in-kernel front-end always runs both stacks: GNN first
        gover = np > glds;
        ngt = (np + 31) >> 5;
        // above the G rows AND above the EPN stack's tables, so that the rows survive the change of layout
        ptl = min(np, max(0, A.lds_words - max(o_gg + (glds + 1) * EPNN_PST, o_x + n * (EPNN_PST + DMS))) / EPNN_ER);
        wave_sync_lds();
        WAVE_STAMP_I();   // pair slots
        // ---- edge coefficients of every pair, one lane per pair: pt[pair] = B^T e(D), the 16 coordinates of the pair's Gaussian
        //      features (charge_gn.py:148-161) in the edge basis.  They are smooth functions of the one variable D: cubic
        //      Lagrange interpolation in a table of 4097 points over [0, cutoff] (built in float64 by epnn_create, error
        //      < 1e-9, part of epnn_edge_basis_residual).  The near flag max_k e_k > tol (charge_gn.py:90-94): wave_near.
        for (int s0 = 0; s0 < np; s0 += 64) {
            if (s0 + lane < np) {
                const int ij = eij[s0 + lane];
                const double D = wave_dist(xs, ij & 0xFF, ij >> 8);
                if (wave_near(A.flip, A.nflip, D)) eij[s0 + lane] = (unsigned short)(ij | 0x80);   // the near flag (charge_gn.py:90-94) rides in the pair's record
                const double tt = D * A.tab_inv_h;
                const int i0 = min(max((int)tt - 1, 0), A.tab_n - 4);
                const float u = (float)(tt - (double)i0);                    // position among the nodes i0 .. i0+3, normally in [1, 2)
                const float um1 = u - 1.f, um2 = u - 2.f, um3 = u - 3.f;
                const float w0 = -(um1 * um2 * um3) * (1.f / 6.f), w1 = (u * um2 * um3) * 0.5f;
                const float w2 = -(u * um1 * um3) * 0.5f, w3 = (u * um1 * um2) * (1.f / 6.f);
                const float *trow = A.etab + (size_t)i0 * EPNN_ER;
                f32x4 tv[4][EPNN_ER / 4];                                    // the four nodes' rows: sixteen requests, ONE round trip
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int g = 0; g < EPNN_ER / 4; ++g) tv[k][g] = w16_ld(trow + k * EPNN_ER + 4 * g);
                f32x4 v[EPNN_ER / 4];
#pragma unroll
                for (int g = 0; g < EPNN_ER / 4; ++g) v[g] = w0 * tv[0][g] + w1 * tv[1][g] + w2 * tv[2][g] + w3 * tv[3][g];
                // two separate predicated stores (LDS / HBM): through one merged pointer they become flat stores
                if (s0 + lane < ptl) {
                    float *prow = sm + A.lds_words - (size_t)(s0 + lane + 1) * EPNN_ER;
#pragma unroll
                    for (int g = 0; g < EPNN_ER / 4; ++g) w16_st(prow + 4 * g, v[g]);
                }
                asm volatile("" ::: "memory");
                if (s0 + lane >= ptl) {
                    float *prow = A.pt + (size_t)(p0 + s0 + lane) * EPNN_ER;
#pragma unroll
                    for (int g = 0; g < EPNN_ER / 4; ++g) w16_st(prow + 4 * g, v[g]);
                }
            }
        }
        wave_sync_all();
        WAVE_STAMP_I();   // edge coordinates
        if (ngt > 0) load_e(0, ge0, ge1);
    } else {
        if (GNN)
            for (int p = lane; p < np; p += 64) {
                const int li = A.pi[p0 + p] - a0, lj = A.pj[p0 + p] - a0;
                pm[li * NPM + lj] = pm_ent(p);                          // message into i = li from j = lj
                if (A.psym[p0 + p]) pm[lj * NPM + li] = pm_ent(p);
            }
        if (EPN)
            for (int p = lane; p < np; p += 64) {
                const int li = A.pi[p0 + p] - a0, lj = A.pj[p0 + p] - a0;
                eij[p] = (unsigned short)(li | (lj << 8));
            }
    }
    if (GNN)
        for (int i = lane; i < EPNN_PST; i += 64) Gl[i] = 0.f;                          // the sweep's zero row
    wave_sync_lds();
    // ---- per-column registers (cb = 0: column n16, cb = 1: column 16 + n16) from the values requested at the top
    const float nm0 = cat0 ? nmv0 : 0.f, nm1 = cat1 ? nmv1 : 0.f;
    float xq0[EPNN_XS], xq1[EPNN_XS];
    {
        if (!A.q_in) qa0 = qa1 = Qb / (float)n;                                            // charge_gn.py:337-338
        const float qv0 = cat0 ? qa0 : 0.f, qv1 = cat1 ? qa1 : 0.f;
#pragma unroll
        for (int s = 0; s < EPNN_XS; ++s) {
            const int phi = 4 * s + q;
            float v0 = 0.f, v1 = 0.f;
            if (phi == 0) { v0 = nm0; v1 = nm1; }
            else if (phi <= nx) { v0 = cat0 ? xv0[s] : 0.f; v1 = cat1 ? xv1[s] : 0.f; }
            else if (phi == nx + 1) { v0 = qv0; v1 = qv1; }
            else if (phi == nx + 2) { v0 = cat0 ? 1.f : 0.f; v1 = cat1 ? 1.f : 0.f; }
            xq0[s] = v0;
            xq1[s] = v1;
        }
    }

    // The per-atom chains (update MLP, the next step's P / R / u1pre, the EPN stack's P / R) on the bf16 matrix pipe as well (32-unit
    // update MLPs; -DEPNN_CHAIN_F32: f32 MFMAs as before): the K = 32 inputs S, u1 and nm u2 are split like the sweep's activations,
    // the xq block (mask, one, charge, x) is one more operand of up to 13 values in the slot order of wave_xq_slot.
#ifdef EPNN_CHAIN_F32
    constexpr bool CHB = false;
#else
    constexpr bool CHB = NRU == 2 && GNN;
#endif
    u32x4 Bp0[3], Bp1[3];                                  // nm u2 of the last step as pieces (CHB)
    u32x4 xqb0[3], xqb1[3];                                // the xq operand's pieces (CHB)
    // (slots: wave_xq_slot -- lane group 0: mask, one, charge, x[0..4]; 1: x[5..9].  Only the charge changes, between EPN steps:
    //  its three pieces go into the low half of dword 1 of the three operand pieces of lane group 0; nothing else is kept)
    float qc0 = 0.f, qc1 = 0.f;                            // the columns' charges, the same bits in every lane
    auto xq_charge = [&]() {
        if (q != 0) return;
        const float a0_ = __uint_as_float(__float_as_uint(qc0) & 0xffff0000u), r0_ = qc0 - a0_;
        const float b0_ = __uint_as_float(__float_as_uint(r0_) & 0xffff0000u), c0_ = r0_ - b0_;
        const float a1_ = __uint_as_float(__float_as_uint(qc1) & 0xffff0000u), r1_ = qc1 - a1_;
        const float b1_ = __uint_as_float(__float_as_uint(r1_) & 0xffff0000u), c1_ = r1_ - b1_;
        const float pc0[3] = {a0_, b0_, c0_}, pc1[3] = {a1_, b1_, c1_};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            xqb0[k][1] = (xqb0[k][1] & 0xffff0000u) | (__float_as_uint(pc0[k]) >> 16);
            xqb1[k][1] = (xqb1[k][1] & 0xffff0000u) | (__float_as_uint(pc1[k]) >> 16);
        }
    };
    if constexpr (CHB) {
        float xs0[8], xs1[8];
        qc0 = cat0 ? qa0 : 0.f;
        qc1 = cat1 ? qa1 : 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            // (unconditional loads of clamped columns, selects afterwards)
            const int k = q == 0 ? s_ - 3 : 5 + s_, kc = min(max(k, 0), nx - 1);
            const float u0 = A.xin[(size_t)ia0 * A.nx + kc], u1 = A.xin[(size_t)ia1 * A.nx + kc];
            const bool isx = q < 2 && k >= 0 && k < nx;
            float v0 = isx && cat0 ? u0 : 0.f, v1 = isx && cat1 ? u1 : 0.f;
            if (q == 0 && s_ == 0) { v0 = nm0; v1 = nm1; }
            if (q == 0 && s_ == 1) { v0 = cat0 ? 1.f : 0.f; v1 = cat1 ? 1.f : 0.f; }
            if (q == 0 && s_ == 2) { v0 = cat0 ? qa0 : 0.f; v1 = cat1 ? qa1 : 0.f; }
            xs0[s_] = v0;
            xs1[s_] = v1;
        }
        w16_split3(xs0, xqb0[0], xqb0[1], xqb0[2]);
        w16_split3(xs1, xqb1[0], xqb1[1], xqb1[2]);
    }
    WAVE_STAMP();   // init done
    const float Nf = (float)A.N, padw = (float)(A.N - n);
    const int Tg = GNN ? A.T : 0, Te = EPN ? A.T : 0;
    const int fo = 4 * q;                                   // this lane's feature offset inside a 16-feature row block

    // G rows of every near pair for the pair MLP whose We is in gw (first e rows in ge0/ge1); rows >= glds go to HBM
    auto gtile = [&](int gt, const float (&e0)[KE], const float (&e1)[KE]) {
        const int s0 = gt * 32 + n16, s1 = s0 + 16;
        f32x4 d0[2] = {w16_splat(0.f), w16_splat(0.f)};
        g_mm(e0, d0);
        // two separate predicated stores per target (LDS / HBM): merged into one pointer they become flat stores
        if (s0 < min(np, glds)) { w16_st(Gl + (s0 + 1) * EPNN_PST + fo, d0[0]); w16_st(Gl + (s0 + 1) * EPNN_PST + 16 + fo, d0[1]); }
        if (gover) {
            asm volatile("" ::: "memory");
            if (s0 >= glds && s0 < np) { w16_st(A.gx + (size_t)(p0 + s0) * 32 + fo, d0[0]); w16_st(A.gx + (size_t)(p0 + s0) * 32 + 16 + fo, d0[1]); }
        }
        if (gt * 32 + 16 < np) {                            // the second 16 pairs of the tile exist
            f32x4 d1[2] = {w16_splat(0.f), w16_splat(0.f)};
            g_mm(e1, d1);
            if (s1 < min(np, glds)) { w16_st(Gl + (s1 + 1) * EPNN_PST + fo, d1[0]); w16_st(Gl + (s1 + 1) * EPNN_PST + 16 + fo, d1[1]); }
            if (gover) {
                asm volatile("" ::: "memory");
                if (s1 >= glds && s1 < np) { w16_st(A.gx + (size_t)(p0 + s1) * 32 + fo, d1[0]); w16_st(A.gx + (size_t)(p0 + s1) * 32 + 16 + fo, d1[1]); }
            }
        }
    };
    auto gtiles = [&]() {                                   // e rows of the next tile are fetched while this one runs
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
    auto gprefetch = [&](int weoff, int weoffb) {
        GW_LD(weoff, weoffb);
        if (ngt > 0) load_e(0, ge0, ge1);
    };
    // 32-vector in natural feature order -> this lane's two groups of four
    auto vec2 = [&](int off, f32x4 (&v)[2]) {
        v[0] = w16_ld(wp + off + fo);
        v[1] = w16_ld(wp + off + 16 + fo);
    };
    auto vecu = [&](int off, f32x4 (&v)[NRU]) {            // one value per unit of an update layer
#pragma unroll
        for (int rb = 0; rb < NRU; ++rb) v[rb] = w16_ld(wp + off + 16 * rb + fo);
    };
    constexpr int KU = 4 * NRU;                            // K steps of nm * u2 / of u1

    // With both stacks in one launch the EPN takes h the way the GNN steps do, through nm*u2 of the last step and the
    // folded matrices (K = 32 + xq instead of 48 + xq, h itself is then needed only when the caller asks for it).
    constexpr bool FOLD = GNN && EPN;
    f32x4 B0[NRU], B1[NRU];                                 // nm*u2 of the two columns
#pragma unroll
    for (int rb = 0; rb < NRU; ++rb) { B0[rb] = w16_splat(0.f); B1[rb] = w16_splat(0.f); }
    // ================================================================== GNN steps (charge_gn.py:60-74)
    if (GNN) {
        f32x4 P0[2], P1[2], U0[NRU], U1[NRU];               // P, u1pre of the two columns
#ifdef EPNN_SWEEP_F32
        float pb[2][8];
#else
        u32x4 pb[2][3];
#endif
        f32x4 b2v[2];
        // ---- step 0: G rows, then P / R / u1pre from (xq | h)
        {
            float wa[2][EPNN_XS], wc[2][EPNN_XS];
            W16_LDX(wa, X.wi0, 2, EPNN_XS, EPNN_XS + 12, 0);      // the xq steps; the h steps only when h is given
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
            f32x4 r0[2] = {w16_splat(0.f), w16_splat(0.f)}, r1[2] = {w16_splat(0.f), w16_splat(0.f)};
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) { P0[rb] = w16_splat(0.f); P1[rb] = w16_splat(0.f); }
#pragma unroll
            for (int rb = 0; rb < NRU; ++rb) { U0[rb] = w16_splat(0.f); U1[rb] = w16_splat(0.f); }
            w16_mm_skip<2, EPNN_XS, EPNN_XS - 1>(wa, xq0, P0, xs3);
            w16_mm_skip<2, EPNN_XS, EPNN_XS - 1>(wc, xq0, r0, xs3);
            if (two) { w16_mm_skip<2, EPNN_XS, EPNN_XS - 1>(wa, xq1, P1, xs3); w16_mm_skip<2, EPNN_XS, EPNN_XS - 1>(wc, xq1, r1, xs3); }
            if (have_h) {                                   // layer-level entry: h given by the caller
                float wh[2][12], hin0[12], hin1[12];
#pragma unroll
                for (int s = 0; s < 12; ++s) { hin0[s] = hk0[s >> 2][s & 3]; hin1[s] = hk1[s >> 2][s & 3]; }
                W16_LDX(wh, X.wi0, 2, 12, EPNN_XS + 12, EPNN_XS);
                w16_mm<2, 12>(wh, hin0, P0);
                if (two) w16_mm<2, 12>(wh, hin1, P1);
                W16_LDX(wh, X.wj0, 2, 12, EPNN_XS + 12, EPNN_XS);
                w16_mm<2, 12>(wh, hin0, r0);
                if (two) w16_mm<2, 12>(wh, hin1, r1);
                // the h block of the update MLP's first layer; masked_input = [h, m] * node_mask (charge_gn.py:72): the mask
                // multiplies the whole pre-activation where the summed messages join it (once: it may be fractional)
                float wu[NRU][12];
                W16_LD(wu, X.u1h0, NRU, 12);
                w16_mm<NRU, 12>(wu, hin0, U0);
                if (two) w16_mm<NRU, 12>(wu, hin1, U1);
            }
            if (cat0) { w16_st(Rl + n16 * EPNN_PST + fo, r0[0]); w16_st(Rl + n16 * EPNN_PST + 16 + fo, r0[1]); }
            if (own1) { w16_st(Rl + col1 * EPNN_PST + fo, r1[0]); w16_st(Rl + col1 * EPNN_PST + 16 + fo, r1[1]); }
        }
        wave_sync_all();
        WAVE_STAMP();   // step-0 G tiles + projections

        // addresses of the sweep's LDS operands (bytes): this lane's 16-byte piece of a row starts at 4 fo
        const unsigned lbase = w16_lds_addr(sm) + 16u * (unsigned)q;
        const unsigned rbase = w16_lds_addr(Rl) + 16u * (unsigned)q, pbase = w16_lds_addr(pm);
#pragma unroll 1
        for (int t = 0; t < Tg; ++t) {
            const WaveGnnPack &M = X.g[t];
            const bool lastg = t + 1 == Tg;
            f32x4 S0[2] = {w16_splat(0.f), w16_splat(0.f)}, S1[2] = {w16_splat(0.f), w16_splat(0.f)};
            float u1s[NRU][8];
            {
                // ---- partner tiles.  Tile k of a block gives column i the partner jp = j0 + k * step: block 0 takes the
                //      partners one by one (j0 = 0, step 1), copy c of block 1 every C1-th from c on -- block 1 is finished
                //      after n / C1 + 1 tiles.  Tiles k < ntr hold real partners only (weight 1); the block's last tile holds
                //      what is left: real partners, the reference's zero-padded partner (jp == n: R = 0, G = 0,
                //      charge_gn.py:70), counted N - n times, and nothing (jp > n, weight 0).
                //      Per tile and lane: the R row (a broadcast in block 0), the pair map's entry (read two tiles ahead)
                //      and the G row it points to; one v_add per running address.
                auto sweep = [&](auto over_tag) {
                    constexpr bool OVER = decltype(over_tag)::value;
                    struct Ops { f32x4 r0, r1, g0, g1; };
                    auto load_rg = [&](Ops &o_, unsigned ra, unsigned ent) {
                        o_.r0 = w16_lds_ld4(ra);
                        o_.r1 = w16_lds_ld4(ra + 64);
                        if (!OVER) {
                            const unsigned ga = lbase + ent;
                            o_.g0 = w16_lds_ld4(ga);
                            o_.g1 = w16_lds_ld4(ga + 64);
                        } else {
                            const bool ov = (ent & 1u) != 0;
                            const unsigned ga = lbase + (ov ? zent : ent);
                            o_.g0 = w16_lds_ld4(ga);
                            o_.g1 = w16_lds_ld4(ga + 64);
                            if (ov) {
                                const float *gp = A.gx + (size_t)(p0 + (int)(ent >> 1)) * 32 + fo;
                                o_.g0 = w16_ld(gp);
                                o_.g1 = w16_ld(gp + 16);
                            }
                        }
                    };
                    // z1 = relu((P + R) + G), acc = W2^T z1 + b2, S += relu(acc)  (charge_gn.py:66-68).  The relu + sum of a tile runs one
                    // tile LATER, in the next tile's block of element-wise work in front of its MFMAs (`dp` = the pending accumulators):
                    // behind its own MFMAs it would wait for them and the compiler would un-pack its packed adds in their shadow.
                    auto tile = [&](const f32x2 (&Pc)[4], f32x4 (&Sc)[2], const Ops &o_, f32x4 (&dp)[2]) {
                        Sc[0] += w16_relu(dp[0]);
                        Sc[1] += w16_relu(dp[1]);
                        const f32x4 za = w16_relu((w16_cat(Pc[0], Pc[1]) + o_.r0) + o_.g0), zb = w16_relu((w16_cat(Pc[2], Pc[3]) + o_.r1) + o_.g1);
                        const float z[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
#ifdef EPNN_SWEEP_F32
                        dp[0] = b2v[0];
                        dp[1] = b2v[1];
                        WAVE_FENCE();
                        w16_mm<2, 8>(pb, z, dp);
#else
                        u32x4 z1, z2, z3;
                        w16_split3(z, z1, z2, z3);
                        dp[0] = b2v[0];
                        dp[1] = b2v[1];
                        WAVE_FENCE();
                        w16_mm_bf(pb, z1, z2, z3, dp);
#endif
                    };
                    // the block's last tile (weight w): the pending tile first, then its own contribution at once
                    auto tile_w = [&](const f32x2 (&Pc)[4], f32x4 (&Sc)[2], const Ops &o_, float w, f32x4 (&dp)[2]) {
                        tile(Pc, Sc, o_, dp);
                        Sc[0] += w * w16_relu(dp[0]);
                        Sc[1] += w * w16_relu(dp[1]);
                    };
                    // one column block.  B1 (compile time): block 1's copies take every C1-th partner (entries and R rows a stride
                    // apart, read one by one); block 0 takes the partners in turn, so the entries of two consecutive tiles are ONE
                    // aligned 32-bit word of the column's pair-map row: one LDS read per two tiles instead of two.
                    auto pass = [&](auto b1_tag) {
                        constexpr bool b1 = decltype(b1_tag)::value;
                        f32x2 Pc[4];
                        f32x4 Sc[2] = {w16_splat(0.f), w16_splat(0.f)};
                        f32x4 dp[2] = {w16_splat(0.f), w16_splat(0.f)};      // no tile pending yet: relu(0) adds nothing
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) {
                            Pc[2 * rb] = b1 ? w16_lo(P1[rb]) : w16_lo(P0[rb]);
                            Pc[2 * rb + 1] = b1 ? w16_hi(P1[rb]) : w16_hi(P0[rb]);
                        }
                        const int step = b1 ? C1 : 1, ntr = b1 ? n / C1 : n;
                        const int j0 = b1 && cat1 ? copy1 : 0;
                        const int col = b1 ? col1 : (cat0 ? n16 : 0);
                        const unsigned rs = 4u * EPNN_PST * (unsigned)step, ps = 2u * (unsigned)step;
                        unsigned ra = rbase + 4u * EPNN_PST * (unsigned)j0, pa = pbase + 2u * (unsigned)(col * NPM + j0);
                        // the last tile's operands: partner jl of this lane
                        const int jl = j0 + ntr * step;
                        const bool rl = jl < n;
                        const float wl = rl ? 1.f : (jl == n ? padw : 0.f);
                        const unsigned ral = rl ? rbase + 4u * EPNN_PST * (unsigned)jl : lbase + zent;
                        // (block 0: its last tile is the zero-padded partner alone -- R = 0, G = 0, nothing to read)
                        const unsigned entl = b1 ? w16_lds_ldu16(pbase + 2u * (unsigned)(col * NPM + min(jl, n))) : zent;
                        auto load_last = [&](Ops &o_) {
                            if (b1) load_rg(o_, ral, entl);
                            else { o_.r0 = w16_splat(0.f); o_.r1 = w16_splat(0.f); o_.g0 = w16_splat(0.f); o_.g1 = w16_splat(0.f); }
                        };
                        Ops oa, ob;
                        unsigned en, en2 = 0;
                        if (b1) {
                            en = w16_lds_ldu16(pa);               // entry of tile 0
                            pa += ps;
                        } else {
                            en2 = w16_lds_ldu32(pa);              // entries of tiles 0 and 1
                            pa += 4u;
                            en = en2 & 0xffffu;
                        }
                        load_rg(oa, ra, en);
                        ra += rs;
                        if (b1) {
                            en = w16_lds_ldu16(pa);               // entry of tile 1
                            pa += ps;
                        } else en = en2 >> 16;
                        int k = 0;
#pragma unroll 1
                        for (; k + 3 <= ntr; k += 2) {            // tiles k, k + 1; tile k + 2 is a real one too
                            load_rg(ob, ra, en);
                            ra += rs;
                            if (b1) {
                                en = w16_lds_ldu16(pa);
                                pa += ps;
                            } else {
                                en2 = w16_lds_ldu32(pa);          // entries of tiles k + 2 and k + 3
                                pa += 4u;
                                en = en2 & 0xffffu;
                            }
                            WAVE_FENCE();
                            tile(Pc, Sc, oa, dp);
                            load_rg(oa, ra, en);
                            ra += rs;
                            if (b1) {
                                en = w16_lds_ldu16(pa);
                                pa += ps;
                            } else en = en2 >> 16;
                            WAVE_FENCE();
                            tile(Pc, Sc, ob, dp);
                        }
                        if constexpr (!CHB) { if (!b1) { W16_LD(u1s, M.u1s, NRU, 8); } }   // first operand of the update MLP
                        if (ntr - k == 2) {                       // real tiles k, k + 1, then the last tile
                            load_rg(ob, ra, en);
                            WAVE_FENCE();
                            tile(Pc, Sc, oa, dp);
                            load_last(oa);
                            WAVE_FENCE();
                            tile(Pc, Sc, ob, dp);
                            tile_w(Pc, Sc, oa, wl, dp);
                        } else {                                  // real tile k, then the last tile
                            load_last(ob);
                            WAVE_FENCE();
                            tile(Pc, Sc, oa, dp);
                            tile_w(Pc, Sc, ob, wl, dp);
                        }
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) {
                            if (b1) S1[rb] = Sc[rb];
                            else S0[rb] = Sc[rb];
                        }
                    };
                    if (two) pass(std::true_type{});              // block 1 first, then block 0
                    pass(std::false_type{});
                };
                if (gover) sweep(std::true_type{});
                else sweep(std::false_type{});
                if (two && C1 > 1) {
                    // every copy of a block-1 atom holds the sum over its own partners: add the copies in a fixed order (all of
                    // them end up with the same bits).  The G rows of this step are dead: their LDS words are the scratch.
                    wave_sync_lds();
                    float *scr = Gl;
                    w16_st(scr + (n16 * 4 + q) * 8, S1[0]);
                    w16_st(scr + (n16 * 4 + q) * 8 + 4, S1[1]);
                    wave_sync_lds();
                    f32x4 t0_ = w16_splat(0.f), t1_ = w16_splat(0.f);
                    for (int k = 0; k < C1; ++k) {
                        const int src = (n16 % m1) + m1 * k;
                        t0_ += w16_ld(scr + (src * 4 + q) * 8);
                        t1_ += w16_ld(scr + (src * 4 + q) * 8 + 4);
                    }
                    S1[0] = t0_;
                    S1[1] = t1_;
                    wave_sync_lds();
                    if (lane < EPNN_PST) Gl[lane] = 0.f;                        // the scratch covered the zero row
                }
            }
            if (t < 2) WAVE_STAMP();   // pair tiles
            // ---- update MLP (charge_gn.py:71-74); the last message Dense is folded into u1s
            if constexpr (CHB) {
                u32x4 w1[2][3], w2b[2][3];
                f32x4 cv[2], bv[2];
                W16_LDB(w1, M.u1sb);
                W16_LDB(w2b, M.u2b);
                vecu(M.cb3, cv);
                vecu(M.bu1, bv);
                WAVE_FENCE();
                f32x4 d0[2] = {U0[0], U0[1]}, d1[2] = {U1[0], U1[1]};
                float in0[8], in1[8];
                u32x4 s1, s2, s3;
                w16_feed<2>(S0, in0);
                w16_split3(in0, s1, s2, s3);
                w16_mm_bf(w1, s1, s2, s3, d0);
                if (two) { w16_feed<2>(S1, in1); w16_split3(in1, s1, s2, s3); w16_mm_bf(w1, s1, s2, s3, d1); }
                f32x4 a0_[2], a1_[2];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    a0_[rb] = w16_relu(nm0 * (d0[rb] + Nf * cv[rb]) + bv[rb]);
                    a1_[rb] = w16_relu(nm1 * (d1[rb] + Nf * cv[rb]) + bv[rb]);
                }
                vecu(M.bu2, bv);
                if (!lastg) gprefetch(FRONT ? X.g[t + 1].we16 : X.g[t + 1].we, X.g[t + 1].we16b);
                else if (Te > 0) { GW_LD(FRONT ? X.e[0].we16 : X.e[0].we, X.e[0].we16b); }
                WAVE_FENCE();
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) { d0[rb] = bv[rb]; d1[rb] = bv[rb]; }
                w16_feed<2>(a0_, in0);
                w16_split3(in0, s1, s2, s3);
                w16_mm_bf(w2b, s1, s2, s3, d0);
                if (two) { w16_feed<2>(a1_, in1); w16_split3(in1, s1, s2, s3); w16_mm_bf(w2b, s1, s2, s3, d1); }
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) { B0[rb] = nm0 * w16_relu(d0[rb]); B1[rb] = nm1 * w16_relu(d1[rb]); }
                // nm u2 as pieces: the three projections below and, after the last step, the EPN stack's take them
                w16_feed<2>(B0, in0);
                w16_split3(in0, Bp0[0], Bp0[1], Bp0[2]);
                w16_feed<2>(B1, in1);
                w16_split3(in1, Bp1[0], Bp1[1], Bp1[2]);
            } else
            {
                float w2[NRU][KU], in0[8], in1[8];
                f32x4 cv[NRU], bv[NRU];
                // (NRU = 2: every operand of the block requested up front, it all fits the registers; 4: the second layer's kernel
                //  is requested once the first layer's operands are dead -- up front it spilled some 130 registers)
                if constexpr (NRU == 2) { W16_LD(w2, M.u2, NRU, KU); }
                vecu(M.cb3, cv);
                vecu(M.bu1, bv);
                WAVE_FENCE();
                f32x4 d0[NRU], d1[NRU];
#pragma unroll
                for (int rb = 0; rb < NRU; ++rb) { d0[rb] = U0[rb]; d1[rb] = U1[rb]; }
                w16_feed<2>(S0, in0);
                w16_mm<NRU, 8>(u1s, in0, d0);
                if (two) { w16_feed<2>(S1, in1); w16_mm<NRU, 8>(u1s, in1, d1); }
                f32x4 a0_[NRU], a1_[NRU];
#pragma unroll
                for (int rb = 0; rb < NRU; ++rb) {
                    a0_[rb] = w16_relu(nm0 * (d0[rb] + Nf * cv[rb]) + bv[rb]);
                    a1_[rb] = w16_relu(nm1 * (d1[rb] + Nf * cv[rb]) + bv[rb]);
                }
                if constexpr (NRU != 2) { W16_LD(w2, M.u2, NRU, KU); }
                vecu(M.bu2, bv);
                if (!lastg) gprefetch(FRONT ? X.g[t + 1].we16 : X.g[t + 1].we, X.g[t + 1].we16b);
                else if (Te > 0) { GW_LD(FRONT ? X.e[0].we16 : X.e[0].we, X.e[0].we16b); }
                WAVE_FENCE();
                float ain0[KU], ain1[KU];
#pragma unroll
                for (int rb = 0; rb < NRU; ++rb) { d0[rb] = bv[rb]; d1[rb] = bv[rb]; }
                w16_feed<NRU>(a0_, ain0);
                w16_mm<NRU, KU>(w2, ain0, d0);
                if (two) { w16_feed<NRU>(a1_, ain1); w16_mm<NRU, KU>(w2, ain1, d1); }
#pragma unroll
                for (int rb = 0; rb < NRU; ++rb) { B0[rb] = nm0 * w16_relu(d0[rb]); B1[rb] = nm1 * w16_relu(d1[rb]); }
            }
            if (t < 2) WAVE_STAMP();   // U1, U2
            if constexpr (CHB) {
              if (!lastg) {
                // next step: G rows, then P / R / u1pre from (nm u2 pieces | xq operand) through the folded matrices' pieces
                u32x4 wh[2][3], wx[2][3];
                W16_LDB(wh, M.pwihb);
                W16_LDB(wx, M.pwixb);
                WAVE_FENCE();
                gtiles();
                if (t < 2) WAVE_STAMP();   // G tiles
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) { P0[rb] = w16_splat(0.f); P1[rb] = w16_splat(0.f); }
                w16_mm_bf(wh, Bp0[0], Bp0[1], Bp0[2], P0);
                w16_mm_bf(wx, xqb0[0], xqb0[1], xqb0[2], P0);
                if (two) { w16_mm_bf(wh, Bp1[0], Bp1[1], Bp1[2], P1); w16_mm_bf(wx, xqb1[0], xqb1[1], xqb1[2], P1); }
                W16_LDB(wh, M.pwjhb);
                W16_LDB(wx, M.pwjxb);
                f32x4 cu[2];
                vecu(M.cu3, cu);
                f32x4 r0[2] = {w16_splat(0.f), w16_splat(0.f)}, r1[2] = {w16_splat(0.f), w16_splat(0.f)};
                w16_mm_bf(wh, Bp0[0], Bp0[1], Bp0[2], r0);
                w16_mm_bf(wx, xqb0[0], xqb0[1], xqb0[2], r0);
                if (two) { w16_mm_bf(wh, Bp1[0], Bp1[1], Bp1[2], r1); w16_mm_bf(wx, xqb1[0], xqb1[1], xqb1[2], r1); }
                if (cat0) { w16_st(Rl + n16 * EPNN_PST + fo, r0[0]); w16_st(Rl + n16 * EPNN_PST + 16 + fo, r0[1]); }
                if (own1) { w16_st(Rl + col1 * EPNN_PST + fo, r1[0]); w16_st(Rl + col1 * EPNN_PST + 16 + fo, r1[1]); }
                W16_LDB(wh, M.pu1b);
#ifdef EPNN_SWEEP_F32
                W16_LD(pb, X.g[t + 1].w2, 2, 8);
#else
                W16_LDB(pb, X.g[t + 1].w2b);
#endif
                vec2(X.g[t + 1].b2, b2v);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) { U0[rb] = nm0 * cu[rb]; U1[rb] = nm1 * cu[rb]; }
                w16_mm_bf(wh, Bp0[0], Bp0[1], Bp0[2], U0);
                if (two) w16_mm_bf(wh, Bp1[0], Bp1[1], Bp1[2], U1);
                wave_sync_all();
                if (t < 2) WAVE_STAMP();   // projections
              }
            } else
            if (!lastg) {
                // next step: G rows, then P / R / u1pre from (nm*u2 | xq) through the folded matrices
                float wa[2][KU + EPNN_XS], wb[2][KU + EPNN_XS], in0[KU + EPNN_XS], in1[KU + EPNN_XS];
                W16_LD(wa, M.pwi, 2, KU + EPNN_XS);
                WAVE_FENCE();
                gtiles();
                if (t < 2) WAVE_STAMP();   // G tiles
#pragma unroll
                for (int s = 0; s < KU; ++s) { in0[s] = B0[s >> 2][s & 3]; in1[s] = B1[s >> 2][s & 3]; }
#pragma unroll
                for (int s = 0; s < EPNN_XS; ++s) { in0[KU + s] = xq0[s]; in1[KU + s] = xq1[s]; }
                W16_LD(wb, M.pwj, 2, KU + EPNN_XS);
                float wu[NRU][KU];
                f32x4 cu[NRU];
                if constexpr (NRU == 2) { W16_LD(wu, M.pu1, NRU, KU); }
                vecu(M.cu3, cu);
                WAVE_FENCE();
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) { P0[rb] = w16_splat(0.f); P1[rb] = w16_splat(0.f); }
                w16_mm_skip<2, KU + EPNN_XS, KU - 1 + EPNN_XS>(wa, in0, P0, xs3);
                if (two) w16_mm_skip<2, KU + EPNN_XS, KU - 1 + EPNN_XS>(wa, in1, P1, xs3);
                f32x4 r0[2] = {w16_splat(0.f), w16_splat(0.f)}, r1[2] = {w16_splat(0.f), w16_splat(0.f)};
                w16_mm_skip<2, KU + EPNN_XS, KU - 1 + EPNN_XS>(wb, in0, r0, xs3);
                if (two) w16_mm_skip<2, KU + EPNN_XS, KU - 1 + EPNN_XS>(wb, in1, r1, xs3);
                if (cat0) { w16_st(Rl + n16 * EPNN_PST + fo, r0[0]); w16_st(Rl + n16 * EPNN_PST + 16 + fo, r0[1]); }
                if (own1) { w16_st(Rl + col1 * EPNN_PST + fo, r1[0]); w16_st(Rl + col1 * EPNN_PST + 16 + fo, r1[1]); }
                if constexpr (NRU != 2) { W16_LD(wu, M.pu1, NRU, KU); }
#ifdef EPNN_SWEEP_F32
                W16_LD(pb, X.g[t + 1].w2, 2, 8);
#else
                W16_LDB(pb, X.g[t + 1].w2b);
#endif
                vec2(X.g[t + 1].b2, b2v);
                WAVE_FENCE();
                float bin0[KU], bin1[KU];
                w16_feed<NRU>(B0, bin0);
                w16_feed<NRU>(B1, bin1);
#pragma unroll
                for (int rb = 0; rb < NRU; ++rb) { U0[rb] = nm0 * cu[rb]; U1[rb] = nm1 * cu[rb]; }
                w16_mm<NRU, KU>(wu, bin0, U0);
                if (two) w16_mm<NRU, KU>(wu, bin1, U1);
                wave_sync_all();
                if (t < 2) WAVE_STAMP();   // projections
            }
        }
        if (!FOLD || A.h_out) {
            // h = node_mask * (Wu3^T u2 + bu3)  (charge_gn.py:73-74) after the last step, straight into the h registers
            float w[3][KU], bin0[KU], bin1[KU];
            W16_LD(w, X.u3, 3, KU);
            f32x4 bv[3];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) bv[rb] = w16_ld(wp + X.bu3 + 16 * rb + fo);
            WAVE_FENCE();
            w16_feed<NRU>(B0, bin0);
            w16_feed<NRU>(B1, bin1);
            // B = nm * u2 already: h = nm * (Wu3^T u2) + nm * bu3
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) { hk0[rb] = nm0 * bv[rb]; hk1[rb] = nm1 * bv[rb]; }
            w16_mm<3, KU>(w, bin0, hk0);
            if (two) w16_mm<3, KU>(w, bin1, hk1);
        }
        if (A.h_out) {
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                if (cat0) w16_st(A.h_out + (size_t)(a0 + n16) * EPNN_EDIM + 16 * rb + fo, hk0[rb]);
                if (own1) w16_st(A.h_out + (size_t)(a0 + col1) * EPNN_EDIM + 16 * rb + fo, hk1[rb]);
            }
        }
    }

    WAVE_STAMP();   // GNN done
    // ================================================================== EPN steps (charge_gn.py:98-118)
    if (EPN) {
        wave_sync_lds();                                    // the GNN's tables are dead: switch to the EPN layout
        for (int i = lane; i < n * DMS; i += 64) Dm[i] = 0.f;
        if (FRONT && ptl < np) {
            // edge coordinates that went through HBM (no room beside the G rows): the EPN stack's layout has room for more of them
            const int pte = min(np, (A.lds_words - (o_x + n * (EPNN_PST + DMS))) / EPNN_ER);
            for (int i = lane; i < (pte - ptl) * (EPNN_ER / 4); i += 64) {
                const int r = ptl + (i >> 2), k = i & 3;
                w16_st(sm + A.lds_words - (size_t)(r + 1) * EPNN_ER + 4 * k, w16_ld(A.pt + (size_t)(p0 + r) * EPNN_ER + 4 * k));
            }
            ptl = max(ptl, pte);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        wave_sync_lds();
        const int qs = (nx + 1) >> 2, ql = (nx + 1) & 3;    // step / lane group of xq that holds q
#pragma unroll 1
        for (int t = 0; t < Te; ++t) {
            const WaveEpnPack &M = X.e[t];
            if constexpr (CHB && FOLD) {
                // P / R of this step from (nm u2 pieces | xq operand): the bf16 pipe (see CHB)
                u32x4 wh[2][3], wx[2][3];
                W16_LDB(wh, M.wifhb);
                W16_LDB(wx, M.wifxb);
                WAVE_FENCE();
                f32x4 d0[2] = {w16_splat(0.f), w16_splat(0.f)}, d1[2] = {w16_splat(0.f), w16_splat(0.f)};
                w16_mm_bf(wh, Bp0[0], Bp0[1], Bp0[2], d0);
                w16_mm_bf(wx, xqb0[0], xqb0[1], xqb0[2], d0);
                if (two) { w16_mm_bf(wh, Bp1[0], Bp1[1], Bp1[2], d1); w16_mm_bf(wx, xqb1[0], xqb1[1], xqb1[2], d1); }
                W16_LDB(wh, M.wjfhb);
                W16_LDB(wx, M.wjfxb);
                if (cat0) { w16_st(Pl + n16 * EPNN_PST + fo, d0[0]); w16_st(Pl + n16 * EPNN_PST + 16 + fo, d0[1]); }
                if (own1) { w16_st(Pl + col1 * EPNN_PST + fo, d1[0]); w16_st(Pl + col1 * EPNN_PST + 16 + fo, d1[1]); }
                d0[0] = w16_splat(0.f); d0[1] = w16_splat(0.f); d1[0] = w16_splat(0.f); d1[1] = w16_splat(0.f);
                w16_mm_bf(wh, Bp0[0], Bp0[1], Bp0[2], d0);
                w16_mm_bf(wx, xqb0[0], xqb0[1], xqb0[2], d0);
                if (two) { w16_mm_bf(wh, Bp1[0], Bp1[1], Bp1[2], d1); w16_mm_bf(wx, xqb1[0], xqb1[1], xqb1[2], d1); }
                if (cat0) { w16_st(Rl + n16 * EPNN_PST + fo, d0[0]); w16_st(Rl + n16 * EPNN_PST + 16 + fo, d0[1]); }
                if (own1) { w16_st(Rl + col1 * EPNN_PST + fo, d1[0]); w16_st(Rl + col1 * EPNN_PST + 16 + fo, d1[1]); }
            } else {
                constexpr int KS = FOLD ? KU + EPNN_XS : EPNN_XS + 12;
                constexpr int SK = FOLD ? KU - 1 + EPNN_XS : EPNN_XS - 1;    // the last xq step
                float wa[2][KS], wb[2][KS], in0[KS], in1[KS];
                if (FOLD) {
#pragma unroll
                    for (int s = 0; s < KU; ++s) { in0[s] = B0[s >> 2][s & 3]; in1[s] = B1[s >> 2][s & 3]; }
#pragma unroll
                    for (int s = 0; s < EPNN_XS; ++s) { in0[KU + s] = xq0[s]; in1[KU + s] = xq1[s]; }
                } else {
#pragma unroll
                    for (int s = 0; s < EPNN_XS; ++s) { in0[s] = xq0[s]; in1[s] = xq1[s]; }
#pragma unroll
                    for (int s = 0; s < 12; ++s) { in0[EPNN_XS + s] = hk0[s >> 2][s & 3]; in1[EPNN_XS + s] = hk1[s >> 2][s & 3]; }
                }
                W16_LD(wa, FOLD ? M.wif : M.wi, 2, KS);
                W16_LD(wb, FOLD ? M.wjf : M.wj, 2, KS);
                WAVE_FENCE();
                f32x4 d0[2] = {w16_splat(0.f), w16_splat(0.f)}, d1[2] = {w16_splat(0.f), w16_splat(0.f)};
                w16_mm_skip<2, KS, SK>(wa, in0, d0, xs3);
                if (two) w16_mm_skip<2, KS, SK>(wa, in1, d1, xs3);
                if (cat0) { w16_st(Pl + n16 * EPNN_PST + fo, d0[0]); w16_st(Pl + n16 * EPNN_PST + 16 + fo, d0[1]); }
                if (own1) { w16_st(Pl + col1 * EPNN_PST + fo, d1[0]); w16_st(Pl + col1 * EPNN_PST + 16 + fo, d1[1]); }
                d0[0] = w16_splat(0.f); d0[1] = w16_splat(0.f); d1[0] = w16_splat(0.f); d1[1] = w16_splat(0.f);
                w16_mm_skip<2, KS, SK>(wb, in0, d0, xs3);
                if (two) w16_mm_skip<2, KS, SK>(wb, in1, d1, xs3);
                if (cat0) { w16_st(Rl + n16 * EPNN_PST + fo, d0[0]); w16_st(Rl + n16 * EPNN_PST + 16 + fo, d0[1]); }
                if (own1) { w16_st(Rl + col1 * EPNN_PST + fo, d1[0]); w16_st(Rl + col1 * EPNN_PST + 16 + fo, d1[1]); }
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
            wave_sync_all();
            if (t < 2) WAVE_STAMP();   // EPN P, R
            {
                // one column per UNORDERED near pair, 16 pairs per column block.  Software pipeline: the pair record
                // (indices in LDS, weights in HBM) is fetched two blocks ahead, the gathered P / R rows and the e row one block ahead
                const int nblk = (np + 15) >> 4;
                struct Rec { int ij; float wi, wj; };
                struct Rows { float e[KE]; f32x4 pi_[2], rj_[2], pj_[2], ri_[2]; };
                auto load_rec = [&](int blk, Rec &r_) {
                    const int sl = blk * 16 + n16 < np ? blk * 16 + n16 : 0;
                    r_.ij = eij[sl];
                    if (FRONT) {
                        r_.wi = r_.wj = (r_.ij & 0x80) ? 1.f : 0.f;
                        r_.ij &= 0xFF7F;
                    } else {
                        r_.wi = A.pwi[p0 + sl];
                        r_.wj = A.pwj[p0 + sl];
                    }
                };
                auto load_rows = [&](int blk, const Rec &r_, Rows &w_) {
                    const int sl = blk * 16 + n16 < np ? blk * 16 + n16 : 0;
                    const int li = r_.ij & 0xFF, lj = r_.ij >> 8;
                    load_e1(sl, w_.e);          // the G term is computed in the block, it never goes through memory
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
                    f32x4 g[2] = {w16_splat(0.f), w16_splat(0.f)};       // G = We^T e of the 16 pairs (charge_gn.py:105, e block)
                    g_mm(w_.e, g);
                    const f32x4 ua = w16_relu((g[0] + w_.pi_[0]) + w_.rj_[0]), ub = w16_relu((g[1] + w_.pi_[1]) + w_.rj_[1]);
                    const f32x4 va = w16_relu((g[0] + w_.pj_[0]) + w_.ri_[0]), vb = w16_relu((g[1] + w_.pj_[1]) + w_.ri_[1]);
                    const float zu[8] = {ua[0], ua[1], ua[2], ua[3], ub[0], ub[1], ub[2], ub[3]};
                    const float zv[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                    f32x4 au[2] = {b2v[0], b2v[1]}, av[2] = {b2v[0], b2v[1]};
#ifdef EPNN_SWEEP_F32
                    WAVE_FENCE();                                          // (the element-wise work first: see the sweep's tile)
                    w16_mm<2, 8>(pb, zu, au);
                    w16_mm<2, 8>(pb, zv, av);
#else
                    u32x4 zu1, zu2, zu3, zv1, zv2, zv3;
                    w16_split3(zu, zu1, zu2, zu3);
                    w16_split3(zv, zv1, zv2, zv3);
                    WAVE_FENCE();                                          // (the element-wise work first: see the sweep's tile)
                    w16_mm_bf(pb, zu1, zu2, zu3, au);
                    w16_mm_bf(pb, zv1, zv2, zv3, av);
#endif
                    WAVE_FENCE();                                          // (... and the block's element-wise tail behind ALL of them)
                    float fd = 0.f;                                    // w3 . (relu(u) - relu(v)) over this lane's 8 features
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        const f32x4 t = w16_relu(au[rb]) - w16_relu(av[rb]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) fd = fmaf(w3[rb][r], t[r], fd);
                    }
                    const float d = 0.5f * w16_sumq(fd);               // charge_gn.py:116; all lanes take part
                    // entries with weight 0 are never written (they stay 0): a one-sided entry (j,i) of the dense
                    // front-end must not clear what the entry (i,j) wrote
                    // (ONE store instruction: lane group q = 0 writes what i receives, q = 1 what j receives)
                    const float wq = q == 0 ? r_.wi : r_.wj;
                    const int to = q == 0 ? li : lj, from = q == 0 ? lj : li;
                    if (q < 2 && valid && wq != 0.f) Dm[to * DMS + ((from & 3) << 3) + (from >> 2)] = q == 0 ? wq * d : -(wq * d);
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
            if (t + 1 < Te) { GW_LD(FRONT ? X.e[t + 1].we16 : X.e[t + 1].we, X.e[t + 1].we16b); }     // on its way during the charge update
            WAVE_FENCE();
            if (t < 2) WAVE_STAMP();   // EPN pair tiles
            // q_i += sum_j antisym_ij (charge_gn.py:118): lane (q, n16) adds columns j = q mod 4 of the rows of its two atoms
            {
                float dq0 = 0.f, dq1 = 0.f;
                // (entries of columns >= n are never written: they add exact zeros, the order of the real ones is j = q, q + 4, ...)
                const unsigned r0a = w16_lds_addr(Dm + (cat0 ? n16 : 0) * DMS) + 32u * (unsigned)q, r1a = w16_lds_addr(Dm + (cat1 ? col1 : 0) * DMS) + 32u * (unsigned)q;
                const f32x4 u0 = w16_lds_ld4(r0a), u1 = w16_lds_ld4(r1a);
#pragma unroll
                for (int k = 0; k < 4; ++k) { dq0 += u0[k]; dq1 += u1[k]; }
                if (two) {
                    const f32x4 v0 = w16_lds_ld4(r0a + 16u), v1 = w16_lds_ld4(r1a + 16u);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { dq0 += v0[k]; dq1 += v1[k]; }
                }
                dq0 = w16_sumq(dq0);
                dq1 = w16_sumq(dq1);
#pragma unroll
                for (int s = 0; s < EPNN_XS; ++s)
                    if (s == qs && q == ql) { xq0[s] += cat0 ? dq0 : 0.f; xq1[s] += cat1 ? dq1 : 0.f; }
                if constexpr (CHB && FOLD) {               // (the same sums in every lane: lane group 0 holds the charge's slot)
                    qc0 += cat0 ? dq0 : 0.f;
                    qc1 += cat1 ? dq1 : 0.f;
                    xq_charge();
                }
            }
            wave_sync_lds();
            if (t < 2) WAVE_STAMP();   // charge update
        }
#pragma unroll
        for (int s = 0; s < EPNN_XS; ++s)
            if (s == qs && q == ql) {
                if (cat0) A.q_out[a0 + n16] = xq0[s];
                if (own1) A.q_out[a0 + col1] = xq1[s];
            }
    }
    if (FRONT && A.handoff && lane == 0) wave_handoff(A, np);
    WAVE_STAMP();
#ifdef EPNN_STAMPS
    if (lane == 0 && A.stamps) {
        // where the wavefront ran: HW_ID (wave slot, SIMD, CU, shader array / engine) and the XCD (tools/wave_clocks.py)
        A.stamps[(size_t)blockIdx.x * 64 + 61] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |
                                                 ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32);
        A.stamps[(size_t)blockIdx.x * 64 + 59] = rt0;
        A.stamps[(size_t)blockIdx.x * 64 + 60] = __builtin_amdgcn_s_memrealtime();
        A.stamps[(size_t)blockIdx.x * 64 + 62] = (unsigned long long)nstamp;
        A.stamps[(size_t)blockIdx.x * 64 + 63] = ((unsigned long long)n << 32) | (unsigned)np;
    }
#endif
    (void)Tg;
}
