// Training step, fused along the pair rows (reference charge_gn.py:393-402; forward :56-75 and :87-119).
//
// One molecule padded to N = 41 is 1681 pair rows and ~2 Gflop per step: every kernel of the layer-by-layer path
// (epnn_train.hip.h) is a short dependent chain on a few dozen wavefronts and lasts ~9 us whatever it computes (kernel
// trace in DESIGN.md section 6b).  Here ONE workgroup owns atom i of a molecule and runs the whole pair MLP over its N
// partner rows -- both orders of the pair for the pass network -- with the rows [a_i | a_j | e_ij] (charge_gn.py:62-66,
// 101-108) assembled in LDS instead of HBM; the backward kernel does the same and leaves one block of weight-gradient
// partials per workgroup, summed in a fixed order by one launch at the end of the step.  2T + 2T + 1 launches instead of ~340.
// The Dense layers and every weight gradient run on the matrix pipe and each launch is laid out for latency (k_tf_pair_fwd,
// k_tb_pair_bwd_mm: HISTORY.md section 6b has the measurements that asked for each change; the scalar-FMA twins of rounds 2-4 were
// removed in round 5 -- the layer-by-layer kernels of epnn_train.hip.h are the second implementation the tests compare).
// All sums have a fixed order: gradients are bit-reproducible.
#pragma once
#include "epnn_host.h"

#define EPNN_TF_NMAX 96          // LDS of the backward kernel: 372 N + 2 k floats
#define EPNN_TF_NT 512           // threads of a pair workgroup: 16 row groups x 32 outputs
#define EPNN_TF_NG (EPNN_TF_NT / 32)
#define EPNN_TF_FMAX 60         // matrix-pipe layers: F = nx + 49 <= 60 (15 K steps of 4 per atom block)

struct TfPair {                  // one sweep of a pair MLP (message network of GNN step t / pass network of EPN step t)
    const float *x, *h, *q;      // the step's per-atom inputs: x [BN][nx], h [BN][48], q [BN]   (a = [x | h | q])
    const float *e;              // [R][48]
    const float *theta;          // flat parameters; the MLP's six tensors start at these offsets
    int oW1, ob1, oW2, ob2, oW3, ob3;
    float *H1, *H2;              // post-activation hidden layers [dir][R][32] (dir 1 = rows [a_j | a_i | e_ij], pass network only)
    float *M;                    // message network: summed messages [BN][32]         (charge_gn.py:70)
    const float *wgt;            // pass network: mask * is_near [R]                   (charge_gn.py:90-94,116)
    float *qn;                   // pass network: q + sum_j 0.5 (f_ij - f_ji) wgt [BN] (charge_gn.py:116-118)
    int N, nx;
    // backward
    const float *dU0, *nm;       // message network: gradient of the update MLP's input [BN][80], node mask [BN]
    float *gq;                   // pass network: gradient of the loss w.r.t. the step's output charges [BN]
    float *dz1;                  // gradient at the first layer's pre-activation [dir][R][32]
    float *part;                 // weight-gradient partials [BN][Pm], Pm = D*32 + 32 + 1024 + 32 + 32*O + O (parameter order)
    float *gacc;                 // atoms kernel: pass network gfeat [BN][48] (+=), message network gh [BN][48] (=)
    // ---- launches folded into their neighbours (round 3): the step is a chain of dependent launches of ~10 us each, so
    // every small kernel between two pair sweeps costs as much as a sweep
    const float *mask;           // forward, first step of a stack: [B][N][N]; the message sweep derives the node masks (nm_w),
    float *nm_w, *wgt_w;         //   the pass sweep the pair weights mask * is_near (wgt_w), and every later launch reads them
    float tol;
    const float *y;              // forward, last pass step: labels -> predictions (pred) and loss terms (lterm) per atom;
    float *pred, *lterm;         //   backward, first pass step (first != 0): gq = -2 (y - pred)
    long long *step_p;           // forward, first message sweep of a replayed hipGraph that also takes the optimizer step: counts the step
    float *out_h;                // forward, last pass step: page-locked host memory for the same two (or null): no download after the step
    int first;                   // backward: this is the first launch of its stack
    // backward: the "atoms" stage of the PREVIOUS backward launch runs as this launch's prologue (k_tb_atoms' arithmetic)
    int pmode;                   // -1: nothing; 1: the previous launch was a pass sweep; 0: a message sweep
    int poW1;                    // that sweep's first Dense
    int pfirst;                  // pmode 1: that sweep was the first one (gfeat starts from it instead of adding to it)
    const float *pdz1;           // that sweep's dz1 (the sweeps alternate between two buffers)
    float *gfeat, *gh, *gqv;     // [BN][48], [BN][48], [BN]
    // matrix-pipe backward (k_tb_pair_bwd_mm): `nsplit` workgroups per atom share its weight-gradient jobs (one molecule is 41
    // workgroups on 256 CUs); what a launch updates in place in the scalar kernel is read from the previous launch's copy here
    int nsplit, natoms;                  // natoms = B N (the grid is padded: see the kernel)
    // coordinate entry (epnn_train_step_xyz): molecule b has moff[b+1] - moff[b] real atoms, the other slots of its N are exact
    // zeros in every input.  The matrix-pipe kernels then leave a padded atom's workgroup at once (it would compute zeros for
    // N rows: half of the 41 slots of an average `mixed` molecule), take the padded partners' h and q as the zeros they are
    // (nobody writes them any more), and the reduction skips the padded atoms' partials (`real`).  Null: every slot is computed.
    const int *moff;
    const int *real;
    const float *gfeat_r, *gq_r, *dU0_r; // the previous launch's gfeat / gq / dU0 (this one writes gfeat / gqv / U.dU0)
    const float *rs_r;                   // [BN][2][32] the previous sweep's row sums of dz1 (listed, swapped rows of each atom)
    float *rs_w;
#ifdef EPNN_TF_CLOCKS
    unsigned long long *clk;     // development build (tools/train_clocks.py): wall_clock64 (100 MHz) of workgroup 0 at the phase boundaries
#endif
};
#ifdef EPNN_TF_CLOCKS
#define TF_CLK_T(k, t) do { if (A.clk && blockIdx.x == 0 && threadIdx.x == (t)) A.clk[k] = wall_clock64(); } while (0)
#define TF_CLK(k) TF_CLK_T(k, 0)
#define TF_CYC(k) do { if (A.clk && blockIdx.x == 0 && threadIdx.x == 0) A.clk[k] = __builtin_readcyclecounter(); } while (0)
#else
#define TF_CYC(k) do { } while (0)
#define TF_CLK_T(k, t) do { } while (0)
#define TF_CLK(k) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------- update MLP arguments
struct TfUpd {
    const float *h, *M, *nm, *theta;          // h [BN][48] (input of the step), M [BN][32]
    int oW0, ob0, oW1, ob1, oW2, ob2;
    float *U0, *U1, *U2, *hn;                 // [BN][80], [BN][32], [BN][32], [BN][48] = update(...) * nm   (charge_gn.py:71-74)
    const float *gh;                          // backward: gradient w.r.t. hn [BN][48]
    float *dU0;                               // [BN][80]
    float *part;                              // [BN][Pu], Pu = 80*32 + 32 + 32*32 + 32 + 32*48 + 48
};
#define EPNN_TF_PU (80 * 32 + 32 + 32 * 32 + 32 + 32 * 48 + 48)


// Matrix-pipe helpers (v_mfma_f32_16x16x4_f32).  Lane (q, x) = (lane >> 4, lane & 15) holds A[m = x][k = q] and B[k = q][n = x];
// accumulator register r of that lane is D[m = 4 q + r][n = x].  Here a COLUMN n is a pair row and m an output feature, so a
// 32-feature vector of a row is two row blocks rb x four registers: feature 16 rb + 4 q + r.
__device__ __forceinline__ f32x4 tm_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 tm_relu(f32x4 v) { return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }
__device__ __forceinline__ f32x4 tm_ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void tm_st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
// four consecutive parameters: the flat parameter vector is only 4-byte aligned at a tensor's start (the pass networks end in a
// 32 x 1 Dense + 1 bias), global_load_dwordx4 takes any dword address
__device__ __forceinline__ void tf_wave_sync() {         // LDS written by this wavefront is read by this wavefront: order only
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ f32x2 tm_ld2u(const float *p) { return *reinterpret_cast<const f32x2u *>(p); }
__device__ __forceinline__ f32x4 tm_ld4u(const float *p) { return *reinterpret_cast<const f32x4u *>(p); }

// Staging loops (tf_stage_w below): a few loads of a thread are in flight before its first LDS store (a plain loop would wait for every
// load in front of its LDS write: the trip counts are run-time values, the compiler does not overlap the iterations); the loads are
// unconditional, of a clamped index.
#define EPNN_TF_SD 4

// ---------------------------------------------------------------------------------------------- forward, pair MLP
// MODE 0: message network (out_dim 32, summed over ALL N partners); MODE 1: pass network (out_dim 1, both orders).
// MODE 0 with U.theta set also runs the update MLP of its atom (charge_gn.py:71-74: it needs only the atom's own h and
// summed message), which used to be a launch of its own per step.
//
// MM: the two hidden layers on the matrix pipe.  A wavefront owns 16 partner rows of one order of the pair (job = tile x order;
// N = 41 is 3 or 6 jobs on the workgroup's 8 wavefronts) and runs [a_i | a_j | e_ij] W1 as 2 F / 4 + 12 K steps of
// v_mfma_f32_16x16x4_f32 with its W1 / W2 fragments read from L2 straight into registers -- issued before the rows are staged, so
// the two latencies overlap and W1 (21 KB, the largest staging) never goes through LDS -- then layer 2 from the accumulators as
// they stand.
template <int MODE>
__global__ __launch_bounds__(EPNN_TF_NT) void k_tf_pair_fwd(TfPair A, TfUpd U) {
    extern __shared__ __attribute__((aligned(16))) float tf_sm[];
    constexpr bool MM = true;                 // (the scalar-FMA twin of rounds 2-4 is gone: the layer-by-layer kernels are the second implementation)
    const int N = A.N, nx = A.nx, F = nx + 49, D = 2 * F + 48, FS = F | 1;
    const int bi = blockIdx.x, b = bi / N, i = bi - b * N;
    const int tid = threadIdx.x, o = tid & 31, g = tid >> 5;
    constexpr int ND = MODE ? 2 : 1;
    TF_CLK(0);
    if (MODE == 0 && A.step_p && bi == 0 && tid == 0) *A.step_p += 1;
    const int nreal = (MM && A.moff) ? A.moff[b + 1] - A.moff[b] : N;       // real atoms of this molecule (MM: see TfPair::moff)
    if (MM && i >= nreal) {
        // a padded slot: q = 0, every transfer weight 0, node mask 0 -- its outputs are zeros nobody reads, except the last
        // pass step's prediction / loss term
        if (MODE == 1 && tid == 0 && A.y) {
            A.pred[bi] = 0.f;
            A.lterm[bi] = A.y[bi] * A.y[bi];
            if (A.out_h) {
                A.out_h[bi] = A.y[bi] * A.y[bi];
                A.out_h[(size_t)gridDim.x + bi] = 0.f;
            }
        }
        return;
    }
    // ---- MM: this wavefront's weight fragments (every job of a wavefront has the same order of the pair: 8 % ND == 0).
    // Row m of output tile rb is FEATURE 2 m + rb: a lane's two tiles are neighbours in memory (one 8-byte read per K step),
    // and its eight accumulators are features 8 lq .. 8 lq + 7 of its row.  W1 goes through LDS once per workgroup (float4
    // staging beside the rows): as registers of every job wavefront it was 27 KB per wavefront through the CU's 64 B/clk
    // vector memory path -- six copies for the pass network, 1 us of its launch.
    constexpr int KA = MM ? (EPNN_TF_FMAX + 3) / 4 : 1;       // K steps of an atom block
    const int wave = tid >> 6, lq = (tid >> 4) & 3, lx = tid & 15;
    const int njobs = ((N + 15) / 16) * ND, jdir = wave % ND;
    f32x2 w2f[MM ? 8 : 1];
    f32x4 b2v[2];
    if (MM && wave < njobs) {
        const float *w2 = A.theta + A.oW2;
#pragma unroll
        for (int s = 0; s < 8; ++s) w2f[s] = tm_ld2u(w2 + (8 * lq + s) * 32 + 2 * lx);     // K step s pairs lane lq with input feature 8 lq + s
        b2v[0] = tm_ld4u(A.theta + A.ob2 + 8 * lq);
        b2v[1] = tm_ld4u(A.theta + A.ob2 + 8 * lq + 4);
    }
    // ---- MM: the a_i term of the first Dense is the same for every row of a workgroup and order of the pair: b1 + a_i W1[own
    // block] is formed ONCE (wavefront 6 for order 0, wavefront 7 for order 1 of the pass network; lane = (k mod 8, float4 of
    // features), then the eight partials) while the job wavefronts run their rows' K steps, and added before the ReLU -- 30 of a
    // job's 100 MFMAs were this one vector recomputed per 16 rows
    const int bw = !MM ? -1 : MODE ? wave - 6 : (wave == 6 ? 0 : -1);
    f32x4 wb[MM ? 8 : 1];
    float b1v = 0.f;
    if (MM && bw >= 0) {
        const int kq = (tid & 63) >> 3, f4 = tid & 7;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = kq + 8 * u;
            const f32x4 t = tm_ld4u(A.theta + A.oW1 + ((bw ? F : 0) + (k < F ? k : 0)) * 32 + 4 * f4);
            wb[u] = k < F ? t : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        b1v = A.theta[A.ob1 + o];
    }
    // ---- MM, message network: the LAST wavefront (never a layer job: N <= 96 is at most 6 tiles) runs the atom's serial tail --
    // column sums, third Dense, update MLP -- with wavefront-level hand-offs, its weights in registers since the kernel's start
    // (a stage that loads its weights when it starts pays a round trip; the tail was 2.9 us of a 9.2 us launch).  Each sum over
    // k is split between the two half-wavefronts (lane = (half, o)).
    constexpr bool TAIL = MM && MODE == 0;
    const bool updf = U.theta != nullptr;
    float w3c[TAIL ? 16 : 1], w0c[TAIL ? 40 : 1], w1c[TAIL ? 16 : 1], w2c[TAIL ? 32 : 1], tb3 = 0.f, tb0 = 0.f, tb1 = 0.f, tb2 = 0.f, hpre = 0.f, nmpre = 0.f;
    if (TAIL && wave == 7) {
        const int half = (tid >> 5) & 1, oc = (tid & 63) < 48 ? (tid & 63) : 0;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) w3c[kk] = A.theta[A.oW3 + (16 * half + kk) * 32 + o];
        tb3 = A.theta[A.ob3 + o];
        if (updf) {
#pragma unroll
            for (int kk = 0; kk < 40; ++kk) w0c[kk] = U.theta[U.oW0 + (40 * half + kk) * 32 + o];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) w1c[kk] = U.theta[U.oW1 + (16 * half + kk) * 32 + o];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) w2c[kk] = U.theta[U.oW2 + kk * 48 + oc];
            tb0 = U.theta[U.ob0 + o];
            tb1 = U.theta[U.ob1 + o];
            tb2 = U.theta[U.ob2 + oc];
            hpre = U.h[(size_t)bi * 48 + oc];
            if (!A.mask) nmpre = U.nm[bi];
        }
    }
    // ---- MM, pass network: the pair weights of this atom's rows are formed by the last two wavefronts; what they need from
    // global memory is loaded here, not behind the barrier
    float wpre = 0.f;
    if (MM && MODE == 1 && tid >= EPNN_TF_NT - 128) {
        const int j = min(tid - (EPNN_TF_NT - 128), N - 1);
        wpre = A.mask ? A.mask[(size_t)bi * N + j] : A.wgt[(size_t)bi * N + j];
    }
    float *As = tf_sm;                        // [N][FS]   a_j of every atom of the molecule
    float *Es = As + N * FS;                  // [N][49]   e_ij of this atom's rows
    float *W1s = Es + N * 49;                 // [D][32]
    float *H1s = W1s + D * 32;                // [ND][N][33]
    float *H2s = H1s + ND * N * 33;           // [ND][N][33]
    float *red = H2s + ND * N * 33;           // [NG][32] + [32]  /  [2][N]
    const size_t a0 = (size_t)b * N, rowbase = (size_t)bi * N;
    const size_t dstride = (size_t)gridDim.x * N * 32;
    // MM, pass network: a 16-row tile whose partners all carry transfer weight 0 (mask = 0 on the stack's first launch, mask * is_near
    // after it: the padding of a molecule, whoever padded it) contributes nothing forward and gets zero gradients: flagged here by the
    // wavefronts that hold the weights, before the staging barrier, and stored as zeros instead of computed (H1s is free in this form)
    int *deadt = reinterpret_cast<int *>(H1s);
    if (MM && MODE == 1 && tid >= EPNN_TF_NT - 128) {
        const int jj = tid - (EPNN_TF_NT - 128), ln = tid & 63;
        const unsigned long long nzb = __ballot(jj < N && wpre != 0.f);
        if (ln < 4) deadt[(jj >> 6) * 4 + ln] = ((nzb >> (16 * ln)) & 0xFFFFull) == 0ull;
    }
    if (MM) {
        // both arrays' loads of a round are in flight before the first LDS write (N = 41 is one round)
        const int nA = N * F, nE = N * 48;
        const int w0 = MODE ? 0 : F, nW = (D - w0) * 8;              // W1 rows the jobs read: the partner blocks and the edge block
        const int rounds = max(max((nA + 5 * EPNN_TF_NT - 1) / (5 * EPNN_TF_NT), (nE + 4 * EPNN_TF_NT - 1) / (4 * EPNN_TF_NT)),
                               (nW + 3 * EPNN_TF_NT - 1) / (3 * EPNN_TF_NT));
        for (int rd = 0; rd < rounds; ++rd) {
            float va[5], ve[4];
            f32x4 vw[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) vw[u] = tm_ld4u(A.theta + A.oW1 + w0 * 32 + 4 * min((3 * rd + u) * EPNN_TF_NT + tid, nW - 1));
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int ia = min((5 * rd + u) * EPNN_TF_NT + tid, nA - 1);
                const int j = ia / F, k = ia - j * F;
                const size_t at = a0 + j;
                const float t = *(k < nx ? A.x + at * nx + k : (k < nx + 48 ? A.h + at * 48 + (k - nx) : A.q + at));
                va[u] = (j >= nreal && k >= nx) ? 0.f : t;               // a padded partner's h and q (x is zero as given)
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) ve[u] = A.e[rowbase * 48 + min((4 * rd + u) * EPNN_TF_NT + tid, nE - 1)];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int ia = (5 * rd + u) * EPNN_TF_NT + tid;
                if (ia < nA) {
                    const int j = ia / F;
                    As[j * FS + (ia - j * F)] = va[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ie = (4 * rd + u) * EPNN_TF_NT + tid;
                if (ie < nE) {
                    const int j = ie / 48;
                    Es[j * 49 + (ie - j * 48)] = ve[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int iw = (3 * rd + u) * EPNN_TF_NT + tid;
                if (iw < nW) tm_st4(W1s + w0 * 32 + 4 * iw, vw[u]);
            }
        }
    }
    float *wl = red + EPNN_TF_NG * 32 + 32 + 2 * N;            // [N] pair weights of this atom's rows (pass network)
    if (MODE == 0 && A.mask && tid < 64) {
        // node mask of this atom (charge_gn.py:59): clip(sum_j mask[j][i], 0, 1); what k_t_nodemask did in a launch of its own
        float sm = 0.f;
        for (int j = tid; j < N; j += 64) sm += A.mask[(a0 + j) * N + i];
        for (int dd = 32; dd >= 1; dd >>= 1) sm += __shfl_xor(sm, dd, 64);
        if (tid == 0) {
            const float v = fminf(fmaxf(sm, 0.f), 1.f);
            A.nm_w[bi] = v;
            wl[0] = v;
        }
    }
    __syncthreads();
    TF_CLK(1);
    if (MM && MODE == 1) {
        // pair weights mask * is_near (charge_gn.py:90-94,116): from the staged e rows on the stack's first launch, else as stored
        const int j = tid - (EPNN_TF_NT - 128);
        if (j >= 0 && j < N) {
            float w = wpre;
            if (A.mask) {
                float mx = 0.f;
                for (int k = 0; k < 48; ++k) mx = fmaxf(mx, Es[j * 49 + k]);
                w = mx > A.tol ? wpre : 0.f;
                A.wgt_w[rowbase + j] = w;
            }
            wl[j] = w;
        }
    }
    if (MM) {
        const float *ai = As + i * FS;
        float *pb = red;                          // [2][8][32] partials of the base vectors; the vectors end up in [d][0][.]
        bool first = true;
        for (int job = wave; first || job < njobs; job += EPNN_TF_NT / 64) {
            const bool has = job < njobs;
            const int j = (job / ND) * 16 + lx;
            const bool jv = has && j < N;
            // pass network: a tile of padded partners only (TfPair::moff) -- every transfer of its rows has weight 0 and its rows
            // get zero gradients: zeros are stored for them, nothing is computed (4 jobs instead of 6 for an average molecule: one
            // per SIMD)
            const bool dead = MODE == 1 && has && ((job / ND) * 16 >= nreal || deadt[job / ND] != 0);
            const float *aj = As + (jv ? j : 0) * FS, *ej = Es + (jv ? j : 0) * 49;
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (first && bw >= 0) {
                const int lane = tid & 63, kq = lane >> 3, f4 = lane & 7;
                f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = kq + 8 * u;
                    const float a = ai[k < F ? k : 0];
#pragma unroll
                    for (int c = 0; c < 4; ++c) p[c] = fmaf(a, wb[u][c], p[c]);             // wb is zero beyond F
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) pb[(bw * 8 + kq) * 32 + 4 * f4 + c] = p[c];
                tf_wave_sync();
                if (lane < 32) {
                    float v = b1v;
#pragma unroll
                    for (int g8 = 0; g8 < 8; ++g8) v += pb[(bw * 8 + g8) * 32 + lane];
                    pb[bw * 256 + lane] = v;          // a lane reads and writes its own column only
                }
            }
            if (has && !dead) {
                // order 0 rows are [a_i | a_j | e_ij]: the partner a_j meets block 1 of W1; order 1 rows are [a_j | a_i | e_ij]: block 0
                const float *wjs = W1s + (jdir ? 0 : F) * 32 + 2 * lx, *wes = W1s + 2 * F * 32 + 2 * lx;
#pragma unroll
                for (int s = 0; s < KA; ++s) {
                    const int k = 4 * s + lq, kc = k < F ? k : 0;
                    const float t = aj[kc], vj = (k < F && jv) ? t : 0.f;
                    const f32x2 w = *reinterpret_cast<const f32x2 *>(wjs + kc * 32);
                    acc[0] = tm_mfma(w[0], vj, acc[0]);
                    acc[1] = tm_mfma(w[1], vj, acc[1]);
                }
#pragma unroll
                for (int s = 0; s < 12; ++s) {
                    const float t = ej[4 * s + lq], ve = jv ? t : 0.f;
                    const f32x2 w = *reinterpret_cast<const f32x2 *>(wes + (4 * s + lq) * 32);
                    acc[0] = tm_mfma(w[0], ve, acc[0]);
                    acc[1] = tm_mfma(w[1], ve, acc[1]);
                }
            }
            if (first) {
                __syncthreads();                      // the base vectors are in LDS (every wavefront passes here exactly once)
                first = false;
            }
            if (has) {
                const float *vb = pb + jdir * 256 + 8 * lq;
                f32x4 d2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                if (!dead) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[0][c] = fmaxf(acc[0][c] + vb[2 * c], 0.f);
                        acc[1][c] = fmaxf(acc[1][c] + vb[2 * c + 1], 0.f);
                        d2[0][c] = b2v[c >> 1][2 * (c & 1)];              // feature 8 lq + 2 c
                        d2[1][c] = b2v[c >> 1][2 * (c & 1) + 1];          // feature 8 lq + 2 c + 1
                    }
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        d2[0] = tm_mfma(w2f[s][0], acc[s & 1][s >> 1], d2[0]);
                        d2[1] = tm_mfma(w2f[s][1], acc[s & 1][s >> 1], d2[1]);
                    }
                    d2[0] = tm_relu(d2[0]);
                    d2[1] = tm_relu(d2[1]);
                }
                if (jv) {
                    float *l2 = H2s + (jdir * N + j) * 33 + 8 * lq;
                    if (A.H1) {                               // (null when nothing will run backward: the dense entry's small calls)
                        float *g1 = A.H1 + jdir * dstride + (rowbase + j) * 32 + 8 * lq, *g2 = A.H2 + jdir * dstride + (rowbase + j) * 32 + 8 * lq;
                        tm_st4(g1, f32x4{acc[0][0], acc[1][0], acc[0][1], acc[1][1]});
                        tm_st4(g1 + 4, f32x4{acc[0][2], acc[1][2], acc[0][3], acc[1][3]});
                        tm_st4(g2, f32x4{d2[0][0], d2[1][0], d2[0][1], d2[1][1]});
                        tm_st4(g2 + 4, f32x4{d2[0][2], d2[1][2], d2[0][3], d2[1][3]});
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        l2[2 * c] = d2[0][c];
                        l2[2 * c + 1] = d2[1][c];
                    }
                }
            }
        }
    }
    __syncthreads();
    TF_CLK(2);
    if (TAIL) {
        if (wave != 7) return;
        const int lane = tid & 63, half = lane >> 5;
        float *S = red, *u0 = red + 32, *u1 = red + 112, *u2 = red + 144;
        // sum_j (H2_j W3 + b3) = (sum_j H2_j) W3 + N b3: column sums (even rows, odd rows, then the two), then one 32x32 product
        float cs = 0.f;
#pragma unroll 8
        for (int j = half; j < N; j += 2) cs += H2s[j * 33 + o];
        cs += __shfl_xor(cs, 32, 64);
        if (half == 0) S[o] = cs;
        tf_wave_sync();
        float mo = 0.f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) mo = fmaf(S[16 * half + kk], w3c[kk], mo);
        mo += __shfl_xor(mo, 32, 64);
        mo += (float)N * tb3;
        const float nm = updf ? (A.mask ? wl[0] : nmpre) : 0.f;
        if (half == 0) {
            if (A.M) A.M[(size_t)bi * 32 + o] = mo;
            if (updf) {
                const float v = mo * nm;
                u0[48 + o] = v;
                if (U.U0) U.U0[(size_t)bi * 80 + 48 + o] = v;
            }
        }
        TF_CLK_T(3, EPNN_TF_NT - 64);
        if (!updf) return;
        if (lane < 48) {
            const float v = hpre * nm;
            u0[lane] = v;
            if (U.U0) U.U0[(size_t)bi * 80 + lane] = v;
        }
        tf_wave_sync();
        float z = 0.f;
#pragma unroll
        for (int kk = 0; kk < 40; ++kk) z = fmaf(u0[40 * half + kk], w0c[kk], z);
        z += __shfl_xor(z, 32, 64);
        z = fmaxf(z + tb0, 0.f);
        if (half == 0) {
            u1[o] = z;
            if (U.U1) U.U1[(size_t)bi * 32 + o] = z;
        }
        tf_wave_sync();
        z = 0.f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) z = fmaf(u1[16 * half + kk], w1c[kk], z);
        z += __shfl_xor(z, 32, 64);
        z = fmaxf(z + tb1, 0.f);
        if (half == 0) {
            u2[o] = z;
            if (U.U2) U.U2[(size_t)bi * 32 + o] = z;
        }
        tf_wave_sync();
        if (lane < 48) {
            z = tb2;
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) z = fmaf(u2[kk], w2c[kk], z);
            U.hn[(size_t)bi * 48 + lane] = z * nm;
        }
        TF_CLK_T(5, EPNN_TF_NT - 64);
        return;
    }
    // ---- pass network: layer 3 (linear, out_dim 1) and the antisymmetric sum over partners (charge_gn.py:110-118)
    {
        float *fs = red;                      // [2][N]
        for (int idx = tid; idx < 2 * N; idx += EPNN_TF_NT) {
            const float *hr = H2s + idx * 33;
            float f = A.theta[A.ob3];
            for (int k = 0; k < 32; ++k) f = fmaf(hr[k], A.theta[A.oW3 + k], f);
            fs[idx] = f;
        }
        __syncthreads();
        TF_CLK(3);
        if (MM && tid >= 64) return;
        float s = 0.f;
        if (MM) {                                 // one wavefront, a fixed butterfly, instead of a serial loop on one lane
            for (int j = tid; j < N; j += 64) s += 0.5f * (fs[j] - fs[N + j]) * wl[j];
#pragma unroll
            for (int dd = 32; dd >= 1; dd >>= 1) s += __shfl_xor(s, dd, 64);
        }
        if (tid == 0) {
            const float qn = A.q[bi] + s;
            A.qn[bi] = qn;
            if (A.pred && !A.y) A.pred[bi] = qn;      // inference: the charges, nothing else
            if (A.y) {                            // last step: prediction and loss term of this atom (charge_gn.py:397)
                const float d = A.y[bi] - qn;
                A.pred[bi] = qn;
                A.lterm[bi] = d * d;
                if (A.out_h) {                    // and straight into the caller's page-locked buffer: loss terms [BN] | predictions [BN]
                    A.out_h[bi] = d * d;
                    A.out_h[(size_t)gridDim.x + bi] = qn;
                }
            }
        }
        TF_CLK(5);
    }
}

// ---------------------------------------------------------------------------------------------- backward, pair MLP, matrix pipe
// The backward of one pair MLP, laid out for latency (clock stamps of one workgroup, tools/train_clocks.py: a scalar
// row-per-thread kernel spent 3.3 us in the prologue, 3.8 us staging, 2.6 us in the update MLP's backward, 2-3 us in dz2 / dz1 / column sums and
// 5.5-6.5 us in the weight-gradient loops, one after the other):
//   * wavefront 0 runs the whole serial chain -- prologue (the previous sweep's gradient at this atom), the update MLP's backward,
//     dM W3^T / df -- with wavefront-level LDS hand-offs, WHILE wavefronts 1-7 stage the rows (float4 rows of 36 / 52 floats);
//   * dz2 is formed in registers by the lane that needs it as the B operand of dz1 = dz2 W2^T (16 MFMAs per 16 rows);
//   * every weight gradient is a product over the pair rows on the matrix pipe: dW[k][o] = sum_rows X[row][k] dz[row][o], 16 weight
//     rows x 16 outputs per job, four K steps (16 pair rows) of LDS reads in flight per pass; the bias gradients are the row "1"
//     of a padded tile (no column-sum phase), the a_i block a constant A operand.
// Sums run in MFMA order instead of row order: fixed, so gradients stay bit-reproducible; they differ from the scalar kernel's by rounding.
#define EPNN_TB_RS 36            // LDS row of a 32-wide array: 16-byte aligned, bank = 4 row + column
#define EPNN_TB_ES 52            // LDS row of e_ij (48)
template <typename T, typename LD, typename ST>
__device__ __forceinline__ void tf_stage_w(int total, int t, int nth, LD &&ld, ST &&st) {
    for (int base = 0; base < total; base += EPNN_TF_SD * nth) {
        T v[EPNN_TF_SD];
#pragma unroll
        for (int u = 0; u < EPNN_TF_SD; ++u) {
            const int idx = base + u * nth + t;
            v[u] = ld(idx < total ? idx : total - 1);
        }
#pragma unroll
        for (int u = 0; u < EPNN_TF_SD; ++u) {
            const int idx = base + u * nth + t;
            if (idx < total) st(idx, v[u]);
        }
    }
}
// sum over `rows` pair rows of a(row) b(row) [+ a2(row) b2(row)] on the matrix pipe: lane (lq, lx) supplies A[m = lx] and B[n = lx] of
// row 4 s + lq; sixteen rows per pass, their LDS reads issued together
template <bool TWO, typename FA, typename FB, typename FA2, typename FB2>
__device__ __forceinline__ f32x4 tb_rows_mm(int rows, int lq, FA &&a, FB &&b, FA2 &&a2, FB2 &&b2) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < rows; r0 += 16) {
        float av[4], bv[4], av2[4], bv2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + 4 * u + lq;
            const bool ok = r < rows;
            const int rc = ok ? r : rows - 1;              // loads are unconditional (a select, not a branch, zeroes the padding rows)
            const float ta = a(rc);
            av[u] = ok ? ta : 0.f;
            bv[u] = b(rc);
            if (TWO) {
                const float ta2 = a2(rc);
                av2[u] = ok ? ta2 : 0.f;
                bv2[u] = b2(rc);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = tm_mfma(av[u], bv[u], acc);
            if (TWO) acc = tm_mfma(av2[u], bv2[u], acc);
        }
    }
    return acc;
}

template <int MODE>
__global__ __launch_bounds__(EPNN_TF_NT) void k_tb_pair_bwd_mm(TfPair A, TfUpd U) {
    extern __shared__ __attribute__((aligned(16))) float tf_sm[];
    const int N = A.N, nx = A.nx, F = nx + 49, D = 2 * F + 48, FS = F | 1;
    // The workgroups of an atom read the same rows: they sit on ONE XCD (consecutive block indices go round the eight XCDs, each
    // with an L2 of its own -- six workgroups of an atom spread over six XCDs fetched its rows from HBM / MALL six times, 15 MB per
    // launch at the moment every workgroup starts).  Block x runs on XCD x % 8; there it is the (x / 8)-th of that XCD's
    // (atom, share) pairs, the XCD's atoms being xcd, xcd + 8, ...; the grid is padded to eight equal parts.
    const int S = A.nsplit, BNa = A.natoms;
    const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3, bi = (kx / S) * 8 + xcd, sub = kx - (kx / S) * S;
    if (bi >= BNa) return;
    const int b = bi / N, i = bi - b * N;
    const int nreal = A.moff ? A.moff[b + 1] - A.moff[b] : N;
    if (i >= nreal) return;                   // a padded slot: every gradient it would form is zero (TfPair::moff); the reduction skips its partials
    const bool own = sub == 0;                // the workgroup of this atom that writes what is not a weight-gradient job
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane >> 4, lx = lane & 15;
    constexpr int ND = MODE ? 2 : 1, O = MODE ? 1 : 32, RS = EPNN_TB_RS, ES = EPNN_TB_ES;
    const int NR = ND * N;                    // pair rows of this workgroup: row d N + j is partner j in order d
    float *H1s = tf_sm;                       // [NR][RS]
    float *H2s = H1s + NR * RS;
    float *D1s = H2s + NR * RS;               // gradient at z1
    float *D2s = D1s + NR * RS;               // gradient at z2
    float *Es = D2s + NR * RS;                // [N][ES]
    float *vec = Es + N * ES;                 // dms [32] | vs [32] | - [32] | df [96]
    float *dms = vec, *vs = vec + 32, *dfs = vec + 96;
    float *ub = vec + 192;                    // update backward: u0 [80] | u1 [32] | u2 [32] | dh [48] | du2 [32] | du1 [32] | dU0 [80]
    float *psh = ub + 336;                    // [4][32] prologue: row / column sums of the previous sweep's dz1
    float *pp = psh + 128;                    // [8][2][32] prologue: the column sums' partials
    float *As = pp + 512;                     // [N][FS]
    const size_t a0 = (size_t)b * N, rowbase = (size_t)bi * N;
    const size_t dstride = (size_t)BNa * N * 32;
    const float *theta = A.theta;
    TF_CLK(0);
    TF_CYC(14);
    // W2 for dz1 = dz2 W2^T: A[m = k][kk = o], K steps in "acc" order o = 16 (s >> 2) + 4 lq + (s & 3) (two float4 of a row per lane)
    f32x4 w2f[2][2], w3v[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) w2f[rb][hh] = tm_ld4u(theta + A.oW2 + (16 * rb + lx) * 32 + 16 * hh + 4 * lq);
    if (MODE == 1) {
        w3v[0] = tm_ld4u(theta + A.oW3 + 4 * lq);
        w3v[1] = tm_ld4u(theta + A.oW3 + 16 + 4 * lq);
    }
    const bool upd = MODE == 0 && U.theta != nullptr;
    if (wave == 0) {
        // ================= the serial chain of this atom, one wavefront, no workgroup barrier.  Every load whose address does not
        // depend on the chain is issued FIRST, as float4 (a wavefront has 63 loads in flight at most: the ~230 scalar loads of
        // the prologue's and the update MLP's weights were four windows of it); a stage that loads its weights when it starts
        // pays a round trip (~0.7 us) per stage.
        const int l32 = lane & 31;
        f32x4 wP[8], wR[8], wA[12], wB[8];
        if (A.pmode >= 0) {
            const int kc = nx + (lane < 49 ? lane : 0);          // h part: k in [nx, nx+48), q part: k = nx + 48
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                wP[c] = tm_ld4u(theta + A.poW1 + kc * 32 + 4 * c);
                wR[c] = tm_ld4u(theta + A.poW1 + (F + kc) * 32 + 4 * c);
            }
        }
        if (upd) {
#pragma unroll
            for (int c = 0; c < 12; ++c) wA[c] = tm_ld4u(U.theta + U.oW2 + l32 * 48 + 4 * c);
#pragma unroll
            for (int c = 0; c < 8; ++c) wB[c] = tm_ld4u(U.theta + U.oW1 + l32 * 32 + 4 * c);
        }
        float wg0 = 0.f, wg1 = 0.f, u0a = 0.f, u0b = 0.f, u12 = 0.f, pre = 0.f, rsv = 0.f;
        const float nm = MODE == 0 ? A.nm[bi] : 0.f;
        if (MODE == 1) {
            wg0 = A.wgt[rowbase + min(lane, N - 1)];
            wg1 = A.wgt[rowbase + min(64 + lane, N - 1)];
        }
        if (upd) {
            u0a = U.U0[(size_t)bi * 80 + lane];
            u0b = U.U0[(size_t)bi * 80 + 64 + (lane & 15)];
            u12 = *(lane < 32 ? U.U1 + (size_t)bi * 32 + lane : U.U2 + (size_t)bi * 32 + lane - 32);
        }
        if (A.pmode == 1) {
            pre = *(lane < 48 ? A.gfeat_r + (size_t)bi * 48 + lane : A.gq_r + bi);
            if (lane < 48 && A.pfirst) pre = 0.f;
        } else if (A.pmode == 0) {
            pre = A.dU0_r[(size_t)bi * 80 + (lane < 48 ? lane : 0)] * nm;
        } else if (MODE == 0) {
            pre = U.gh[(size_t)bi * 48 + (lane < 48 ? lane : 0)];
        }
        if (A.pmode >= 0) rsv = A.rs_r[(size_t)bi * 64 + lane];
        float gq = 0.f, ghv = pre;
        if (MODE == 1 && A.first) gq = -2.f * (A.y[bi] - A.pred[bi]);           // charge_gn.py:397-398
        else if (MODE == 1 && A.pmode < 0) gq = A.gq_r[bi];
        // ---- prologue: the gradient that reaches a_i through the first Dense of the PREVIOUS sweep (k_tb_atoms' arithmetic).
        // It needs four sums of that sweep's dz1 over N rows each:
        //   0: listed rows (a, j), a is the first block;  1: listed rows (i, a), a is the second block;
        //   2: swapped rows (i, a) = [a_a | a_i | e], first block;  3: swapped rows (a, j) = [a_j | a_a | e], second block
        // 0 and 3 run over this atom's own rows: the previous launch left them (rs_r; the "1" rows of its weight-gradient
        // tiles).  1 and 2 run down a column of the molecule's rows: float4 loads, lane = (row group of 8, float4 of the row),
        // 2 x ceil(N / 8) loads per lane in flight at once, then the eight partials in a fixed order.
        if (A.pmode >= 0) {
            const int rg = lane >> 3, c4 = lane & 7;
            const bool two = A.pmode == 1;
            const float *pc = A.pdz1 + (a0 * N + i) * 32 + 4 * c4;
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
            for (int t0 = 0; t0 < nreal; t0 += 48) {
                f32x4 v0[6], v1[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int tc = min(t0 + 8 * u + rg, nreal - 1);
                    v0[u] = tm_ld4(pc + (size_t)tc * N * 32);
                    v1[u] = tm_ld4(pc + dstride + (size_t)tc * N * 32);
                }
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const bool ok = t0 + 8 * u + rg < nreal;          // (rows of padded atoms are exact zeros, or were never written)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        s0[c] += ok ? v0[u][c] : 0.f;
                        s1[c] += ok && two ? v1[u][c] : 0.f;
                    }
                }
            }
            tm_st4(pp + (rg * 2 + 0) * 32 + 4 * c4, s0);
            tm_st4(pp + (rg * 2 + 1) * 32 + 4 * c4, s1);
            tf_wave_sync();
            {
                // lane (half, o): half 0 holds sum 0 (rsv) and forms sum 1, half 1 holds sum 3 and forms sum 2; the first block's
                // gradient needs sP = 0 + 2, the second block's sR = 1 + 3: one exchange between the halves
                const int half = lane >> 5;
                float col = 0.f;
#pragma unroll
                for (int g8 = 0; g8 < 8; ++g8) col += pp[(g8 * 2 + half) * 32 + l32];
                const float row = half && !two ? 0.f : rsv;
                const float ocol = __shfl_xor(col, 32, 64), orow = __shfl_xor(row, 32, 64);
                if (half == 0) {
                    psh[2 * l32] = row + ocol;                            // sP[o]
                    psh[2 * l32 + 1] = col + orow;                        // sR[o]
                }
            }
            tf_wave_sync();
            TF_CLK(4);
            if (lane < 49) {
                float daP = 0.f, daR = 0.f;                               // two chains of 32 instead of one of 64
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const f32x4 sa = tm_ld4(psh + 8 * c), sb = tm_ld4(psh + 8 * c + 4);      // sP, sR of outputs 4 c .. 4 c + 3
                    daP = fmaf(sa[0], wP[c][0], daP);
                    daR = fmaf(sa[1], wR[c][0], daR);
                    daP = fmaf(sa[2], wP[c][1], daP);
                    daR = fmaf(sa[3], wR[c][1], daR);
                    daP = fmaf(sb[0], wP[c][2], daP);
                    daR = fmaf(sb[1], wR[c][2], daR);
                    daP = fmaf(sb[2], wP[c][3], daP);
                    daR = fmaf(sb[3], wR[c][3], daR);
                }
                const float da = daP + daR;
                if (A.pmode == 1) {
                    if (lane < 48) {
                        ghv = pre + da;
                        if (own) A.gfeat[(size_t)bi * 48 + lane] = ghv;
                        if (MODE == 0 && own) A.gh[(size_t)bi * 48 + lane] = ghv;     // the pass stack is done: the feature gradient enters the GNN
                    } else {
                        gq = pre + da;
                    }
                } else if (lane < 48) {
                    ghv = pre + da;
                    if (own) A.gh[(size_t)bi * 48 + lane] = ghv;
                }
            }
        }
        TF_CLK(5);
        if (MODE == 1) {
            if (A.pmode >= 0) gq = __shfl(gq, 48, 64);
            if (lane == 0 && own) A.gqv[bi] = gq;
            const float gqi = 0.5f * gq;
            if (lane < N) dfs[lane] = gqi * wg0;                                  // df_ij; the swapped row gets -df_ij
            if (lane + 64 < N) dfs[lane + 64] = gqi * wg1;
        } else {
            f32x4 w3r[8];
            if (upd) {
                float *u0 = ub, *u1 = ub + 80, *u2 = ub + 112, *dh = ub + 144, *du2 = ub + 192, *du1 = ub + 224, *dU = ub + 256;
                // the last layer's weights (dU0 = du1 W0^T: 80 outputs on 64 lanes) and W3 follow the prologue's into the registers it freed
                f32x4 wC[8], wD[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    wC[c] = tm_ld4u(U.theta + U.oW0 + lane * 32 + 4 * c);
                    wD[c] = tm_ld4u(U.theta + U.oW0 + (64 + (lane & 15)) * 32 + 4 * c);
                    w3r[c] = tm_ld4u(theta + A.oW3 + l32 * 32 + 4 * c);
                }
                u0[lane] = u0a;
                if (lane < 16) u0[64 + lane] = u0b;
                u1[lane] = u12;                                               // u1 | u2 are adjacent
                if (lane < 48) dh[lane] = ghv * nm;
                tf_wave_sync();
                if (lane < 32) {
                    float sa = 0.f, sb = 0.f;                                     // even / odd float4s: two chains
#pragma unroll
                    for (int c = 0; c < 12; c += 2) {
                        const f32x4 x0 = tm_ld4(dh + 4 * c), x1 = tm_ld4(dh + 4 * c + 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            sa = fmaf(x0[r], wA[c][r], sa);
                            sb = fmaf(x1[r], wA[c + 1][r], sb);
                        }
                    }
                    du2[lane] = u2[lane] > 0.f ? sa + sb : 0.f;
                }
                tf_wave_sync();
                if (lane < 32) {
                    float sa = 0.f, sb = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; c += 2) {
                        const f32x4 x0 = tm_ld4(du2 + 4 * c), x1 = tm_ld4(du2 + 4 * c + 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            sa = fmaf(x0[r], wB[c][r], sa);
                            sb = fmaf(x1[r], wB[c + 1][r], sb);
                        }
                    }
                    du1[lane] = u1[lane] > 0.f ? sa + sb : 0.f;
                }
                tf_wave_sync();
                {
                    float sc = 0.f, sd = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const f32x4 x0 = tm_ld4(du1 + 4 * c);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            sc = fmaf(x0[r], wC[c][r], sc);
                            sd = fmaf(x0[r], wD[c][r], sd);
                        }
                    }
                    dU[lane] = sc;
                    if (own) U.dU0[(size_t)bi * 80 + lane] = sc;
                    if (lane < 16) {
                        dU[64 + lane] = sd;
                        if (own) U.dU0[(size_t)bi * 80 + 64 + lane] = sd;
                    }
                }
                tf_wave_sync();
                if (lane < 32) dms[lane] = dU[48 + lane] * nm;                    // dM_i: the same for every partner row
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) w3r[c] = tm_ld4u(theta + A.oW3 + l32 * 32 + 4 * c);
                if (lane < 32) dms[lane] = A.dU0[(size_t)bi * 80 + 48 + lane] * nm;
            }
            tf_wave_sync();
            TF_CLK(6);
            if (lane < 32) {
                float va = 0.f, vb = 0.f;
#pragma unroll
                for (int c = 0; c < 8; c += 2) {
                    const f32x4 x0 = tm_ld4(dms + 4 * c), x1 = tm_ld4(dms + 4 * c + 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        va = fmaf(x0[r], w3r[c][r], va);
                        vb = fmaf(x1[r], w3r[c + 1][r], vb);
                    }
                }
                vs[lane] = va + vb;
            }
        }
        TF_CLK(7);
    } else {
        // ================= wavefronts 1-7: the rows into LDS.  EVERY array's loads of a round are issued before the first LDS
        // write (array after array, each is a round trip of its own: 7 of them were 4.5 us); N = 41 is one round
        const int t = tid - 64, nth = EPNN_TF_NT - 64;
        const int nH = NR * 8, nE = N * 12, nA = N * F;
        const int rounds = max(max((nH + 2 * nth - 1) / (2 * nth), (nE + 2 * nth - 1) / (2 * nth)), (nA + 6 * nth - 1) / (6 * nth));
        for (int rd = 0; rd < rounds; ++rd) {
            f32x4 vh1[2], vh2[2], ve[2];
            float va[6];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ih = min((2 * rd + u) * nth + t, nH - 1), ie = min((2 * rd + u) * nth + t, nE - 1);
                const int row = ih >> 3, d = row >= N;                        // row d N + j
                const size_t src = (d ? dstride : 0) + (rowbase + (row - d * N)) * 32 + 4 * (ih & 7);
                vh1[u] = tm_ld4(A.H1 + src);
                vh2[u] = tm_ld4(A.H2 + src);
                ve[u] = tm_ld4(A.e + rowbase * 48 + (size_t)ie * 4);
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int ia = min((6 * rd + u) * nth + t, nA - 1);
                const int j = ia / F, k = ia - j * F;
                const size_t at = a0 + j;
                const float tv = *(k < nx ? A.x + at * nx + k : (k < nx + 48 ? A.h + at * 48 + (k - nx) : A.q + at));     // one load of a selected address
                va[u] = (j >= nreal && k >= nx) ? 0.f : tv;                   // a padded partner's h and q
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ih = (2 * rd + u) * nth + t;
                if (ih < nH) {
                    tm_st4(H1s + (ih >> 3) * RS + 4 * (ih & 7), vh1[u]);
                    tm_st4(H2s + (ih >> 3) * RS + 4 * (ih & 7), vh2[u]);
                }
                if (ih < nE) {
                    const int j = ih / 12;
                    tm_st4(Es + j * ES + 4 * (ih - 12 * j), ve[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int ia = (6 * rd + u) * nth + t;
                if (ia < nA) {
                    const int j = ia / F;
                    As[j * FS + (ia - j * F)] = va[u];
                }
            }
        }
        TF_CLK_T(8, 64);
    }
    __syncthreads();
    TF_CLK(1);
    // ---- dz2 = [H2 > 0] * (dOut W3^T) in registers, dz1 = [H1 > 0] * (dz2 W2^T) on the matrix pipe: a wavefront per 16 rows
    for (int job = wave; job < ((N + 15) / 16) * ND; job += EPNN_TF_NT / 64) {
        const int d = job % ND, j = (job / ND) * 16 + lx;
        const bool jv = j < N;
        const int r = d * N + (jv ? j : 0);
        f32x4 h2[2], dz2[2];
        if (MODE == 1 && (job / ND) * 16 >= nreal) {
            // a tile of padded partners: df = 0 for every row of it -- zeros, without the products (the forward stored zeros too)
            if (jv) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c4 = 0; c4 < 8; ++c4) {
                    tm_st4(D2s + r * RS + 4 * c4, z);
                    tm_st4(D1s + r * RS + 4 * c4, z);
                    if (own) tm_st4(A.dz1 + d * dstride + (rowbase + j) * 32 + 4 * c4, z);
                }
            }
            continue;
        }
        h2[0] = tm_ld4(H2s + r * RS + 4 * lq);
        h2[1] = tm_ld4(H2s + r * RS + 16 + 4 * lq);
        if (MODE == 0) {
            const f32x4 va = tm_ld4(vs + 4 * lq), vb = tm_ld4(vs + 16 + 4 * lq);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                dz2[0][c] = jv && h2[0][c] > 0.f ? va[c] : 0.f;
                dz2[1][c] = jv && h2[1][c] > 0.f ? vb[c] : 0.f;
            }
        } else {
            const float df = d ? -dfs[jv ? j : 0] : dfs[jv ? j : 0];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                dz2[0][c] = jv && h2[0][c] > 0.f ? df * w3v[0][c] : 0.f;
                dz2[1][c] = jv && h2[1][c] > 0.f ? df * w3v[1][c] : 0.f;
            }
        }
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            acc[0] = tm_mfma(w2f[0][s >> 2][s & 3], dz2[s >> 2][s & 3], acc[0]);
            acc[1] = tm_mfma(w2f[1][s >> 2][s & 3], dz2[s >> 2][s & 3], acc[1]);
        }
        TF_CLK(9);
        if (jv) {
            tm_st4(D2s + r * RS + 4 * lq, dz2[0]);
            tm_st4(D2s + r * RS + 16 + 4 * lq, dz2[1]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const f32x4 h1 = tm_ld4(H1s + r * RS + 16 * rb + 4 * lq);
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = h1[c] > 0.f ? acc[rb][c] : 0.f;
                tm_st4(D1s + r * RS + 16 * rb + 4 * lq, v);
                if (own) tm_st4(A.dz1 + d * dstride + (rowbase + j) * 32 + 16 * rb + 4 * lq, v);
            }
        }
    }
    TF_CLK(10);
    if (upd && own) {
        // weight-gradient partials of the update MLP for this atom (rank one per layer), parameter order
        const float *u0 = ub, *u1 = ub + 80, *u2 = ub + 112, *dh = ub + 144, *du2 = ub + 192, *du1 = ub + 224;
        float *Pu = U.part + (size_t)bi * EPNN_TF_PU;
        for (int idx = tid; idx < 80 * 32; idx += EPNN_TF_NT) Pu[idx] = u0[idx >> 5] * du1[idx & 31];
        Pu += 80 * 32;
        if (tid < 32) Pu[tid] = du1[tid];
        Pu += 32;
        for (int idx = tid; idx < 32 * 32; idx += EPNN_TF_NT) Pu[idx] = u1[idx >> 5] * du2[idx & 31];
        Pu += 32 * 32;
        if (tid < 32) Pu[tid] = du2[tid];
        Pu += 32;
        for (int idx = tid; idx < 32 * 48; idx += EPNN_TF_NT) Pu[idx] = u2[idx / 48] * dh[idx % 48];
        Pu += 32 * 48;
        if (tid < 48) Pu[tid] = dh[tid];
    }
    TF_CLK(11);
    __syncthreads();
    TF_CLK(2);
    // ---- weight-gradient partials of this workgroup, in parameter order: W1 [D][32] | b1 | W2 [32][32] | b2 | W3 [32][O] | b3
    const int Pm = D * 32 + 32 + 1024 + 32 + 32 * O + O;
    float *P = A.part + (size_t)bi * Pm;
    float *Pb1 = P + D * 32, *PW2 = Pb1 + 32, *Pb2 = PW2 + 1024, *PW3 = Pb2 + 32, *Pb3 = PW3 + 32 * O;
    const float *ai = As + i * FS;
    // atom-block tiles.  Block 0 carries three rows of ones behind its F weight rows: F = the sum of dz1 over the listed rows,
    // F + 1 = over the swapped rows (the NEXT launch's prologue needs them: rs_w), F + 2 = both (the bias gradient)
    const int KTA = (F + 18) / 16;
    const int nA = 2 * KTA, nE = 6, nW2 = 6, nW3 = MODE ? 2 : 4;
    const int njobs = 2 * nA + nE + nW2 + nW3;
    auto none = [](int) { return 0.f; };
    // pass network: the rows of padded partners are exact zeros in dz1 and dz2 (TfPair::moff): the products stop at the last tile
    // that holds a real partner, in both orders of the pair (two passes of sixteen rows instead of three for an average molecule)
    const int NK = MODE ? min(N, ((nreal + 15) / 16) * 16) : N;
    auto rm = [&](int r) { return (MODE && r >= NK) ? N + (r - NK) : r; };       // row of the flat (order, partner) index over 2 NK rows
    // jobs are dealt to the atom's workgroups first, then to the wavefronts of each (two wavefronts of a SIMD share its matrix pipe)
    for (int job = sub + S * wave; job < njobs; job += (EPNN_TF_NT / 64) * S) {
        f32x4 acc;
        TF_CLK(12);
        if (job < 2 * nA) {
            // first Dense, atom blocks.  Block 0: listed rows carry a_i there, swapped rows a_j; block 1 the other way round
            const int blk = job >= nA, jj = job - blk * nA, kt = jj >> 1, ob = jj & 1;
            const int k = 16 * kt + lx, o = 16 * ob + lx;
            const int kc = k < F ? k : 0;
            const float aic = ai[kc], oneN = blk == 0 && (k == F || k == F + 2) ? 1.f : 0.f, oneT = blk == 0 && (k == F + 1 || k == F + 2) ? 1.f : 0.f;
            auto dN = [&](int r) { return D1s[r * RS + o]; };
            auto dT = [&](int r) { return D1s[(N + r) * RS + o]; };
            if (blk == 0) {                   // a_i meets the listed rows, a_j the swapped ones
                auto ac = [&](int) { return k < F ? aic : oneN; };
                auto aj = [&](int r) { const float t = As[r * FS + kc]; return k < F ? t : oneT; };
                acc = MODE == 0 ? tb_rows_mm<false>(NK, lq, ac, dN, none, none) : tb_rows_mm<true>(NK, lq, ac, dN, aj, dT);
            } else {
                auto ac = [&](int) { return k < F ? aic : 0.f; };
                auto aj = [&](int r) { const float t = As[r * FS + kc]; return k < F ? t : 0.f; };
                acc = MODE == 0 ? tb_rows_mm<false>(NK, lq, aj, dN, none, none) : tb_rows_mm<true>(NK, lq, aj, dN, ac, dT);
            }
            TF_CLK(13);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int kk = 16 * kt + 4 * lq + c;
                if (kk < F) P[(blk * F + kk) * 32 + o] = acc[c];
                else if (blk == 0 && kk == F) A.rs_w[(size_t)bi * 64 + o] = acc[c];
                else if (blk == 0 && kk == F + 1) A.rs_w[(size_t)bi * 64 + 32 + o] = acc[c];
                else if (blk == 0 && kk == F + 2) Pb1[o] = acc[c];
            }
        } else if (job < 2 * nA + nE) {
            // first Dense, edge block: the same e_ij in both orders
            const int jj = job - 2 * nA, kt = jj >> 1, ob = jj & 1, k = 16 * kt + lx, o = 16 * ob + lx;
            auto ae = [&](int r) { return Es[r * ES + k]; };
            auto dS = [&](int r) { return MODE ? D1s[r * RS + o] + D1s[(N + r) * RS + o] : D1s[r * RS + o]; };
            acc = tb_rows_mm<false>(NK, lq, ae, dS, none, none);
#pragma unroll
            for (int c = 0; c < 4; ++c) P[(2 * F + 16 * kt + 4 * lq + c) * 32 + o] = acc[c];
        } else if (job < 2 * nA + nE + nW2) {
            // second Dense over both orders; tile 2 is the row "1": b2.  The pass network's dz2 of the two orders of a pair cancel
            // (+df w3 against -df w3 wherever both rows are active): b2 sums them pair by pair, not one order after the other
            const int jj = job - 2 * nA - nE, kt = jj >> 1, ob = jj & 1, k = 16 * kt + lx, o = 16 * ob + lx;
            if (kt < 2) {
                auto ah = [&](int r) { return H1s[rm(r) * RS + k]; };
                auto d2 = [&](int r) { return D2s[rm(r) * RS + o]; };
                acc = tb_rows_mm<false>(ND * NK, lq, ah, d2, none, none);
            } else {
                auto a1 = [&](int) { return lx == 0 ? 1.f : 0.f; };
                auto d2 = [&](int r) { return MODE ? D2s[r * RS + o] + D2s[(N + r) * RS + o] : D2s[r * RS + o]; };
                acc = tb_rows_mm<false>(NK, lq, a1, d2, none, none);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int kk = 16 * kt + 4 * lq + c;
                if (kk < 32) PW2[kk * 32 + o] = acc[c];
                else if (kk == 32) Pb2[o] = acc[c];
            }
        } else {
            // third Dense: message network sum_j H2[j][k] dM[o]; pass network sum over both orders of H2[row][k] (+-df)
            const int jj = job - 2 * nA - nE - nW2, kt = MODE ? jj : jj >> 1, ob = MODE ? 0 : jj & 1, k = 16 * kt + lx, o = 16 * ob + lx;
            auto ah = [&](int r) { return H2s[rm(r) * RS + k]; };
            if (MODE == 0) {
                const float dmo = dms[o];
                auto bm = [&](int) { return dmo; };
                acc = tb_rows_mm<false>(N, lq, ah, bm, none, none);
#pragma unroll
                for (int c = 0; c < 4; ++c) PW3[(16 * kt + 4 * lq + c) * 32 + o] = acc[c];
            } else {
                auto bd = [&](int r) { const float t = dfs[r < NK ? r : r - NK]; return lx == 0 ? (r < NK ? t : -t) : 0.f; };
                acc = tb_rows_mm<false>(2 * NK, lq, ah, bd, none, none);
                if (lx == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) PW3[16 * kt + 4 * lq + c] = acc[c];
                }
            }
        }
    }
    if (!own) {
    } else if (MODE == 0) {
        if (tid < 32) Pb3[tid] = (float)N * dms[tid];
    } else if (tid == 0) {
        Pb3[0] = 0.f;                         // sum over the rows of df and of -df, each in the same order: exactly 0
    }
    TF_CLK(3);
    TF_CYC(15);
}

// ---------------------------------------------------------------------------------------------- gradient = sum of partials
#define EPNN_TF_MAXRED 24
// Keras-2 Adam's step size lr sqrt(1 - b2^t) / (1 - b1^t) (charge_gn.py:419) with the powers by repeated squaring in double: the
// same bits on the host and in a kernel (a replayed hipGraph reads t from device memory)
__host__ __device__ inline float epnn_adam_alpha(float lr, float b1, float b2, long long t) {
    double p1 = 1.0, p2 = 1.0, x1 = (double)b1, x2 = (double)b2;
    for (long long e = t; e > 0; e >>= 1) {
        if (e & 1) {
            p1 *= x1;
            p2 *= x2;
        }
        x1 *= x1;
        x2 *= x2;
    }
    return (float)((double)lr * sqrt(1.0 - p2) / (1.0 - p1));
}
struct TfReduce {
    int n;
    int theta_off[EPNN_TF_MAXRED], len[EPNN_TF_MAXRED], nblk[EPNN_TF_MAXRED];
    size_t part_off[EPNN_TF_MAXRED];
    // with `adam` set the same launch takes the optimizer step on the element it has just summed (Keras-2 Adam,
    // charge_gn.py:419: theta -= alpha m / (sqrt(v) + eps), alpha = lr sqrt(1 - b2^t) / (1 - b1^t))
    int adam;
    float alpha, b1, b2, eps;
    float *theta, *m, *v;
    const long long *step_p;      // hipGraph replay: the step number lives on the device (the step's first launch has counted it)
    float lr;
    const int *real;              // [natoms] 0 for a padded slot whose block of partials nobody wrote (block index mod natoms), or null
    int natoms;
};
// grad[theta_off + idx] = sum_blk part[part_off + blk * len + idx]: four quarter sums (one per wavefront), combined in order
__global__ __launch_bounds__(256) void k_tb_wreduce(TfReduce T, const float *part, float *grad) {
    __shared__ float sh[4][64];
    const int en = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int len = T.len[en], idx = blockIdx.x * 64 + lane;
    if ((int)blockIdx.x * 64 >= len) return;
    const int nblk = T.nblk[en];
    const int lo = (int)((long long)nblk * w / 4), hi = (int)((long long)nblk * (w + 1) / 4);
    // everything this thread will need is requested before the first wait: the optimizer state of its element (wavefront 0) and
    // sixteen partials per pass, unconditional loads of a clamped block index (a remainder loop of single loads, then m / v, then
    // theta were five round trips of the step's last 7 us)
    const bool fin = w == 0 && idx < len;
    const int ei = T.theta_off[en] + (idx < len ? idx : 0);
    float m0 = 0.f, v0 = 0.f, th0 = 0.f;
    long long stp = 0;
    if (T.adam && fin) {
        if (T.step_p) stp = *T.step_p;
        m0 = T.m[ei];
        v0 = T.v[ei];
        th0 = T.theta[ei];
    }
    float s = 0.f;
    if (idx < len) {
        const float *p = part + T.part_off[en] + idx;
        for (int blk = lo; blk < hi; blk += 16) {
            float pv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) pv[u] = p[(size_t)min(blk + u, hi - 1) * len];
            if (T.real) {                         // (a padded atom's partials are exact zeros when they are computed: the same bits)
                const int b0 = blk % T.natoms;     // block index -> atom slot: one division per pass (the update MLP's entry has T natoms blocks)
                int rl[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    int sl = b0 + u;
                    if (T.natoms >= 16) sl -= sl >= T.natoms ? T.natoms : 0;
                    else sl %= T.natoms;
                    rl[u] = T.real[sl];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (!rl[u]) pv[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (blk + u < hi) s += pv[u];
        }
    }
    sh[w][lane] = s;
    __syncthreads();
    if (fin) {
        const int i = ei;
        const float g = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
        grad[i] = g;
        if (T.adam) {
            const float alpha = T.step_p ? epnn_adam_alpha(T.lr, T.b1, T.b2, stp) : T.alpha;
            const float mi = T.b1 * m0 + (1.f - T.b1) * g;
            const float vi = T.b2 * v0 + (1.f - T.b2) * g * g;
            T.m[i] = mi;
            T.v[i] = vi;
            T.theta[i] = th0 - alpha * mi / (sqrtf(vi) + T.eps);
        }
    }
}
