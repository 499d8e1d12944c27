// Front-end kernels: coordinates -> near-pair list with Gaussian edge features.
//
// Replaces get_init_edges (reference charge_gn.py:122-163) and the dense e/mask tensors that
// gen_padded_init_state builds from it (charge_gn.py:331-364).  Instead of an (n,n,48) tensor per molecule the
// device keeps one entry per UNORDERED pair {i<j} with D_ij < cutoff (the only pairs whose e is non-zero), in
// row-major (i, then j) order:  pi, pj (flat atom index), pe[48] (float32, computed in float64 like NumPy does),
// wi = wj = near flag (max_k e_k > tol as float32, charge_gn.py:90-94) and sym = 1 (e_ij == e_ji).
#pragma once
#include "epnn_common.h"

struct FrontArgs {
    const float *xyz;      // [A][3]
    const int *mol_of;     // [A]
    const int *moff;       // [B+1]
    int A;
    double cutoff, eta;
    float tol;
    int e_dim;
    const double *mu;      // [e_dim] linspace(0.1, cutoff, e_dim) evaluated like NumPy
    int *row_cnt;          // [A]
    int *row_off;          // [A+1]
    int pcap;
    int *pi, *pj, *psym;
    float *pe, *pwi, *pwj;
    int *status;
};

// distance exactly as scipy.spatial.distance_matrix on float32 coordinates promoted to float64
// (charge_gn.py:124): sqrt((dx*dx + dy*dy) + dz*dz).
__device__ __forceinline__ double epnn_dist(const float *xyz, int i, int j) {
    double dx = (double)xyz[3 * j + 0] - (double)xyz[3 * i + 0];
    double dy = (double)xyz[3 * j + 1] - (double)xyz[3 * i + 1];
    double dz = (double)xyz[3 * j + 2] - (double)xyz[3 * i + 2];
    return sqrt(__dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz)));
}

// one wave per row i: number of j > i (same molecule) with D < cutoff
__global__ __launch_bounds__(256) void k_front_count(FrontArgs F) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= F.A) return;
    const int end = F.moff[F.mol_of[row] + 1];
    int cnt = 0;
    for (int j0 = row + 1; j0 < end; j0 += 64) {
        const int j = j0 + lane;
        bool near = false;
        if (j < end) near = epnn_dist(F.xyz, row, j) < F.cutoff;
        cnt += __popcll(__ballot(near));
    }
    if (lane == 0) F.row_cnt[row] = cnt;
}

// exclusive scan of row_cnt[0..A) -> row_off[0..A] in two passes: (1) every block of 256 threads scans 2048
// elements locally and publishes its total, (2) every block adds the totals of the blocks before it.
#define EPNN_SCAN_ELEMS 2048
__global__ __launch_bounds__(256) void k_front_scan1(FrontArgs F, int *bsum) {
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = blockIdx.x * EPNN_SCAN_ELEMS + tid * 8;
    int v[8], tot = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k] = base + k < F.A ? F.row_cnt[base + k] : 0;
        tot += v[k];
    }
    int incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int run = woff + incl - tot;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (base + k < F.A) F.row_off[base + k] = run;
        run += v[k];
    }
    if (tid == 255) {
        bsum[blockIdx.x] = run;
        if (gridDim.x == 1) {                      // a single block: the total is known here, k_front_scan2 is not launched
            F.row_off[F.A] = run;
            if (run > F.pcap) atomicOr(F.status, EPNN_ST_PAIR_OVERFLOW);
        }
    }
}
__global__ __launch_bounds__(256) void k_front_scan2(FrontArgs F, const int *bsum) {
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int part = 0;
    for (int b = tid; b < (int)blockIdx.x; b += 256) part += bsum[b];
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    const int boff = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const int base = blockIdx.x * EPNN_SCAN_ELEMS + tid * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (base + k < F.A) F.row_off[base + k] += boff;
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        const int total = boff + bsum[blockIdx.x];
        F.row_off[F.A] = total;
        if (total > F.pcap) atomicOr(F.status, EPNN_ST_PAIR_OVERFLOW);
    }
}

// one wave per row: write the row's pairs
__global__ __launch_bounds__(256) void k_front_fill(FrontArgs F) {
    __shared__ int s_j[4][64];
    __shared__ double s_D[4][64];
    __shared__ double s_C[4][64];
    __shared__ int s_max[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= F.A) return;
    if (F.row_off[F.A] > F.pcap) return;   // overflow: host regrows and reruns
    const int end = F.moff[F.mol_of[row] + 1];
    int slot0 = F.row_off[row];
    const double pi_d = 3.141592653589793;
    for (int j0 = row + 1; j0 < end; j0 += 64) {
        const int j = j0 + lane;
        bool near = false;
        double D = 0.0;
        if (j < end) {
            D = epnn_dist(F.xyz, row, j);
            near = D < F.cutoff;
        }
        const unsigned long long bal = __ballot(near);
        const int m = __popcll(bal);
        if (m == 0) continue;
        if (near) {
            const int rank = __popcll(bal & ((1ull << lane) - 1ull));
            s_j[wave][rank] = j;
            s_D[wave][rank] = D;
            // charge_gn.py:148-152: C = (cos(pi*D/cutoff)+1)/2 ; C[D<=0] = 1 (D>=cutoff excluded, diagonal not listed)
            double C = (cos(pi_d * (D - 0.0) / F.cutoff) + 1.0) / 2.0;
            if (D <= 0.0) C = 1.0;
            s_C[wave][rank] = C;
            s_max[wave][rank] = 0;
        }
        __builtin_amdgcn_wave_barrier();
        const int total = m * F.e_dim;
        for (int idx = lane; idx < total; idx += 64) {
            const int pr = idx / F.e_dim, ch = idx - pr * F.e_dim;
            const double d = s_D[wave][pr] - F.mu[ch];
            const double val = s_C[wave][pr] * exp(-F.eta * (d * d));      // charge_gn.py:160
            const float ef = (float)val;                                      // :161
            F.pe[(size_t)(slot0 + pr) * F.e_dim + ch] = ef;
            atomicMax(&s_max[wave][pr], __float_as_int(ef));                  // e >= 0: int order == float order
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < m) {
            const int s = slot0 + lane;
            F.pi[s] = row;
            F.pj[s] = s_j[wave][lane];
            const float w = __int_as_float(s_max[wave][lane]) > F.tol ? 1.0f : 0.0f;   // charge_gn.py:90-94
            F.pwi[s] = w;
            F.pwj[s] = w;
            F.psym[s] = 1;
        }
        __builtin_amdgcn_wave_barrier();
        slot0 += m;
    }
}

// epnn_edges: dense (n,n,e_dim) tensor exactly like get_init_edges, one thread per (i,j,ch)
__global__ __launch_bounds__(256) void k_edges_dense(const float *xyz, int n, int e_dim, double cutoff, double eta,
                                                     const double *mu, float *e_out, double *c_out) {
    const size_t total = (size_t)n * n * e_dim;
    const double pi_d = 3.141592653589793;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int ch = (int)(idx % e_dim);
        const size_t pr = idx / e_dim;
        const int j = (int)(pr % n), i = (int)(pr / n);
        const double D = epnn_dist(xyz, i, j);
        double C = (cos(pi_d * (D - 0.0) / cutoff) + 1.0) / 2.0;
        if (D >= cutoff) C = 0.0;
        if (D <= 0.0) C = 1.0;
        if (i == j) C = 0.0;
        const double d = D - mu[ch];
        e_out[idx] = (float)(C * exp(-eta * (d * d)));
        if (c_out && ch == 0) c_out[pr] = C;                  // the cutoff weights the reference returns tiled (charge_gn.py:163)
    }
}
