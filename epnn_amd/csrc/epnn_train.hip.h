// Training step (reference charge_gn.py:393-402): forward, loss = sum_atoms (y - p)^2, backward, Adam.
//
// The reference trains with ONE molecule per optimizer step (charge_gn.py:443-451); data-parallel training puts
// one molecule (or a few) on each GPU and sums the gradients with one RCCL all-reduce of the flat 296 KB gradient.
// At that size a step is launch/latency-bound, not throughput-bound, so this path keeps the reference's literal
// formulation -- materialised [a_i | a_j | e_ij] rows and three Dense layers per sweep -- because its backward is
// the textbook Dense backward and every intermediate can be checked against the oracle.  (The fused / tiled
// inference kernels are not used here.)  All reductions have a fixed order: gradients are bit-reproducible.
//
// Master weights, gradients and Adam moments live on the device as flat float32 vectors in the order of Keras'
// model.trainable_variables (charge_gn.py:371-374): update MLP, message MLPs t=0.., pass MLPs t=0..; kernel, bias.
#pragma once
#include "epnn_host.h"
#include "epnn_train_fused.hip.h"

struct TDense {            // one Dense inside the flat parameter vector
    int offW, offB, n_in, n_out;
};

// ------------------------------------------------------------------------------------------------ kernels
// X[(b,i,j)] = swap ? [a_j | a_i | e_ij] : [a_i | a_j | e_ij]          (charge_gn.py:62-66, 101-108)
__global__ __launch_bounds__(256) void k_t_rows(const float *a, const float *e, float *X, int B, int N, int F, int E, int swap) {
    const int D = 2 * F + E;
    const size_t total = (size_t)B * N * N * D;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int k = (int)(idx % D);
        const size_t r = idx / D;
        const int j = (int)(r % N), i = (int)((r / N) % N), b = (int)(r / ((size_t)N * N));
        float v;
        if (k < F) v = a[((size_t)b * N + (swap ? j : i)) * F + k];
        else if (k < 2 * F) v = a[((size_t)b * N + (swap ? i : j)) * F + (k - F)];
        else v = e[r * E + (k - 2 * F)];
        X[idx] = v;
    }
}
// Y = act(X W + b)      X [R][K], W [K][O], Y [R][O]
__global__ __launch_bounds__(256) void k_t_dense(const float *X, const float *W, const float *b, float *Y, int R, int K,
                                                 int O, int relu) {
    const size_t total = (size_t)R * O;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int o = (int)(idx % O);
        const size_t r = idx / O;
        const float *x = X + r * K;
        float s = b[o];
        for (int k = 0; k < K; ++k) s = fmaf(x[k], W[(size_t)k * O + o], s);
        Y[idx] = relu ? fmaxf(s, 0.f) : s;
    }
}
// dX = (dY * [Ypost > 0]) W^T        (Ypost == nullptr: linear layer)
__global__ __launch_bounds__(256) void k_t_dense_dx(const float *dY, const float *Ypost, const float *W, float *dX, int R,
                                                    int K, int O) {
    const size_t total = (size_t)R * K;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int k = (int)(idx % K);
        const size_t r = idx / K;
        float s = 0.f;
        for (int o = 0; o < O; ++o) {
            float g = dY[r * O + o];
            if (Ypost && !(Ypost[r * O + o] > 0.f)) g = 0.f;
            s = fmaf(g, W[(size_t)k * O + o], s);
        }
        dX[idx] = s;
    }
}
// partial dW[slice][k][o] = sum_{r in slice} X[r][k] * dYp[r][o];  k == K: bias row (sum of dYp)
__global__ __launch_bounds__(256) void k_t_dense_dw(const float *X, const float *dY, const float *Ypost, float *part, int R,
                                                    int K, int O, int nslice) {
    const int slice = blockIdx.y;
    const int rlo = (int)((size_t)R * slice / nslice), rhi = (int)((size_t)R * (slice + 1) / nslice);
    const int total = (K + 1) * O;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int o = idx % O, k = idx / O;
        float s = 0.f;
        for (int r = rlo; r < rhi; ++r) {
            float g = dY[(size_t)r * O + o];
            if (Ypost && !(Ypost[(size_t)r * O + o] > 0.f)) g = 0.f;
            s = fmaf(k < K ? X[(size_t)r * K + k] : 1.f, g, s);
        }
        part[(size_t)slice * total + idx] = s;
    }
}

// ---- the same three products on v_mfma_f32_32x32x2_f32 (one 32x32 output tile per wavefront, four per workgroup).
// Lane (c = lane & 31, hh = lane >> 5): A operand = A[row c][k = 2s + hh], B operand = B[k = 2s + hh][col c],
// accumulator register r = D[row kappa(hh,r)][col c].  Every sum has a fixed order (bit-reproducible gradients).
// forward: Y[r][o] = act(sum_k X[r][k] W[k][o] + b[o]);  tiles: rows x ceil(O/32)
__global__ __launch_bounds__(256) void k_t_mm_fwd(const float *X, const float *W, const float *b, float *Y, int R, int K, int O,
                                                   int relu, int ntile_o) {
    const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rt = tile / ntile_o, ot = tile - rt * ntile_o;
    const int r0 = rt * 32, o0 = ot * 32;
    if (r0 >= R) return;
    const int row = min(r0 + c, R - 1), col = min(o0 + c, O - 1);
    const float *xr = X + (size_t)row * K;
    f32x16 acc = epnn_splat16(0.f);
    const int steps = (K + 1) >> 1;
    for (int s = 0; s < steps; ++s) {
        const int k = 2 * s + hh;
        const float a = k < K ? xr[k] : 0.f;
        const float w = k < K ? W[(size_t)k * O + col] : 0.f;
        acc = epnn_mfma(a, w, acc);
    }
    if (o0 + c < O) {
        const float bias = b[o0 + c];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = r0 + epnn_kappa(hh, r);
            if (rr < R) {
                const float v = acc[r] + bias;
                Y[(size_t)rr * O + o0 + c] = relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
}
// backward to the input: dX[r][k] = sum_o g[r][o] W[k][o],  g = dY * [Ypost > 0];  tiles: rows x ceil(K/32)
__global__ __launch_bounds__(256) void k_t_mm_dx(const float *dY, const float *Ypost, const float *W, float *dX, int R, int K,
                                                  int O, int ntile_k) {
    const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rt = tile / ntile_k, kt = tile - rt * ntile_k;
    const int r0 = rt * 32, k0 = kt * 32;
    if (r0 >= R) return;
    const int row = min(r0 + c, R - 1), kk = min(k0 + c, K - 1);
    f32x16 acc = epnn_splat16(0.f);
    const int steps = (O + 1) >> 1;
    for (int s = 0; s < steps; ++s) {
        const int o = 2 * s + hh;
        float g = 0.f, w = 0.f;
        if (o < O) {
            g = dY[(size_t)row * O + o];
            if (Ypost && !(Ypost[(size_t)row * O + o] > 0.f)) g = 0.f;
            w = W[(size_t)kk * O + o];
        }
        acc = epnn_mfma(g, w, acc);
    }
    if (k0 + c < K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = r0 + epnn_kappa(hh, r);
            if (rr < R) dX[(size_t)rr * K + k0 + c] = acc[r];
        }
    }
}
// weight gradient: part[slice][k][o] = sum_{r in slice} [X[r][:] | 1][k] g[r][o];  tiles: slices x ceil((K+1)/32) x ceil(O/32)
__global__ __launch_bounds__(256) void k_t_mm_dw(const float *X, const float *dY, const float *Ypost, float *part, int R, int K,
                                                  int O, int nslice, int ntile_k, int ntile_o) {
    const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int slice = tile / (ntile_k * ntile_o), rem = tile - slice * (ntile_k * ntile_o);
    const int kt = rem / ntile_o, ot = rem - kt * ntile_o;
    if (slice >= nslice) return;
    const int rlo = (int)((size_t)R * slice / nslice), rhi = (int)((size_t)R * (slice + 1) / nslice);
    const int k0 = kt * 32, o0 = ot * 32;
    const int krow = k0 + c, col = min(o0 + c, O - 1);      // A operand row = weight row k (k == K: the bias row)
    f32x16 acc = epnn_splat16(0.f);
    for (int r2 = rlo; r2 < rhi; r2 += 2) {
        const int r = r2 + hh;
        float a = 0.f, g = 0.f;
        if (r < rhi) {
            a = krow < K ? X[(size_t)r * K + krow] : (krow == K ? 1.f : 0.f);
            g = dY[(size_t)r * O + col];
            if (Ypost && !(Ypost[(size_t)r * O + col] > 0.f)) g = 0.f;
        }
        acc = epnn_mfma(a, g, acc);
    }
    if (o0 + c < O) {
        const int total = (K + 1) * O;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = k0 + epnn_kappa(hh, r);
            if (kr <= K) part[(size_t)slice * total + (size_t)kr * O + o0 + c] = acc[r];
        }
    }
}
// grad[offW..] += sum_slices part (fixed order); bias row goes to offB
__global__ __launch_bounds__(256) void k_t_dw_reduce(const float *part, float *grad, int offW, int offB, int K, int O, int nslice) {
    const int total = (K + 1) * O;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        float s = 0.f;
        for (int sl = 0; sl < nslice; ++sl) s += part[(size_t)sl * total + idx];
        const int o = idx % O, k = idx / O;
        if (k < K) grad[offW + k * O + o] += s;
        else grad[offB + o] += s;
    }
}
// da[(b,i)][f] += sum over the pair rows in which atom i appears: block 0 of X is atom `first`, block 1 the other
__global__ __launch_bounds__(256) void k_t_rows_bwd(const float *dX, float *da, int B, int N, int F, int E, int swap) {
    const int D = 2 * F + E;
    const size_t total = (size_t)B * N * F;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int f = (int)(idx % F);
        const int i = (int)((idx / F) % N), b = (int)(idx / ((size_t)F * N));
        const size_t mb = (size_t)b * N * N;
        float s = 0.f;
        // rows (i, j): atom i sits in block (swap ? 1 : 0);  rows (j, i): atom i sits in block (swap ? 0 : 1)
        for (int j = 0; j < N; ++j) s += dX[(mb + (size_t)i * N + j) * D + (swap ? F : 0) + f];
        for (int j = 0; j < N; ++j) s += dX[(mb + (size_t)j * N + i) * D + (swap ? 0 : F) + f];
        da[idx] += s;
    }
}
// M[(b,i)][o] = sum_j Mij[(b,i,j)][o]      (charge_gn.py:70)
__global__ __launch_bounds__(256) void k_t_sumj(const float *Mij, float *M, int B, int N, int O) {
    const size_t total = (size_t)B * N * O;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int o = (int)(idx % O);
        const size_t bi = idx / O;
        float s = 0.f;
        for (int j = 0; j < N; ++j) s += Mij[(bi * N + j) * O + o];
        M[idx] = s;
    }
}
// a = [x | h | q]
__global__ __launch_bounds__(256) void k_t_assemble(const float *x, const float *h, const float *q, float *a, int BN, int nx, int H) {
    const int F = nx + H + 1;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < BN * F; idx += gridDim.x * 256) {
        const int f = idx % F, at = idx / F;
        a[idx] = f < nx ? x[at * nx + f] : (f < nx + H ? h[at * H + (f - nx)] : q[at]);
    }
}
// U0 = [h | M] * nm            (charge_gn.py:71-72)
__global__ __launch_bounds__(256) void k_t_u0(const float *h, const float *M, const float *nm, float *U0, int BN, int H, int O) {
    const int W = H + O;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < BN * W; idx += gridDim.x * 256) {
        const int f = idx % W, at = idx / W;
        U0[idx] = (f < H ? h[at * H + f] : M[at * O + (f - H)]) * nm[at];
    }
}
__global__ __launch_bounds__(256) void k_t_scale_rows(const float *src, const float *nm, float *dst, int BN, int W) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < BN * W; idx += gridDim.x * 256) dst[idx] = src[idx] * nm[idx / W];
}
// wgt = max(mask, -1) * is_near(e)        (charge_gn.py:90-94,116)
__global__ __launch_bounds__(256) void k_t_wgt(const float *e, const float *mask, float *wgt, int R, int E, float tol) {
    for (int r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256) {
        float mx = 0.f;
        for (int k = 0; k < E; ++k) mx = fmaxf(mx, e[(size_t)r * E + k]);
        wgt[r] = mx > tol ? mask[r] : 0.f;
    }
}
__global__ __launch_bounds__(256) void k_t_nodemask(const float *mask, float *nm, int B, int N) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < B * N; idx += gridDim.x * 256) {
        const int k = idx % N, b = idx / N;
        float s = 0.f;
        for (int j = 0; j < N; ++j) s += mask[((size_t)b * N + j) * N + k];
        nm[idx] = fminf(fmaxf(s, 0.f), 1.f);
    }
}
// q_new[i] = q[i] + sum_j 0.5 (fN - fT) wgt      (charge_gn.py:116-118)
__global__ __launch_bounds__(256) void k_t_epn_q(const float *fN, const float *fT, const float *wgt, const float *q, float *qn, int BN, int N) {
    for (int at = blockIdx.x * 256 + threadIdx.x; at < BN; at += gridDim.x * 256) {
        float s = 0.f;
        for (int j = 0; j < N; ++j) {
            const size_t r = (size_t)at * N + j;
            s += 0.5f * (fN[r] - fT[r]) * wgt[r];
        }
        qn[at] = q[at] + s;
    }
}
// dfN = 0.5 gq_i wgt, dfT = -dfN
__global__ __launch_bounds__(256) void k_t_epn_df(const float *gq, const float *wgt, float *dfN, float *dfT, int BN, int N) {
    for (int r = blockIdx.x * 256 + threadIdx.x; r < BN * N; r += gridDim.x * 256) {
        const float g = 0.5f * gq[r / N] * wgt[r];
        dfN[r] = g;
        dfT[r] = -g;
    }
}
// gq = -2 (y - q); loss per molecule (fixed order)
__global__ __launch_bounds__(64) void k_t_loss(const float *y, const float *q, float *gq, float *loss, int N) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < N; ++i) {
            const float d = y[b * N + i] - q[b * N + i];
            s += d * d;
            gq[b * N + i] = -2.f * d;
        }
        loss[b] = s;
    }
}
// predictions and per-atom loss terms (the row-fused forward writes them from its last pass sweep)
__global__ __launch_bounds__(256) void k_t_loss_terms(const float *y, const float *q, float *pred, float *lterm, int BN) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < BN; i += gridDim.x * 256) {
        const float d = y[i] - q[i];
        pred[i] = q[i];
        lterm[i] = d * d;
    }
}
// dm_ij = gM_i  (broadcast over j);  gM = second block of dU0 * nm
__global__ __launch_bounds__(256) void k_t_bcast_gm(const float *dU0, const float *nm, float *dm, int BN, int N, int H, int O) {
    const size_t total = (size_t)BN * N * O;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int o = (int)(idx % O);
        const size_t at = idx / ((size_t)N * O);
        dm[idx] = dU0[at * (H + O) + H + o] * nm[at];
    }
}
// gh_prev = dU0[:, :H] * nm + da[:, nx:nx+H]
__global__ __launch_bounds__(256) void k_t_gh_prev(const float *dU0, const float *nm, const float *da, float *gh, int BN, int nx, int H, int O) {
    const int F = nx + H + 1;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < BN * H; idx += gridDim.x * 256) {
        const int f = idx % H, at = idx / H;
        gh[idx] = dU0[at * (H + O) + f] * nm[at] + da[at * F + nx + f];
    }
}
// after one EPN step's backward: gfeat += da[h part]; gq += da[q part]
__global__ __launch_bounds__(256) void k_t_epn_fold(const float *da, float *gfeat, float *gq, int BN, int nx, int H) {
    const int F = nx + H + 1;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < BN * (H + 1); idx += gridDim.x * 256) {
        const int f = idx % (H + 1), at = idx / (H + 1);
        if (f < H) gfeat[at * H + f] += da[at * F + nx + f];
        else gq[at] += da[at * F + nx + H];
    }
}
// Keras-2 Adam (charge_gn.py:419): theta -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + eps)
__global__ __launch_bounds__(256) void k_t_adam(float *theta, const float *grad, float *m, float *v, int n, float alpha,
                                                float b1, float b2, float eps) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float g = grad[i];
        const float mi = b1 * m[i] + (1.f - b1) * g;
        const float vi = b2 * v[i] + (1.f - b2) * g * g;
        m[i] = mi;
        v[i] = vi;
        theta[i] -= alpha * mi / (sqrtf(vi) + eps);
    }
}

// ================================================================================================ host side

struct TrainState {
    bool ready = false;
    int P = 0;                                   // number of parameters
    TDense upd[3], msg[EPNN_MAXT][3], pas[EPNN_MAXT][3];
    std::vector<TDense> updv;                    // the update MLP's layers when make_model was given other `layers` than [32, 32] (h->updg)
    DevBuf theta, grad, m, v, part, arena, loss;
    long step = 0;
    float lr = 1e-3f, b1 = 0.9f, b2 = 0.999f, eps = 1e-7f;
    bool dev_newer = false;                      // device masters are ahead of the host copies
    // the ~340 launches of one forward + backward, captured once per (B, N, buffer set) and replayed as one hipGraph
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    std::vector<const void *> gkey;
    // the last few captures are kept (a training loop alternates between its batch size and the last, smaller batch of an epoch:
    // re-instantiating on every change would cost more than the replay saves)
    struct KeptGraph { std::vector<const void *> key; hipGraph_t graph; hipGraphExec_t exec; };
    std::vector<KeptGraph> kept;
    bool fused_attr = false;                     // dynamic LDS limit of the row-fused kernels raised
    hipEvent_t ev_fwd = nullptr;                 // recorded behind the step's last forward launch: loss terms and predictions are on the host
    bool inflight = false;                       //   ... the caller was handed them while the backward pass and the optimizer step were still running
    int last_B = 0, last_N = 0;
    DevBuf d_step;                               // hipGraph replay with the optimizer step inside: the step number on the device
    long dev_step = -1;                          //   ... and the value the host last put there
    bool host_out = false;                       // the last forward launch wrote loss terms | predictions into the caller's page-locked buffer
#ifdef EPNN_TF_CLOCKS
    DevBuf clk;                                  // [launch][16] phase clocks of workgroup 0 (development build)
#endif
};

static TrainState *train_state(epnn_handle *h) {
    if (!h->train) h->train = new TrainState();
    return reinterpret_cast<TrainState *>(h->train);
}

// the update MLP of a state: its three standard layers, or (TrainState only) the layers of epnn_set_update_layers
template <typename STATE> static std::vector<TDense> *updv_of(STATE *) { return nullptr; }
static std::vector<TDense> *updv_of(TrainState *ts) { return &ts->updv; }
template <typename STATE>
static bool upd_is_generic(epnn_handle *h, STATE *ts) { return h->upd_generic && updv_of(ts) != nullptr; }

template <typename STATE>
static void train_layout(epnn_handle *h, STATE *ts) {
    int off = 0;
    auto put = [&](TDense &d, const HostDense &hd) {
        d.n_in = hd.n_in;
        d.n_out = hd.n_out;
        d.offW = off;
        off += hd.n_in * hd.n_out;
        d.offB = off;
        off += hd.n_out;
    };
    if (upd_is_generic(h, ts)) {
        std::vector<TDense> &v = *updv_of(ts);
        v.resize(h->updg.size());
        for (size_t l = 0; l < v.size(); ++l) put(v[l], h->updg[l]);
    } else
        for (int l = 0; l < 3; ++l) put(ts->upd[l], h->upd[l]);
    for (int t = 0; t < h->cfg.T; ++t)
        for (int l = 0; l < 3; ++l) put(ts->msg[t][l], h->msg[t][l]);
    for (int t = 0; t < h->cfg.T; ++t)
        for (int l = 0; l < 3; ++l) put(ts->pas[t][l], h->pas[t][l]);
    ts->P = off;
}

// host copies <-> flat vector (Keras trainable_variables order)
template <typename STATE>
static void train_gather_host(epnn_handle *h, STATE *ts, std::vector<float> &flat) {
    flat.assign(ts->P, 0.f);
    auto cp = [&](const TDense &d, const HostDense &hd) {
        memcpy(flat.data() + d.offW, hd.W.data(), hd.W.size() * 4);
        memcpy(flat.data() + d.offB, hd.b.data(), hd.b.size() * 4);
    };
    if (upd_is_generic(h, ts))
        for (size_t l = 0; l < h->updg.size(); ++l) cp((*updv_of(ts))[l], h->updg[l]);
    else
        for (int l = 0; l < 3; ++l) cp(ts->upd[l], h->upd[l]);
    for (int t = 0; t < h->cfg.T; ++t)
        for (int l = 0; l < 3; ++l) { cp(ts->msg[t][l], h->msg[t][l]); cp(ts->pas[t][l], h->pas[t][l]); }
}
static void train_scatter_host(epnn_handle *h, TrainState *ts, const std::vector<float> &flat) {
    auto cp = [&](const TDense &d, HostDense &hd) {
        memcpy(hd.W.data(), flat.data() + d.offW, hd.W.size() * 4);
        memcpy(hd.b.data(), flat.data() + d.offB, hd.b.size() * 4);
    };
    if (upd_is_generic(h, ts))
        for (size_t l = 0; l < h->updg.size(); ++l) cp(ts->updv[l], h->updg[l]);
    else
        for (int l = 0; l < 3; ++l) cp(ts->upd[l], h->upd[l]);
    for (int t = 0; t < h->cfg.T; ++t)
        for (int l = 0; l < 3; ++l) { cp(ts->msg[t][l], h->msg[t][l]); cp(ts->pas[t][l], h->pas[t][l]); }
}

// pull the trained masters back into the host copies (so inference / get_weights / save_weights see them)
static int train_sync_to_host(epnn_handle *h) {
    if (!h->train) return 0;
    TrainState *ts = train_state(h);
    if (!ts->ready || !ts->dev_newer) return 0;
    std::vector<float> flat(ts->P);
    HIPCHK(hipMemcpyAsync(flat.data(), ts->theta.p, (size_t)ts->P * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    train_scatter_host(h, ts, flat);
    ts->dev_newer = false;
    h->weights_dirty = true;
    return 0;
}

static int train_init(epnn_handle *h, float lr, float b1, float b2, float eps) {
    TrainState *ts = train_state(h);
    if (train_sync_to_host(h)) return 1;
    train_layout(h, ts);
    const size_t bytes = (size_t)ts->P * 4;
    if (ts->theta.ensure(bytes) || ts->grad.ensure(bytes) || ts->m.ensure(bytes) || ts->v.ensure(bytes)) return 1;
    std::vector<float> flat;
    train_gather_host(h, ts, flat);
    HIPCHK(hipMemcpyAsync(ts->theta.p, flat.data(), bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(ts->grad.p, 0, bytes, h->stream));
    HIPCHK(hipMemsetAsync(ts->m.p, 0, bytes, h->stream));
    HIPCHK(hipMemsetAsync(ts->v.p, 0, bytes, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    ts->lr = lr; ts->b1 = b1; ts->b2 = b2; ts->eps = eps;
    ts->step = 0;
    ts->ready = true;
    ts->dev_newer = false;
    return 0;
}

static inline unsigned t_grid(size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, 16384); }

// Forward + backward of the literal dense algorithm for B molecules padded to N.  Device pointers:
// e [B][N][N][48], mask [B][N][N], x [B][N][nx], h0 [B][N][48], q0 [B][N], y [B][N] -> pred [B][N], loss [B];
// gradients are ADDED into ts->grad (caller zeroes it).
static int train_fwd_bwd(epnn_handle *h, int B, int N, const float *d_e, const float *d_mask, const float *d_x,
                         const float *d_h0, const float *d_q0, const float *d_y, float *d_pred, float *d_loss,
                         bool size_only = false) {
    TrainState *ts = train_state(h);
    if (!ts->ready) EPNN_FAIL("training: call epnn_train_init first");
    const int T = h->cfg.T, nx = h->cfg.nx, H = EPNN_EDIM, E = EPNN_EDIM, F = nx + H + 1, D = 2 * F + E;
    const int BN = B * N;
    const size_t R = (size_t)BN * N;
    hipStream_t st = h->stream;
    const float *theta = ts->theta.as<float>();
    float *grad = ts->grad.as<float>();
    const int NSL = 256;          // most row slices of one dW reduction (fixed by the row count: reproducible)
    // the update MLP: three layers of the reference's own [32, 32], or what make_model(layers, ...) asked for (charge_gn.py:371)
    const bool genu = h->upd_generic;
    const int NU = genu ? (int)ts->updv.size() : 3;
    const TDense *UL = genu ? ts->updv.data() : ts->upd;
    size_t part_elems = (size_t)(D + 1) * 48;
    for (int l = 0; l < NU; ++l) part_elems = std::max(part_elems, (size_t)(UL[l].n_in + 1) * UL[l].n_out);
    if (ts->part.ensure((size_t)NSL * part_elems * 4)) return 1;
    // ---- arena
    size_t need = 0;
    auto sz = [&](size_t n) { size_t o = need; need += (n + 63) & ~size_t(63); return o; };
    struct GStep { size_t a, X, H1, H2, Mij, M, U[EPNN_GMLP_LMAX], hraw, hn; } gs[EPNN_MAXT];      // U[l]: input rows of update layer l
    struct EStep { size_t a, XN, H1N, H2N, fN, XT, H1T, H2T, fT, qn; } es[EPNN_MAXT];
    const size_t o_nm = sz(BN), o_wgt = sz(R);
    for (int t = 0; t < T; ++t) {
        GStep &g = gs[t];
        g.a = sz((size_t)BN * F); g.X = sz(R * D); g.H1 = sz(R * 32); g.H2 = sz(R * 32); g.Mij = sz(R * 32); g.M = sz((size_t)BN * 32);
        for (int l = 0; l < NU; ++l) g.U[l] = sz((size_t)BN * UL[l].n_in);
        g.hraw = sz((size_t)BN * H); g.hn = sz((size_t)BN * H);
    }
    for (int t = 0; t < T; ++t) {
        es[t] = {sz((size_t)BN * F), sz(R * D), sz(R * 32), sz(R * 32), sz(R), sz(R * D), sz(R * 32), sz(R * 32), sz(R), sz(BN)};
    }
    const size_t o_dX = sz(R * D), o_dA = sz(R * 32), o_dB = sz(R * 32), o_dC = sz(R * 32), o_da = sz((size_t)BN * F);
    size_t o_dU[EPNN_GMLP_LMAX];                       // gradient with respect to the input rows of update layer l
    for (int l = 0; l < NU; ++l) o_dU[l] = sz((size_t)BN * UL[l].n_in);
    const size_t o_dU0 = o_dU[0], o_dh = sz((size_t)BN * H),
                 o_gh = sz((size_t)BN * H), o_gfeat = sz((size_t)BN * H), o_gq = sz(BN), o_dfN = sz(R), o_dfT = sz(R);
    if (ts->arena.ensure(need * 4)) return 1;
    if (size_only) return 0;            // scratch is allocated: nothing below calls the allocator (graph capture)
    float *ar = ts->arena.as<float>();
    auto P = [&](size_t off) { return ar + off; };
    float *nm = P(o_nm), *wgt = P(o_wgt);

    auto dense = [&](const float *X, const TDense &d, float *Y, size_t rows, int relu) {
        if (d.n_out >= 16) {         // matrix pipe: one 32x32 output tile per wavefront
            const int nto = (d.n_out + 31) / 32;
            const size_t tiles = ((rows + 31) / 32) * nto;
            hipLaunchKernelGGL(k_t_mm_fwd, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, X, theta + d.offW, theta + d.offB, Y,
                               (int)rows, d.n_in, d.n_out, relu, nto);
        } else {
            hipLaunchKernelGGL(k_t_dense, dim3(t_grid(rows * d.n_out)), dim3(256), 0, st, X, theta + d.offW, theta + d.offB, Y,
                               (int)rows, d.n_in, d.n_out, relu);
        }
    };
    // backward of one Dense: dX (optional) and gradient accumulation
    auto dense_bwd = [&](const float *X, const float *dY, const float *Ypost, const TDense &d, float *dX, size_t rows) {
        if (dX) {
            const int ntk = (d.n_in + 31) / 32;
            const size_t tiles = ((rows + 31) / 32) * ntk;
            hipLaunchKernelGGL(k_t_mm_dx, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, dY, Ypost, theta + d.offW, dX,
                               (int)rows, d.n_in, d.n_out, ntk);
        }
        // ~64 rows per slice: the pair-row GEMMs (B*N*N rows) spread over the whole GPU, the per-atom ones stay small
        const int nsl = (int)std::min<size_t>(NSL, std::max<size_t>(1, rows / 64));
        const int tot = (d.n_in + 1) * d.n_out;
        {
            const int ntk = (d.n_in + 1 + 31) / 32, nto = (d.n_out + 31) / 32;
            const size_t tiles = (size_t)nsl * ntk * nto;
            hipLaunchKernelGGL(k_t_mm_dw, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, X, dY, Ypost, ts->part.as<float>(),
                               (int)rows, d.n_in, d.n_out, nsl, ntk, nto);
        }
        hipLaunchKernelGGL(k_t_dw_reduce, dim3((tot + 255) / 256), dim3(256), 0, st, ts->part.as<float>(), grad, d.offW,
                           d.offB, d.n_in, d.n_out, nsl);
    };

    hipLaunchKernelGGL(k_t_nodemask, dim3(t_grid(BN)), dim3(256), 0, st, d_mask, nm, B, N);
    hipLaunchKernelGGL(k_t_wgt, dim3(t_grid(R)), dim3(256), 0, st, d_e, d_mask, wgt, (int)R, E, h->cfg.near_tol);
    // ================================================================ forward: GNN (charge_gn.py:60-74)
    const float *hcur = d_h0;
    for (int t = 0; t < T; ++t) {
        const GStep &g = gs[t];
        hipLaunchKernelGGL(k_t_assemble, dim3(t_grid((size_t)BN * F)), dim3(256), 0, st, d_x, hcur, d_q0, P(g.a), BN, nx, H);
        hipLaunchKernelGGL(k_t_rows, dim3(t_grid(R * D)), dim3(256), 0, st, P(g.a), d_e, P(g.X), B, N, F, E, 0);
        dense(P(g.X), ts->msg[t][0], P(g.H1), R, 1);
        dense(P(g.H1), ts->msg[t][1], P(g.H2), R, 1);
        dense(P(g.H2), ts->msg[t][2], P(g.Mij), R, 0);
        hipLaunchKernelGGL(k_t_sumj, dim3(t_grid((size_t)BN * 32)), dim3(256), 0, st, P(g.Mij), P(g.M), B, N, 32);
        hipLaunchKernelGGL(k_t_u0, dim3(t_grid((size_t)BN * 80)), dim3(256), 0, st, hcur, P(g.M), nm, P(g.U[0]), BN, H, 32);
        for (int l = 0; l < NU; ++l) dense(P(g.U[l]), UL[l], l + 1 < NU ? P(g.U[l + 1]) : P(g.hraw), BN, l + 1 < NU ? 1 : 0);
        hipLaunchKernelGGL(k_t_scale_rows, dim3(t_grid((size_t)BN * H)), dim3(256), 0, st, P(g.hraw), nm, P(g.hn), BN, H);
        hcur = P(g.hn);
    }
    const float *feats = hcur;
    // ================================================================ forward: EPN (charge_gn.py:98-118)
    const float *qcur = d_q0;
    for (int t = 0; t < T; ++t) {
        const EStep &s = es[t];
        hipLaunchKernelGGL(k_t_assemble, dim3(t_grid((size_t)BN * F)), dim3(256), 0, st, d_x, feats, qcur, P(s.a), BN, nx, H);
        hipLaunchKernelGGL(k_t_rows, dim3(t_grid(R * D)), dim3(256), 0, st, P(s.a), d_e, P(s.XN), B, N, F, E, 0);
        hipLaunchKernelGGL(k_t_rows, dim3(t_grid(R * D)), dim3(256), 0, st, P(s.a), d_e, P(s.XT), B, N, F, E, 1);
        dense(P(s.XN), ts->pas[t][0], P(s.H1N), R, 1);
        dense(P(s.H1N), ts->pas[t][1], P(s.H2N), R, 1);
        dense(P(s.H2N), ts->pas[t][2], P(s.fN), R, 0);
        dense(P(s.XT), ts->pas[t][0], P(s.H1T), R, 1);
        dense(P(s.H1T), ts->pas[t][1], P(s.H2T), R, 1);
        dense(P(s.H2T), ts->pas[t][2], P(s.fT), R, 0);
        hipLaunchKernelGGL(k_t_epn_q, dim3(t_grid(BN)), dim3(256), 0, st, P(s.fN), P(s.fT), wgt, qcur, P(s.qn), BN, N);
        qcur = P(s.qn);
    }
    HIPCHK(hipMemcpyAsync(d_pred, qcur, (size_t)BN * 4, hipMemcpyDeviceToDevice, st));
    // ================================================================ loss (charge_gn.py:397-398)
    float *gq = P(o_gq), *gfeat = P(o_gfeat), *gh = P(o_gh), *da = P(o_da);
    hipLaunchKernelGGL(k_t_loss, dim3(B), dim3(64), 0, st, d_y, qcur, gq, d_loss, N);
    HIPCHK(hipMemsetAsync(gfeat, 0, (size_t)BN * H * 4, st));
    // ================================================================ backward: EPN
    for (int t = T - 1; t >= 0; --t) {
        const EStep &s = es[t];
        hipLaunchKernelGGL(k_t_epn_df, dim3(t_grid(R)), dim3(256), 0, st, gq, wgt, P(o_dfN), P(o_dfT), BN, N);
        HIPCHK(hipMemsetAsync(da, 0, (size_t)BN * F * 4, st));
        for (int dir = 0; dir < 2; ++dir) {
            const float *X = P(dir ? s.XT : s.XN), *H1 = P(dir ? s.H1T : s.H1N), *H2 = P(dir ? s.H2T : s.H2N);
            const float *df = P(dir ? o_dfT : o_dfN);
            dense_bwd(H2, df, nullptr, ts->pas[t][2], P(o_dA), R);          // dH2 (post-activation gradient)
            dense_bwd(H1, P(o_dA), H2, ts->pas[t][1], P(o_dB), R);          // masks dH2 by H2 > 0
            dense_bwd(X, P(o_dB), H1, ts->pas[t][0], P(o_dX), R);
            hipLaunchKernelGGL(k_t_rows_bwd, dim3(t_grid((size_t)BN * F)), dim3(256), 0, st, P(o_dX), da, B, N, F, E, dir);
        }
        hipLaunchKernelGGL(k_t_epn_fold, dim3(t_grid((size_t)BN * (H + 1))), dim3(256), 0, st, da, gfeat, gq, BN, nx, H);
    }
    // ================================================================ backward: GNN
    HIPCHK(hipMemcpyAsync(gh, gfeat, (size_t)BN * H * 4, hipMemcpyDeviceToDevice, st));
    for (int t = T - 1; t >= 0; --t) {
        const GStep &g = gs[t];
        // h_{t+1} = hraw * nm
        hipLaunchKernelGGL(k_t_scale_rows, dim3(t_grid((size_t)BN * H)), dim3(256), 0, st, gh, nm, P(o_dh), BN, H);
        for (int l = NU - 1; l >= 0; --l)              // (a hidden layer's gradient is masked by its own output > 0: the next layer's input)
            dense_bwd(P(g.U[l]), l + 1 < NU ? P(o_dU[l + 1]) : P(o_dh), l + 1 < NU ? P(g.U[l + 1]) : nullptr, UL[l], P(o_dU[l]), BN);
        // U0 = [h | M] * nm ; M_i = sum_j m_ij
        hipLaunchKernelGGL(k_t_bcast_gm, dim3(t_grid(R * 32)), dim3(256), 0, st, P(o_dU0), nm, P(o_dC), BN, N, H, 32);
        dense_bwd(P(g.H2), P(o_dC), nullptr, ts->msg[t][2], P(o_dA), R);
        dense_bwd(P(g.H1), P(o_dA), P(g.H2), ts->msg[t][1], P(o_dB), R);
        dense_bwd(P(g.X), P(o_dB), P(g.H1), ts->msg[t][0], P(o_dX), R);
        HIPCHK(hipMemsetAsync(da, 0, (size_t)BN * F * 4, st));
        hipLaunchKernelGGL(k_t_rows_bwd, dim3(t_grid((size_t)BN * F)), dim3(256), 0, st, P(o_dX), da, B, N, F, E, 0);
        hipLaunchKernelGGL(k_t_gh_prev, dim3(t_grid((size_t)BN * H)), dim3(256), 0, st, P(o_dU0), nm, da, gh, BN, nx, H, 32);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// The same forward + backward with the row-fused kernels of epnn_train_fused.hip.h (N <= EPNN_TF_NMAX).  2T forward and 2T
// backward launches and one that sums the weight-gradient partials (and, with `adam_now`, takes the optimizer step):
// node masks, pair weights, loss terms and the "atoms" stage between two backward sweeps ride in their neighbours (round 3;
// they were 18 launches of their own).  d_loss receives one loss term per atom slot [B][N].
static int train_fwd_bwd_fused(epnn_handle *h, int B, int N, const float *d_e, const float *d_mask, const float *d_x,
                               const float *d_h0, const float *d_q0, const float *d_y, float *d_pred, float *d_loss,
                               bool size_only = false, bool adam_now = false, float *out_host = nullptr, bool step_on_device = false) {
    TrainState *ts = train_state(h);
    if (!ts->ready) EPNN_FAIL("training: call epnn_train_init first");
    ts->host_out = out_host != nullptr;
    if (step_on_device && ts->d_step.ensure(8)) return 1;
    const int T = h->cfg.T, nx = h->cfg.nx, H = EPNN_EDIM, E = EPNN_EDIM, F = nx + H + 1, D = 2 * F + E, FS = F | 1;
    const int BN = B * N;
    const size_t R = (size_t)BN * N;
    hipStream_t st = h->stream;
    const float *theta = ts->theta.as<float>();
    float *grad = ts->grad.as<float>();
    const int Pm0 = D * 32 + 32 + 1024 + 32 + 32 * 32 + 32, Pm1 = D * 32 + 32 + 1024 + 32 + 32 + 1;
    if (2 * T + 1 > EPNN_TF_MAXRED) EPNN_FAIL("training: T = %d exceeds the fused step's reduction table", T);
    if (T < 1) EPNN_FAIL("training: T must be at least 1");
    // ---- arena
    size_t need = 0;
    auto sz = [&](size_t n) { size_t o = need; need += (n + 63) & ~size_t(63); return o; };
    struct GStep { size_t H1, H2, M, U0, U1, U2, hn; } gs[EPNN_MAXT];
    struct EStep { size_t H1, H2, qn; } es[EPNN_MAXT];
    const size_t o_nm = sz(BN), o_wgt = sz(R);
    for (int t = 0; t < T; ++t)
        gs[t] = {sz(R * 32), sz(R * 32), sz((size_t)BN * 32), sz((size_t)BN * 80), sz((size_t)BN * 32), sz((size_t)BN * 32), sz((size_t)BN * H)};
    for (int t = 0; t < T; ++t) es[t] = {sz(2 * R * 32), sz(2 * R * 32), sz(BN)};
    const size_t o_dz1a = sz(2 * R * 32), o_dz1b = sz(2 * R * 32), o_dU0 = sz((size_t)BN * 80), o_gh = sz((size_t)BN * H),
                 o_gfeat = sz((size_t)BN * H), o_gq = sz(BN);
    // matrix-pipe backward: second copies of what the scalar kernels update in place (several workgroups per atom read the
    // previous launch's copy while one of them writes this launch's), and the row sums a sweep leaves for the next prologue
    const size_t o_dU0b = sz((size_t)BN * 80), o_gfeatb = sz((size_t)BN * H), o_gqb = sz(BN), o_rsa = sz((size_t)BN * 64), o_rsb = sz((size_t)BN * 64);
    size_t o_pm[EPNN_MAXT], o_pp[EPNN_MAXT];
    for (int t = 0; t < T; ++t) o_pm[t] = sz((size_t)BN * Pm0);
    for (int t = 0; t < T; ++t) o_pp[t] = sz((size_t)BN * Pm1);
    const size_t o_pu = sz((size_t)T * BN * EPNN_TF_PU);
    if (ts->arena.ensure(need * 4)) return 1;
    const size_t lds_fwd = ((size_t)N * FS + (size_t)N * 49 + (size_t)D * 32 + 4 * (size_t)N * 33 + EPNN_TF_NG * 32 + 32 + 3 * (size_t)N) * 4;
    if (!ts->fused_attr) {
        const int cap = 160 * 1024;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tf_pair_fwd<0>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tf_pair_fwd<1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tb_pair_bwd_mm<0>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tb_pair_bwd_mm<1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        ts->fused_attr = true;
    }
    if (size_only) return 0;            // scratch is allocated: nothing below calls the allocator (graph capture)
    float *ar = ts->arena.as<float>();
    auto P = [&](size_t off) { return ar + off; };
    float *nm = P(o_nm), *wgt = P(o_wgt);
    float *gq = P(o_gq), *gfeat = P(o_gfeat), *gh = P(o_gh);
    // (the row-fused kernels stage float4 rows of e: every buffer this library hands them is 256-byte aligned; nx + 49 <= EPNN_TF_FMAX by epnn_create)
    if (((uintptr_t)d_e & 15) != 0 || nx + 49 > EPNN_TF_FMAX) EPNN_FAIL("training: internal error (row-fused kernels: e not 16-byte aligned or %d atom features > %d)", nx + 49, EPNN_TF_FMAX);
#ifdef EPNN_TF_CLOCKS
    int nclk = 0;
    if (ts->clk.ensure(64 * 16 * 8)) return 1;
    HIPCHK(hipMemsetAsync(ts->clk.p, 0, 64 * 16 * 8, st));
#endif
    auto pair_args = [&](const TDense *mlp, const float *hh, const float *qq) {
        TfPair A{};
        A.x = d_x; A.h = hh; A.q = qq; A.e = d_e; A.theta = theta;
        A.oW1 = mlp[0].offW; A.ob1 = mlp[0].offB; A.oW2 = mlp[1].offW; A.ob2 = mlp[1].offB; A.oW3 = mlp[2].offW; A.ob3 = mlp[2].offB;
        A.N = N; A.nx = nx; A.wgt = wgt; A.nm = nm;
        A.nm_w = nm; A.wgt_w = wgt; A.tol = h->cfg.near_tol; A.pmode = -1;
        A.gfeat = gfeat; A.gh = gh; A.gqv = gq; A.gq = gq; A.dU0 = P(o_dU0);
        A.moff = h->tr_moff; A.real = h->tr_real;
#ifdef EPNN_TF_CLOCKS
        A.clk = ts->clk.as<unsigned long long>() + 16 * (nclk++);
#endif
        return A;
    };
    auto upd_args = [&](int t, const float *hh) {
        TfUpd U{};
        U.h = hh; U.M = P(gs[t].M); U.nm = nm; U.theta = theta;
        U.oW0 = ts->upd[0].offW; U.ob0 = ts->upd[0].offB; U.oW1 = ts->upd[1].offW; U.ob1 = ts->upd[1].offB;
        U.oW2 = ts->upd[2].offW; U.ob2 = ts->upd[2].offB;
        U.U0 = P(gs[t].U0); U.U1 = P(gs[t].U1); U.U2 = P(gs[t].U2); U.hn = P(gs[t].hn);
        U.gh = gh; U.dU0 = P(o_dU0); U.part = P(o_pu) + (size_t)t * BN * EPNN_TF_PU;
        return U;
    };
    auto lds_bwd_mm = [&](int nd) { return ((size_t)4 * nd * N * EPNN_TB_RS + (size_t)N * EPNN_TB_ES + 1168 + (size_t)N * FS) * 4; };
    // one molecule per step is N workgroups on 256 CUs: up to six workgroups per atom share its weight-gradient jobs
    // (an XCD has 32 CUs and a workgroup of these kernels has a CU to itself: the shares of an XCD's atoms must fit it in one round)
    const int nsplit = h->opt_train_split ? h->opt_train_split : std::max(1, std::min(6, 32 / ((BN + 7) / 8)));
    const unsigned bwd_grid = 8u * (unsigned)((BN + 7) / 8) * (unsigned)nsplit;          // eight XCDs, equal parts (k_tb_pair_bwd_mm)
    // ================================================================ forward: GNN (charge_gn.py:60-74)
    const float *hcur = d_h0;
    for (int t = 0; t < T; ++t) {
        TfPair A = pair_args(ts->msg[t], hcur, d_q0);
        A.H1 = P(gs[t].H1); A.H2 = P(gs[t].H2); A.M = P(gs[t].M);
        if (t == 0) A.mask = d_mask;                                    // ... and the node masks
        if (t == 0 && adam_now && step_on_device) A.step_p = ts->d_step.as<long long>();
        hipLaunchKernelGGL((k_tf_pair_fwd<0>), dim3(BN), dim3(EPNN_TF_NT), lds_fwd, st, A, upd_args(t, hcur));    // + update MLP
        hcur = P(gs[t].hn);
    }
    const float *feats = hcur;
    // ================================================================ forward: EPN (charge_gn.py:98-118), loss terms (:397)
    const float *qcur = d_q0;
    for (int t = 0; t < T; ++t) {
        TfPair A = pair_args(ts->pas[t], feats, qcur);
        A.H1 = P(es[t].H1); A.H2 = P(es[t].H2); A.qn = P(es[t].qn);
        if (t == 0) A.mask = d_mask;                                    // ... and the pair weights
        if (t == T - 1) {
            A.y = d_y; A.pred = d_pred; A.lterm = d_loss;
            A.out_h = out_host;
        }
        hipLaunchKernelGGL((k_tf_pair_fwd<1>), dim3(BN), dim3(EPNN_TF_NT), lds_fwd, st, A, TfUpd{});
        qcur = P(es[t].qn);
    }
    if (ts->host_out && ts->ev_fwd) HIPCHK(hipEventRecord(ts->ev_fwd, st));        // (an event-record node when the step is being captured)
    // ================================================================ backward: EPN, then GNN.  Every launch starts with the
    // "atoms" stage of the one before it (the gradient that reached the atoms through that sweep's first Dense); the
    // sweeps alternate between two dz1 buffers, so a launch may still read the previous one's while it writes its own.
    int nb = 0;
    int pmode = -1, poW1 = 0, pfirst = 0;
    auto chain = [&](TfPair &A) {
        A.dz1 = P((nb & 1) ? o_dz1b : o_dz1a);
        A.pdz1 = P((nb & 1) ? o_dz1a : o_dz1b);
        A.pmode = pmode; A.poW1 = poW1; A.pfirst = pfirst;
        const bool odd = nb & 1;
        A.nsplit = nsplit; A.natoms = BN;
        A.gfeat = P(odd ? o_gfeatb : o_gfeat); A.gfeat_r = P(odd ? o_gfeat : o_gfeatb);
        A.gqv = P(odd ? o_gqb : o_gq); A.gq_r = P(odd ? o_gq : o_gqb);
        A.dU0_r = P(odd ? o_dU0 : o_dU0b);
        A.rs_w = P(odd ? o_rsb : o_rsa); A.rs_r = P(odd ? o_rsa : o_rsb);
        nb += 1;
    };
    for (int t = T - 1; t >= 0; --t) {
        TfPair A = pair_args(ts->pas[t], feats, t ? P(es[t - 1].qn) : d_q0);
        A.H1 = P(es[t].H1); A.H2 = P(es[t].H2); A.part = P(o_pp[t]);
        A.first = t == T - 1; A.y = d_y; A.pred = d_pred;
        chain(A);
        hipLaunchKernelGGL(k_tb_pair_bwd_mm<1>, dim3(bwd_grid), dim3(EPNN_TF_NT), lds_bwd_mm(2), st, A, TfUpd{});
        pmode = 1; poW1 = ts->pas[t][0].offW; pfirst = t == T - 1;
    }
    for (int t = T - 1; t >= 0; --t) {
        const float *hin = t ? P(gs[t - 1].hn) : d_h0;
        TfPair A = pair_args(ts->msg[t], hin, d_q0);
        A.H1 = P(gs[t].H1); A.H2 = P(gs[t].H2); A.part = P(o_pm[t]);
        chain(A);
        TfUpd Ub = upd_args(t, hin);                                   // (the update MLP's backward first)
        Ub.dU0 = P(((nb - 1) & 1) ? o_dU0b : o_dU0);                   // chain() has counted this launch
        hipLaunchKernelGGL(k_tb_pair_bwd_mm<0>, dim3(bwd_grid), dim3(EPNN_TF_NT), lds_bwd_mm(1), st, A, Ub);
        pmode = 0; poW1 = ts->msg[t][0].offW; pfirst = 0;
    }
    // ================================================================ gradient = sum of the workgroups' partials (+ Adam)
    TfReduce Rd{};
    int maxlen = 0;
    auto entry = [&](int theta_off, int len, int nblk_, size_t part_off) {
        Rd.theta_off[Rd.n] = theta_off; Rd.len[Rd.n] = len; Rd.nblk[Rd.n] = nblk_; Rd.part_off[Rd.n] = part_off;
        Rd.n += 1;
        maxlen = std::max(maxlen, len);
    };
    entry(ts->upd[0].offW, EPNN_TF_PU, T * BN, o_pu);
    for (int t = 0; t < T; ++t) entry(ts->msg[t][0].offW, Pm0, BN, o_pm[t]);
    for (int t = 0; t < T; ++t) entry(ts->pas[t][0].offW, Pm1, BN, o_pp[t]);
    if (h->tr_real) { Rd.real = h->tr_real; Rd.natoms = BN; }
    if (adam_now) {
        Rd.adam = 1;
        if (step_on_device) {                       // captured into a hipGraph: the caller keeps ts->step and the device counter in step
            Rd.step_p = ts->d_step.as<long long>();
            Rd.lr = ts->lr;
        } else {
            ts->step += 1;
            Rd.alpha = epnn_adam_alpha(ts->lr, ts->b1, ts->b2, ts->step);
        }
        Rd.b1 = ts->b1; Rd.b2 = ts->b2; Rd.eps = ts->eps;
        Rd.theta = ts->theta.as<float>(); Rd.m = ts->m.as<float>(); Rd.v = ts->v.as<float>();
        ts->dev_newer = true;
    }
    hipLaunchKernelGGL(k_tb_wreduce, dim3((unsigned)((maxlen + 63) / 64), (unsigned)Rd.n), dim3(256), 0, st, Rd, ar, grad);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ the forward alone
// The row-fused forward kernels as an INFERENCE path for the dense entry's small calls (one or a few molecules whose largest
// fills most of the padded size): 2T launches of B N workgroups, one per atom slot, against one molecule on 1-4 wavefronts
// of ONE CU in the fused inference kernels (58-141 us at N = 41).  The literal arithmetic of the reference (rows
// [a_i | a_j | e_ij], all N slots, both orders of a pair), nothing stored for a backward pass.  The weights are a flat
// parameter vector of their own, refreshed when the handle's weights change.
struct InferFused {
    long gen = -1;
    int P = 0;
    TDense upd[3], msg[EPNN_MAXT][3], pas[EPNN_MAXT][3];
    DevBuf theta, arena;
    bool attr = false;
};
static bool infer_rowfused_fits(const epnn_handle *h, int N) { return N <= EPNN_TF_NMAX && h->cfg.nx + 49 <= EPNN_TF_FMAX && h->cfg.T >= 1 && !h->upd_generic; }
static int infer_rowfused_forward(epnn_handle *h, int B, int N, const float *d_e, const float *d_mask, const float *d_x,
                                  const float *d_h0, const float *d_q0, float *d_out) {
    if (!h->infer_fused) h->infer_fused = new InferFused();
    InferFused *is = reinterpret_cast<InferFused *>(h->infer_fused);
    const int T = h->cfg.T, nx = h->cfg.nx, H = EPNN_EDIM, F = nx + H + 1, D = 2 * F + H, FS = F | 1;
    const int BN = B * N;
    const size_t R = (size_t)BN * N;
    hipStream_t st = h->stream;
    if (is->gen != h->weights_gen) {
        train_layout(h, is);
        if (is->theta.ensure((size_t)is->P * 4)) return 1;
        std::vector<float> flat;
        train_gather_host(h, is, flat);
        HIPCHK(hipMemcpyAsync(is->theta.p, flat.data(), (size_t)is->P * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));                      // (`flat` is pageable and goes out of scope)
        is->gen = h->weights_gen;
    }
    size_t need = 0;
    auto sz = [&](size_t n) { size_t o = need; need += (n + 63) & ~size_t(63); return o; };
    const size_t o_nm = sz(BN), o_wgt = sz(R), o_ha = sz((size_t)BN * H), o_hb = sz((size_t)BN * H), o_qa = sz(BN), o_qb = sz(BN);
    if (is->arena.ensure(need * 4)) return 1;
    float *ar = is->arena.as<float>();
    const size_t lds_fwd = ((size_t)N * FS + (size_t)N * 49 + (size_t)D * 32 + 4 * (size_t)N * 33 + EPNN_TF_NG * 32 + 32 + 3 * (size_t)N) * 4;
    if (!is->attr) {
        const int cap = 160 * 1024;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tf_pair_fwd<0>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tf_pair_fwd<1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        is->attr = true;
    }
    const float *theta = is->theta.as<float>();
    auto pair_args = [&](const TDense *mlp, const float *hh, const float *qq) {
        TfPair A{};
        A.x = d_x; A.h = hh; A.q = qq; A.e = d_e; A.theta = theta;
        A.oW1 = mlp[0].offW; A.ob1 = mlp[0].offB; A.oW2 = mlp[1].offW; A.ob2 = mlp[1].offB; A.oW3 = mlp[2].offW; A.ob3 = mlp[2].offB;
        A.N = N; A.nx = nx; A.wgt = ar + o_wgt; A.nm = ar + o_nm;
        A.nm_w = ar + o_nm; A.wgt_w = ar + o_wgt; A.tol = h->cfg.near_tol; A.pmode = -1;
        return A;
    };
    const float *hcur = d_h0;
    for (int t = 0; t < T; ++t) {
        TfPair A = pair_args(is->msg[t], hcur, d_q0);
        if (t == 0) A.mask = d_mask;
        TfUpd U{};
        U.h = hcur; U.nm = ar + o_nm; U.theta = theta;
        U.oW0 = is->upd[0].offW; U.ob0 = is->upd[0].offB; U.oW1 = is->upd[1].offW; U.ob1 = is->upd[1].offB;
        U.oW2 = is->upd[2].offW; U.ob2 = is->upd[2].offB;
        U.hn = ar + ((t & 1) ? o_hb : o_ha);
        hipLaunchKernelGGL((k_tf_pair_fwd<0>), dim3(BN), dim3(EPNN_TF_NT), lds_fwd, st, A, U);
        hcur = U.hn;
    }
    const float *qcur = d_q0;
    for (int t = 0; t < T; ++t) {
        TfPair A = pair_args(is->pas[t], hcur, qcur);
        if (t == 0) A.mask = d_mask;
        A.qn = t == T - 1 ? d_out : ar + ((t & 1) ? o_qb : o_qa);
        hipLaunchKernelGGL((k_tf_pair_fwd<1>), dim3(BN), dim3(EPNN_TF_NT), lds_fwd, st, A, TfUpd{});
        qcur = A.qn;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static int train_apply(epnn_handle *h) {
    TrainState *ts = train_state(h);
    if (comm_collectives(h)) {
        // every rank says "my gradient is ready" before anyone enqueues the all-reduce (comm_guard, epnn_host.h)
        if (comm_guard(h, 0, "train step (gradient all-reduce)")) return 1;
        ncclResult_t rc = ncclAllReduce(ts->grad.p, ts->grad.p, (size_t)ts->P, ncclFloat, ncclSum, h->comm, h->stream);
        if (rc != ncclSuccess) EPNN_FAIL("ncclAllReduce failed: %s", ncclGetErrorString(rc));
    }
    ts->step += 1;
    const float alpha = epnn_adam_alpha(ts->lr, ts->b1, ts->b2, ts->step);
    hipLaunchKernelGGL(k_t_adam, dim3(t_grid(ts->P)), dim3(256), 0, h->stream, ts->theta.as<float>(), ts->grad.as<float>(),
                       ts->m.as<float>(), ts->v.as<float>(), ts->P, alpha, ts->b1, ts->b2, ts->eps);
    HIPCHK(hipGetLastError());
    ts->dev_newer = true;
    return 0;
}
