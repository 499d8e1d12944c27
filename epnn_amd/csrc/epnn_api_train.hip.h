// train_step (charge_gn.py:393-402): entry points, hipGraph replay, the RCCL communicator and its collectives; debug exports.
// Part of the one translation unit epnn_api.hip.
#pragma once

// ------------------------------------------------------------------------------------------------ training
// dense (B,N,N,.) make_model inputs from a flat coordinate batch: what gen_padded_init_state builds on the host
// the body: `in(k)` reads word k of the staged block  offsets | xyz | x | Q | y  (word offsets o_*), wherever that block is
template <typename IN>
__device__ __forceinline__ void t_pad_inputs_body(IN &&in, int o_xyz, int o_x, int o_Q, int o_y, int B, int N, int nx, int E, double cutoff,
                                                  double eta, const double *mu, float *e, float *mask, float *xs, float *hs, float *qs,
                                                  float *ys, int *real_out, int *moff_out) {
    // a thread per (pair, four channels): one thread per pair was 48 double-precision exp in a row on 7 workgroups (13 us of a
    // 0.27 ms one-molecule step); the distance and the cutoff are recomputed by the 12 threads of a pair
    const size_t pairs = (size_t)B * N * N;
    const int G = (E + 3) / 4;
    const double pi_d = 3.141592653589793;
    if (moff_out && blockIdx.x == 0 && (int)threadIdx.x <= B) moff_out[threadIdx.x] = __float_as_int(in((int)threadIdx.x));
    for (size_t it = (size_t)blockIdx.x * 256 + threadIdx.x; it < pairs * G; it += (size_t)gridDim.x * 256) {
        const size_t r = it / G;
        const int cg = (int)(it - r * G);
        const int j = (int)(r % N), i = (int)((r / N) % N), b = (int)(r / ((size_t)N * N));
        const int a0 = __float_as_int(in(b)), n = __float_as_int(in(b + 1)) - a0;
        const bool real = i < n && j < n;
        if (cg == 0) mask[r] = real ? 1.f : 0.f;
        double D = 0, Cc = 0;
        if (real) {
            // distance exactly as scipy.spatial.distance_matrix on float32 coordinates promoted to float64 (epnn_dist)
            const int pi_ = o_xyz + 3 * (a0 + i), pj_ = o_xyz + 3 * (a0 + j);
            const double dx = (double)in(pj_) - (double)in(pi_), dy = (double)in(pj_ + 1) - (double)in(pi_ + 1),
                         dz = (double)in(pj_ + 2) - (double)in(pi_ + 2);
            D = sqrt(__dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz)));
            Cc = (cos(pi_d * (D - 0.0) / cutoff) + 1.0) / 2.0;
            if (D >= cutoff) Cc = 0.0;
            if (D <= 0.0) Cc = 1.0;
            if (i == j) Cc = 0.0;
        }
        for (int ch = 4 * cg; ch < min(E, 4 * cg + 4); ++ch) {
            const double d = D - mu[ch];
            e[r * E + ch] = real ? (float)(Cc * exp(-eta * (d * d))) : 0.f;
        }
        if (j == 0 && cg == 0) {
            const size_t at = (size_t)b * N + i;
            for (int f = 0; f < nx; ++f) xs[at * nx + f] = i < n ? in(o_x + (a0 + i) * nx + f) : 0.f;
            for (int f = 0; f < EPNN_EDIM; ++f) hs[at * EPNN_EDIM + f] = 0.f;
            qs[at] = i < n ? in(o_Q + b) / (float)n : 0.f;
            ys[at] = i < n ? in(o_y + a0 + i) : 0.f;
            real_out[at] = i < n;
        }
    }
}
// the staged block in device memory (uploaded before the launch)
__global__ __launch_bounds__(256) void k_t_pad_inputs(const float *blk, int o_xyz, int o_x, int o_Q, int o_y, int B, int N, int nx, int E,
                                                      double cutoff, double eta, const double *mu, float *e, float *mask, float *xs,
                                                      float *hs, float *qs, float *ys, int *real_out) {
    t_pad_inputs_body([&](int k) { return blk[k]; }, o_xyz, o_x, o_Q, o_y, B, N, nx, E, cutoff, eta, mu, e, mask, xs, hs, qs, ys, real_out, nullptr);
}
// ... or riding in the kernel's own argument block (up to 3.6 KB: one molecule of up to ~69 atoms): no upload, i.e. no copy kernel
// and no launch boundary in front of the step (5 us of a 0.22 ms one-molecule step); the offsets are left in device memory for the
// step's kernels (moff_out)
#define EPNN_PAD_INLINE_WORDS 900
struct PadInline { float w[EPNN_PAD_INLINE_WORDS]; };
__global__ __launch_bounds__(256) void k_t_pad_inputs_inline(const PadInline P, int o_xyz, int o_x, int o_Q, int o_y, int B, int N, int nx,
                                                             int E, double cutoff, double eta, const double *mu, float *e, float *mask,
                                                             float *xs, float *hs, float *qs, float *ys, int *real_out, int *moff_out) {
    t_pad_inputs_body([&](int k) { return P.w[k]; }, o_xyz, o_x, o_Q, o_y, B, N, nx, E, cutoff, eta, mu, e, mask, xs, hs, qs, ys, real_out, moff_out);
}

// A step may have returned while its backward pass was still running ("train_async"): before anything it reads can be reallocated
// or destroyed, wait for it
static int train_quiesce(epnn_handle *h) {
    if (h->train && train_state(h)->inflight) {
        HIPCHK(hipStreamSynchronize(h->stream));
        train_state(h)->inflight = false;
    }
    return 0;
}
extern "C" int epnn_train_init(epnn_handle *h, float lr, float beta1, float beta2, float eps) {
    if (!h) EPNN_FAIL("epnn_train_init: null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->pending.active && finish_forward(h)) return 1;
    if (train_quiesce(h)) return 1;
    return train_init(h, lr, beta1, beta2, eps);
}
// A model with h_dim below 48 (epnn_host.h: model_dim): where the MODEL's parameters -- Keras order, the model's own shapes -- sit
// in the flat vector of the padded layers the handle trains (the padding's gradients are zero and are not part of the interface).
static void model_flat_index(epnn_handle *h, std::vector<int> &idx) {
    TrainState ts;
    train_layout(h, &ts);
    idx.clear();
    auto add = [&](const TDense &d, const HostDense &hd, int which, int layer) {
        ModelMaps M;
        model_maps(h, which, layer, hd, M);
        for (int r : M.rows)
            for (int c : M.cols) idx.push_back(d.offW + r * d.n_out + c);
        for (int c : M.cols) idx.push_back(d.offB + c);
    };
    if (upd_is_generic(h, &ts))
        for (size_t l = 0; l < h->updg.size(); ++l) add(ts.updv[l], h->updg[l], EPNN_W_UPD, (int)l);
    else
        for (int l = 0; l < 3; ++l) add(ts.upd[l], h->upd[l], EPNN_W_UPD, l);
    for (int t = 0; t < h->cfg.T; ++t)
        for (int l = 0; l < 3; ++l) add(ts.msg[t][l], h->msg[t][l], EPNN_W_MSG, l);
    for (int t = 0; t < h->cfg.T; ++t)
        for (int l = 0; l < 3; ++l) add(ts.pas[t][l], h->pas[t][l], EPNN_W_PAS, l);
}
extern "C" int epnn_param_count(epnn_handle *h, int64_t *out) {
    if (!h || !out) EPNN_FAIL("epnn_param_count: null argument");
    if (h->model_dim != EPNN_EDIM) {
        std::vector<int> idx;
        model_flat_index(h, idx);
        *out = (int64_t)idx.size();
        return 0;
    }
    TrainState ts;
    train_layout(h, &ts);
    *out = ts.P;
    return 0;
}
extern "C" int epnn_get_gradients(epnn_handle *h, float *out, int64_t count) {
    if (!h || !out) EPNN_FAIL("epnn_get_gradients: null argument");
    TrainState *ts = train_state(h);
    HIPCHK(hipSetDevice(h->device));
    if (h->model_dim != EPNN_EDIM) {
        std::vector<int> idx;
        model_flat_index(h, idx);
        if (!ts->ready || count != (int64_t)idx.size()) EPNN_FAIL("epnn_get_gradients: training not initialised or wrong count (%zu parameters)", idx.size());
        std::vector<float> flat(ts->P);
        HIPCHK(hipMemcpyAsync(flat.data(), ts->grad.p, (size_t)ts->P * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (size_t k = 0; k < idx.size(); ++k) out[k] = flat[idx[k]];
        return 0;
    }
    if (!ts->ready || count != ts->P) EPNN_FAIL("epnn_get_gradients: training not initialised or wrong count (%d parameters)", ts->P);
    HIPCHK(hipMemcpyAsync(out, ts->grad.p, (size_t)ts->P * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int epnn_set_gradients(epnn_handle *h, const float *in, int64_t count) {
    if (!h || !in) EPNN_FAIL("epnn_set_gradients: null argument");
    TrainState *ts = train_state(h);
    HIPCHK(hipSetDevice(h->device));
    if (h->model_dim != EPNN_EDIM) {
        std::vector<int> idx;
        model_flat_index(h, idx);
        if (!ts->ready || count != (int64_t)idx.size()) EPNN_FAIL("epnn_set_gradients: training not initialised or wrong count (%zu parameters)", idx.size());
        std::vector<float> flat(ts->P, 0.f);
        for (size_t k = 0; k < idx.size(); ++k) flat[idx[k]] = in[k];
        HIPCHK(hipMemcpyAsync(ts->grad.p, flat.data(), (size_t)ts->P * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return 0;
    }
    if (!ts->ready || count != ts->P) EPNN_FAIL("epnn_set_gradients: training not initialised or wrong count (%d parameters)", ts->P);
    HIPCHK(hipMemcpyAsync(ts->grad.p, in, (size_t)ts->P * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
// all-reduce (when a communicator is attached) + Adam on the current gradient buffer
extern "C" int epnn_train_apply(epnn_handle *h) {
    if (!h) EPNN_FAIL("epnn_train_apply: null handle");
    HIPCHK(hipSetDevice(h->device));
    if (!train_state(h)->ready) EPNN_FAIL("epnn_train_apply: call epnn_train_init first");
    if (train_apply(h)) return 1;
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
// (entry points whose step ends in the gradient all-reduce: a failure in front of it is reported to the peers, comm_guard)
static bool train_step_guarded(const epnn_handle *h, int apply) { return h && apply && comm_collectives(h); }

// shared tail of the two train-step entry points: slot arrays are on the device
// forward + backward of one batch: row-fused kernels when the padded size fits their LDS budget ("train_fused", default 1),
// else (or with the option at 0) the layer-by-layer kernels
static bool train_is_fused(const epnn_handle *h, int N) { return h->opt_train_fused && N <= EPNN_TF_NMAX && !h->upd_generic; }   // (other update layers than [32, 32]: one launch per Dense layer)
// d_loss: [B][N] loss terms (the layer-by-layer path fills one per molecule and leaves the rest zero); adam_now: the fused
// path's last launch also takes the optimizer step
static int train_fb(epnn_handle *h, int B, int N, const float *d_e, const float *d_mask, const float *d_x, const float *d_h0,
                    const float *d_q0, const float *d_y, float *d_pred, float *d_loss, bool size_only = false, bool adam_now = false,
                    float *out_host = nullptr, bool step_on_device = false) {
    if (train_is_fused(h, N))
        return train_fwd_bwd_fused(h, B, N, d_e, d_mask, d_x, d_h0, d_q0, d_y, d_pred, d_loss, size_only, adam_now, out_host, step_on_device);
    if (!size_only) {
        TrainState *ts = train_state(h);
        HIPCHK(hipMemsetAsync(ts->grad.p, 0, (size_t)ts->P * 4, h->stream));       // its launches ADD their parts of the gradient
        HIPCHK(hipMemsetAsync(d_loss, 0, (size_t)B * N * 4, h->stream));
    }
    return train_fwd_bwd(h, B, N, d_e, d_mask, d_x, d_h0, d_q0, d_y, d_pred, d_loss, size_only);
}

static int train_step_slots(epnn_handle *h, int B, int N, const float *d_e, const float *d_mask, const float *d_x,
                            const float *d_h0, const float *d_q0, const float *d_y, float *pred_host, float *loss_host, int apply) {
    TrainState *ts = train_state(h);
    if (!ts->ready) EPNN_FAIL("train step: call epnn_train_init first");
    const size_t BN = (size_t)B * N;
    if ((B != ts->last_B || N != ts->last_N) && train_quiesce(h)) return 1;      // buffers may grow: nothing of the previous step may still be running
    ts->last_B = B;
    ts->last_N = N;
    if (!ts->ev_fwd) HIPCHK(hipEventCreateWithFlags(&ts->ev_fwd, hipEventDisableTiming));
    if (ts->loss.ensure(2 * BN * 4)) return 1;
    float *d_loss = ts->loss.as<float>(), *d_pred = d_loss + BN;
    // the optimizer step rides in the gradient reduction's launch when nothing has to happen between the two (no all-reduce
    // over ranks); the last forward launch of the row-fused path writes loss terms | predictions into page-locked host memory as
    // well: no download (a 4 us copy kernel and its launch) between the last launch and the caller
    const bool rowfused = train_is_fused(h, N);
    const bool adam_now = apply && rowfused && !comm_collectives(h);
    if (h->pin_tout.ensure(2 * BN * 4)) return 1;
    float *out_host = rowfused ? h->pin_tout.as<float>() : nullptr;
    // "train_async": the step returns behind its forward pass.  That needs an event the host can wait for between the forward and the
    // backward launches -- an event recorded inside a replayed graph is not one (measured: the wait returns at once) -- so such a
    // step is launched kernel by kernel (the replay was worth 1 %, returning early is worth 8 %).
    // (several ranks: the all-reduce and the optimizer step are enqueued behind the backward pass on the same stream -- every rank
    // enqueues the same sequence --, so such a step returns behind its forward pass too)
    const bool early_ok = h->opt_train_async && rowfused;
    if (h->opt_train_graph && !early_ok) {
        // The step is a chain of dependent launches a few microseconds long: recorded once per (B, N, buffer set, apply) and
        // replayed as one hipGraph.  The step number Adam's step size depends on then lives on the device: the graph's first
        // launch counts it, its last one reads it (the host keeps its own count in step and repairs the device's when they differ).
        if (train_fb(h, B, N, d_e, d_mask, d_x, d_h0, d_q0, d_y, d_pred, d_loss, true, adam_now, out_host, true)) return 1;
        const std::vector<const void *> key = {(const void *)(size_t)B, (const void *)(size_t)N, (const void *)(size_t)(h->opt_train_fused + 16 * h->opt_train_split + 256 * (int)adam_now), d_e, d_mask, d_x, d_h0, d_q0, d_y, d_pred,
                                               d_loss, ts->arena.p, ts->part.p, ts->theta.p, ts->grad.p, out_host, ts->d_step.p, h->tr_moff, h->tr_real};
        if (!ts->gexec || key != ts->gkey) {
            // park the current capture, look for one made with this key
            if (ts->gexec) {
                if (ts->kept.size() >= 4) {
                    if (train_quiesce(h)) return 1;                       // (the capture that goes may be the one still running)
                    (void)hipGraphExecDestroy(ts->kept.front().exec);
                    (void)hipGraphDestroy(ts->kept.front().graph);
                    ts->kept.erase(ts->kept.begin());
                }
                ts->kept.push_back({ts->gkey, ts->graph, ts->gexec});
                ts->gexec = nullptr;
                ts->graph = nullptr;
            }
            for (size_t k = 0; k < ts->kept.size(); ++k)
                if (ts->kept[k].key == key) {
                    ts->graph = ts->kept[k].graph;
                    ts->gexec = ts->kept[k].exec;
                    ts->gkey = key;
                    ts->kept.erase(ts->kept.begin() + k);
                    break;
                }
        }
        if (!ts->gexec || key != ts->gkey) {
            const long step_before = ts->step;
            HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            const int bad = train_fb(h, B, N, d_e, d_mask, d_x, d_h0, d_q0, d_y, d_pred, d_loss, false, adam_now, out_host, true);
            const hipError_t ec = hipStreamEndCapture(h->stream, &ts->graph);
            ts->step = step_before;
            if (bad || ec != hipSuccess) {
                if (ts->graph) { (void)hipGraphDestroy(ts->graph); ts->graph = nullptr; }
                if (!bad) EPNN_FAIL("train step: hipStreamEndCapture failed: %s", hipGetErrorString(ec));
                return 1;
            }
            HIPCHK(hipGraphInstantiate(&ts->gexec, ts->graph, nullptr, nullptr, 0));
            ts->gkey = key;
        }
        if (adam_now && ts->dev_step != ts->step) {
            const long long sv = ts->step;
            HIPCHK(hipMemcpyAsync(ts->d_step.p, &sv, 8, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            ts->dev_step = ts->step;
        }
        HIPCHK(hipGraphLaunch(ts->gexec, h->stream));
        if (adam_now) {
            ts->step += 1;
            ts->dev_step = ts->step;
            ts->dev_newer = true;
        }
    } else {
        if (train_fb(h, B, N, d_e, d_mask, d_x, d_h0, d_q0, d_y, d_pred, d_loss, false, adam_now, out_host)) return 1;
    }
    if (apply && !adam_now && train_apply(h)) return 1;
    // the step's loss terms and predictions are neighbours on the device: one download into page-locked memory
    const size_t nback = BN + (pred_host ? BN : 0);
    if (!ts->host_out) HIPCHK(hipMemcpyAsync(h->pin_tout.p, d_loss, nback * 4, hipMemcpyDeviceToHost, h->stream));
    // What the caller gets back -- loss terms, predictions -- is on the host when the FORWARD is done.  The backward pass and the
    // optimizer step run on behind the return: the next step's launches queue up behind them on the stream, everything that looks at
    // gradients or weights synchronises first (epnn_get_gradients, the host copies of the weights, any inference call).  A loop of
    // steps then costs its GPU time, not GPU time + the host's wake-up and launch latencies.
    const bool early = early_ok && ts->host_out && ts->ev_fwd;
    if (early) {
        HIPCHK(hipEventSynchronize(ts->ev_fwd));
        ts->inflight = true;
    } else {
        HIPCHK(hipStreamSynchronize(h->stream));
        ts->inflight = false;
    }
    const float *back = h->pin_tout.as<float>();
    if (pred_host) memcpy(pred_host, back + BN, BN * 4);
    if (loss_host) {
        double s = 0;
        for (size_t k = 0; k < BN; ++k) s += back[k];
        *loss_host = (float)s;
    }
    return 0;
}

// train_step (charge_gn.py:393-402) on the literal make_model inputs; y and pred are (B,N,1).  apply = 0 leaves the
// gradient in place (epnn_get_gradients) without touching the weights.
static int train_step_dense_impl(epnn_handle *h, int B, int N, const float *h_inp, const float *e_inp, const float *x_inp,
                                 const float *q_inp, const float *mask_inp, const float *y, float *pred_out,
                                 float *loss_out, int apply);
extern "C" int epnn_train_step_dense(epnn_handle *h, int B, int N, const float *h_inp, const float *e_inp, const float *x_inp,
                                     const float *q_inp, const float *mask_inp, const float *y, float *pred_out,
                                     float *loss_out, int apply) {
    std::vector<float> ph, pe;
    if (h && h->model_dim != EPNN_EDIM && h_inp && e_inp && B > 0 && N > 0) {      // h_dim = e_dim below 48: zero channels added (epnn_host.h)
        pad_channels(h_inp, (size_t)B * N * N, h->model_dim, ph);
        pad_channels(e_inp, (size_t)B * N * N, h->model_dim, pe);
        h_inp = ph.data();
        e_inp = pe.data();
    }
    if (!train_step_guarded(h, apply)) return train_step_dense_impl(h, B, N, h_inp, e_inp, x_inp, q_inp, mask_inp, y, pred_out, loss_out, apply);
    h->guard_pending = true;
    return comm_guard_exit(h, train_step_dense_impl(h, B, N, h_inp, e_inp, x_inp, q_inp, mask_inp, y, pred_out, loss_out, apply), "train step (gradient all-reduce)");
}
static int train_step_dense_impl(epnn_handle *h, int B, int N, const float *h_inp, const float *e_inp, const float *x_inp,
                                 const float *q_inp, const float *mask_inp, const float *y, float *pred_out,
                                 float *loss_out, int apply) {
    if (!h || !h_inp || !e_inp || !x_inp || !q_inp || !mask_inp || !y) EPNN_FAIL("epnn_train_step_dense: null argument");
    if (B < 1 || N < 1) EPNN_FAIL("epnn_train_step_dense: B and N must be positive");
    HIPCHK(hipSetDevice(h->device));
    if (h->pending.active && finish_forward(h)) return 1;
    if (train_quiesce(h)) return 1;               // (this entry re-uploads the tensors the previous step's kernels read)
    const int nx = h->cfg.nx;
    const size_t pairs = (size_t)B * N * N, slots = (size_t)B * N;
    if (h->dn_xs.ensure(slots * nx * 4) || h->dn_hs.ensure(slots * EPNN_EDIM * 4) || h->dn_qs.ensure(slots * 4) ||
        h->dn_nms.ensure(slots * 4) || h->dn_flag.ensure(slots * 4))
        return 1;
    DenseArgs D{};
    D.B = B; D.N = N; D.nx = nx; D.model_level = 1;
    const float *d_y;
    auto up256 = [](size_t bytes) { return (bytes + 255) & ~size_t(255); };
    const size_t b_he = pairs * EPNN_EDIM * 4, o_e = up256(b_he), o_x = o_e + up256(b_he), o_q = o_x + up256(pairs * nx * 4),
                 o_m = o_q + up256(pairs * 4), o_y = o_m + up256(pairs * 4), in_bytes = o_y + slots * 4;
    if (in_bytes <= ((size_t)4 << 20)) {
        // one molecule per step (the reference's loop): one page-locked staging buffer, one upload (as in dense_host)
        if (h->pin_train.ensure(in_bytes) || h->s_train.ensure(in_bytes)) return 1;
        char *stage = h->pin_train.as<char>();
        memcpy(stage, h_inp, b_he);
        memcpy(stage + o_e, e_inp, b_he);
        memcpy(stage + o_x, x_inp, pairs * nx * 4);
        memcpy(stage + o_q, q_inp, pairs * 4);
        memcpy(stage + o_m, mask_inp, pairs * 4);
        memcpy(stage + o_y, y, slots * 4);
        HIPCHK(hipMemcpyAsync(h->s_train.p, stage, in_bytes, hipMemcpyHostToDevice, h->stream));
        const char *dev = h->s_train.as<char>();
        D.h_in = reinterpret_cast<const float *>(dev); D.e_in = reinterpret_cast<const float *>(dev + o_e);
        D.x_in = reinterpret_cast<const float *>(dev + o_x); D.q_in = reinterpret_cast<const float *>(dev + o_q);
        D.mask_in = reinterpret_cast<const float *>(dev + o_m);
        d_y = reinterpret_cast<const float *>(dev + o_y);
    } else {
        if (h->sd_h.ensure(b_he) || h->sd_e.ensure(b_he) || h->sd_x.ensure(pairs * nx * 4) || h->sd_q.ensure(pairs * 4) ||
            h->sd_mask.ensure(pairs * 4) || h->sd_out.ensure(slots * 4))
            return 1;
        HIPCHK(hipMemcpyAsync(h->sd_h.p, h_inp, b_he, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->sd_e.p, e_inp, b_he, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->sd_x.p, x_inp, pairs * nx * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->sd_q.p, q_inp, pairs * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->sd_mask.p, mask_inp, pairs * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->sd_out.p, y, slots * 4, hipMemcpyHostToDevice, h->stream));
        D.h_in = h->sd_h.as<float>(); D.e_in = h->sd_e.as<float>(); D.x_in = h->sd_x.as<float>();
        D.q_in = h->sd_q.as<float>(); D.mask_in = h->sd_mask.as<float>();
        d_y = h->sd_out.as<float>();
    }
    D.xs = h->dn_xs.as<float>(); D.hs = h->dn_hs.as<float>(); D.qs = h->dn_qs.as<float>(); D.nms = h->dn_nms.as<float>();
    D.flag = h->dn_flag.as<int>(); D.tol = h->cfg.near_tol;
    if (slots * N <= 65536) {
        // a step on one or a few molecules: the per-atom reductions alone (charge_gn.py:382-384), one launch -- training needs
        // neither the flags nor the e scan of the inference front-end
        if (h->dn_den.ensure(slots * 4)) return 1;
        D.model_level = 1;
        const int fb = (N * (EPNN_EDIM + nx + 1) + 255) / 256;
        hipLaunchKernelGGL(k_dn_front_small, dim3((unsigned)fb, (unsigned)B), dim3(256), 0, h->stream, D, h->dn_den.as<float>(), 1, fb);
        HIPCHK(hipGetLastError());
    } else if (launch_dense_atoms(h, D)) {
        return 1;
    }
    return train_step_slots(h, B, N, D.e_in, D.mask_in, D.xs, D.hs, D.qs, d_y, pred_out, loss_out, apply);
}

// train_step from a flat coordinate batch: y_flat / q_out_flat are per real atom [A]
static int train_step_xyz_impl(epnn_handle *h, int B, int N, const int32_t *offsets, const float *xyz, const float *x,
                               const float *Q, const float *y_flat, float *q_out_flat, float *loss_out, int apply);
extern "C" int epnn_train_step_xyz(epnn_handle *h, int B, int N, const int32_t *offsets, const float *xyz, const float *x,
                                   const float *Q, const float *y_flat, float *q_out_flat, float *loss_out, int apply) {
    if (!train_step_guarded(h, apply)) return train_step_xyz_impl(h, B, N, offsets, xyz, x, Q, y_flat, q_out_flat, loss_out, apply);
    h->guard_pending = true;
    return comm_guard_exit(h, train_step_xyz_impl(h, B, N, offsets, xyz, x, Q, y_flat, q_out_flat, loss_out, apply), "train step (gradient all-reduce)");
}
static int train_step_xyz_impl(epnn_handle *h, int B, int N, const int32_t *offsets, const float *xyz, const float *x,
                               const float *Q, const float *y_flat, float *q_out_flat, float *loss_out, int apply) {
    if (!h || !offsets || !xyz || !x || !Q || !y_flat) EPNN_FAIL("epnn_train_step_xyz: null argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->pending.active && finish_forward(h)) return 1;
    if (B < 1 || N < 1 || offsets[0] != 0) EPNN_FAIL("epnn_train_step_xyz: B and N must be positive and offsets[0] must be 0");
    const int nx = h->cfg.nx, A = offsets[B];
    for (int b = 0; b < B; ++b)
        if (offsets[b + 1] - offsets[b] > N || offsets[b + 1] - offsets[b] < 1) EPNN_FAIL("epnn_train_step_xyz: molecule %d does not fit N=%d", b, N);
    const size_t pairs = (size_t)B * N * N, slots = (size_t)B * N;
    // ONE upload per step: offsets | xyz | x | Q | y staged in page-locked memory, same layout on the device (five separate
    // copies from pageable memory were ~50 us of a 0.5 ms step before its first kernel could start)
    auto up256 = [](size_t bytes) { return (bytes + 255) & ~size_t(255); };
    const size_t o_xyz = up256((size_t)(B + 1) * 4), o_x = o_xyz + up256((size_t)A * 3 * 4), o_Q = o_x + up256((size_t)A * nx * 4),
                 o_y = o_Q + up256((size_t)B * 4), in_bytes = o_y + (size_t)A * 4;
    if (h->train) {
        // the previous step's backward pass may still be running ("train_async"): it reads the buffers below
        TrainState *ts0 = train_state(h);
        if ((B != ts0->last_B || N != ts0->last_N || in_bytes > h->s_train.cap || in_bytes > h->pin_train.cap) && train_quiesce(h)) return 1;
    }
    if (h->pin_train.ensure(in_bytes) || h->s_train.ensure(in_bytes) || h->sd_e.ensure(pairs * EPNN_EDIM * 4) ||
        h->sd_mask.ensure(pairs * 4) || h->dn_xs.ensure(slots * nx * 4) || h->dn_hs.ensure(slots * EPNN_EDIM * 4) ||
        h->dn_qs.ensure(slots * 4) || h->sd_out.ensure(slots * 4) || h->tr_realbuf.ensure(slots * 4))
        return 1;
    char *stage = h->pin_train.as<char>();
    const char *dev = h->s_train.as<char>();
    const int w_xyz = (int)(o_xyz / 4), w_x = (int)(o_x / 4), w_Q = (int)(o_Q / 4), w_y = (int)(o_y / 4);
    const unsigned pgrid = t_grid(pairs * ((h->cfg.e_dim + 3) / 4));
    // one molecule (or two small ones): the inputs are few enough to ride in the padding kernel's argument block, packed (no 256-byte
    // sections) -- no upload; otherwise ONE upload of the staged block
    const size_t packed_words = (size_t)(B + 1) + (size_t)A * (3 + nx + 1) + B;
    if (h->opt_train_inline && packed_words <= EPNN_PAD_INLINE_WORDS) {
        PadInline P;
        int k = 0;
        memcpy(P.w + k, offsets, (size_t)(B + 1) * 4); k += B + 1;
        const int p_xyz = k; memcpy(P.w + k, xyz, (size_t)A * 3 * 4); k += A * 3;
        const int p_x = k; memcpy(P.w + k, x, (size_t)A * nx * 4); k += A * nx;
        const int p_Q = k; memcpy(P.w + k, Q, (size_t)B * 4); k += B;
        const int p_y = k; memcpy(P.w + k, y_flat, (size_t)A * 4); k += A;
        hipLaunchKernelGGL(k_t_pad_inputs_inline, dim3(pgrid), dim3(256), 0, h->stream, P, p_xyz, p_x, p_Q, p_y, B, N, nx, h->cfg.e_dim,
                           (double)h->cfg.cutoff, (double)h->cfg.eta, h->d_mu.as<double>(), h->sd_e.as<float>(), h->sd_mask.as<float>(),
                           h->dn_xs.as<float>(), h->dn_hs.as<float>(), h->dn_qs.as<float>(), h->sd_out.as<float>(), h->tr_realbuf.as<int>(),
                           reinterpret_cast<int *>(h->s_train.p));
    } else {
        memcpy(stage, offsets, (size_t)(B + 1) * 4);             // (the previous step ended with a stream synchronisation)
        memcpy(stage + o_xyz, xyz, (size_t)A * 3 * 4);
        memcpy(stage + o_x, x, (size_t)A * nx * 4);
        memcpy(stage + o_Q, Q, (size_t)B * 4);
        memcpy(stage + o_y, y_flat, (size_t)A * 4);
        HIPCHK(hipMemcpyAsync(h->s_train.p, stage, in_bytes, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_t_pad_inputs, dim3(pgrid), dim3(256), 0, h->stream, reinterpret_cast<const float *>(dev), w_xyz, w_x, w_Q, w_y, B, N, nx,
                           h->cfg.e_dim, (double)h->cfg.cutoff, (double)h->cfg.eta, h->d_mu.as<double>(), h->sd_e.as<float>(), h->sd_mask.as<float>(),
                           h->dn_xs.as<float>(), h->dn_hs.as<float>(), h->dn_qs.as<float>(), h->sd_out.as<float>(), h->tr_realbuf.as<int>());
    }
    HIPCHK(hipGetLastError());
    std::vector<float> pred(q_out_flat ? slots : 0);
    // the padded slots of a coordinate batch are exact zeros in every input: the matrix-pipe kernels skip their workgroups
    if (h->opt_train_skip_padded) { h->tr_moff = reinterpret_cast<const int *>(dev); h->tr_real = h->tr_realbuf.as<int>(); }
    const int rc_step = train_step_slots(h, B, N, h->sd_e.as<float>(), h->sd_mask.as<float>(), h->dn_xs.as<float>(), h->dn_hs.as<float>(),
                                         h->dn_qs.as<float>(), h->sd_out.as<float>(), q_out_flat ? pred.data() : nullptr, loss_out, apply);
    h->tr_moff = nullptr;
    h->tr_real = nullptr;
    if (rc_step) return 1;
    if (q_out_flat)
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < offsets[b + 1] - offsets[b]; ++i) q_out_flat[offsets[b] + i] = pred[(size_t)b * N + i];
    return 0;
}

// RCCL communicator for the gradient all-reduce (one rank per GPU).  The 128-byte id is created on rank 0 and
// handed to the other ranks by the caller (torch.distributed broadcast, a file, ...).
extern "C" int epnn_comm_unique_id(char *out128) {
    if (!out128) EPNN_FAIL("epnn_comm_unique_id: null argument");
    ncclUniqueId id;
    ncclResult_t rc = ncclGetUniqueId(&id);
    if (rc != ncclSuccess) EPNN_FAIL("ncclGetUniqueId failed: %s", ncclGetErrorString(rc));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out128, &id, 128);
    return 0;
}
extern "C" int epnn_comm_init(epnn_handle *h, const char *id128, int rank, int world) {
    if (!h || !id128 || world < 1 || rank < 0 || rank >= world) EPNN_FAIL("epnn_comm_init: bad argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->pending.active && finish_forward(h)) return 1;
    if (h->comm) { (void)ncclCommDestroy(h->comm); h->comm = nullptr; }
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclResult_t rc = ncclCommInitRank(&h->comm, world, id, rank);
    if (rc != ncclSuccess) EPNN_FAIL("ncclCommInitRank failed: %s", ncclGetErrorString(rc));
    h->comm_world = world;
    h->comm_rank = rank;
    return 0;
}

// Number of ranks that joined the handle's communicator (ncclCommCount): proof that N processes really met over RCCL.
extern "C" int epnn_comm_count(epnn_handle *h, int32_t *ranks_out) {
    if (!h || !ranks_out) EPNN_FAIL("epnn_comm_count: null argument");
    if (!h->comm) EPNN_FAIL("epnn_comm_count: no communicator (epnn_comm_init)");
    int n = 0;
    ncclResult_t rc = ncclCommCount(h->comm, &n);
    if (rc != ncclSuccess) EPNN_FAIL("ncclCommCount failed: %s", ncclGetErrorString(rc));
    *ranks_out = n;
    return 0;
}
// A small all-reduce of host doubles over the handle's communicator, on the handle's stream and waited for: the barrier
// and the MAX / SUM over ranks a multi-process driver (bench.py --gpus N) needs, through the product's own RCCL path.
// op: 0 sum, 1 max.
extern "C" int epnn_comm_allreduce(epnn_handle *h, double *inout, int32_t n, int32_t op) {
    if (!h || !inout || n < 1 || n > 1024 || (op != 0 && op != 1)) EPNN_FAIL("epnn_comm_allreduce: bad argument");
    if (!h->comm) EPNN_FAIL("epnn_comm_allreduce: no communicator (epnn_comm_init)");
    HIPCHK(hipSetDevice(h->device));
    if (h->pending.active && finish_forward(h)) return 1;
    if (h->s_misc.ensure(1024 * sizeof(double))) return 1;
    HIPCHK(hipMemcpyAsync(h->s_misc.p, inout, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    ncclResult_t rc = ncclAllReduce(h->s_misc.p, h->s_misc.p, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, h->comm, h->stream);
    if (rc != ncclSuccess) EPNN_FAIL("ncclAllReduce failed: %s", ncclGetErrorString(rc));
    HIPCHK(hipMemcpyAsync(inout, h->s_misc.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// The pair list the separate front-end (or the dense front-end) built for the last forward: indices and near weights of up
// to `cap` pairs (tests: the device's D < cutoff and is_near decisions against a host count).  Returns the number of pairs.
#ifdef EPNN_LG_CLOCKS
// development build only (tools/large_clocks.py): phase clocks of workgroup 0 of the tiled path's tail and EPN-step launches of the last forward
extern "C" int epnn_debug_large_clocks(epnn_handle *h, unsigned long long *dst, int n) {
    if (!h || !dst || !h->lg_clk.p) EPNN_FAIL("epnn_debug_large_clocks: bad argument");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(dst, h->lg_clk.p, (size_t)std::min(n, 128 + 4 * 1024) * 8, hipMemcpyDeviceToHost));
    return 0;
}
#endif
#ifdef EPNN_TF_CLOCKS
// development build only (tools/train_clocks.py): phase clocks of workgroup 0 of every row-fused training launch of the last step
extern "C" int epnn_debug_train_clocks(epnn_handle *h, unsigned long long *dst, int n) {
    if (!h || !dst || !h->train) EPNN_FAIL("epnn_debug_train_clocks: bad argument");
    TrainState *ts = train_state(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(dst, ts->clk.p, (size_t)std::min(n, 64 * 16) * 8, hipMemcpyDeviceToHost));
    return 0;
}
#endif

extern "C" int epnn_debug_pairs(epnn_handle *h, int32_t *pi, int32_t *pj, float *pwi, int64_t cap, int64_t *count_out) {
    if (!h || !pi || !pj || !pwi || !count_out || cap < 0) EPNN_FAIL("epnn_debug_pairs: bad argument");
    HIPCHK(hipSetDevice(h->device));
    if (finish_forward(h)) return 1;
    if (!h->plan.valid || !h->d_rowoff.p || !h->d_pi.p) EPNN_FAIL("epnn_debug_pairs: no pair list (the last forward used the in-kernel front-end)");
    int np = 0;
    HIPCHK(hipMemcpy(&np, h->d_rowoff.as<int>() + h->plan.A, sizeof(int), hipMemcpyDeviceToHost));
    if (np < 0 || np > h->pcap) EPNN_FAIL("epnn_debug_pairs: the list holds %d pairs, capacity %d", np, h->pcap);
    const size_t n = (size_t)std::min<int64_t>(np, cap);
    HIPCHK(hipMemcpy(pi, h->d_pi.p, n * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pj, h->d_pj.p, n * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pwi, h->d_pwi.p, n * sizeof(float), hipMemcpyDeviceToHost));
    *count_out = np;
    return 0;
}
