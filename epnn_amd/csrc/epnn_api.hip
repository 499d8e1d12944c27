// C ABI of libepnn_hip.so (see include/epnn.h).  gfx950 only.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <mutex>
#include <numeric>
#include <sched.h>

#include "epnn_host.h"
#include "epnn_frontend.hip.h"
#include "epnn_wave.hip.h"
#include "epnn_wave2.hip.h"
#include "epnn_mlp.hip.h"
#include "epnn_large.hip.h"
#include "epnn_dense.hip.h"
#include "epnn_train.hip.h"

thread_local std::string g_epnn_err;

extern "C" const char *epnn_last_error(void) { return g_epnn_err.c_str(); }
extern "C" int epnn_version(void) { return 1; }
extern "C" int epnn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------------ create
static void shape_layers(epnn_handle *h) {
    const int F = h->cfg.nx + h->cfg.h_dim + 1, H = h->cfg.hidden, E = h->cfg.e_dim;
    auto set = [](HostDense &d, int i, int o) {
        d.n_in = i;
        d.n_out = o;
        d.W.assign((size_t)i * o, 0.f);
        d.b.assign(o, 0.f);
    };
    for (int t = 0; t < h->cfg.T; ++t) {
        set(h->msg[t][0], 2 * F + E, H);
        set(h->msg[t][1], H, H);
        set(h->msg[t][2], H, H);                 // message width is hard-coded 32 (charge_gn.py:52)
        set(h->pas[t][0], 2 * F + E, H);
        set(h->pas[t][1], H, H);
        set(h->pas[t][2], H, 1);
    }
    set(h->upd[0], h->cfg.h_dim + H, H);
    set(h->upd[1], H, H);
    set(h->upd[2], H, h->cfg.h_dim);
}

static double edge_basis(const epnn_config &cfg, const std::vector<double> &mu, std::vector<double> &Bout, std::vector<float> &tab);
static int create_resources(epnn_handle *h);
extern "C" int epnn_destroy(epnn_handle *h);

// The HIP runtime maps a process's streams round-robin onto its hardware queues (GPU_MAX_HW_QUEUES of them) in the order the
// streams are created, and which queues a pipeline's lanes sit on matters: eight lanes on every other queue run the bench batch at
// 220 M atoms/s, on eight consecutive queues at 213 M (round 4, measured both ways several times).  A caller that builds several
// handles can therefore leave queues out between them: n placeholder streams are created here and kept in a process-level list;
// n == 0 destroys every placeholder made so far on `device` (engine.Pipeline.close does: a long-lived process that builds many
// pipelines does not accumulate them).  The placement is only defined for the FIRST pipeline a process builds -- later streams
// land wherever the runtime's round robin has got to.
static std::mutex g_skip_mutex;
static std::vector<std::pair<int, hipStream_t>> g_skip_streams;
extern "C" int epnn_skip_hw_queues(int device, int n) {
    if (n < 0 || n > 64) EPNN_FAIL("epnn_skip_hw_queues: n must be in 0..64");
    HIPCHK(hipSetDevice(device));
    std::lock_guard<std::mutex> lock(g_skip_mutex);
    if (n == 0) {
        for (size_t k = 0; k < g_skip_streams.size();) {
            if (g_skip_streams[k].first != device) { ++k; continue; }
            (void)hipStreamDestroy(g_skip_streams[k].second);
            g_skip_streams.erase(g_skip_streams.begin() + (long)k);
        }
        return 0;
    }
    for (int k = 0; k < n; ++k) {
        hipStream_t s;
        HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));      // (never used: it only holds its place)
        g_skip_streams.emplace_back(device, s);
    }
    return 0;
}

extern "C" int epnn_create(const epnn_config *cfg, int device, epnn_handle **out) {
    if (!cfg || !out) EPNN_FAIL("epnn_create: null argument");
    if (cfg->h_dim != cfg->e_dim) EPNN_FAIL("epnn_create: e_dim (%d) must equal h_dim (%d): make_model gives e_inp h_dim channels (charge_gn.py:377)", cfg->e_dim, cfg->h_dim);
    if (cfg->h_dim < 1 || cfg->h_dim > EPNN_EDIM)
        EPNN_FAIL("epnn_create: h_dim = e_dim must be in 1..%d (the kernels hold %d channels; a smaller model runs zero-padded, a larger one is not built)", EPNN_EDIM, EPNN_EDIM);
    if (cfg->hidden != EPNN_HID) EPNN_FAIL("epnn_create: hidden must be %d", EPNN_HID);
    if (cfg->T < 1 || cfg->T > EPNN_MAXT) EPNN_FAIL("epnn_create: T must be in 1..%d", EPNN_MAXT);
    if (cfg->nx < 1 || cfg->nx + EPNN_EDIM + 1 > EPNN_F1) EPNN_FAIL("epnn_create: nx must be in 1..%d", EPNN_F1 - EPNN_EDIM - 1);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        EPNN_FAIL("epnn_create: no HIP device visible; the EPNN hot path has no CPU fallback");
    if (device < 0 || device >= ndev) EPNN_FAIL("epnn_create: device %d out of range (0..%d)", device, ndev - 1);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        EPNN_FAIL("epnn_create: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    epnn_handle *h = new epnn_handle();
    h->cfg = *cfg;
    h->model_dim = cfg->h_dim;
    h->cfg.h_dim = h->cfg.e_dim = EPNN_EDIM;     // what the kernels see (epnn_host.h: model_dim)
    h->device = device;
    if (create_resources(h)) {                   // the message is set; nothing of a half-built handle stays behind
        const std::string why = g_epnn_err;
        epnn_destroy(h);
        g_epnn_err = why;
        return 1;
    }
    *out = h;
    return 0;
}
static int create_resources(epnn_handle *h) {
    const epnn_config *cfg = &h->cfg;
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&h->ev_t0));
    HIPCHK(hipEventCreateWithFlags(&h->ev_ctl, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&h->ev_t1));
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&h->h_status_base), 12 * sizeof(int), hipHostMallocDefault));      // two status slots of 4, the guard word (comm_guard)
    memset(h->h_status_base, 0, 12 * sizeof(int));
    h->h_status = h->h_status_base;
    for (auto &e : h->ev_done) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (h->d_status.ensure(4 * sizeof(int))) return 1;
    HIPCHK(hipMemsetAsync(h->d_status.p, 0, 4 * sizeof(int), h->stream));
    // mu = np.linspace(0.1, cutoff, e_dim): arange(num)*step + start, last element forced to stop (num == 1: [start]).  The
    // channels beyond the model's e_dim get a centre so far away that exp(-eta (D - mu)^2) is 0 for every distance: zero channels
    const int ME = h->model_dim;
    std::vector<double> mu(cfg->e_dim, 1.0e4);
    const double start = 0.1, stop = (double)cfg->cutoff;
    const double step = ME > 1 ? (stop - start) / (double)(ME - 1) : 0.0;
    for (int k = 0; k < ME; ++k) mu[k] = (double)k * step + start;
    if (ME > 1) mu[ME - 1] = stop;
    if (h->d_mu.ensure(mu.size() * sizeof(double))) return 1;
    HIPCHK(hipMemcpy(h->d_mu.p, mu.data(), mu.size() * sizeof(double), hipMemcpyHostToDevice));
    shape_layers(h);
    {
        std::vector<float> tab;
        h->edge_res = cfg->e_dim == EPNN_EDIM ? edge_basis(h->cfg, mu, h->edge_B, tab) : 1.0;
        if (!tab.empty()) {
            if (h->d_etab.ensure(tab.size() * sizeof(float))) return 1;
            HIPCHK(hipMemcpy(h->d_etab.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    {   // dsafe: largest D up to which a lower bound of max_k e_k stays above 2 tol (every pair closer than that is a near
        // pair, charge_gn.py:90-94).  Bound: C(D), decreasing in D, times the Gaussian at the largest possible distance
        // `gap` from D to its nearest mu_k (half the widest spacing; mu_0 itself for D below it).
        double gap = std::max(mu[0], stop - mu[ME - 1]);
        for (int k = 0; k + 1 < ME; ++k) gap = std::max(gap, 0.5 * (mu[k + 1] - mu[k]));
        const double floor_g = exp(-(double)cfg->eta * gap * gap);
        double lo = 0.0, hi = stop;
        for (int it = 0; it < 60; ++it) {
            const double m = 0.5 * (lo + hi), L = (cos(3.141592653589793 * m / stop) + 1.0) / 2.0 * floor_g;
            if (L > 2.0 * (double)cfg->near_tol) lo = m; else hi = m;
        }
        h->dsafe = lo;
    }
    {   // Beyond dsafe the near flag (float)(C(D) exp(-eta min_k (D - mu_k)^2)) > tol (charge_gn.py:90-94 on get_init_edges' rows) is a
        // function of the ONE variable D: the fused kernels' own front-end does not evaluate cos / exp per pair, it counts how many
        // of the function's flips lie below D.  The flips are found here, in float64 with the reference's expression: a scan of
        // (dsafe, cutoff) over the nodes listed below, every change bisected down to two adjacent doubles (flip = the last D of the old value).
        // With the reference's parameters there is exactly one, 6.0e-3 below the cutoff.
        const double eta = (double)cfg->eta;
        const float tol = cfg->near_tol;
        const int E = ME;
        auto nearf = [&](double D) -> bool {
            if (D >= stop) return false;
            const double C = (cos(3.141592653589793 * (D - 0.0) / stop) + 1.0) / 2.0;
            double best = 1e300;
            for (int k = 0; k < E; ++k) best = std::min(best, (D - mu[k]) * (D - mu[k]));
            return (float)(C * exp(-eta * best)) > tol;
        };
        // Scan nodes: 20 000 equal steps AND every local extremum of the flag's argument -- the Gaussian centres mu_k (maxima of
        // exp(-eta min_k (D - mu_k)^2)) and the midpoints between neighbouring centres (its minima), with one double on either side of
        // each: between two consecutive nodes the argument is then monotone up to the slowly falling C(D), so a near window
        // narrower than a grid step (tol just under C(mu_k) at a centre, needle-like Gaussians) is still seen from both ends.
        std::vector<double> nodes;
        const int M = 20000;
        for (int i = 1; i < M; ++i) nodes.push_back(h->dsafe + (stop - h->dsafe) * (double)i / (double)M);
        for (int k = 0; k < E; ++k) {
            const double cand[2] = {mu[k], k + 1 < E ? 0.5 * (mu[k] + mu[k + 1]) : stop};
            for (double c : cand)
                for (double v : {nextafter(c, 0.0), c, nextafter(c, 1e300)})
                    if (v > h->dsafe && v < stop) nodes.push_back(v);
        }
        std::sort(nodes.begin(), nodes.end());
        nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
        nodes.push_back(stop);
        std::vector<double> flips;
        bool prev = true;                                   // (every D <= dsafe is near by the bound above)
        double xprev = h->dsafe;
        if (!nearf(xprev)) { flips.push_back(xprev); prev = false; }
        for (size_t i = 0; i < nodes.size() && flips.size() <= EPNN_NFLIP_MAX; ++i) {
            const double xi = nodes[i];
            const bool cur = nearf(xi);
            if (cur != prev) {
                double a = xprev, b = xi;
                for (;;) {
                    const double mid = a + (b - a) * 0.5;
                    if (!(mid > a && mid < b)) break;
                    if (nearf(mid) == prev) a = mid; else b = mid;
                }
                flips.push_back(a);
                prev = cur;
            }
            xprev = xi;
        }
        if (flips.size() > EPNN_NFLIP_MAX) h->edge_res = 1.0;      // (needle-like Gaussians: the in-kernel front-end is not used at all)
        else {
            h->nflip = (int)flips.size();
            flips.resize(EPNN_NFLIP_MAX, 1e300);
            if (h->d_flip.ensure(flips.size() * sizeof(double))) return 1;
            HIPCHK(hipMemcpy(h->d_flip.p, flips.data(), flips.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    return 0;
}

extern "C" int epnn_destroy(epnn_handle *h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    DevBuf *bufs[] = {&h->d_wpack, &h->d_updgen, &h->d_mu, &h->d_flip, &h->d_mu_ex, &h->d_ctl, &h->d_rowcnt, &h->d_rowoff,
                      &h->d_status, &h->d_guard, &h->d_bsum, &h->d_pi, &h->d_pj, &h->d_psym, &h->d_pe, &h->d_pwi, &h->d_pwj, &h->s_xyz,
                      &h->s_train, &h->s_misc, &h->s_gx, &h->s_pt, &h->f_pw, &h->d_etab, &h->l_a, &h->l_P, &h->l_R, &h->l_zp, &h->l_S0,
                      &h->l_corr, &h->l_dl, &h->l_tiles, &h->l_csr_off, &h->l_csr_ent, &h->l_cnt, &h->l_nm,
                      &h->d_deg, &h->d_incoff, &h->d_nbr, &h->d_nearbits, &h->d_desti, &h->d_destj, &h->d_prec, &h->l_Nn, &h->l_Yb, &h->l_qbuf,
                      &h->l_Pst, &h->l_Rst, &h->l_lmol, &h->l_typrow, &h->l_typtab, &h->l_stype, &h->l_typhash,
                      &h->l_stasks, &h->l_schunk, &h->l_sfin, &h->l_sfrac, &h->l_stasks2, &h->l_sfrac2, &h->dn_xs, &h->dn_hs, &h->dn_qs, &h->dn_nms,
                      &h->dn_flag, &h->dn_neff, &h->dn_den, &h->tr_realbuf, &h->dn_xf, &h->dn_hf, &h->dn_qf, &h->dn_nmf, &h->dn_out, &h->sd_h,
                      &h->sd_e, &h->sd_x, &h->sd_q, &h->sd_mask, &h->sd_out};
    for (DevBuf *b : bufs) b->release();
    if (h->train) {
        TrainState *ts = train_state(h);
        if (ts->gexec) (void)hipGraphExecDestroy(ts->gexec);
        if (ts->graph) (void)hipGraphDestroy(ts->graph);
        for (auto &kg : ts->kept) { (void)hipGraphExecDestroy(kg.exec); (void)hipGraphDestroy(kg.graph); }
        if (ts->ev_fwd) (void)hipEventDestroy(ts->ev_fwd);
        for (DevBuf *b : {&ts->theta, &ts->grad, &ts->m, &ts->v, &ts->part, &ts->arena, &ts->loss, &ts->d_step}) b->release();
        delete ts;
        h->train = nullptr;
    }
    if (h->infer_fused) {
        InferFused *is = reinterpret_cast<InferFused *>(h->infer_fused);
        is->theta.release();
        is->arena.release();
        delete is;
        h->infer_fused = nullptr;
    }
    if (h->comm) (void)ncclCommDestroy(h->comm);
    if (h->h_status_base) (void)hipHostFree(h->h_status_base);
    for (auto &e : h->ev_done) if (e) (void)hipEventDestroy(e);
    if (h->ev_t0) (void)hipEventDestroy(h->ev_t0);
    if (h->ev_ctl) (void)hipEventDestroy(h->ev_ctl);
    h->pin_ctl.release();
    h->pin_train.release();
    h->pin_tout.release();
    h->pin_out.release();
    h->pin_neff.release();
    if (h->ev_t1) (void)hipEventDestroy(h->ev_t1);
    for (auto &e : h->evpool) (void)hipEventDestroy(e);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}


#include "epnn_api_weights.hip.h"
#include "epnn_api_forward.hip.h"
// ------------------------------------------------------------------------------------------------ plumbing
extern "C" int epnn_dev_alloc(epnn_handle *h, size_t bytes, void **out) {
    if (!h || !out) EPNN_FAIL("epnn_dev_alloc: null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMalloc(out, bytes ? bytes : 1));
    return 0;
}
extern "C" int epnn_dev_free(epnn_handle *h, void *p) {
    if (!h) EPNN_FAIL("epnn_dev_free: null handle");
    HIPCHK(hipSetDevice(h->device));
    if (p) HIPCHK(hipFree(p));
    return 0;
}
extern "C" int epnn_memcpy_h2d(epnn_handle *h, void *dst, const void *src, size_t bytes) {
    if (!h) EPNN_FAIL("epnn_memcpy_h2d: null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int epnn_memcpy_d2h(epnn_handle *h, void *dst, const void *src, size_t bytes) {
    if (!h) EPNN_FAIL("epnn_memcpy_d2h: null handle");
    HIPCHK(hipSetDevice(h->device));
    if (finish_forward(h)) return 1;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int epnn_timer_begin(epnn_handle *h) {
    if (!h) EPNN_FAIL("epnn_timer_begin: null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventRecord(h->ev_t0, h->stream));
    return 0;
}
extern "C" int epnn_timer_end(epnn_handle *h, float *elapsed_ms) {
    if (!h || !elapsed_ms) EPNN_FAIL("epnn_timer_end: null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventRecord(h->ev_t1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev_t1));
    HIPCHK(hipEventElapsedTime(elapsed_ms, h->ev_t0, h->ev_t1));
    return 0;
}
extern "C" int epnn_timing_at(epnn_handle *h, int idx, float *out4) {
    if (!h || !out4) EPNN_FAIL("epnn_timing_at: null argument");
    if (h->opt_profile <= 0) EPNN_FAIL("epnn_timing_at: profiling is off (epnn_set_option(\"profile\", pool_size))");
    if (idx < 0 || idx >= h->ev_next || idx < h->ev_next - h->opt_profile)
        EPNN_FAIL("epnn_timing_at: forward %d is not in the event pool (recorded %d, pool %d)", idx, h->ev_next, h->opt_profile);
    HIPCHK(hipSetDevice(h->device));
    if (finish_forward(h)) return 1;
    hipEvent_t *ev = h->evpool.data() + 4 * (idx % h->opt_profile);
    // boundaries that were not recorded (a stage without kernels of its own) coincide with the one before: that stage reads 0
    const unsigned char has = (idx % h->opt_profile) < (int)h->ev_recorded.size() ? h->ev_recorded[idx % h->opt_profile] : 0;
    hipEvent_t b[4] = {ev[0], (has & 1) ? ev[1] : ev[0], ev[2], ev[3]};
    b[2] = (has & 2) ? ev[2] : (has & 4) ? ev[3] : b[1];     // (a forward of fused molecules only: the fused stage is all of it)
    for (int k = 0; k < 3; ++k) {
        out4[k] = 0.f;
        if (b[k] != b[k + 1]) HIPCHK(hipEventElapsedTime(&out4[k], b[k], b[k + 1]));
    }
    HIPCHK(hipEventElapsedTime(&out4[3], ev[0], ev[3]));
    return 0;
}
extern "C" int epnn_last_timing(epnn_handle *h, float *out4) {
    if (!h || !out4) EPNN_FAIL("epnn_last_timing: null argument");
    return epnn_timing_at(h, h->ev_next - 1, out4);
}
extern "C" int epnn_last_stats(epnn_handle *h, int64_t *out4) {
    if (!h || !out4) EPNN_FAIL("epnn_last_stats: null argument");
    memcpy(out4, h->stats, sizeof(h->stats));
    return 0;
}
extern "C" double epnn_edge_basis_residual(epnn_handle *h) { return h ? h->edge_res : 1.0; }
extern "C" int epnn_set_option(epnn_handle *h, const char *name, int value) {
    if (!h || !name) EPNN_FAIL("epnn_set_option: null argument");
    if (!strcmp(name, "profile")) {
        HIPCHK(hipSetDevice(h->device));
        if (finish_forward(h)) return 1;
        value = std::max(0, std::min(value, 4096));
        while ((int)h->evpool.size() < 4 * value) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            h->evpool.push_back(e);
        }
        h->opt_profile = value;
        h->ev_next = 0;
    }
    else if (!strcmp(name, "force_path")) { if (value < 0 || value > 2) EPNN_FAIL("epnn_set_option: force_path must be 0, 1 or 2"); h->opt_force_path = value; h->plan.valid = false; }
    else if (!strcmp(name, "pair_cap_per_atom")) { h->pair_cap_per_atom = std::max(1, value); }
    else if (!strcmp(name, "wave2")) { if (value != -1 && value != 0 && (value < 17 || value > 32)) EPNN_FAIL("epnn_set_option: wave2 must be -1, 0 or 17..32"); h->opt_wave2 = value; h->plan.valid = false; }
    else if (!strcmp(name, "wave3")) { h->opt_wave3 = value != 0; h->plan.valid = false; }
    else if (!strcmp(name, "sync_spin_us")) { h->opt_sync_spin_us = std::max(0, value); }
    else if (!strcmp(name, "large_dedupe")) { h->opt_large_dedupe = value != 0; h->types_overflowed = false; }
    else if (!strcmp(name, "train_fused")) { if (value != 0 && value != 1) EPNN_FAIL("epnn_set_option: train_fused must be 0 (one launch per Dense layer) or 1 (row-fused kernels)"); h->opt_train_fused = value; }
    else if (!strcmp(name, "train_async")) { h->opt_train_async = value != 0; }
    else if (!strcmp(name, "train_graph")) { h->opt_train_graph = value != 0; }
    // ---- developer switches (include/epnn_dev.h)
    else if (!strcmp(name, "wave_lds")) { h->wave_lds = value; }
    else if (!strcmp(name, "wave_front")) { h->opt_wave_front = value; h->plan.valid = false; }
    else if (!strcmp(name, "wave_prio")) { h->opt_wave_prio = value; }
    else if (!strcmp(name, "wave_order")) { h->opt_wave_order = value; h->plan.valid = false; }
    else if (!strcmp(name, "large_fused")) { h->opt_large_fused = value; }
    else if (!strcmp(name, "large_merge")) { h->opt_large_merge = value; }
    else if (!strcmp(name, "large_chunks")) { h->opt_large_chunks = value; h->plan.valid = false; }
    else if (!strcmp(name, "front_inline")) { h->opt_front_inline = value != 0; }
    else if (!strcmp(name, "front_bits")) { h->opt_front_bits = value != 0; }
    else if (!strcmp(name, "large_sweep_old")) { h->opt_large_sweep_old = value != 0; h->plan.valid = false; }
    else if (!strcmp(name, "part_collective")) { h->opt_part_collective = value; }
    else if (!strcmp(name, "comm_guard")) { h->opt_comm_guard = value != 0; }
    else if (!strcmp(name, "comm_inject_fail")) { h->opt_comm_inject_fail = value != 0; }
    else if (!strcmp(name, "forward_ahead")) { h->opt_forward_ahead = value; }
    else if (!strcmp(name, "dense_small")) { h->opt_dense_small = value; }
    else if (!strcmp(name, "dense_rowfused")) { h->opt_dense_rowfused = value; }
    else if (!strcmp(name, "train_skip_padded")) { h->opt_train_skip_padded = value; }
    else if (!strcmp(name, "train_inline")) { h->opt_train_inline = value; }
    else if (!strcmp(name, "train_split")) { if (value < 0 || value > 8) EPNN_FAIL("epnn_set_option: train_split must be 0 (automatic) .. 8"); h->opt_train_split = value; }
    else EPNN_FAIL("epnn_set_option: unknown option '%s'", name);
    return 0;
}


#include "epnn_api_dense.hip.h"
#include "epnn_api_train.hip.h"
