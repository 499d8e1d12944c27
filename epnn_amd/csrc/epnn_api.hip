// C ABI of libepnn_hip.so (see include/epnn.h).  gfx950 only.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <numeric>

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
// handles can therefore leave queues out between them: n streams are created here and kept until the process ends.
extern "C" int epnn_skip_hw_queues(int device, int n) {
    if (n < 0 || n > 64) EPNN_FAIL("epnn_skip_hw_queues: n must be in 0..64");
    HIPCHK(hipSetDevice(device));
    for (int k = 0; k < n; ++k) {
        hipStream_t s;
        HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));      // (never used, never destroyed: it only holds its place)
    }
    return 0;
}

extern "C" int epnn_create(const epnn_config *cfg, int device, epnn_handle **out) {
    if (!cfg || !out) EPNN_FAIL("epnn_create: null argument");
    if (cfg->h_dim != EPNN_EDIM || cfg->e_dim != EPNN_EDIM)
        EPNN_FAIL("epnn_create: h_dim and e_dim must both be %d (charge_gn.py:377 requires e_dim == h_dim)", EPNN_EDIM);
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
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&h->h_status_base), 8 * sizeof(int), hipHostMallocDefault));
    memset(h->h_status_base, 0, 8 * sizeof(int));
    h->h_status = h->h_status_base;
    for (auto &e : h->ev_done) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (h->d_status.ensure(4 * sizeof(int))) return 1;
    HIPCHK(hipMemsetAsync(h->d_status.p, 0, 4 * sizeof(int), h->stream));
    // mu = np.linspace(0.1, cutoff, e_dim): arange(num)*step + start, last element forced to stop
    std::vector<double> mu(cfg->e_dim);
    const double start = 0.1, stop = (double)cfg->cutoff;
    const double step = (stop - start) / (double)(cfg->e_dim - 1);
    for (int k = 0; k < cfg->e_dim; ++k) mu[k] = (double)k * step + start;
    mu[cfg->e_dim - 1] = stop;
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
        double gap = std::max(mu[0], stop - mu[cfg->e_dim - 1]);
        for (int k = 0; k + 1 < cfg->e_dim; ++k) gap = std::max(gap, 0.5 * (mu[k + 1] - mu[k]));
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
        // (dsafe, cutoff) in 20 000 steps, every change bisected down to two adjacent doubles (flip = the last D of the old value).
        // With the reference's parameters there is exactly one, 6.0e-3 below the cutoff.
        const double eta = (double)cfg->eta;
        const float tol = cfg->near_tol;
        const int E = cfg->e_dim;
        auto nearf = [&](double D) -> bool {
            if (D >= stop) return false;
            const double C = (cos(3.141592653589793 * (D - 0.0) / stop) + 1.0) / 2.0;
            double best = 1e300;
            for (int k = 0; k < E; ++k) best = std::min(best, (D - mu[k]) * (D - mu[k]));
            return (float)(C * exp(-eta * best)) > tol;
        };
        std::vector<double> flips;
        bool prev = true;                                   // (every D <= dsafe is near by the bound above)
        double xprev = h->dsafe;
        if (!nearf(xprev)) { flips.push_back(xprev); prev = false; }
        const int M = 20000;
        for (int i = 1; i <= M && flips.size() <= EPNN_NFLIP_MAX; ++i) {
            const double xi = i == M ? stop : h->dsafe + (stop - h->dsafe) * (double)i / (double)M;
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
                      &h->d_status, &h->d_bsum, &h->d_pi, &h->d_pj, &h->d_psym, &h->d_pe, &h->d_pwi, &h->d_pwj, &h->s_xyz,
                      &h->s_train, &h->s_misc, &h->s_gx, &h->s_pt, &h->f_pw, &h->d_etab, &h->l_a, &h->l_P, &h->l_R, &h->l_zp, &h->l_S0,
                      &h->l_corr, &h->l_dl, &h->l_tiles, &h->l_csr_off, &h->l_csr_ent, &h->l_cnt, &h->l_nm,
                      &h->d_deg, &h->d_incoff, &h->d_nbr, &h->d_desti, &h->d_destj, &h->d_prec, &h->l_Nn, &h->l_Yb, &h->l_qbuf,
                      &h->l_Pst, &h->l_Rst, &h->l_lmol, &h->l_typrow, &h->l_typtab, &h->l_stype, &h->l_typhash,
                      &h->l_stasks, &h->l_schunk, &h->l_sfin, &h->l_sfrac, &h->dn_xs, &h->dn_hs, &h->dn_qs, &h->dn_nms,
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


// ------------------------------------------------------------------------------------------------ edge-feature basis
// The 48 Gaussian edge features of a distance, e_k(D) = C(D) exp(-eta (D - mu_k)^2) (charge_gn.py:148-161), are 48 heavily
// overlapping bumps of ONE variable: as vectors they stay in a 16-dimensional subspace to 5e-10 (relative to max e = 1)
// for every D in [0, cutoff].  With B = the 16 leading right singular vectors of the family (orthonormal, 48 x 16),
//   G = We^T e = (B^T We)^T (B^T e)   up to |We| * 5e-10,
// i.e. far below the float32 rounding of e itself.  The fused kernel's own front-end (which produces e from coordinates,
// so e IS of that family) projects every pair's e once and runs all 2T G products with K = 16 instead of 48.
// One-sided Jacobi (Hestenes) SVD in float64: accurate also for the small singular directions.  Returns the residual
// max |E - E B B^T| over the sampling grid.
static double edge_basis(const epnn_config &cfg, const std::vector<double> &mu, std::vector<double> &Bout, std::vector<float> &tab) {
    const int K = cfg.e_dim, R = EPNN_ER, ND = 1025;
    std::vector<double> E((size_t)ND * K), E0;
    const double pi_d = 3.141592653589793, cut = (double)cfg.cutoff, eta = (double)cfg.eta;
    for (int i = 0; i < ND; ++i) {
        const double D = cut * (double)i / (double)(ND - 1);
        double C = (cos(pi_d * D / cut) + 1.0) / 2.0;
        if (D <= 0.0) C = 1.0;
        if (D >= cut) C = 0.0;
        for (int k = 0; k < K; ++k) {
            const double d = D - mu[k];
            E[(size_t)i * K + k] = C * exp(-eta * d * d);
        }
    }
    E0 = E;
    std::vector<double> V((size_t)K * K, 0.0);
    for (int k = 0; k < K; ++k) V[(size_t)k * K + k] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < K; ++p)
            for (int q = p + 1; q < K; ++q) {
                double a = 0, b = 0, g = 0;
                for (int i = 0; i < ND; ++i) {
                    const double x = E[(size_t)i * K + p], y = E[(size_t)i * K + q];
                    a += x * x; b += y * y; g += x * y;
                }
                if (a == 0.0 || b == 0.0 || fabs(g) <= 1e-15 * sqrt(a * b)) continue;
                off = std::max(off, fabs(g) / sqrt(a * b));
                const double zeta = (b - a) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < ND; ++i) {
                    const double x = E[(size_t)i * K + p], y = E[(size_t)i * K + q];
                    E[(size_t)i * K + p] = c * x - sn * y;
                    E[(size_t)i * K + q] = sn * x + c * y;
                }
                for (int i = 0; i < K; ++i) {
                    const double x = V[(size_t)i * K + p], y = V[(size_t)i * K + q];
                    V[(size_t)i * K + p] = c * x - sn * y;
                    V[(size_t)i * K + q] = sn * x + c * y;
                }
            }
        if (off < 1e-14) break;
    }
    std::vector<std::pair<double, int>> sv(K);
    for (int k = 0; k < K; ++k) {
        double a = 0;
        for (int i = 0; i < ND; ++i) a += E[(size_t)i * K + k] * E[(size_t)i * K + k];
        sv[k] = {sqrt(a), k};
    }
    std::sort(sv.begin(), sv.end(), [](const std::pair<double, int> &x, const std::pair<double, int> &y) { return x.first > y.first; });
    Bout.assign((size_t)K * R, 0.0);
    for (int r = 0; r < R; ++r)
        for (int k = 0; k < K; ++k) Bout[(size_t)k * R + r] = V[(size_t)k * K + sv[r].second];
    double res = 0.0;
    std::vector<double> c(R);
    for (int i = 0; i < ND; ++i) {
        for (int r = 0; r < R; ++r) {
            double a = 0;
            for (int k = 0; k < K; ++k) a += E0[(size_t)i * K + k] * Bout[(size_t)k * R + r];
            c[r] = a;
        }
        for (int k = 0; k < K; ++k) {
            double a = 0;
            for (int r = 0; r < R; ++r) a += c[r] * Bout[(size_t)k * R + r];
            res = std::max(res, fabs(E0[(size_t)i * K + k] - a));
        }
    }
    // table of the coordinates B^T e(D) for the kernel's cubic interpolation, and its error at off-grid distances
    auto coords = [&](double D, double *out) {
        double C = (cos(pi_d * D / cut) + 1.0) / 2.0;
        if (D <= 0.0) C = 1.0;
        if (D >= cut) C = 0.0;
        for (int r = 0; r < R; ++r) out[r] = 0.0;
        for (int k = 0; k < K; ++k) {
            const double d = D - mu[k], e = C * exp(-eta * d * d);
            for (int r = 0; r < R; ++r) out[r] += e * Bout[(size_t)k * R + r];
        }
    };
    const int NT = EPNN_ETAB_N;
    tab.assign((size_t)NT * R, 0.f);
    std::vector<double> tabd((size_t)NT * R), row(R);
    for (int i = 0; i < NT; ++i) {
        coords(cut * (double)i / (double)(NT - 1), &tabd[(size_t)i * R]);
        for (int r = 0; r < R; ++r) tab[(size_t)i * R + r] = (float)tabd[(size_t)i * R + r];
    }
    // interpolation error of the method (float64 nodes; the float32 storage of the nodes is the same 6e-8 relative
    // rounding every float32 operand of the kernel has, like the reference's own float32 cast of e)
    const double inv_h = (double)(NT - 1) / cut;
    for (int t = 0; t < 20000; ++t) {
        const double D = cut * ((double)t + 0.37) / 20000.0;
        const double tt = D * inv_h;
        const int i0 = std::min(std::max((int)tt - 1, 0), NT - 4);
        const double u = tt - i0;
        const double w[4] = {-(u - 1) * (u - 2) * (u - 3) / 6.0, u * (u - 2) * (u - 3) / 2.0, -u * (u - 1) * (u - 3) / 2.0,
                             u * (u - 1) * (u - 2) / 6.0};
        coords(D, row.data());
        for (int r = 0; r < R; ++r) {
            double a = 0;
            for (int j = 0; j < 4; ++j) a += w[j] * tabd[(size_t)(i0 + j) * R + r];
            res = std::max(res, fabs(a - row[r]));
        }
    }
    return res;
}

// ------------------------------------------------------------------------------------------------ weights
static HostDense *find_layer(epnn_handle *h, int which, int t, int layer) {
    if (!h || layer < 0) return nullptr;
    if (which == EPNN_W_UPD && h->upd_generic) return layer < (int)h->updg.size() ? &h->updg[layer] : nullptr;
    if (layer > 2) return nullptr;
    if (which == EPNN_W_UPD) return &h->upd[layer];
    if (t < 0 || t >= h->cfg.T) return nullptr;
    if (which == EPNN_W_MSG) return &h->msg[t][layer];
    if (which == EPNN_W_PAS) return &h->pas[t][layer];
    return nullptr;
}

static int finish_forward(epnn_handle *h);
extern "C" int epnn_set_update_layers(epnn_handle *h, int n_hidden, const int32_t *widths) {
    if (!h || !widths) EPNN_FAIL("epnn_set_update_layers: null argument");
    if (n_hidden < 1 || n_hidden + 1 > EPNN_GMLP_LMAX) EPNN_FAIL("epnn_set_update_layers: %d hidden layers (1 .. %d are built)", n_hidden, EPNN_GMLP_LMAX - 1);
    for (int l = 0; l < n_hidden; ++l)
        if (widths[l] < 1 || widths[l] > EPNN_GMLP_WMAX) EPNN_FAIL("epnn_set_update_layers: width %d of layer %d (1 .. %d are built)", widths[l], l, EPNN_GMLP_WMAX);
    HIPCHK(hipSetDevice(h->device));
    if (h->pending.active && finish_forward(h)) return 1;
    if (h->train) EPNN_FAIL("epnn_set_update_layers: the handle already holds training state (set the layers before epnn_train_init)");
    const int H = h->cfg.hidden;
    h->upd_generic = !(n_hidden == 2 && widths[0] == H && widths[1] == H);
    h->updg.clear();
    if (h->upd_generic) {
        int n_in = h->cfg.h_dim + H;                       // [h | summed messages] (charge_gn.py:71)
        for (int l = 0; l <= n_hidden; ++l) {
            HostDense d;
            d.n_in = n_in;
            d.n_out = l < n_hidden ? widths[l] : h->cfg.h_dim;
            d.W.assign((size_t)d.n_in * d.n_out, 0.f);
            d.b.assign(d.n_out, 0.f);
            n_in = d.n_out;
            h->updg.push_back(std::move(d));
        }
    }
    h->weights_dirty = true;
    h->plan.valid = false;
    return 0;
}

extern "C" int epnn_weight_shape(epnn_handle *h, int which, int t, int layer, int32_t *n_in, int32_t *n_out) {
    HostDense *d = find_layer(h, which, t, layer);
    if (!d) EPNN_FAIL("epnn_weight_shape: bad (which=%d, t=%d, layer=%d)", which, t, layer);
    if (n_in) *n_in = d->n_in;
    if (n_out) *n_out = d->n_out;
    return 0;
}

extern "C" int epnn_set_weights(epnn_handle *h, int which, int t, int layer, const float *kernel, const float *bias) {
    HostDense *d = find_layer(h, which, t, layer);
    if (!d || !kernel || !bias) EPNN_FAIL("epnn_set_weights: bad (which=%d, t=%d, layer=%d) or null pointer", which, t, layer);
    if (h->train) { if (train_sync_to_host(h)) return 1; train_state(h)->ready = false; }   // masters are stale now
    memcpy(d->W.data(), kernel, d->W.size() * sizeof(float));
    memcpy(d->b.data(), bias, d->b.size() * sizeof(float));
    h->weights_dirty = true;
    return 0;
}

extern "C" int epnn_get_weights(epnn_handle *h, int which, int t, int layer, float *kernel, float *bias) {
    HostDense *d = find_layer(h, which, t, layer);
    if (!d) EPNN_FAIL("epnn_get_weights: bad (which=%d, t=%d, layer=%d)", which, t, layer);
    if (train_sync_to_host(h)) return 1;
    if (kernel) memcpy(kernel, d->W.data(), d->W.size() * sizeof(float));
    if (bias) memcpy(bias, d->b.data(), d->b.size() * sizeof(float));
    return 0;
}

// Re-lay the Keras kernels into MFMA fragment order (see epnn_common.h) and upload.
static int pack_weights(epnn_handle *h) {
    if (h->train && train_state(h)->inflight) {   // a training step's backward pass may still be running: inference (re)allocates shared buffers
        HIPCHK(hipStreamSynchronize(h->stream));
        train_state(h)->inflight = false;
    }
    if (train_sync_to_host(h)) return 1;          // weights trained on the device are the current ones
    if (!h->weights_dirty) return 0;
    const int nx = h->cfg.nx, F = nx + EPNN_EDIM + 1, T = h->cfg.T;
    std::vector<float> buf;
    auto alloc = [&](size_t n) {
        size_t off = (buf.size() + 63) & ~size_t(63);      // 256-byte aligned sections
        buf.resize(off + n, 0.f);
        return (int)off;
    };
    auto pack_pair = [&](HostDense (&m)[3], PairMlpPack &pk, bool is_pass) {
        const float *W1 = m[0].W.data(), *b1 = m[0].b.data(), *W2 = m[1].W.data(), *b2 = m[1].b.data();
        pk.wiF = alloc(EPNN_KA * 64);
        pk.wjF = alloc(EPNN_KA * 64);
        for (int s = 0; s < EPNN_KA; ++s)
            for (int l = 0; l < 64; ++l) {
                const int c = l & 31, hh = l >> 5, f = 2 * s + hh;
                buf[pk.wiF + s * 64 + l] = f < F ? W1[(size_t)f * 32 + c] : (f == EPNN_F1 ? b1[c] : 0.f);
                buf[pk.wjF + s * 64 + l] = f < F ? W1[(size_t)(F + f) * 32 + c] : 0.f;
            }
        pk.b1p = alloc(32);
        pk.b2p = alloc(32);
        pk.b2 = alloc(32);
        pk.w3p = alloc(32);
        pk.wqi = alloc(32);
        pk.wqj = alloc(32);
        for (int hh = 0; hh < 2; ++hh)
            for (int r = 0; r < 16; ++r) {
                buf[pk.wqi + hh * 16 + r] = W1[(size_t)(F - 1) * 32 + epnn_kappa(hh, r)];          // q is the last atom feature
                buf[pk.wqj + hh * 16 + r] = W1[(size_t)(2 * F - 1) * 32 + epnn_kappa(hh, r)];
                buf[pk.b1p + hh * 16 + r] = b1[epnn_kappa(hh, r)];
                buf[pk.b2p + hh * 16 + r] = b2[epnn_kappa(hh, r)];
                buf[pk.w3p + hh * 16 + r] = is_pass ? m[2].W[epnn_kappa(hh, r)] : 0.f;
            }
        for (int c = 0; c < 32; ++c) buf[pk.b2 + c] = b2[c];
        pk.weF = alloc(24 * 64);
        for (int s = 0; s < 24; ++s)
            for (int l = 0; l < 64; ++l) {
                const int c = l & 31, hh = l >> 5;
                buf[pk.weF + s * 64 + l] = W1[(size_t)(2 * F + 24 * hh + s) * 32 + c];
            }
        pk.w2F = alloc(16 * 64);
        for (int s = 0; s < 16; ++s)
            for (int l = 0; l < 64; ++l) {
                const int c = l & 31, hh = l >> 5;
                buf[pk.w2F + s * 64 + l] = W2[(size_t)epnn_kappa(hh, s) * 32 + c];
            }
    };
    for (int t = 0; t < T; ++t) {
        pack_pair(h->msg[t], h->widx.msg[t], false);
        pack_pair(h->pas[t], h->widx.pas[t], true);
    }
    const float *Wu1 = h->upd[0].W.data(), *Wu2 = h->upd[1].W.data(), *Wu3 = h->upd[2].W.data();
    for (int t = 0; t < T; ++t) {
        UpdPack &U = h->widx.upd[t];
        const float *W3 = h->msg[t][2].W.data(), *b3 = h->msg[t][2].b.data();
        // fold the last message Dense into the first update Dense:  Wu1_M^T (W3^T S + N b3)
        std::vector<double> fold(32 * 32), cb3(32);
        for (int o = 0; o < 32; ++o)
            for (int k = 0; k < 32; ++k) {
                double a = 0;
                for (int m = 0; m < 32; ++m) a += (double)W3[o * 32 + m] * (double)Wu1[(size_t)(EPNN_EDIM + m) * 32 + k];
                fold[o * 32 + k] = a;
            }
        for (int k = 0; k < 32; ++k) {
            double a = 0;
            for (int m = 0; m < 32; ++m) a += (double)b3[m] * (double)Wu1[(size_t)(EPNN_EDIM + m) * 32 + k];
            cb3[k] = a;
        }
        U.u1F = alloc(40 * 64);
        for (int s = 0; s < 40; ++s)
            for (int l = 0; l < 64; ++l) {
                const int c = l & 31, hh = l >> 5;
                float v;
                if (s < 24) {
                    const int u0 = (nx - hh + 1) >> 1;
                    const int fp = 2 * (u0 + s) + hh - nx;          // h feature read at a_eo[hh*32 + u0 + s]
                    v = Wu1[(size_t)fp * 32 + c];
                } else {
                    v = (float)fold[(2 * (s - 24) + hh) * 32 + c];
                }
                buf[U.u1F + s * 64 + l] = v;
            }
        U.cb3p = alloc(32);
        U.bu1p = alloc(32);
        U.bu2p = alloc(32);
        U.bu3p = alloc(64);
        for (int hh = 0; hh < 2; ++hh)
            for (int r = 0; r < 16; ++r) {
                const int k = epnn_kappa(hh, r);
                buf[U.cb3p + hh * 16 + r] = (float)cb3[k];
                buf[U.bu1p + hh * 16 + r] = h->upd[0].b[k];
                buf[U.bu2p + hh * 16 + r] = h->upd[1].b[k];
                buf[U.bu3p + hh * 16 + r] = h->upd[2].b[k];
                buf[U.bu3p + 32 + hh * 16 + r] = 32 + k < EPNN_EDIM ? h->upd[2].b[32 + k] : 0.f;
            }
        U.u2F = alloc(16 * 64);
        U.u3F = alloc(2 * 16 * 64);
        for (int s = 0; s < 16; ++s)
            for (int l = 0; l < 64; ++l) {
                const int c = l & 31, hh = l >> 5, k = epnn_kappa(hh, s);
                buf[U.u2F + s * 64 + l] = Wu2[(size_t)k * 32 + c];
                buf[U.u3F + s * 64 + l] = Wu3[(size_t)k * EPNN_EDIM + c];
                buf[U.u3F + (16 + s) * 64 + l] = 32 + c < EPNN_EDIM ? Wu3[(size_t)k * EPNN_EDIM + 32 + c] : 0.f;
            }
    }
    // ------------------------------------------------------------ fragments of the fused kernel (epnn_wave.hip.h)
    {
        WaveIndex &X = h->wvidx;
        const float *bu1 = h->upd[0].b.data(), *bu2 = h->upd[1].b.data(), *bu3 = h->upd[2].b.data();
        auto vec = [&](int len, auto &&fn) {
            const int off = alloc(len);
            for (int k = 0; k < len; ++k) buf[off + k] = (float)fn(k);
            return off;
        };
        // [nrb][steps / 4][64][4]: lane (q,m) of (rb, step s) = fn(s, q, 16rb + m)  (input selector, output feature); a lane's
        // four consecutive steps are 16 contiguous bytes (one dwordx4 load, W16_LDX in epnn_wave.hip.h)
        auto frag = [&](int nrb, int steps, auto &&fn) {
            const int off = alloc((size_t)nrb * steps * 64);
            for (int rb = 0; rb < nrb; ++rb)
                for (int s = 0; s < steps; ++s)
                    for (int l = 0; l < 64; ++l)
                        buf[off + ((rb * (steps / 4) + s / 4) * 64 + l) * 4 + (s & 3)] = (float)fn(s, l >> 4, 16 * rb + (l & 15));
            return off;
        };
        auto accf = [](int s, int q) { return 16 * (s >> 2) + 4 * q + (s & 3); };       // "acc" K order
        auto xq_row = [&](const float *W1, const float *b1, int r0, int phi, int m, double nmrow) -> double {
            if (phi == 0) return nmrow;
            if (phi <= nx) return W1[(size_t)(r0 + phi - 1) * 32 + m];
            if (phi == nx + 1) return W1[(size_t)(r0 + nx + EPNN_EDIM) * 32 + m];
            if (phi == nx + 2) return b1 ? b1[m] : 0.0;
            return 0.0;
        };
        auto unfolded = [&](const float *W1, const float *b1, int r0) {      // xq steps, then the 12 h steps
            return frag(2, EPNN_XS + 12, [&](int s, int q, int m) -> double {
                if (s < EPNN_XS) return xq_row(W1, b1, r0, 4 * s + q, m, 0.0);
                return W1[(size_t)(r0 + nx + accf(s - EPNN_XS, q)) * 32 + m];
            });
        };
        auto folded = [&](const float *W1, const float *b1, int r0) {        // 8 acc steps (Wu3 M_h), then the xq steps
            std::vector<double> prod(32 * 32), cb(32);
            for (int k = 0; k < 32; ++k)
                for (int m = 0; m < 32; ++m) {
                    double a = 0;
                    for (int f = 0; f < EPNN_EDIM; ++f) a += (double)Wu3[(size_t)k * EPNN_EDIM + f] * (double)W1[(size_t)(r0 + nx + f) * 32 + m];
                    prod[k * 32 + m] = a;
                }
            for (int m = 0; m < 32; ++m) {
                double a = 0;
                for (int f = 0; f < EPNN_EDIM; ++f) a += (double)bu3[f] * (double)W1[(size_t)(r0 + nx + f) * 32 + m];
                cb[m] = a;
            }
            return frag(2, 8 + EPNN_XS, [&](int s, int q, int m) -> double {
                if (s < 8) return prod[accf(s, q) * 32 + m];
                return xq_row(W1, b1, r0, 4 * (s - 8) + q, m, cb[m]);
            });
        };
        const bool have_basis = (int)h->edge_B.size() == EPNN_EDIM * EPNN_ER;
        auto pair_common = [&](HostDense (&mm)[3], int &we, int &we16, int &w2, int &b2) {
            const float *W1 = mm[0].W.data(), *W2 = mm[1].W.data(), *bb2 = mm[1].b.data();
            we = frag(2, 12, [&](int s, int q, int m) { return (double)W1[(size_t)(2 * F + 12 * q + s) * 32 + m]; });
            we16 = frag(2, EPNN_ER / 4, [&](int s, int q, int m) -> double {      // (B^T We)[4q + s][m]
                if (!have_basis) return 0.0;
                double a = 0;
                for (int ch = 0; ch < EPNN_EDIM; ++ch) a += h->edge_B[(size_t)ch * EPNN_ER + 4 * q + s] * (double)W1[(size_t)(2 * F + ch) * 32 + m];
                return a;
            });
            w2 = frag(2, 8, [&](int s, int q, int m) { return (double)W2[(size_t)accf(s, q) * 32 + m]; });
            b2 = vec(32, [&](int k) { return (double)bb2[k]; });
        };
        std::vector<double> pu1(32 * 32), cu3(32);
        for (int k = 0; k < 32; ++k)
            for (int m = 0; m < 32; ++m) {
                double a = 0;
                for (int f = 0; f < EPNN_EDIM; ++f) a += (double)Wu3[(size_t)k * EPNN_EDIM + f] * (double)Wu1[(size_t)f * 32 + m];
                pu1[k * 32 + m] = a;
            }
        for (int m = 0; m < 32; ++m) {
            double a = 0;
            for (int f = 0; f < EPNN_EDIM; ++f) a += (double)bu3[f] * (double)Wu1[(size_t)f * 32 + m];
            cu3[m] = a;
        }
        const int off_pu1 = frag(2, 8, [&](int s, int q, int m) { return pu1[accf(s, q) * 32 + m]; });
        const int off_cu3 = vec(32, [&](int k) { return cu3[k]; });
        const int off_u2 = frag(2, 8, [&](int s, int q, int m) { return (double)Wu2[(size_t)accf(s, q) * 32 + m]; });
        const int off_bu1 = vec(32, [&](int k) { return (double)bu1[k]; });
        const int off_bu2 = vec(32, [&](int k) { return (double)bu2[k]; });
        for (int t = 0; t < T; ++t) {
            WaveGnnPack &G = X.g[t];
            pair_common(h->msg[t], G.we, G.we16, G.w2, G.b2);
            const float *W3 = h->msg[t][2].W.data(), *b3 = h->msg[t][2].b.data();
            std::vector<double> fold(32 * 32), cb3(32);
            for (int k = 0; k < 32; ++k)
                for (int m = 0; m < 32; ++m) {
                    double a = 0;
                    for (int j = 0; j < 32; ++j) a += (double)W3[k * 32 + j] * (double)Wu1[(size_t)(EPNN_EDIM + j) * 32 + m];
                    fold[k * 32 + m] = a;
                }
            for (int m = 0; m < 32; ++m) {
                double a = 0;
                for (int j = 0; j < 32; ++j) a += (double)b3[j] * (double)Wu1[(size_t)(EPNN_EDIM + j) * 32 + m];
                cb3[m] = a;
            }
            G.u1s = frag(2, 8, [&](int s, int q, int m) { return fold[accf(s, q) * 32 + m]; });
            G.cb3 = vec(32, [&](int k) { return cb3[k]; });
            G.bu1 = off_bu1;
            G.u2 = off_u2;
            G.bu2 = off_bu2;
            G.pu1 = off_pu1;
            G.cu3 = off_cu3;
            if (t + 1 < T) {
                const float *N1 = h->msg[t + 1][0].W.data(), *nb1 = h->msg[t + 1][0].b.data();
                G.pwi = folded(N1, nb1, 0);
                G.pwj = folded(N1, nullptr, F);
            } else {
                G.pwi = G.pwj = 0;
            }
        }
        X.wi0 = unfolded(h->msg[0][0].W.data(), h->msg[0][0].b.data(), 0);
        X.wj0 = unfolded(h->msg[0][0].W.data(), nullptr, F);
        X.u1h0 = frag(2, 12, [&](int s, int q, int m) { return (double)Wu1[(size_t)accf(s, q) * 32 + m]; });
        X.u3 = frag(3, 8, [&](int s, int q, int m) { return (double)Wu3[(size_t)accf(s, q) * EPNN_EDIM + m]; });
        X.bu3 = vec(48, [&](int k) { return (double)bu3[k]; });
        for (int t = 0; t < T; ++t) {
            WaveEpnPack &E = X.e[t];
            pair_common(h->pas[t], E.we, E.we16, E.w2, E.b2);
            const float *W1 = h->pas[t][0].W.data(), *b1 = h->pas[t][0].b.data();
            E.w3 = vec(32, [&](int k) { return (double)h->pas[t][2].W[k]; });
            E.wi = unfolded(W1, b1, 0);
            E.wj = unfolded(W1, nullptr, F);
            E.wif = folded(W1, b1, 0);
            E.wjf = folded(W1, nullptr, F);
        }
    }
    std::vector<float> gbuf;
    if (h->upd_generic) {
        // the generic update stage takes its kernels as they are: [W3_t | b3_t] of every message MLP, then the update MLP's layers
        auto put = [&](const std::vector<float> &v) {
            const int off = (int)gbuf.size();
            gbuf.insert(gbuf.end(), v.begin(), v.end());
            return off;
        };
        for (int t = 0; t < T; ++t) {
            h->gen_w3[t] = put(h->msg[t][2].W);
            h->gen_b3[t] = put(h->msg[t][2].b);
        }
        GenMlp &G = h->gen_upd;
        G.n = (int)h->updg.size();
        G.dims[0] = h->updg[0].n_in;
        for (int l = 0; l < G.n; ++l) {
            G.dims[l + 1] = h->updg[l].n_out;
            G.offW[l] = put(h->updg[l].W);
            G.offB[l] = put(h->updg[l].b);
        }
        G.w = nullptr;
        if (h->d_updgen.ensure(gbuf.size() * sizeof(float))) return 1;
        HIPCHK(hipMemcpyAsync(h->d_updgen.p, gbuf.data(), gbuf.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    }
    if (h->d_wpack.ensure(buf.size() * sizeof(float))) return 1;
    HIPCHK(hipMemcpyAsync(h->d_wpack.p, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));     // buf / gbuf are locals
    h->weights_dirty = false;
    h->weights_gen += 1;
    return 0;
}

// ------------------------------------------------------------------------------------------------ plan
// allow_mid: the block-per-wavefront kernel may be used (compact entry and the literal make_model entry: both stacks in one launch)
// payload_bytes / ctl_fresh (host entry): room for the call's inputs behind the index arrays, in the page-locked staging and
// in its device mirror alike, so that ONE host-to-device copy carries everything a forward needs (plan_payload_offset);
// with ctl_fresh given, the upload of freshly built index arrays is left to the caller, who sends them with the payload.
// index arrays of a plan: wblk [4B + 4] int4 | moff [B + 1] | mflag [B] | molof [A]   (wblk: one entry per wavefront of the fused
// kernels; the block-per-wavefront kernel has two to four per workgroup)
static size_t plan_wblk_cap(int B) { return 4 * (size_t)B + 4; }
static size_t plan_ctl_ints(int B, int A) { return 4 * plan_wblk_cap(B) + 2 * (size_t)B + 1 + (size_t)A; }
static size_t plan_payload_offset(int B, int A) { return (plan_ctl_ints(B, A) * sizeof(int) + 255) & ~size_t(255); }
static int build_plan(epnn_handle *h, int B, int N, const int32_t *offsets, bool allow_mid = false, size_t payload_bytes = 0,
                      bool *ctl_fresh = nullptr) {
    Plan &P = h->plan;
    allow_mid = allow_mid && h->opt_force_path == 0 && !h->upd_generic;
    if (ctl_fresh) *ctl_fresh = false;
    if (P.valid && P.B == B && P.N == N && P.allow_mid == allow_mid && (int)P.offsets.size() == B + 1 &&
        memcmp(P.offsets.data(), offsets, (B + 1) * sizeof(int)) == 0 &&
        (payload_bytes == 0 || plan_payload_offset(B, P.A) + payload_bytes <= std::min(h->d_ctl.cap, h->pin_ctl.cap)))
        return 0;
    if (B < 1) EPNN_FAIL("forward: batch must have at least one molecule");
    if (offsets[0] != 0) EPNN_FAIL("forward: offsets[0] must be 0");
    for (int b = 0; b < B; ++b) {                   // before anything is sized by offsets[B] or indexed by an offset
        const long long n = (long long)offsets[b + 1] - (long long)offsets[b];
        if (n < 1) EPNN_FAIL("forward: molecule %d has %lld atoms", b, n);
        if (n > N) EPNN_FAIL("forward: molecule %d has %lld atoms but the padded size N is %d", b, n, N);
    }
    P.valid = false;
    P.B = B;
    P.N = N;
    P.A = offsets[B];
    P.offsets.assign(offsets, offsets + B + 1);
    P.allow_mid = allow_mid;
    P.small_order.clear();
    P.split_order.clear();
    P.single_order.clear();
    P.split3_order.clear();
    P.split4_order.clear();
    P.pair_wgs = 0;
    P.large_list.clear();
    P.small_nmax = 0;
    // index arrays of the plan (plan_ctl_ints), written straight into page-locked memory and uploaded without waiting
    if (h->ctl_uploading) {                         // the previous plan's upload must have run before its staging is reused
        HIPCHK(hipEventSynchronize(h->ev_ctl));
        h->ctl_uploading = false;
    }
    const size_t ctl_total = plan_payload_offset(B, P.A) + payload_bytes;
    if (h->pin_ctl.ensure(ctl_total)) return 1;
    int4 *c_wblk = h->pin_ctl.as<int4>();
    int *c_moff = h->pin_ctl.as<int>() + 4 * plan_wblk_cap(B), *c_mflag = c_moff + B + 1, *c_molof = c_mflag + B;
    // the block-per-wavefront kernel takes molecules of >= thr2 atoms (split) and of <= 16 atoms (in pairs); 0: not used
    const int want2 = h->opt_wave2 >= 0 ? h->opt_wave2 : (B <= EPNN_W2_AUTO_MAX ? 17 : 0);
    const int thr2 = (allow_mid && want2 > 0) ? std::max(17, want2) : 0;
    std::vector<int> pbase(B);
    int count[EPNN_SMALL_NMAX + 2] = {0};
    long long run = 0;
    const bool wave_ok = h->cfg.nx + 3 <= 4 * EPNN_XS;       // the fused kernel's xq block holds nx + 3 inputs
    for (int b = 0; b < B; ++b) {
        const int n = offsets[b + 1] - offsets[b];
        for (int a = offsets[b]; a < offsets[b + 1]; ++a) c_molof[a] = b;
        // (an update MLP of other widths than [32, 32]: the fused kernels are not built for it, everything is tiled)
        const bool small = !h->upd_generic && ((h->opt_force_path == 1) || (h->opt_force_path == 0 && n <= EPNN_SMALL_NMAX && wave_ok));
        if (h->upd_generic && h->opt_force_path == 1) EPNN_FAIL("forward: force_path=1 with an update MLP other than [32, 32] (tiled path only)");
        if (small && (n > EPNN_SMALL_NMAX || !wave_ok))
            EPNN_FAIL("forward: force_path=1 but molecule %d has %d atoms (fused kernel: n <= %d, nx <= %d)", b, n, EPNN_SMALL_NMAX, 4 * EPNN_XS - 3);
        const bool mid = !small && allow_mid && h->opt_wave3 && wave_ok && n > EPNN_SMALL_NMAX && n <= EPNN_W2_NMAX4;
        c_mflag[b] = small || mid ? 0 : 1 + (int)P.large_list.size();   // 1 + its place among the tiled molecules
        if (small) {
            P.small_nmax = std::max(P.small_nmax, n);
            if (thr2 && n >= thr2) P.split_order.push_back(b);
            else if (thr2 && n <= 16) P.single_order.push_back(b);
            else {
                P.small_order.push_back(b);
                count[n] += 1;
            }
        } else if (mid) {
            (n <= EPNN_W2_NMAX3 ? P.split3_order : P.split4_order).push_back(b);     // three / four wavefronts each
        } else {
            P.large_list.push_back(b);
        }
        // pair slots of the in-kernel front-end: every i<j pair of every molecule
        pbase[b] = (int)run;
        run += small || mid ? (long long)n * (n - 1) / 2 : 0;
    }
    if (run > 0x7fffffffLL / 64) EPNN_FAIL("forward: batch too large (%lld pair slots)", run);
    P.pair_slots = (int)run;
    memcpy(c_moff, offsets, (size_t)(B + 1) * sizeof(int));
    {   // largest molecules first (their wavefronts run longest), equal sizes in batch order: counting sort on n
        int start[EPNN_SMALL_NMAX + 2], at = 0;
        for (int n = EPNN_SMALL_NMAX; n >= 0; --n) { start[n] = at; at += count[n]; }
        std::vector<int> sorted(P.small_order.size());
        for (int b : P.small_order) sorted[start[offsets[b + 1] - offsets[b]]++] = b;
        if (h->opt_wave_order == 1) {            // developer switch: largest, smallest, second largest, second smallest, ...
            std::vector<int> mix(sorted.size());
            size_t lo = 0, hi = sorted.size();
            for (size_t k = 0; k < sorted.size(); ++k) mix[k] = (k & 1) ? sorted[--hi] : sorted[lo++];
            sorted.swap(mix);
        } else if (h->opt_wave_order == 2) {     // smallest first
            std::reverse(sorted.begin(), sorted.end());
        }
        P.small_order.swap(sorted);
        for (size_t k = 0; k < P.small_order.size(); ++k) {
            const int b = P.small_order[k];
            c_wblk[k] = make_int4(b, offsets[b], offsets[b + 1] - offsets[b], pbase[b]);
        }
        auto larger_first = [&](int a, int c) { return offsets[a + 1] - offsets[a] > offsets[c + 1] - offsets[c]; };
        // the block-per-wavefront kernel's workgroups behind them: two entries each, split molecules first
        std::stable_sort(P.split_order.begin(), P.split_order.end(), larger_first);
        std::stable_sort(P.single_order.begin(), P.single_order.end(), larger_first);
        int4 *c_pair = c_wblk + P.small_order.size();
        size_t e = 0;
        for (int b : P.split_order) {
            const int4 ent = make_int4(b, offsets[b], (offsets[b + 1] - offsets[b]) | (EPNN_W2_SPLIT << 8), pbase[b]);
            c_pair[e++] = ent;
            c_pair[e++] = ent;
        }
        for (int b : P.single_order) c_pair[e++] = make_int4(b, offsets[b], (offsets[b + 1] - offsets[b]) | (EPNN_W2_SINGLE << 8), pbase[b]);
        if (e & 1) c_pair[e++] = make_int4(0, 0, EPNN_W2_IDLE << 8, 0);
        P.pair_wgs = (int)(e / 2);
        // the molecules of 33..48 atoms behind them: three entries each
        std::stable_sort(P.split3_order.begin(), P.split3_order.end(), larger_first);
        for (int b : P.split3_order) {
            const int4 ent = make_int4(b, offsets[b], (offsets[b + 1] - offsets[b]) | (EPNN_W2_SPLIT << 8), pbase[b]);
            for (int k = 0; k < 3; ++k) c_pair[e++] = ent;
        }
        std::stable_sort(P.split4_order.begin(), P.split4_order.end(), larger_first);
        for (int b : P.split4_order) {
            const int4 ent = make_int4(b, offsets[b], (offsets[b + 1] - offsets[b]) | (EPNN_W2_SPLIT << 8), pbase[b]);
            for (int k = 0; k < 4; ++k) c_pair[e++] = ent;
        }
    }
    // the device copy has the same layout: ONE upload per plan
    const size_t ctl_ints = plan_ctl_ints(B, P.A);
    if (h->d_ctl.ensure(ctl_total) || h->d_rowcnt.ensure((P.A + 1) * sizeof(int)) ||
        h->d_rowoff.ensure((P.A + 1) * sizeof(int)))
        return 1;
    h->p_wblk = h->d_ctl.as<int4>();
    h->p_moff = h->d_ctl.as<int>() + 4 * plan_wblk_cap(B);
    h->p_mflag = h->p_moff + B + 1;
    h->p_molof = h->p_mflag + B;
    if (ctl_fresh) {
        *ctl_fresh = true;                          // the caller uploads index arrays + payload in one copy
    } else {
        HIPCHK(hipMemcpyAsync(h->d_ctl.p, h->pin_ctl.p, ctl_ints * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipEventRecord(h->ev_ctl, h->stream));
        h->ctl_uploading = true;
    }
    if (large_plan(h)) return 1;
    P.valid = true;
    return 0;
}

static int ensure_pairs(epnn_handle *h, int pcap) {
    if (pcap <= h->pcap) return 0;
    if (h->d_pi.ensure((size_t)pcap * sizeof(int)) || h->d_pj.ensure((size_t)pcap * sizeof(int)) ||
        h->d_psym.ensure((size_t)pcap * sizeof(int)) || h->d_pwi.ensure((size_t)pcap * sizeof(float)) ||
        h->d_pwj.ensure((size_t)pcap * sizeof(float)) || h->d_pe.ensure((size_t)pcap * EPNN_EDIM * sizeof(float)) ||
        h->d_nbr.ensure(2 * (size_t)pcap * sizeof(int)) || h->d_desti.ensure((size_t)pcap * sizeof(int)) ||
        h->d_destj.ensure((size_t)pcap * sizeof(int)) || h->d_prec.ensure(2 * ((size_t)pcap + 256) * sizeof(int4)))
        return 1;
    h->pcap = pcap;
    return 0;
}

// cut2 = smallest double whose (correctly rounded, monotone) sqrt is >= cutoff: D < cutoff <=> D^2 < cut2, no sqrt per candidate
static double cutoff_squared(double cutoff) {
    double t = cutoff * cutoff;
    while (sqrt(t) >= cutoff) t = nextafter(t, 0.0);
    while (sqrt(t) < cutoff) t = nextafter(t, 1e300);
    return t;
}

struct PairSource {     // where the fused / tiled kernels read atoms and pairs from
    const float *d_x = nullptr, *d_Q = nullptr, *d_hin = nullptr, *d_qin = nullptr, *d_nm = nullptr;
    float *d_q = nullptr, *d_hout = nullptr;
    int run_gnn = 1, run_epn = 1;
    const float *d_xyz = nullptr;    // set: the wave kernel builds the pair lists of its molecules itself
    int handoff = 0;                 // ... and its last wave hands status + pair count to the host (no other kernel ran)
};

static int launch_large(epnn_handle *h, const PairSource &S, bool have_inc = false, const FrontArgs *front = nullptr) {
    return launch_large_impl(h, S.d_x, S.d_Q, S.d_hin, S.d_qin, S.d_nm, S.d_q, S.d_hout, S.run_gnn, S.run_epn, have_inc, front);
}


// Wave-autonomous fused kernel: one 64-thread workgroup (one wavefront) per molecule, fixed LDS budget per wave.
static int launch_wave(epnn_handle *h, const PairSource &S) {
    const Plan &P = h->plan;
    WaveArgs A{};
    A.wpack = h->d_wpack.as<float>();
    A.xin = S.d_x;
    A.Q = S.d_Q;
    A.wblk = h->p_wblk;
    A.row_off = h->d_rowoff.as<int>();
    A.pi = h->d_pi.as<int>();
    A.pj = h->d_pj.as<int>();
    A.psym = h->d_psym.as<int>();
    if (S.d_xyz) {      // in-kernel front-end: its own pair scratch, one slot per i<j pair of every small molecule
        const size_t slots = (size_t)std::max(1, P.pair_slots);
        if (h->f_pw.ensure(slots * 2 * 4) || h->s_pt.ensure(slots * EPNN_ER * 4)) return 1;
        A.pe = nullptr;            // the 48-channel rows are never materialised on this path
        A.pwi = h->f_pw.as<float>();
        A.pwj = h->f_pw.as<float>() + slots;
        A.pt = h->s_pt.as<float>();
    } else {
        A.pe = h->d_pe.as<float>();
        A.pwi = h->d_pwi.as<float>();
        A.pwj = h->d_pwj.as<float>();
    }
    A.handoff = S.handoff;
    A.prio_n = h->opt_wave_prio;
    A.q_out = S.d_q;
    A.h_out = S.d_hout;
    A.h_in = S.d_hin;
    A.q_in = S.d_qin;
    A.nm_in = S.d_nm;
    A.status = h->d_status.as<int>();
    A.N = P.N;
    A.T = h->cfg.T;
    A.nx = h->cfg.nx;
    A.A = P.A;
    if (h->s_gx.ensure((size_t)std::max(h->pcap, P.pair_slots) * 32 * 4)) return 1;
    A.gx = h->s_gx.as<float>();
    // worst case inside the budget: n = 32, every unordered pair + diagonal entries (528 records) and >= 1 G row
    const int lds = std::min(std::max(h->wave_lds, 16384), 65536) & ~15;
    A.lds_words = lds / 4;
#ifdef EPNN_STAMPS
    if (h->l_nm.ensure(P.small_order.size() * 4 * 64 * 8)) return 1;
    A.stamps = h->l_nm.as<unsigned long long>();
#endif
    A.xyz = S.d_xyz;
    A.cut2 = cutoff_squared((double)h->cfg.cutoff);       // D < cutoff decided without the sqrt
    A.host_status = h->h_status;          // pinned, device-visible
    A.etab = h->d_etab.as<float>();
    A.tab_n = EPNN_ETAB_N;
    A.tab_inv_h = (double)(EPNN_ETAB_N - 1) / (double)h->cfg.cutoff;
    A.flip = h->d_flip.as<double>();
    A.nflip = h->nflip;
    const dim3 grid((unsigned)P.small_order.size());
    const WaveIndex &X = h->wvidx;
    A.total_waves = (int)P.fused_count();          // reports to the hand-off: one per molecule
    bool side_mid = false;
    if (!P.split3_order.empty() || !P.split4_order.empty()) {
        // Molecules of 33..48 atoms (the reference's `mixed` set goes up to 41) and 49..64: three / four wavefronts each.  On a
        // lone handle ("wave2" != 0) these launches run BESIDE the launch of the smaller molecules,
        // on the handle's second stream -- on one stream they run one after the other (0.12 + 0.16 ms for the reference's
        // validation batch).  Pipeline lanes keep everything on their one stream.
        if (!(S.run_gnn && S.run_epn)) EPNN_FAIL("forward: internal error (block-per-wavefront kernel for a single stack)");
        if (!h->wave23_attr) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wave_forward2<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wave_forward2<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wave_forward2<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wave_forward2<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            h->wave23_attr = true;
        }
        side_mid = h->opt_wave2 != 0 && (P.pair_wgs > 0 || !P.small_order.empty());     // a lone handle (engine.Pipeline sets 0 on its
                                                                                        // lanes) with a launch to run beside
        // (the second stream is created when a handle first needs it: every stream takes one of the process's hardware queues,
        // and a pipeline of many handles wants them for its lanes)
        if (side_mid && !h->stream2) HIPCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
        hipStream_t st = side_mid ? h->stream2 : h->stream;
        if (side_mid) {
            HIPCHK(hipEventRecord(h->ev_fork, h->stream));
            HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
        }
        WaveArgs A2 = A;
        A2.wblk = A.wblk + P.small_order.size() + 2 * (size_t)P.pair_wgs;
        if (!P.split3_order.empty()) {
            const int lds23 = std::min(3 * lds, 131072);
            A2.lds_words = lds23 / 4;
            const dim3 g3((unsigned)P.split3_order.size());
            if (S.d_xyz) hipLaunchKernelGGL((k_wave_forward2<3, true>), g3, dim3(192), (size_t)lds23, st, A2, h->wvidx);
            else hipLaunchKernelGGL((k_wave_forward2<3, false>), g3, dim3(192), (size_t)lds23, st, A2, h->wvidx);
            HIPCHK(hipGetLastError());
        }
        if (!P.split4_order.empty()) {
            A2.wblk += 3 * P.split3_order.size();
            const int lds24 = std::min(4 * lds, 131072);
            A2.lds_words = lds24 / 4;
            const dim3 g4((unsigned)P.split4_order.size());
            if (S.d_xyz) hipLaunchKernelGGL((k_wave_forward2<4, true>), g4, dim3(256), (size_t)lds24, st, A2, h->wvidx);
            else hipLaunchKernelGGL((k_wave_forward2<4, false>), g4, dim3(256), (size_t)lds24, st, A2, h->wvidx);
            HIPCHK(hipGetLastError());
        }
        if (side_mid) HIPCHK(hipEventRecord(h->ev_join, h->stream2));
    }
    if (P.pair_wgs > 0) {
        // block-per-wavefront kernel: 128-thread workgroups, twice the LDS budget of a wavefront of k_wave_forward
        if (!(S.run_gnn && S.run_epn)) EPNN_FAIL("forward: internal error (block-per-wavefront kernel for a single stack)");
        WaveArgs A2 = A;
        A2.wblk = A.wblk + P.small_order.size();
        const int lds2 = std::min(2 * lds, 131072);
        A2.lds_words = lds2 / 4;
        if (!h->wave2_attr) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wave_forward2<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wave_forward2<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            h->wave2_attr = true;
        }
        if (S.d_xyz) hipLaunchKernelGGL((k_wave_forward2<2, true>), dim3((unsigned)P.pair_wgs), dim3(128), (size_t)lds2, h->stream, A2, h->wvidx);
        else hipLaunchKernelGGL((k_wave_forward2<2, false>), dim3((unsigned)P.pair_wgs), dim3(128), (size_t)lds2, h->stream, A2, h->wvidx);
        HIPCHK(hipGetLastError());
    }
    if (side_mid) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    if (P.small_order.empty()) return 0;
    if (S.d_xyz) hipLaunchKernelGGL((k_wave_forward<true, true, true>), grid, dim3(64), (size_t)lds, h->stream, A, X);
    else if (S.run_gnn && S.run_epn) hipLaunchKernelGGL((k_wave_forward<true, true, false>), grid, dim3(64), (size_t)lds, h->stream, A, X);
    else if (S.run_gnn) hipLaunchKernelGGL((k_wave_forward<true, false, false>), grid, dim3(64), (size_t)lds, h->stream, A, X);
    else hipLaunchKernelGGL((k_wave_forward<false, true, false>), grid, dim3(64), (size_t)lds, h->stream, A, X);
    HIPCHK(hipGetLastError());
    return 0;
}

static int launch_small(epnn_handle *h, const PairSource &S) {
    if (h->plan.fused_count() == 0) return 0;
    return launch_wave(h, S);
}

// arguments of the front-end's launches: pair list + incidence rows from coordinates (epnn_frontend.hip.h)
static int make_front_args(epnn_handle *h, const float *d_xyz, FrontArgs &F) {
    const Plan &P = h->plan;
    if (h->d_deg.ensure(((size_t)P.A + 1) * sizeof(int)) || h->d_incoff.ensure(((size_t)P.A + 1) * sizeof(int))) return 1;
    F = FrontArgs{};
    F.xyz = d_xyz;
    F.mol_of = h->p_molof;
    F.moff = h->p_moff;
    F.mflag = h->p_mflag;
    F.A = P.A;
    F.cutoff = (double)h->cfg.cutoff;
    F.cut2 = cutoff_squared(F.cutoff);
    F.eta = (double)h->cfg.eta;
    F.tol = h->cfg.near_tol;
    F.e_dim = h->cfg.e_dim;
    F.mu = h->d_mu.as<double>();
    F.row_cnt = h->d_rowcnt.as<int>();
    F.row_off = h->d_rowoff.as<int>();
    F.deg = h->d_deg.as<int>();
    F.inc_off = h->d_incoff.as<int>();
    F.nbr = h->d_nbr.as<int>();
    F.dest_i = h->d_desti.as<int>();
    F.dest_j = h->d_destj.as<int>();
    F.prec = h->d_prec.as<int4>();
    F.pcap = h->pcap;
    F.pi = h->d_pi.as<int>();
    F.pj = h->d_pj.as<int>();
    F.psym = h->d_psym.as<int>();
    F.pe = h->d_pe.as<float>();
    F.pwi = h->d_pwi.as<float>();
    F.pwj = h->d_pwj.as<float>();
    F.status = h->d_status.as<int>();
    return 0;
}
static int run_frontend_xyz(epnn_handle *h, const FrontArgs &F) {
    const unsigned rows = (unsigned)((F.A + 3) / 4);
    hipLaunchKernelGGL(k_front_count, dim3(rows), dim3(256), 0, h->stream, F);
    hipLaunchKernelGGL(k_front_scan_both, dim3(1), dim3(1024), 0, h->stream, F);
    hipLaunchKernelGGL(k_front_fill, dim3(rows), dim3(256), 0, h->stream, F);
    hipLaunchKernelGGL(k_front_link, dim3((unsigned)std::min<size_t>(((size_t)h->pcap + 255) / 256, 1024)), dim3(256), 0, h->stream, F);
    HIPCHK(hipGetLastError());
    return 0;
}

static bool wave_front_ok(const epnn_handle *h) { return h->opt_wave_front && h->cfg.e_dim == EPNN_EDIM && h->edge_res < 1e-8; }
static int enqueue_forward_xyz(epnn_handle *h, int B, int N, const int32_t *offsets, const float *d_xyz,
                               const float *d_x, const float *d_Q, float *d_q) {
    HIPCHK(hipSetDevice(h->device));
    if (pack_weights(h)) return 1;
    const bool front_ok = wave_front_ok(h);
    if (build_plan(h, B, N, offsets, front_ok)) return 1;
    const Plan &P = h->plan;
    // Small molecules (fused kernel): the wavefront builds its molecule's pair list itself (slots for every i<j pair, so
    // nothing can overflow; G products in the 16-dimensional edge basis, used only when it represents the features to
    // 1e-8).  Which path a molecule takes does not depend on what else is in the batch.  With small molecules only no
    // other kernel runs and the kernel's last wave also hands status + pair count to the host.
    const bool front_small = front_ok && P.fused_count() > 0;
    const bool pure = front_small && P.large_list.empty();
    if (!pure && ensure_pairs(h, std::max(h->pcap, std::max(1024, P.A * h->pair_cap_per_atom)))) return 1;
    // (the control words are left clear by the last wavefront of a fused-only forward and by the tiled path's hand-over)
    if (!h->ctl_clean) HIPCHK(hipMemsetAsync(h->d_status.p, 0, 4 * sizeof(int), h->stream));
    h->ctl_clean = false;
    h->last_front = pure;
    hipEvent_t *ev = nullptr;
    if (h->opt_profile > 0) {
        ev = h->evpool.data() + 4 * (h->ev_next % h->opt_profile);
        h->ev_next += 1;
    }
    if (ev) HIPCHK(hipEventRecord(ev[0], h->stream));
    // The separate front-end (pair list + incidence rows) serves the tiled kernels and, when the in-kernel front-end is off,
    // the fused ones.  With only tiled molecules waiting for it, the tiled path drives its launches itself, merged with the work
    // that needs just the atoms (feature rows, atom types, first projections: k_lg_first / k_lg_second); otherwise it runs
    // here as four launches of its own.
    FrontArgs F{};
    const FrontArgs *front_later = nullptr;
    if (!pure) {
        if (make_front_args(h, d_xyz, F)) return 1;
        if (h->opt_large_merge && !P.large_list.empty() && (front_small || P.fused_count() == 0)) front_later = &F;
        else if (run_frontend_xyz(h, F)) return 1;
    }
    if (ev) HIPCHK(hipEventRecord(ev[1], h->stream));
    PairSource S;
    S.d_x = d_x;
    S.d_Q = d_Q;
    S.d_q = d_q;
    S.d_xyz = front_small ? d_xyz : nullptr;
    S.handoff = pure;
    if (launch_small(h, S)) return 1;
    if (ev) HIPCHK(hipEventRecord(ev[2], h->stream));
    h->want_large_handoff = !pure;
    h->did_large_handoff = false;
    const int rc_large = launch_large(h, S, true, front_later);
    h->want_large_handoff = false;
    if (rc_large) return 1;
    if (ev) HIPCHK(hipEventRecord(ev[3], h->stream));
    // status + pair count come back with the results (from the tiled path's last kernel when it ran)
    if (!pure && !h->did_large_handoff) {
        HIPCHK(hipMemcpyAsync(h->h_status, h->d_status.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(h->h_status + 1, h->d_rowoff.as<int>() + P.A, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    }
    h->ctl_clean = pure || h->did_large_handoff;
    h->stats[1] = (int64_t)P.fused_count();
    h->stats[2] = (int64_t)P.large_list.size();
    return 0;
}

// Wait for the handle's stream: poll its status for up to "sync_spin_us" microseconds, then sleep in hipStreamSynchronize.  A host
// thread that sleeps on the completion interrupt wakes up tens of microseconds after the stream is done and now and then a
// millisecond later (one in ~30 of bench.py's 2 ms timed regions read 128 M atoms/s instead of 210 M with normal enqueue times);
// forwards of this library last 0.1 .. 0.6 ms, so the poll usually sees the end itself.
static hipError_t wait_stream(epnn_handle *h) {
    if (h->opt_sync_spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t e = hipStreamQuery(h->stream);
            if (e != hipErrorNotReady) return e;
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > h->opt_sync_spin_us) break;
        }
    }
    return hipStreamSynchronize(h->stream);
}
// wait for the stream; if the last forward overflowed a capacity, grow it and run again
static int finish_forward(epnn_handle *h) {
    for (int attempt = 0; attempt < 8; ++attempt) {
        HIPCHK(wait_stream(h));
        if (!h->pending.active) {
            if (h->last_front) h->stats[0] = h->h_status[1];       // written by the last wave of the last forward
            return 0;
        }
        const int st = h->h_status[0];
        h->stats[0] = h->h_status[1];
        if (st == 0) {
            h->pending.active = false;
            return 0;
        }
        h->stats[3] += 1;
        if (st & EPNN_ST_PAIR_OVERFLOW) {
            if (ensure_pairs(h, h->h_status[1] + h->h_status[1] / 8 + 1024)) return 1;
        }
        if (st & EPNN_ST_TYPE_OVERFLOW) h->types_overflowed = true;     // this handle sweeps all pairs in the first step from now on
        if (h->pending.redo()) return 1;
    }
    EPNN_FAIL("forward: capacity regrow did not converge");
}

extern "C" int epnn_forward_xyz_dev(epnn_handle *h, int B, int N, const int32_t *offsets, const float *d_xyz,
                                    const float *d_x, const float *d_Q, float *d_q_out) {
    if (!h || !offsets || !d_xyz || !d_x || !d_Q || !d_q_out) EPNN_FAIL("epnn_forward_xyz_dev: null argument");
    auto &pd = h->pending;
    const void *key[4] = {d_xyz, d_x, d_Q, d_q_out};
    // The SAME forward again (same batch, same device buffers: a trajectory, a benchmark loop) while the previous one may still need
    // a look at its status: enqueue first, check after -- nothing is (re)allocated for a plan that is reused, the two forwards
    // report through two status slots, and a forward that did overflow is redone with its successor behind it.  (Waiting for the
    // previous forward before enqueueing left the GPU idle for the host's 15 us between any two forwards of the tiled path.)
    const Plan &P0 = h->plan;
    const bool ahead = h->opt_forward_ahead && pd.active && memcmp(pd.key, key, sizeof(key)) == 0 && P0.valid && P0.B == B && P0.N == N &&
                       (int)P0.offsets.size() == B + 1 && memcmp(P0.offsets.data(), offsets, (B + 1) * sizeof(int)) == 0 && h->part_world == 1;
    if (!ahead && pd.active && finish_forward(h)) return 1;     // previous call may still need a regrow
    const int old_slot = pd.slot;
    std::function<int()> old_redo;
    if (ahead) old_redo = pd.redo;
    h->st_slot = ahead ? (old_slot ^ 1) : h->st_slot;
    h->h_status = h->h_status_base + 4 * h->st_slot;
    if (enqueue_forward_xyz(h, B, N, offsets, d_xyz, d_x, d_Q, d_q_out)) return 1;
    if (h->last_front && !ahead) {            // nothing can overflow with the in-kernel front-end: no need to look at this forward
        pd.active = false;                    // again, the caller may queue the next one right away (the headline loop: nothing else
        return 0;                             // is done per call)
    }
    HIPCHK(hipEventRecord(h->ev_done[h->st_slot], h->stream));
    const bool new_active = !h->last_front;
    std::vector<int> offs(offsets, offsets + B + 1);
    auto redo = [h, B, N, offs, d_xyz, d_x, d_Q, d_q_out]() {
        const int rc = enqueue_forward_xyz(h, B, N, offs.data(), d_xyz, d_x, d_Q, d_q_out);
        if (!rc) (void)hipEventRecord(h->ev_done[h->st_slot], h->stream);
        return rc;
    };
    if (ahead) {
        // the forward before this one: its own event, its own slot
        HIPCHK(hipEventSynchronize(h->ev_done[old_slot]));
        const int *os = h->h_status_base + 4 * old_slot;
        if (os[0] != 0) {
            // it overflowed a capacity (this one, enqueued behind it with the same capacities, gave up at its first kernel as well):
            // wait for everything, regrow, run the old one again until it fits, then this one
            HIPCHK(hipStreamSynchronize(h->stream));
            const int new_slot = h->st_slot;
            h->st_slot = old_slot;
            h->h_status = h->h_status_base + 4 * old_slot;
            pd.active = true;
            pd.redo = old_redo;
            pd.slot = old_slot;
            h->stats[3] += 1;
            if (os[0] & EPNN_ST_PAIR_OVERFLOW) {
                if (ensure_pairs(h, os[1] + os[1] / 8 + 1024)) return 1;
            }
            if (os[0] & EPNN_ST_TYPE_OVERFLOW) h->types_overflowed = true;
            if (pd.redo()) return 1;
            if (finish_forward(h)) return 1;
            h->st_slot = new_slot;
            h->h_status = h->h_status_base + 4 * new_slot;
            if (redo()) return 1;
        }
    }
    pd.active = new_active;
    pd.redo = redo;
    pd.slot = h->st_slot;
    memcpy(pd.key, key, sizeof(key));
    return 0;
}

// Row-block partition of a SINGLE large system over `world` processes: the all-pairs sweep -- all of
// the cost of the tiled path -- is split by atom tiles; after every GNN step `exchange` must complete the rows of S this
// process did not compute (it is called with the device pointer, the row length, the number of rows and this process's
// own row range, on a synchronised stream; epnn_memcpy_d2h / _h2d move rows).  Everything else is computed by every
// process, so all of them end with all the charges.  world = 1 switches the partition off.
extern "C" int epnn_set_partition(epnn_handle *h, int rank, int world, epnn_exchange_fn exchange, void *ctx) {
    if (!h || world < 1 || rank < 0 || rank >= world) EPNN_FAIL("epnn_set_partition: bad argument");
    if (world > 1 && !exchange && !(h->comm && h->comm_world == world && h->comm_rank == rank))
        EPNN_FAIL("epnn_set_partition: world %d needs an exchange function or a communicator of that size (epnn_comm_init) with this rank", world);
    if (h->pending.active && finish_forward(h)) return 1;
    h->part_rank = rank;
    h->part_world = world;
    h->part_exchange = exchange;
    h->part_ctx = ctx;
    h->plan.valid = false;
    return 0;
}

extern "C" int epnn_sync(epnn_handle *h) {
    if (!h) EPNN_FAIL("epnn_sync: null handle");
    HIPCHK(hipSetDevice(h->device));
    return finish_forward(h);
}

// Host entry in two halves.  begin: the inputs are copied into the handle's page-locked staging (the caller may reuse
// its arrays at once), uploads + kernel + download of the charges are queued, and the call returns without waiting for
// the GPU.  end: waits and hands the charges over.  One forward per handle between begin and end; several handles
// (engine.Pipeline) keep several batches in flight.
extern "C" int epnn_forward_xyz_begin(epnn_handle *h, int B, int N, const int32_t *offsets, const float *xyz,
                                      const float *x, const float *Q) {
    if (!h || !offsets || !xyz || !x || !Q) EPNN_FAIL("epnn_forward_xyz_begin: null argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->hostcall.active) EPNN_FAIL("epnn_forward_xyz_begin: collect the previous forward with epnn_forward_xyz_end first");
    if (B < 1) EPNN_FAIL("epnn_forward_xyz: empty batch");
    const int A = offsets[B];
    if (A < 1) EPNN_FAIL("epnn_forward_xyz: no atoms");
    const int nx = h->cfg.nx;
    if (h->pending.active && finish_forward(h)) return 1;
    auto up256 = [](size_t bytes) { return (bytes + 255) & ~size_t(255); };
    const size_t n_xyz = (size_t)A * 3, n_x = (size_t)A * nx;
    const size_t o_x = up256(n_xyz * 4), o_Q = o_x + up256(n_x * 4), in_bytes = o_Q + (size_t)B * 4;
    // ONE host-to-device copy per forward: the inputs are staged behind the plan's index arrays in the same page-locked
    // buffer, whose device mirror has the same layout.  (Separate copies for xyz, x, Q and the index arrays kept the copy
    // engine busy 73 us per batch of 1024 molecules -- of the 86 us the GPU needs for it -- and the kernels of different
    // handles then started one after the other: 2.9 launches in flight instead of 8.)
    bool fresh = false;
    if (build_plan(h, B, N, offsets, wave_front_ok(h), in_bytes, &fresh)) return 1;
    if (h->pin_out.ensure((size_t)A * 4)) return 1;
    const size_t off = plan_payload_offset(B, A);
    if (h->ctl_uploading) {                         // (a cached plan: build_plan did not wait for the staging's last upload)
        HIPCHK(hipEventSynchronize(h->ev_ctl));
        h->ctl_uploading = false;
    }
    char *stage = h->pin_ctl.as<char>() + off;
    memcpy(stage, xyz, n_xyz * 4);
    memcpy(stage + o_x, x, n_x * 4);
    memcpy(stage + o_Q, Q, (size_t)B * 4);
    const size_t from = fresh ? 0 : off;
    HIPCHK(hipMemcpyAsync(h->d_ctl.as<char>() + from, h->pin_ctl.as<char>() + from, off + in_bytes - from, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipEventRecord(h->ev_ctl, h->stream));
    h->ctl_uploading = true;
    const char *dev = h->d_ctl.as<char>() + off;
    const float *d_xyz = reinterpret_cast<const float *>(dev), *d_x = reinterpret_cast<const float *>(dev + o_x),
                *d_Q = reinterpret_cast<const float *>(dev + o_Q);
    // The charges are written by the kernels straight into the page-locked result buffer (device-visible host memory):
    // no device-to-host copy is queued.  With one, the copy engine's queue holds "results of batch k" (which waits for
    // kernel k) in front of "inputs of batch k+1", and the kernels of different handles run one after the other instead
    // of side by side (kernel trace: 0.75 instead of 4.2 kernels in flight).
    if (epnn_forward_xyz_dev(h, B, N, offsets, d_xyz, d_x, d_Q, h->pin_out.as<float>())) return 1;
    h->hostcall.active = true;
    h->hostcall.A = A;
    return 0;
}

extern "C" int epnn_forward_xyz_end(epnn_handle *h, float *q_out) {
    if (!h || !q_out) EPNN_FAIL("epnn_forward_xyz_end: null argument");
    HIPCHK(hipSetDevice(h->device));
    if (!h->hostcall.active) EPNN_FAIL("epnn_forward_xyz_end: no forward was begun on this handle");
    h->hostcall.active = false;
    if (finish_forward(h)) return 1;                 // waits; re-runs the forward if a pair list had to grow
    memcpy(q_out, h->pin_out.p, (size_t)h->hostcall.A * 4);
    return 0;
}

extern "C" int epnn_forward_xyz(epnn_handle *h, int B, int N, const int32_t *offsets, const float *xyz,
                                const float *x, const float *Q, float *q_out) {
    if (!h || !offsets || !xyz || !x || !Q || !q_out) EPNN_FAIL("epnn_forward_xyz: null argument");
    if (epnn_forward_xyz_begin(h, B, N, offsets, xyz, x, Q)) return 1;
    return epnn_forward_xyz_end(h, q_out);
}

// ------------------------------------------------------------------------------------------------ edges
static int edges_impl(epnn_handle *h, int n, const float *xyz, int num, double cutoff, double eta, const double *d_mu,
                      float *e_out, double *c_out) {
    const size_t total = (size_t)n * n * num, nn = (size_t)n * n;
    if (h->s_xyz.ensure((size_t)n * 3 * 4) || h->s_misc.ensure(total * 4 + (c_out ? nn * 8 + 8 : 0))) return 1;
    double *d_c = c_out ? reinterpret_cast<double *>(h->s_misc.as<char>() + ((total * 4 + 7) & ~size_t(7))) : nullptr;
    HIPCHK(hipMemcpyAsync(h->s_xyz.p, xyz, (size_t)n * 3 * 4, hipMemcpyHostToDevice, h->stream));
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_edges_dense, dim3(grid), dim3(256), 0, h->stream, h->s_xyz.as<float>(), n, num, cutoff, eta, d_mu,
                       h->s_misc.as<float>(), d_c);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(e_out, h->s_misc.p, total * 4, hipMemcpyDeviceToHost, h->stream));
    if (c_out) HIPCHK(hipMemcpyAsync(c_out, d_c, nn * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int epnn_edges(epnn_handle *h, int n, const float *xyz, float *e_out) {
    if (!h || !xyz || !e_out || n < 1) EPNN_FAIL("epnn_edges: bad argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->pending.active && finish_forward(h)) return 1;
    return edges_impl(h, n, xyz, h->cfg.e_dim, (double)h->cfg.cutoff, (double)h->cfg.eta, h->d_mu.as<double>(), e_out, nullptr);
}

// get_init_edges with the reference's own parameters (charge_gn.py:122: num, and the constants 3.0 / 2.0 of :148-161 as
// arguments): any number of channels, plus the cutoff weights C[n][n] (float64) the reference returns tiled.
extern "C" int epnn_edges_ex(epnn_handle *h, int n, const float *xyz, int num, double cutoff, double eta, float *e_out,
                             double *c_out) {
    if (!h || !xyz || !e_out || n < 1 || num < 2 || !(cutoff > 0.1)) EPNN_FAIL("epnn_edges_ex: bad argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->pending.active && finish_forward(h)) return 1;
    // mu = np.linspace(0.1, cutoff, num): arange(num)*step + start, last element forced to stop
    std::vector<double> mu(num);
    const double step = (cutoff - 0.1) / (double)(num - 1);
    for (int k = 0; k < num; ++k) mu[k] = (double)k * step + 0.1;
    mu[num - 1] = cutoff;
    if (h->d_mu_ex.ensure((size_t)num * sizeof(double))) return 1;
    HIPCHK(hipMemcpyAsync(h->d_mu_ex.p, mu.data(), (size_t)num * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));      // mu is a local
    return edges_impl(h, n, xyz, num, cutoff, eta, h->d_mu_ex.as<double>(), e_out, c_out);
}

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
    for (int k = 0; k < 3; ++k) HIPCHK(hipEventElapsedTime(&out4[k], ev[k], ev[k + 1]));
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
    else if (!strcmp(name, "force_path")) { h->opt_force_path = value; h->plan.valid = false; }
    else if (!strcmp(name, "pair_cap_per_atom")) { h->pair_cap_per_atom = std::max(1, value); }
    else if (!strcmp(name, "wave_lds")) { h->wave_lds = value; }
    else if (!strcmp(name, "wave_front")) { h->opt_wave_front = value; h->plan.valid = false; }
    else if (!strcmp(name, "wave3")) { h->opt_wave3 = value; h->plan.valid = false; }
    else if (!strcmp(name, "wave2")) { h->opt_wave2 = value; h->plan.valid = false; }
    else if (!strcmp(name, "wave_prio")) { h->opt_wave_prio = value; }
    else if (!strcmp(name, "large_fused")) { h->opt_large_fused = value; }
    else if (!strcmp(name, "sync_spin_us")) { h->opt_sync_spin_us = value; }
    else if (!strcmp(name, "large_dedupe")) { h->opt_large_dedupe = value; h->types_overflowed = false; }
    else if (!strcmp(name, "large_merge")) { h->opt_large_merge = value; }
    else if (!strcmp(name, "large_chunks")) { h->opt_large_chunks = value; h->plan.valid = false; }
    else if (!strcmp(name, "wave_order")) { h->opt_wave_order = value; h->plan.valid = false; }
    else if (!strcmp(name, "part_collective")) { h->opt_part_collective = value; }
    else if (!strcmp(name, "train_graph")) { h->opt_train_graph = value; }
    else if (!strcmp(name, "forward_ahead")) { h->opt_forward_ahead = value; }
    else if (!strcmp(name, "dense_small")) { h->opt_dense_small = value; }
    else if (!strcmp(name, "dense_rowfused")) { h->opt_dense_rowfused = value; }
    else if (!strcmp(name, "train_skip_padded")) { h->opt_train_skip_padded = value; }
    else if (!strcmp(name, "train_inline")) { h->opt_train_inline = value; }
    else if (!strcmp(name, "train_async")) { h->opt_train_async = value; }
    else if (!strcmp(name, "train_fused")) { h->opt_train_fused = value; }
    else if (!strcmp(name, "train_split")) { if (value < 0 || value > 8) EPNN_FAIL("epnn_set_option: train_split must be 0 (automatic) .. 8"); h->opt_train_split = value; }
    else EPNN_FAIL("epnn_set_option: unknown option '%s'", name);
    return 0;
}


// ------------------------------------------------------------------------------------------------ dense entries
// per-slot atom features, node mask and "non-trivial" flags from the dense inputs (replaces one monolithic kernel:
// every pass below is a coalesced stream)
static int launch_dense_atoms(epnn_handle *h, DenseArgs &D) {
    const size_t slots = (size_t)D.B * D.N;
    if (h->dn_den.ensure(slots * 4)) return 1;
    float *den = h->dn_den.as<float>();
    HIPCHK(hipMemsetAsync(D.flag, 0, slots * sizeof(int), h->stream));
    if (D.model_level && slots * D.N <= 65536) {            // one or a few molecules: one launch instead of four
        const int N = D.N, CT = EPNN_EDIM + D.nx + 1;
        hipLaunchKernelGGL(k_dn_feat_all, dim3((unsigned)((N * CT + 255) / 256), (unsigned)D.B), dim3(256), 0, h->stream, D, den);
        hipLaunchKernelGGL(k_dn_escan, dim3((unsigned)std::min<size_t>((slots * D.N + 255) / 256, 16384)), dim3(256), 0, h->stream, D);
        HIPCHK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k_dn_den, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, h->stream, D, den);
    if (D.model_level) {
        const int N = D.N;
        hipLaunchKernelGGL(k_dn_feat<0>, dim3((unsigned)((N * EPNN_EDIM + 255) / 256), (unsigned)D.B), dim3(256), 0, h->stream, D, den);
        hipLaunchKernelGGL(k_dn_feat<1>, dim3((unsigned)((N * D.nx + 255) / 256), (unsigned)D.B), dim3(256), 0, h->stream, D, den);
        hipLaunchKernelGGL(k_dn_feat<2>, dim3((unsigned)((N + 255) / 256), (unsigned)D.B), dim3(256), 0, h->stream, D, den);
    } else {
        hipLaunchKernelGGL(k_dn_copy_atoms, dim3((unsigned)std::min<size_t>((slots * (D.nx + EPNN_EDIM + 1) + 255) / 256, 8192)),
                           dim3(256), 0, h->stream, D);
    }
    hipLaunchKernelGGL(k_dn_escan, dim3((unsigned)std::min<size_t>((slots * D.N + 255) / 256, 16384)), dim3(256), 0, h->stream, D);
    HIPCHK(hipGetLastError());
    return 0;
}

// mode 0: make_model (model-level inputs, both stacks); 1: GNN_layer.call; 2: EPN_layer.call.  Device pointers.
static int enqueue_dense(epnn_handle *h, int B, int N, int mode, const float *d_h, const float *d_e, const float *d_x,
                         const float *d_q, const float *d_mask, float *d_out) {
    HIPCHK(hipSetDevice(h->device));
    if (B < 1 || N < 1) EPNN_FAIL("dense forward: B and N must be positive");
    if (pack_weights(h)) return 1;
    const int nx = h->cfg.nx;
    const size_t slots = (size_t)B * N;
    if (h->dn_xs.ensure(slots * nx * 4) || h->dn_hs.ensure(slots * EPNN_EDIM * 4) || h->dn_qs.ensure(slots * 4) ||
        h->dn_nms.ensure(slots * 4) || h->dn_flag.ensure(slots * 4) || h->dn_neff.ensure((size_t)B * 4))
        return 1;
    DenseArgs D{};
    D.B = B;
    D.N = N;
    D.nx = nx;
    D.model_level = mode == 0;
    D.h_in = d_h;
    D.e_in = d_e;
    D.x_in = d_x;
    D.q_in = d_q;
    D.mask_in = d_mask;
    D.xs = h->dn_xs.as<float>();
    D.hs = h->dn_hs.as<float>();
    D.qs = h->dn_qs.as<float>();
    D.nms = h->dn_nms.as<float>();
    D.flag = h->dn_flag.as<int>();
    D.neff = h->dn_neff.as<int>();
    D.tol = h->cfg.near_tol;
    // one or a few molecules: the call is made of latencies -- everything the host waits for is one launch that writes the
    // effective atom counts into page-locked memory itself (dense entry)
    const bool small_call = mode == 0 && h->opt_dense_small && slots * N <= 65536;
    h->dn_neff_host.resize(B);
    if (small_call) {
        if (h->pin_neff.ensure((size_t)B * 4) || h->dn_den.ensure(slots * 4)) return 1;
        // a flag counts when it equals this call's generation number: numbers start at 2 (the general sequence writes 0 / 1 into
        // the same array), and a new allocation or a wrapped counter starts from a cleared array
        if (h->dn_flag.p != h->dn_flag_seen || h->dn_gen >= 0x7ffffff0) {
            HIPCHK(hipMemsetAsync(h->dn_flag.p, 0, h->dn_flag.cap, h->stream));
            h->dn_flag_seen = h->dn_flag.p;
            h->dn_gen = 1;
        }
        h->dn_gen += 1;
        const int fb = (N * (EPNN_EDIM + nx + 1) + 255) / 256, eb = (N * N + 255) / 256;
        if (h->opt_dense_rowfused && (size_t)B * N <= 256 && N <= 48 && infer_rowfused_fits(h, N)) {
            // a padded size this small: the row-fused forward is at least as fast as the fused kernel on one CU whatever the
            // molecule's real size, and it does not need the effective atom counts -- no host synchronisation in the middle of
            // the call (the per-atom features are all it takes from the front-end: the feature blocks alone)
            hipLaunchKernelGGL(k_dn_front_small, dim3((unsigned)fb, (unsigned)B), dim3(256), 0, h->stream, D, h->dn_den.as<float>(), h->dn_gen, fb);
            HIPCHK(hipGetLastError());
            if (infer_rowfused_forward(h, B, N, d_e, d_mask, D.xs, D.hs, D.qs, d_out)) return 1;
            h->h_status[0] = 0;
            h->h_status[1] = 0;
            h->last_front = false;
            h->stats[1] = 0;
            h->stats[2] = 0;
            return 0;
        }
        hipLaunchKernelGGL(k_dn_front_small, dim3((unsigned)(fb + eb), (unsigned)B), dim3(256), 0, h->stream, D, h->dn_den.as<float>(), h->dn_gen, fb);
        hipLaunchKernelGGL(k_dn_neff_small, dim3((unsigned)B), dim3(64), 0, h->stream, D, h->dn_gen, h->pin_neff.as<int>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
        memcpy(h->dn_neff_host.data(), h->pin_neff.p, (size_t)B * 4);
    } else {
        if (launch_dense_atoms(h, D)) return 1;
        hipLaunchKernelGGL(k_dn_neff, dim3((unsigned)B), dim3(64), 0, h->stream, D);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h->dn_neff_host.data(), D.neff, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));       // the host plans tiles from the effective atom counts
    }
    std::vector<int> offsets(B + 1, 0);
    for (int b = 0; b < B; ++b) offsets[b + 1] = offsets[b] + h->dn_neff_host[b];
    if (small_call && h->opt_dense_rowfused && (size_t)B * N <= 256 && infer_rowfused_fits(h, N) &&
        20 * *std::max_element(h->dn_neff_host.begin(), h->dn_neff_host.end()) >= 11 * N) {
        // a lone molecule that fills most of its padded size: one workgroup per atom slot through the row-fused forward kernels
        // instead of one CU for the whole molecule (dense entry).  No pair list, no scatter: the kernels
        // take the dense tensors and write (B,N,1); padded slots come out as exact zeros (q = 0, every transfer weight 0).
        if (infer_rowfused_forward(h, B, N, d_e, d_mask, D.xs, D.hs, D.qs, d_out)) return 1;
        h->h_status[0] = 0;
        h->h_status[1] = 0;
        h->last_front = false;
        h->stats[1] = 0;                          // (neither the fused nor the tiled kernels: that is how epnn_last_stats shows this path)
        h->stats[2] = 0;
        return 0;
    }
    if (build_plan(h, B, N, offsets.data(), mode == 0)) return 1;     // both stacks: the block-per-wavefront kernel may take part
    const Plan &P = h->plan;
    const size_t A = (size_t)P.A;
    const int C = mode == 1 ? EPNN_EDIM : 1;
    if (h->dn_xf.ensure(A * nx * 4) || h->dn_hf.ensure(A * EPNN_EDIM * 4) || h->dn_qf.ensure(A * 4) ||
        h->dn_nmf.ensure(A * 4) || h->dn_out.ensure(A * EPNN_EDIM * 4))
        return 1;
    if (ensure_pairs(h, std::max(h->pcap, std::max(1024, P.A * h->pair_cap_per_atom)))) return 1;
    D.A = P.A;
    D.moff = h->p_moff;
    D.mol_of = h->p_molof;
    D.xf = h->dn_xf.as<float>();
    D.hf = h->dn_hf.as<float>();
    D.qf = h->dn_qf.as<float>();
    D.nmf = h->dn_nmf.as<float>();
    D.row_cnt = h->d_rowcnt.as<int>();
    D.row_off = h->d_rowoff.as<int>();
    D.pcap = h->pcap;
    D.pi = h->d_pi.as<int>();
    D.pj = h->d_pj.as<int>();
    D.psym = h->d_psym.as<int>();
    D.pe = h->d_pe.as<float>();
    D.pwi = h->d_pwi.as<float>();
    D.pwj = h->d_pwj.as<float>();
    D.status = h->d_status.as<int>();
    h->ctl_clean = false;
    h->last_front = false;
    if (small_call && P.A <= 1024) {
        const unsigned rows = (unsigned)((P.A + 3) / 4);
        hipLaunchKernelGGL(k_dn_pairs_count_small, dim3(rows), dim3(256), 0, h->stream, D);
        hipLaunchKernelGGL(k_dn_pairs_fill_small, dim3(rows), dim3(256), 0, h->stream, D);
    } else {
        HIPCHK(hipMemsetAsync(h->d_status.p, 0, 4 * sizeof(int), h->stream));
        const unsigned rows = (unsigned)((P.A + 3) / 4);
        hipLaunchKernelGGL(k_dn_pairs<0>, dim3(rows), dim3(256), 0, h->stream, D);
        FrontArgs F{};
        F.A = P.A;
        F.row_cnt = h->d_rowcnt.as<int>();
        F.row_off = h->d_rowoff.as<int>();
        F.pcap = h->pcap;
        F.status = h->d_status.as<int>();
        {
            const unsigned nsb = (unsigned)((F.A + EPNN_SCAN_ELEMS - 1) / EPNN_SCAN_ELEMS);
            if (h->d_bsum.ensure((size_t)nsb * sizeof(int))) return 1;
            hipLaunchKernelGGL(k_front_scan1, dim3(nsb), dim3(256), 0, h->stream, F, h->d_bsum.as<int>());
            if (nsb > 1) hipLaunchKernelGGL(k_front_scan2, dim3(nsb), dim3(256), 0, h->stream, F, h->d_bsum.as<int>());
        }
        hipLaunchKernelGGL(k_dn_pairs<1>, dim3(rows), dim3(256), 0, h->stream, D);
    }
    HIPCHK(hipGetLastError());
    PairSource S;
    S.d_x = D.xf;
    S.d_hin = D.hf;
    S.d_qin = D.qf;
    S.d_nm = D.nmf;
    S.run_gnn = mode != 2;
    S.run_epn = mode != 1;
    float *flat = h->dn_out.as<float>();
    S.d_q = mode == 1 ? nullptr : flat;
    S.d_hout = mode == 1 ? flat : nullptr;
    if (mode == 1) S.d_q = h->dn_qf.as<float>();      // the fused kernel always stores q; keep it off the h buffer
    if (launch_small(h, S)) return 1;
    if (launch_large(h, S)) return 1;
    D.out = d_out;
    D.src = flat;
    D.C = C;
    const unsigned gO = (unsigned)std::min<size_t>((slots * C + 255) / 256, 8192);
    D.host_status = h->h_status;                  // pinned, device-visible: written by the scatter kernel
    hipLaunchKernelGGL(k_dn_scatter, dim3(gO), dim3(256), 0, h->stream, D);
    HIPCHK(hipGetLastError());
    h->stats[1] = (int64_t)P.fused_count();
    h->stats[2] = (int64_t)P.large_list.size();
    return 0;
}

static int dense_dev(epnn_handle *h, int B, int N, int mode, const float *d_h, const float *d_e, const float *d_x,
                     const float *d_q, const float *d_mask, float *d_out) {
    if (!h || !d_h || !d_e || !d_x || !d_q || !d_mask || !d_out) EPNN_FAIL("dense forward: null argument");
    if (h->pending.active && finish_forward(h)) return 1;
    if (enqueue_dense(h, B, N, mode, d_h, d_e, d_x, d_q, d_mask, d_out)) return 1;
    h->pending.active = true;
    h->pending.key[0] = nullptr;                  // (not a compact forward: the next one waits for this one the usual way)
    h->pending.redo = [=]() { return enqueue_dense(h, B, N, mode, d_h, d_e, d_x, d_q, d_mask, d_out); };
    return 0;
}

static int dense_host(epnn_handle *h, int B, int N, int mode, const float *hh, const float *e, const float *x,
                      const float *q, const float *mask, float *out) {
    if (!h || !hh || !e || !x || !q || !mask || !out) EPNN_FAIL("dense forward: null argument");
    HIPCHK(hipSetDevice(h->device));
    const int nx = h->cfg.nx;
    const size_t pairs = (size_t)B * N * N, atoms = (size_t)B * N;
    const size_t nh = (mode == 0 ? pairs : atoms) * EPNN_EDIM, nxx = (mode == 0 ? pairs : atoms) * nx,
                 nq = mode == 0 ? pairs : atoms, ne = pairs * EPNN_EDIM, nm = pairs;
    const size_t nout = atoms * (mode == 1 ? EPNN_EDIM : 1);
    auto up256 = [](size_t bytes) { return (bytes + 255) & ~size_t(255); };
    const size_t o_e = up256(nh * 4), o_x = o_e + up256(ne * 4), o_q = o_x + up256(nxx * 4), o_m = o_q + up256(nq * 4),
                 in_bytes = o_m + nm * 4;
    if (in_bytes <= ((size_t)4 << 20)) {
        // A call on one or a few molecules (the reference's loop, infer.py:62-76) is made of latencies: the five tensors go
        // through ONE page-locked staging buffer and ONE upload, the result comes back through page-locked memory (five
        // uploads from pageable memory and a pageable download were ~50 us of a 0.3 ms call).  Larger batches keep the
        // direct copies (staging 369 MB by hand would cost more than it saves).
        if (h->pin_train.ensure(in_bytes) || h->s_train.ensure(in_bytes) || h->sd_out.ensure(nout * 4) || h->pin_tout.ensure(nout * 4))
            return 1;
        char *stage = h->pin_train.as<char>();
        memcpy(stage, hh, nh * 4);
        memcpy(stage + o_e, e, ne * 4);
        memcpy(stage + o_x, x, nxx * 4);
        memcpy(stage + o_q, q, nq * 4);
        memcpy(stage + o_m, mask, nm * 4);
        HIPCHK(hipMemcpyAsync(h->s_train.p, stage, in_bytes, hipMemcpyHostToDevice, h->stream));
        const char *dev = h->s_train.as<char>();
        if (dense_dev(h, B, N, mode, reinterpret_cast<const float *>(dev), reinterpret_cast<const float *>(dev + o_e),
                      reinterpret_cast<const float *>(dev + o_x), reinterpret_cast<const float *>(dev + o_q),
                      reinterpret_cast<const float *>(dev + o_m), h->sd_out.as<float>()))
            return 1;
        if (finish_forward(h)) return 1;
        HIPCHK(hipMemcpyAsync(h->pin_tout.p, h->sd_out.p, nout * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        memcpy(out, h->pin_tout.p, nout * 4);
        return 0;
    }
    if (h->sd_h.ensure(nh * 4) || h->sd_e.ensure(ne * 4) || h->sd_x.ensure(nxx * 4) || h->sd_q.ensure(nq * 4) ||
        h->sd_mask.ensure(nm * 4) || h->sd_out.ensure(nout * 4))
        return 1;
    HIPCHK(hipMemcpyAsync(h->sd_h.p, hh, nh * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->sd_e.p, e, ne * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->sd_x.p, x, nxx * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->sd_q.p, q, nq * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->sd_mask.p, mask, nm * 4, hipMemcpyHostToDevice, h->stream));
    if (dense_dev(h, B, N, mode, h->sd_h.as<float>(), h->sd_e.as<float>(), h->sd_x.as<float>(), h->sd_q.as<float>(),
                  h->sd_mask.as<float>(), h->sd_out.as<float>()))
        return 1;
    if (finish_forward(h)) return 1;
    HIPCHK(hipMemcpyAsync(out, h->sd_out.p, nout * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int epnn_model_forward_dense(epnn_handle *h, int B, int N, const float *h_inp, const float *e_inp,
                                        const float *x_inp, const float *q_inp, const float *mask_inp, float *q_out) {
    return dense_host(h, B, N, 0, h_inp, e_inp, x_inp, q_inp, mask_inp, q_out);
}
extern "C" int epnn_model_forward_dense_dev(epnn_handle *h, int B, int N, const float *d_h_inp, const float *d_e_inp,
                                            const float *d_x_inp, const float *d_q_inp, const float *d_mask_inp,
                                            float *d_q_out) {
    return dense_dev(h, B, N, 0, d_h_inp, d_e_inp, d_x_inp, d_q_inp, d_mask_inp, d_q_out);
}
extern "C" int epnn_gnn_forward(epnn_handle *h, int B, int N, const float *hin, const float *e, const float *x,
                                const float *q, const float *mask, float *h_out) {
    return dense_host(h, B, N, 1, hin, e, x, q, mask, h_out);
}
extern "C" int epnn_epn_forward(epnn_handle *h, int B, int N, const float *hin, const float *e, const float *x,
                                const float *q, const float *mask, float *q_out) {
    return dense_host(h, B, N, 2, hin, e, x, q, mask, q_out);
}

// ------------------------------------------------------------------------------------------------ MLP_layer.call
extern "C" int epnn_mlp_forward(epnn_handle *h, int rows, int n_in, int n_out, const float *W1, const float *b1,
                                const float *W2, const float *b2, const float *W3, const float *b3, const float *x,
                                float *out) {
    if (!h || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !x || !out) EPNN_FAIL("epnn_mlp_forward: null argument");
    if (rows < 1 || n_in < 1 || n_out < 1) EPNN_FAIL("epnn_mlp_forward: rows, n_in and n_out must be positive");
    HIPCHK(hipSetDevice(h->device));
    if (h->pending.active && finish_forward(h)) return 1;
    const size_t nw = (size_t)n_in * 32 + 32 + 32 * 32 + 32 + (size_t)32 * n_out + n_out;
    const size_t nxs = (size_t)rows * n_in, no = (size_t)rows * n_out;
    if (h->s_misc.ensure((nw + nxs + no) * 4)) return 1;
    float *d = h->s_misc.as<float>();
    MlpArgs M{};
    size_t off = 0;
    auto up = [&](const float *src, size_t n) -> const float * {
        float *dst = d + off;
        (void)hipMemcpyAsync(dst, src, n * 4, hipMemcpyHostToDevice, h->stream);
        off += n;
        return dst;
    };
    M.W1 = up(W1, (size_t)n_in * 32);
    M.b1 = up(b1, 32);
    M.W2 = up(W2, 32 * 32);
    M.b2 = up(b2, 32);
    M.W3 = up(W3, (size_t)32 * n_out);
    M.b3 = up(b3, n_out);
    M.x = up(x, nxs);
    M.out = d + off;
    M.rows = rows;
    M.n_in = n_in;
    M.n_out = n_out;
    hipLaunchKernelGGL(k_mlp_forward, dim3((unsigned)((rows + 127) / 128)), dim3(256), 0, h->stream, M);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, M.out, no * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int epnn_mlp_forward_layers(epnn_handle *h, int rows, int n_layers, const int32_t *dims, const float *const *W,
                                       const float *const *b, const float *x, float *out) {
    if (!h || !dims || !W || !b || !x || !out) EPNN_FAIL("epnn_mlp_forward_layers: null argument");
    if (rows < 1 || n_layers < 1 || n_layers > EPNN_GMLP_LMAX) EPNN_FAIL("epnn_mlp_forward_layers: rows >= 1 and 1 .. %d Dense layers", EPNN_GMLP_LMAX);
    for (int l = 0; l <= n_layers; ++l)
        if (dims[l] < 1 || dims[l] > EPNN_GMLP_WMAX) EPNN_FAIL("epnn_mlp_forward_layers: width %d (1 .. %d are built)", dims[l], EPNN_GMLP_WMAX);
    for (int l = 0; l < n_layers; ++l)
        if (!W[l] || !b[l]) EPNN_FAIL("epnn_mlp_forward_layers: null kernel / bias of layer %d", l);
    HIPCHK(hipSetDevice(h->device));
    if (h->pending.active && finish_forward(h)) return 1;
    GenMlp G{};
    G.n = n_layers;
    size_t nw = 0;
    for (int l = 0; l <= n_layers; ++l) G.dims[l] = dims[l];
    for (int l = 0; l < n_layers; ++l) {
        G.offW[l] = (int)nw;
        nw += (size_t)dims[l] * dims[l + 1];
        G.offB[l] = (int)nw;
        nw += (size_t)dims[l + 1];
    }
    const size_t nxs = (size_t)rows * dims[0], no = (size_t)rows * dims[n_layers];
    if (h->s_misc.ensure((nw + nxs + no) * 4)) return 1;
    float *d = h->s_misc.as<float>();
    for (int l = 0; l < n_layers; ++l) {
        (void)hipMemcpyAsync(d + G.offW[l], W[l], (size_t)dims[l] * dims[l + 1] * 4, hipMemcpyHostToDevice, h->stream);
        (void)hipMemcpyAsync(d + G.offB[l], b[l], (size_t)dims[l + 1] * 4, hipMemcpyHostToDevice, h->stream);
    }
    (void)hipMemcpyAsync(d + nw, x, nxs * 4, hipMemcpyHostToDevice, h->stream);
    G.w = d;
    hipLaunchKernelGGL(k_mlp_generic, dim3((unsigned)((rows + EPNN_GMLP_ROWS - 1) / EPNN_GMLP_ROWS)), dim3(256), 0, h->stream, G, d + nw, d + nw + nxs, rows);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d + nw + nxs, no * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

#ifdef EPNN_STAMPS
// diagnostic build only: copy the per-wave phase stamps of the last fused launch
extern "C" int epnn_debug_stamps(epnn_handle *h, unsigned long long *out, size_t count) {
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->l_nm.p, count * 8, hipMemcpyDeviceToHost));
    return 0;
}
#endif

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
extern "C" int epnn_param_count(epnn_handle *h, int64_t *out) {
    if (!h || !out) EPNN_FAIL("epnn_param_count: null argument");
    TrainState ts;
    train_layout(h, &ts);
    *out = ts.P;
    return 0;
}
extern "C" int epnn_get_gradients(epnn_handle *h, float *out, int64_t count) {
    if (!h || !out) EPNN_FAIL("epnn_get_gradients: null argument");
    TrainState *ts = train_state(h);
    if (!ts->ready || count != ts->P) EPNN_FAIL("epnn_get_gradients: training not initialised or wrong count (%d parameters)", ts->P);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(out, ts->grad.p, (size_t)ts->P * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int epnn_set_gradients(epnn_handle *h, const float *in, int64_t count) {
    if (!h || !in) EPNN_FAIL("epnn_set_gradients: null argument");
    TrainState *ts = train_state(h);
    if (!ts->ready || count != ts->P) EPNN_FAIL("epnn_set_gradients: training not initialised or wrong count (%d parameters)", ts->P);
    HIPCHK(hipSetDevice(h->device));
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
    const bool adam_now = apply && rowfused && !(h->comm && h->comm_world > 1);
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
extern "C" int epnn_train_step_dense(epnn_handle *h, int B, int N, const float *h_inp, const float *e_inp, const float *x_inp,
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
extern "C" int epnn_train_step_xyz(epnn_handle *h, int B, int N, const int32_t *offsets, const float *xyz, const float *x,
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
