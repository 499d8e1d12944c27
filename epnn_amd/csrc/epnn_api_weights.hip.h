// Weights of a handle: the edge-feature basis of the fused kernels' front-end, set / get of the Keras kernels, and their re-layout into
// MFMA fragment order (pack_weights).  Part of the one translation unit epnn_api.hip (included there, in this order).
#pragma once

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

// A model whose h_dim = e_dim = d is below the kernels' 48 (epnn_host.h: model_dim): where the rows / columns of the MODEL's Dense
// layer (which, layer) sit in the padded layer the handle holds.  Inputs of a pair MLP are [x_i, h_i, q_i, x_j, h_j, q_j, e]
// (charge_gn.py:61-64, 103-107), of the update MLP [h, messages] (:71), its output is h (:73).
struct ModelMaps { std::vector<int> rows, cols; };
static void model_maps(const epnn_handle *h, int which, int layer, const HostDense &D, ModelMaps &M) {
    const int d = h->model_dim, nx = h->cfg.nx, F = nx + EPNN_EDIM + 1;
    M.rows.clear();
    M.cols.clear();
    const bool first = layer == 0;
    const bool last_upd = which == EPNN_W_UPD && layer == (h->upd_generic ? (int)h->updg.size() - 1 : 2);
    if (d != EPNN_EDIM && first && which != EPNN_W_UPD) {
        for (int side = 0; side < 2; ++side) {
            for (int k = 0; k < nx; ++k) M.rows.push_back(side * F + k);
            for (int k = 0; k < d; ++k) M.rows.push_back(side * F + nx + k);
            M.rows.push_back(side * F + nx + EPNN_EDIM);
        }
        for (int k = 0; k < d; ++k) M.rows.push_back(2 * F + k);
    } else if (d != EPNN_EDIM && first) {
        for (int k = 0; k < d; ++k) M.rows.push_back(k);
        for (int k = 0; k < D.n_in - EPNN_EDIM; ++k) M.rows.push_back(EPNN_EDIM + k);
    } else {
        for (int k = 0; k < D.n_in; ++k) M.rows.push_back(k);
    }
    const int nc = d != EPNN_EDIM && last_upd ? d : D.n_out;
    for (int k = 0; k < nc; ++k) M.cols.push_back(k);
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
    // One or two hidden layers of at most 32 units fit INSIDE the tuned kernels' 80 -> 32 -> 32 -> 48 update MLP exactly: missing units
    // are units with zero weights and zero bias (relu(0) = 0 feeds nothing), a missing second layer is the identity on the first
    // layer's outputs (they are >= 0 behind their ReLU, so relu(I u + 0) = u).  pack_weights builds that padded copy; such a model
    // runs every inference kernel of the [32, 32] model, not the generic update stage.
    h->upd_embed = h->upd_generic && n_hidden <= 2 && widths[0] <= H && (n_hidden == 1 || widths[1] <= H);
    // One or two hidden layers of at most 64 units: the same embedding into [64, 64], which the one-wavefront-per-molecule kernel
    // is also built for (k_wave_forward<.., NRU = 4>); molecules of more than 32 atoms take the tiled path's generic update stage.
    h->upd_wide = h->upd_generic && !h->upd_embed && n_hidden <= 2 && widths[0] <= 2 * H && (n_hidden == 1 || widths[1] <= 2 * H) && H == 32;
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
    ModelMaps M;
    model_maps(h, which, layer, *d, M);
    if (n_in) *n_in = (int)M.rows.size();
    if (n_out) *n_out = (int)M.cols.size();
    return 0;
}

extern "C" int epnn_set_weights(epnn_handle *h, int which, int t, int layer, const float *kernel, const float *bias) {
    HostDense *d = find_layer(h, which, t, layer);
    if (!d || !kernel || !bias) EPNN_FAIL("epnn_set_weights: bad (which=%d, t=%d, layer=%d) or null pointer", which, t, layer);
    if (h->train) { if (train_sync_to_host(h)) return 1; train_state(h)->ready = false; }   // masters are stale now
    if (h->model_dim == EPNN_EDIM) {
        memcpy(d->W.data(), kernel, d->W.size() * sizeof(float));
        memcpy(d->b.data(), bias, d->b.size() * sizeof(float));
    } else {                                      // the model's rows / columns into the zero-padded layer
        ModelMaps M;
        model_maps(h, which, layer, *d, M);
        std::fill(d->W.begin(), d->W.end(), 0.f);
        std::fill(d->b.begin(), d->b.end(), 0.f);
        const size_t nc = M.cols.size();
        for (size_t r = 0; r < M.rows.size(); ++r)
            for (size_t c = 0; c < nc; ++c) d->W[(size_t)M.rows[r] * d->n_out + M.cols[c]] = kernel[r * nc + c];
        for (size_t c = 0; c < nc; ++c) d->b[M.cols[c]] = bias[c];
    }
    h->weights_dirty = true;
    return 0;
}

extern "C" int epnn_get_weights(epnn_handle *h, int which, int t, int layer, float *kernel, float *bias) {
    HostDense *d = find_layer(h, which, t, layer);
    if (!d) EPNN_FAIL("epnn_get_weights: bad (which=%d, t=%d, layer=%d)", which, t, layer);
    if (train_sync_to_host(h)) return 1;
    if (h->model_dim == EPNN_EDIM) {
        if (kernel) memcpy(kernel, d->W.data(), d->W.size() * sizeof(float));
        if (bias) memcpy(bias, d->b.data(), d->b.size() * sizeof(float));
        return 0;
    }
    ModelMaps M;
    model_maps(h, which, layer, *d, M);
    const size_t nc = M.cols.size();
    if (kernel)
        for (size_t r = 0; r < M.rows.size(); ++r)
            for (size_t c = 0; c < nc; ++c) kernel[r * nc + c] = d->W[(size_t)M.rows[r] * d->n_out + M.cols[c]];
    if (bias)
        for (size_t c = 0; c < nc; ++c) bias[c] = d->b[M.cols[c]];
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
    if (h->upd_embed || h->upd_wide) {
        // the update MLP of `layers` = [w1] or [w1, w2] as a [HP, HP] one (epnn_set_update_layers): exact.  HP = 32: `upd`, every tuned
        // kernel; HP = 64: `updw`, the 64-unit variant of the one-wavefront-per-molecule kernel
        const int HP = h->upd_embed ? h->cfg.hidden : 2 * h->cfg.hidden, nh = (int)h->updg.size() - 1, w1 = h->updg[0].n_out, wl = h->updg[nh - 1].n_out;
        HostDense *dst = h->upd_embed ? h->upd : h->updw;
        const int dims[4] = {h->updg[0].n_in, HP, HP, h->cfg.h_dim};
        for (int l = 0; l < 3; ++l) {
            dst[l].n_in = dims[l];
            dst[l].n_out = dims[l + 1];
            dst[l].W.assign((size_t)dims[l] * dims[l + 1], 0.f);
            dst[l].b.assign(dims[l + 1], 0.f);
        }
        for (int i = 0; i < dst[0].n_in; ++i)
            for (int o = 0; o < w1; ++o) dst[0].W[(size_t)i * HP + o] = h->updg[0].W[(size_t)i * w1 + o];
        for (int o = 0; o < w1; ++o) dst[0].b[o] = h->updg[0].b[o];
        if (nh == 2) {
            for (int i = 0; i < w1; ++i)
                for (int o = 0; o < wl; ++o) dst[1].W[(size_t)i * HP + o] = h->updg[1].W[(size_t)i * wl + o];
            for (int o = 0; o < wl; ++o) dst[1].b[o] = h->updg[1].b[o];
        } else {
            for (int i = 0; i < w1; ++i) dst[1].W[(size_t)i * HP + i] = 1.f;
        }
        const HostDense &last = h->updg[nh];
        for (int i = 0; i < wl; ++i)
            for (int o = 0; o < last.n_out; ++o) dst[2].W[(size_t)i * last.n_out + o] = last.W[(size_t)i * last.n_out + o];
        for (int o = 0; o < last.n_out; ++o) dst[2].b[o] = last.b[o];
    }
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
        // (HU units in the update MLP's hidden layers: 32, or 64 for the k_wave_forward<.., NRU = 4> variant -- NRU = HU / 16 row
        //  blocks per layer, KU = HU / 4 K steps of nm * u2; the block-per-wavefront kernels only ever see HU = 32)
        WaveIndex &X = h->wvidx;
        const HostDense *ud = h->upd_wide ? h->updw : h->upd;
        const int HU = h->upd_wide ? 64 : 32, NRU = HU / 16, KU = HU / 4;
        const float *Wu1 = ud[0].W.data(), *Wu2 = ud[1].W.data(), *Wu3 = ud[2].W.data();
        const float *bu1 = ud[0].b.data(), *bu2 = ud[1].b.data(), *bu3 = ud[2].b.data();
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
        // a K = 32 kernel as three bf16 pieces per weight (epnn_wave.hip.h, w16_split3: truncation, exact), [2][3][64 lanes][4 dwords]:
        // lane (q, m) of row block rb: K slots s = 0..7 = input feature accf(s, q), output feature 16 rb + m; dword j = slots 2j | 2j+1 << 16
        // frag_bf3s: the K slot (lane group q, slot s) -> value mapping is the caller's (fn(q, s, out)); frag_bf3: acc order
        auto frag_bf3s = [&](auto &&fn) {
            const int off = alloc((size_t)2 * 3 * 64 * 4);
            for (int rb = 0; rb < 2; ++rb)
                for (int l = 0; l < 64; ++l) {
                    uint32_t pc[3][8];
                    for (int s = 0; s < 8; ++s) {
                        float v = (float)fn(l >> 4, s, 16 * rb + (l & 15));
                        for (int k = 0; k < 3; ++k) {
                            uint32_t bits;
                            memcpy(&bits, &v, 4);
                            bits &= 0xffff0000u;
                            float top;
                            memcpy(&top, &bits, 4);
                            pc[k][s] = bits >> 16;
                            v = v - top;
                        }
                    }
                    for (int k = 0; k < 3; ++k)
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t word = pc[k][2 * j] | pc[k][2 * j + 1] << 16;
                            memcpy(&buf[off + ((rb * 3 + k) * 64 + l) * 4 + j], &word, 4);
                        }
                }
            return off;
        };
        auto frag_bf3 = [&](auto &&fn) { return frag_bf3s([&](int q, int s, int out) { return fn(accf(s, q), out); }); };
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
        struct FoldedOff { int f32, hb, xb; };
        auto folded = [&](const float *W1, const float *b1, int r0) -> FoldedOff {        // KU acc steps (Wu3 M_h), then the xq steps
            std::vector<double> prod((size_t)HU * 32), cb(32);
            for (int k = 0; k < HU; ++k)
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
            FoldedOff o{0, 0, 0};
            o.f32 = frag(2, KU + EPNN_XS, [&](int s, int q, int m) -> double {
                if (s < KU) return prod[accf(s, q) * 32 + m];
                return xq_row(W1, b1, r0, 4 * (s - KU) + q, m, cb[m]);
            });
            if (HU == 32) {      // the same two blocks as bf16 pieces (the per-atom chains on the bf16 pipe; 32-unit update MLPs)
                o.hb = frag_bf3([&](int in, int out) { return prod[(size_t)in * 32 + out]; });
                o.xb = frag_bf3s([&](int q, int s, int out) -> double {
                    const int phi = wave_xq_slot(q, s, nx);
                    return phi < 0 ? 0.0 : xq_row(W1, b1, r0, phi, out, cb[out]);
                });
            }
            return o;
        };
        const bool have_basis = (int)h->edge_B.size() == EPNN_EDIM * EPNN_ER;
        // a K = 16 kernel as three bf16 pieces per weight for v_mfma_f32_16x16x16_bf16, [2][3][64 lanes][2 dwords]: lane (q, m) of row
        // block rb: K slots s = 0..3 = fn(q, s, 16 rb + m); dword j = slots 2j | 2j+1 << 16
        auto frag_bf3k16 = [&](auto &&fn) {
            const int off = alloc((size_t)2 * 3 * 64 * 2);
            for (int rb = 0; rb < 2; ++rb)
                for (int l = 0; l < 64; ++l) {
                    uint32_t pc[3][4];
                    for (int s = 0; s < 4; ++s) {
                        float v = (float)fn(l >> 4, s, 16 * rb + (l & 15));
                        for (int k = 0; k < 3; ++k) {
                            uint32_t bits;
                            memcpy(&bits, &v, 4);
                            bits &= 0xffff0000u;
                            float top;
                            memcpy(&top, &bits, 4);
                            pc[k][s] = bits >> 16;
                            v = v - top;
                        }
                    }
                    for (int k = 0; k < 3; ++k)
                        for (int j = 0; j < 2; ++j) {
                            const uint32_t word = pc[k][2 * j] | pc[k][2 * j + 1] << 16;
                            memcpy(&buf[off + ((rb * 3 + k) * 64 + l) * 2 + j], &word, 4);
                        }
                }
            return off;
        };
        auto pair_common = [&](HostDense (&mm)[3], int &we, int &we16, int &we16b, int &w2, int &b2) {
            const float *W1 = mm[0].W.data(), *W2 = mm[1].W.data(), *bb2 = mm[1].b.data();
            we = frag(2, 12, [&](int s, int q, int m) { return (double)W1[(size_t)(2 * F + 12 * q + s) * 32 + m]; });
            we16 = frag(2, EPNN_ER / 4, [&](int s, int q, int m) -> double {      // (B^T We)[4q + s][m]
                if (!have_basis) return 0.0;
                double a = 0;
                for (int ch = 0; ch < EPNN_EDIM; ++ch) a += h->edge_B[(size_t)ch * EPNN_ER + 4 * q + s] * (double)W1[(size_t)(2 * F + ch) * 32 + m];
                return a;
            });
            we16b = frag_bf3k16([&](int q, int s, int m) -> double {                // the same, K slot s of lane group q = coefficient 4q + s
                if (!have_basis) return 0.0;
                double a = 0;
                for (int ch = 0; ch < EPNN_EDIM; ++ch) a += h->edge_B[(size_t)ch * EPNN_ER + 4 * q + s] * (double)W1[(size_t)(2 * F + ch) * 32 + m];
                return a;
            });
            w2 = frag(2, 8, [&](int s, int q, int m) { return (double)W2[(size_t)accf(s, q) * 32 + m]; });
            b2 = vec(32, [&](int k) { return (double)bb2[k]; });
        };
        std::vector<double> pu1((size_t)HU * HU), cu3(HU);
        for (int k = 0; k < HU; ++k)
            for (int m = 0; m < HU; ++m) {
                double a = 0;
                for (int f = 0; f < EPNN_EDIM; ++f) a += (double)Wu3[(size_t)k * EPNN_EDIM + f] * (double)Wu1[(size_t)f * HU + m];
                pu1[(size_t)k * HU + m] = a;
            }
        for (int m = 0; m < HU; ++m) {
            double a = 0;
            for (int f = 0; f < EPNN_EDIM; ++f) a += (double)bu3[f] * (double)Wu1[(size_t)f * HU + m];
            cu3[m] = a;
        }
        const int off_pu1 = frag(NRU, KU, [&](int s, int q, int m) { return pu1[(size_t)accf(s, q) * HU + m]; });
        const int off_cu3 = vec(HU, [&](int k) { return cu3[k]; });
        const int off_u2 = frag(NRU, KU, [&](int s, int q, int m) { return (double)Wu2[(size_t)accf(s, q) * HU + m]; });
        const int off_u2b = HU == 32 ? frag_bf3([&](int in, int out) { return (double)Wu2[(size_t)in * HU + out]; }) : 0;
        const int off_pu1b = HU == 32 ? frag_bf3([&](int in, int out) { return pu1[(size_t)in * HU + out]; }) : 0;
        const int off_bu1 = vec(HU, [&](int k) { return (double)bu1[k]; });
        const int off_bu2 = vec(HU, [&](int k) { return (double)bu2[k]; });
        for (int t = 0; t < T; ++t) {
            WaveGnnPack &G = X.g[t];
            pair_common(h->msg[t], G.we, G.we16, G.we16b, G.w2, G.b2);
            {
                const float *W2m = h->msg[t][1].W.data();
                G.w2b = frag_bf3([&](int in, int out) { return (double)W2m[(size_t)in * 32 + out]; });
            }
            const float *W3 = h->msg[t][2].W.data(), *b3 = h->msg[t][2].b.data();
            std::vector<double> fold((size_t)32 * HU), cb3(HU);
            for (int k = 0; k < 32; ++k)
                for (int m = 0; m < HU; ++m) {
                    double a = 0;
                    for (int j = 0; j < 32; ++j) a += (double)W3[k * 32 + j] * (double)Wu1[(size_t)(EPNN_EDIM + j) * HU + m];
                    fold[(size_t)k * HU + m] = a;
                }
            for (int m = 0; m < HU; ++m) {
                double a = 0;
                for (int j = 0; j < 32; ++j) a += (double)b3[j] * (double)Wu1[(size_t)(EPNN_EDIM + j) * HU + m];
                cb3[m] = a;
            }
            G.u1s = frag(NRU, 8, [&](int s, int q, int m) { return fold[(size_t)accf(s, q) * HU + m]; });
            G.u1sb = G.u2b = G.pu1b = G.pwihb = G.pwixb = G.pwjhb = G.pwjxb = 0;
            if (HU == 32) {
                G.u1sb = frag_bf3([&](int in, int out) { return fold[(size_t)in * HU + out]; });
                G.u2b = off_u2b;
                G.pu1b = off_pu1b;
            }
            G.cb3 = vec(HU, [&](int k) { return cb3[k]; });
            G.bu1 = off_bu1;
            G.u2 = off_u2;
            G.bu2 = off_bu2;
            G.pu1 = off_pu1;
            G.cu3 = off_cu3;
            if (t + 1 < T) {
                const float *N1 = h->msg[t + 1][0].W.data(), *nb1 = h->msg[t + 1][0].b.data();
                const FoldedOff fi = folded(N1, nb1, 0), fj = folded(N1, nullptr, F);
                G.pwi = fi.f32; G.pwihb = fi.hb; G.pwixb = fi.xb;
                G.pwj = fj.f32; G.pwjhb = fj.hb; G.pwjxb = fj.xb;
            } else {
                G.pwi = G.pwj = 0;
            }
        }
        X.wi0 = unfolded(h->msg[0][0].W.data(), h->msg[0][0].b.data(), 0);
        X.wj0 = unfolded(h->msg[0][0].W.data(), nullptr, F);
        X.u1h0 = frag(NRU, 12, [&](int s, int q, int m) { return (double)Wu1[(size_t)accf(s, q) * HU + m]; });
        X.u3 = frag(3, KU, [&](int s, int q, int m) { return (double)Wu3[(size_t)accf(s, q) * EPNN_EDIM + m]; });
        X.bu3 = vec(48, [&](int k) { return (double)bu3[k]; });
        for (int t = 0; t < T; ++t) {
            WaveEpnPack &E = X.e[t];
            pair_common(h->pas[t], E.we, E.we16, E.we16b, E.w2, E.b2);
            {
                const float *W2p = h->pas[t][1].W.data();
                E.w2b = frag_bf3([&](int in, int out) { return (double)W2p[(size_t)in * 32 + out]; });
            }
            const float *W1 = h->pas[t][0].W.data(), *b1 = h->pas[t][0].b.data();
            E.w3 = vec(32, [&](int k) { return (double)h->pas[t][2].W[k]; });
            E.wi = unfolded(W1, b1, 0);
            E.wj = unfolded(W1, nullptr, F);
            const FoldedOff fi = folded(W1, b1, 0), fj = folded(W1, nullptr, F);
            E.wif = fi.f32; E.wifhb = fi.hb; E.wifxb = fi.xb;
            E.wjf = fj.f32; E.wjfhb = fj.hb; E.wjfxb = fj.xb;
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
