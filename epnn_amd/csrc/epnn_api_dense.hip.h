// The reference's dense (B,N,N,.) entries -- make_model call, GNN_layer.call, EPN_layer.call -- and MLP_layer.call.
// Part of the one translation unit epnn_api.hip.
#pragma once

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

// rows of d channels -> rows of EPNN_EDIM channels, the rest zero (a model with h_dim = e_dim = d < 48: epnn_host.h, model_dim)
static void pad_channels(const float *src, size_t rows, int d, std::vector<float> &dst) {
    dst.assign(rows * EPNN_EDIM, 0.f);
    for (size_t r = 0; r < rows; ++r) memcpy(dst.data() + r * EPNN_EDIM, src + r * (size_t)d, (size_t)d * sizeof(float));
}
static int dense_host_48(epnn_handle *h, int B, int N, int mode, const float *hh, const float *e, const float *x,
                         const float *q, const float *mask, float *out);
static int dense_host(epnn_handle *h, int B, int N, int mode, const float *hh, const float *e, const float *x,
                      const float *q, const float *mask, float *out) {
    if (!h || !hh || !e || !x || !q || !mask || !out) EPNN_FAIL("dense forward: null argument");
    if (h->model_dim == EPNN_EDIM) return dense_host_48(h, B, N, mode, hh, e, x, q, mask, out);
    if (B < 1 || N < 1) EPNN_FAIL("dense forward: B and N must be positive");
    // h_dim = e_dim below 48: the tensors' channels are padded with zeros on the way in, h_out is cut on the way out
    const int d = h->model_dim;
    const size_t pairs = (size_t)B * N * N, atoms = (size_t)B * N;
    std::vector<float> ph, pe, po;
    pad_channels(hh, mode == 0 ? pairs : atoms, d, ph);
    pad_channels(e, pairs, d, pe);
    if (mode != 1) return dense_host_48(h, B, N, mode, ph.data(), pe.data(), x, q, mask, out);
    po.resize(atoms * EPNN_EDIM);
    if (dense_host_48(h, B, N, mode, ph.data(), pe.data(), x, q, mask, po.data())) return 1;
    for (size_t r = 0; r < atoms; ++r) memcpy(out + r * (size_t)d, po.data() + r * EPNN_EDIM, (size_t)d * sizeof(float));
    return 0;
}
static int dense_host_48(epnn_handle *h, int B, int N, int mode, const float *hh, const float *e, const float *x,
                         const float *q, const float *mask, float *out) {
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
    if (h && h->model_dim != EPNN_EDIM)
        EPNN_FAIL("epnn_model_forward_dense_dev: device tensors are read as they are, with %d channels; a model with h_dim = %d goes through epnn_model_forward_dense", EPNN_EDIM, h->model_dim);
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
                                       const float *const *b, const float *x, float *out, int activation) {
    if (!h || !dims || !W || !b || !x || !out) EPNN_FAIL("epnn_mlp_forward_layers: null argument");
    if (activation < EPNN_ACT_RELU || activation > EPNN_ACT_SIGMOID) EPNN_FAIL("epnn_mlp_forward_layers: activation %d (0 relu, 1 linear, 2 tanh, 3 sigmoid are built)", activation);
    if (rows < 1 || n_layers < 1 || n_layers > EPNN_GMLP_LMAX) EPNN_FAIL("epnn_mlp_forward_layers: rows >= 1 and 1 .. %d Dense layers", EPNN_GMLP_LMAX);
    for (int l = 0; l <= n_layers; ++l)
        if (dims[l] < 1 || dims[l] > EPNN_GMLP_WMAX) EPNN_FAIL("epnn_mlp_forward_layers: width %d (1 .. %d are built)", dims[l], EPNN_GMLP_WMAX);
    for (int l = 0; l < n_layers; ++l)
        if (!W[l] || !b[l]) EPNN_FAIL("epnn_mlp_forward_layers: null kernel / bias of layer %d", l);
    HIPCHK(hipSetDevice(h->device));
    if (h->pending.active && finish_forward(h)) return 1;
    GenMlp G{};
    G.n = n_layers;
    G.act = activation;
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
