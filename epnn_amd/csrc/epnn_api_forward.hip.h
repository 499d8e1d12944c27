// Inference dispatch of the compact entry: the plan of a batch (which kernel family takes which molecule), the launches of the fused
// kernels, epnn_forward_xyz[_dev|_begin|_end], epnn_set_partition, epnn_edges.  Part of the one translation unit epnn_api.hip.
#pragma once

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
    allow_mid = allow_mid && h->opt_force_path == 0 && !upd_generic_stage(h);
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
        const bool small = !upd_tiled_only(h) && ((h->opt_force_path == 1) || (h->opt_force_path == 0 && n <= EPNN_SMALL_NMAX && wave_ok));
        if (upd_tiled_only(h) && h->opt_force_path == 1) EPNN_FAIL("forward: force_path=1 with an update MLP that does not fit [32, 32] (tiled path only)");
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
        // (Pipeline lanes with a side stream each, on the hardware queue next to the lane's own, were measured in round 5: the
        //  validation batch at eight lanes 187 -> 175 M atoms/s -- the GPU is full either way and the fork / join cost stream time)
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
    // (the side stream joins BEHIND the launch of the small molecules: until round 5 it joined in front of it, so that only the
    //  block-per-wavefront launch of a small batch ran beside the 33..64-atom molecules and the two launches of a batch of more
    //  than 1024 molecules, or with "wave2" > 17, one after the other)
    if (P.small_order.empty()) {
        if (side_mid) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
        return 0;
    }
    if (h->upd_wide && S.run_gnn) {              // update MLP of up to [64, 64] (the EPN stack alone has no update MLP)
        if (S.d_xyz) hipLaunchKernelGGL((k_wave_forward<true, true, true, 4>), grid, dim3(64), (size_t)lds, h->stream, A, X);
        else if (S.run_epn) hipLaunchKernelGGL((k_wave_forward<true, true, false, 4>), grid, dim3(64), (size_t)lds, h->stream, A, X);
        else hipLaunchKernelGGL((k_wave_forward<true, false, false, 4>), grid, dim3(64), (size_t)lds, h->stream, A, X);
    } else if (S.d_xyz) hipLaunchKernelGGL((k_wave_forward<true, true, true>), grid, dim3(64), (size_t)lds, h->stream, A, X);
    else if (S.run_gnn && S.run_epn) hipLaunchKernelGGL((k_wave_forward<true, true, false>), grid, dim3(64), (size_t)lds, h->stream, A, X);
    else if (S.run_gnn) hipLaunchKernelGGL((k_wave_forward<true, false, false>), grid, dim3(64), (size_t)lds, h->stream, A, X);
    else hipLaunchKernelGGL((k_wave_forward<false, true, false>), grid, dim3(64), (size_t)lds, h->stream, A, X);
    HIPCHK(hipGetLastError());
    if (side_mid) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
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
    // the count pass's decisions as a bit per candidate (epnn_frontend.hip.h): while a row's molecule has at most 4096 atoms and
    // the rows' words fit 64 MB
    int nmax = 0;
    for (int b = 0; b < P.B; ++b) nmax = std::max(nmax, P.offsets[b + 1] - P.offsets[b]);
    const int bw = (nmax + 63) / 64;
    if (h->opt_front_bits && bw <= 64 && (size_t)P.A * bw * 8 <= ((size_t)64 << 20)) {
        if (h->d_nearbits.ensure((size_t)P.A * bw * 8)) return 1;
        F.bits = h->d_nearbits.as<unsigned long long>();
        F.bits_w = bw;
    }
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
static int enqueue_forward_planned(epnn_handle *h, const float *d_xyz, const float *d_x, const float *d_Q, float *d_q, bool front_ok);
static int enqueue_forward_xyz(epnn_handle *h, int B, int N, const int32_t *offsets, const float *d_xyz,
                               const float *d_x, const float *d_Q, float *d_q) {
    HIPCHK(hipSetDevice(h->device));
    if (pack_weights(h)) return 1;
    const bool front_ok = wave_front_ok(h);
    if (build_plan(h, B, N, offsets, front_ok)) return 1;
    // from here to the first row exchange of a partitioned system a failure is reported to the other processes (comm_guard)
    if (large_exchanges_over_rccl(h, 1)) h->guard_pending = true;
    return comm_guard_exit(h, enqueue_forward_planned(h, d_xyz, d_x, d_Q, d_q, front_ok), "partitioned forward (row exchange)");
}
static int enqueue_forward_planned(epnn_handle *h, const float *d_xyz, const float *d_x, const float *d_Q, float *d_q, bool front_ok) {
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
    // stage events ("profile"): a boundary is recorded only where a stage with kernels of its own ends -- a timed event record
    // between two launches is ~5 us of stream time, and a forward of tiled molecules only has neither a separate front-end
    // nor fused kernels (its events used to read 11 us for those two empty stages)
    hipEvent_t *ev = nullptr;
    unsigned char *ev_has = nullptr;
    if (h->opt_profile > 0) {
        const int slot = h->ev_next % h->opt_profile;
        ev = h->evpool.data() + 4 * slot;
        if ((int)h->ev_recorded.size() < h->opt_profile) h->ev_recorded.resize(h->opt_profile, 0);
        ev_has = &h->ev_recorded[slot];
        *ev_has = 0;
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
        else {
            if (run_frontend_xyz(h, F)) return 1;
            if (ev) { HIPCHK(hipEventRecord(ev[1], h->stream)); *ev_has |= 1; }
        }
    }
    PairSource S;
    S.d_x = d_x;
    S.d_Q = d_Q;
    S.d_q = d_q;
    S.d_xyz = front_small ? d_xyz : nullptr;
    S.handoff = pure;
    if (launch_small(h, S)) return 1;
    if (ev && P.fused_count() > 0 && !P.large_list.empty()) { HIPCHK(hipEventRecord(ev[2], h->stream)); *ev_has |= 2; }
    if (ev && P.large_list.empty()) *ev_has |= 4;           // (no tiled stage: the fused stage runs to the forward's end)
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
            sched_yield();                       // ranks or Pipeline.map workers that share a core get it between two polls
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
    return edges_impl(h, n, xyz, h->model_dim, (double)h->cfg.cutoff, (double)h->cfg.eta, h->d_mu.as<double>(), e_out, nullptr);
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
