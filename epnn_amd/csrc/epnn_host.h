// Host-side state of one epnn_handle (one per device).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "../../include/epnn.h"
#include "epnn_common.h"

extern thread_local std::string g_epnn_err;

#define EPNN_FAIL(...)                                        \
    do {                                                      \
        char _buf[512];                                       \
        snprintf(_buf, sizeof(_buf), __VA_ARGS__);            \
        g_epnn_err = _buf;                                    \
        return 1;                                             \
    } while (0)

#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) EPNN_FAIL("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// growable device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) HIPCHK(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        HIPCHK(hipMalloc(&p, want));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct PinBuf {               // page-locked host memory: asynchronous copies really are asynchronous from / to it
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) HIPCHK(hipHostFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        HIPCHK(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct HostDense {            // one Keras Dense: kernel [in][out], bias [out]
    int n_in = 0, n_out = 0;
    std::vector<float> W, b;
};

struct Plan {                 // what the host derives from `offsets`
    int B = 0, A = 0, N = 0;
    std::vector<int> offsets;
    std::vector<int> small_order;   // molecules on the fused path, largest first
    // block-per-wavefront kernel (epnn_wave2.hip.h; compact entry, option "wave2"): molecules split over two wavefronts and
    // molecules of at most 16 atoms that share a workgroup in pairs, both largest first
    std::vector<int> split_order, single_order, split3_order, split4_order;    // (split3 / split4: 33..48 / 49..64 atoms over three / four
                                                                                // wavefronts, behind the pair entries)
    int pair_wgs = 0;               // that kernel's workgroups: split molecules + pairs of single ones (two wblk entries each,
                                    // behind the small_order entries; the three-block kernel's entries follow)
    size_t fused_count() const { return small_order.size() + split_order.size() + single_order.size() + split3_order.size() + split4_order.size(); }
    bool allow_mid = false;
    std::vector<int> large_list;    // molecules on the tiled path
    int small_nmax = 0;
    int pair_slots = 0;             // sum of n(n-1)/2 over the small molecules (capacity of the in-kernel front-end)
    bool valid = false;
};

struct epnn_handle {
    epnn_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;    // side stream of a lone handle (the launch of the 33..64-atom molecules beside the others'), forked / joined by
                                      // events; created at its first use
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    std::vector<hipEvent_t> evpool;   // 4 stage events per profiled forward ("profile" option = pool size)
    int ev_next = 0;                  // forwards recorded since the option was set
    std::vector<unsigned char> ev_recorded;   // per pool slot: bit 0 / 1 = the boundary behind the front-end / the fused kernels was recorded
    // weights
    HostDense msg[EPNN_MAXT][3], pas[EPNN_MAXT][3], upd[3];
    // make_model(layers != [32, 32]) / MLP_layer(nodes) as update_fn (charge_gn.py:369-371): the update MLP with any hidden widths
    // (epnn_set_update_layers).  Every molecule then takes the tiled path with one kernel per stage, whose update stage is the
    // generic Dense stack of epnn_mlp.hip.h; `upd` above keeps its standard shapes (zeros) so that the fragment packers run unchanged.
    std::vector<HostDense> updg;
    // make_model(layers, h_dim, ...) with h_dim (= the channels of e, charge_gn.py:376-377) below the 48 the kernels are built for:
    // the model runs as a 48-channel one whose extra channels of h and e are zero and whose extra kernel rows / columns are zero
    // (exact: they feed nothing and stay zero through every step, and their gradients are zero).  `cfg` holds the kernels' 48;
    // model_dim the caller's h_dim: the weight entries and the dense entries translate at the boundary (model_maps, pad_channels).
    int model_dim = EPNN_EDIM;
    bool upd_generic = false;
    bool upd_wide = false;            // ... of one or two hidden layers of <= 64 units (not embedded): k_wave_forward<.., NRU = 4> takes molecules of <= 32 atoms, `updw` holds the layers zero-padded to [64, 64]
    HostDense updw[3];
    bool upd_embed = false;           // ... of one or two hidden layers of <= 32 units: `upd` holds it zero-padded to [32, 32] (exact) and every tuned inference kernel runs
    DevBuf d_updgen;                  // raw kernels / biases of updg + the message MLPs' last Dense (W3_t, b3_t), see pack_weights
    GenMlp gen_upd{};                 // offsets into d_updgen
    int gen_w3[EPNN_MAXT] = {0}, gen_b3[EPNN_MAXT] = {0};
    bool weights_dirty = true;
    long weights_gen = 0;             // counts the changes of the weights (device-side copies made elsewhere compare it)
    void *infer_fused = nullptr;      // InferFused (epnn_train.hip.h)
    WeightIndex widx{};
    WaveIndex wvidx{};
    std::vector<double> edge_B;       // [48][EPNN_ER] orthonormal basis of the edge-feature family (epnn_api.hip edge_basis)
    double edge_res = 1.0;            // its residual max |e - B B^T e| over D in [0, cutoff]
    DevBuf d_wpack;
    DevBuf d_mu;
    DevBuf d_flip;                    // [EPNN_NFLIP_MAX] the near flag's flips beyond dsafe (epnn_create), padded with 1e300
    int nflip = 0;
    // plan + workspace
    Plan plan;
    DevBuf d_mu_ex;                   // Gaussian centres of an epnn_edges_ex call with its own num / cutoff
    DevBuf d_ctl;                     // the plan's index arrays, one upload: wblk [B] int4 | moff [B+1] | mflag [B] | molof [A]
    int4 *p_wblk = nullptr;
    int *p_moff = nullptr, *p_mflag = nullptr, *p_molof = nullptr;
    DevBuf d_rowcnt, d_rowoff, d_status, d_bsum;
    DevBuf d_pi, d_pj, d_psym, d_pe, d_pwi, d_pwj;
    DevBuf d_deg, d_incoff, d_nbr, d_desti, d_destj, d_prec;   // incidence rows of the pair list (epnn_frontend.hip.h)
    DevBuf d_nearbits;                // the count pass's D < cutoff decisions, a bit per candidate (FrontArgs::bits)
    int opt_front_inline = 1;         // developer switch: 0 = the prefix sums of the pair list always in a launch of their own
    int opt_front_bits = 1;           // developer switch: 0 = the fill pass measures every distance again
    int pcap = 0;
    int pair_cap_per_atom = 16;
    int *h_status = nullptr;      // pinned: [0] status bits, [1] total near pairs -- of the forward being enqueued: one of two slots of
    int *h_status_base = nullptr; //   this block (a repeated device-resident forward is enqueued before the one before it is checked)
    int st_slot = 0;
    hipEvent_t ev_done[2] = {nullptr, nullptr};                // end of the forward that used slot 0 / 1
    int opt_forward_ahead = 1;    // 1: epnn_forward_xyz_dev on the same batch and buffers as the call before enqueues first, checks the previous one after
    bool want_large_handoff = false, did_large_handoff = false;    // the tiled path's last kernel writes h_status itself
    // staging for the host-pointer entry points
    DevBuf s_xyz, s_misc, s_gx, s_pt;
    double dsafe = 0.0;               // distance up to which every pair is a near pair (epnn_create)
    DevBuf f_pw, d_etab;              // fused kernel's own front-end: near weights of its pairs; table of B^T e(D)
    bool last_front = false;          // the last forward used the in-kernel front-end (status words come from its last wave)
    bool ctl_clean = false;           // d_status is known to be all zero (left so by the last wave of the previous wave-front forward)
    int opt_dense_rowfused = 1;       // dense entry, small calls whose largest molecule fills > 55 % of N: the row-fused forward (a workgroup per atom slot)
    int opt_dense_small = 1;          // dense entry, one or a few molecules per call: the front-end as four launches instead of two memsets, seven kernels, a download
    int dn_gen = 1;                   //   ... whose flags are generation numbers (no memset per call)
    void *dn_flag_seen = nullptr;
    int opt_train_async = 1;          // training: a step returns when its forward is done (loss, predictions); backward + optimizer run on behind it
    int opt_train_inline = 1;         // training, coordinate entry: the inputs of a one-molecule step ride in the padding kernel's argument block (no upload)
    int opt_train_skip_padded = 1;    // training, coordinate entry, matrix-pipe kernels: padded atom slots leave their workgroups at once (the same bits)
    const int *tr_moff = nullptr, *tr_real = nullptr;      // ... set around the step by epnn_train_step_xyz
    int opt_train_split = 0;          // training, matrix-pipe backward: workgroups per atom (0 = as many as fit the atom's XCD, at most 6)
    int opt_train_fused = 1;          // training: 1 = row-fused pair-MLP kernels, Dense layers and weight gradients on the matrix pipe;
                                      // 0 = the layer-by-layer kernels
    int opt_train_graph = 1;          // training: 1 replays the step's launch sequence (optimizer step included) as a hipGraph: 0.22 vs 0.245 ms
                                      // per one-molecule step once the kernels were short enough for the launch boundaries to show
                                      // (round 2: 0.47 vs 0.45); a new (B, N, buffer set) means a new capture
    int opt_wave_front = 1;           // xyz entry, small molecules only: pair list built inside the wave kernel (no front-end kernels)
    int opt_wave2 = -1;               // compact entry, block-per-wavefront kernel (epnn_wave2.hip.h): molecules with at least this many
                                      // atoms (17..32) are split over two wavefronts, those of at most 16 run in pairs, the rest on
                                      // k_wave_forward; 0: k_wave_forward for all; -1 (default): 17 for a batch of at most
                                      // EPNN_W2_AUTO_MAX molecules -- a lone small batch leaves the GPU half empty and lasts as
                                      // long as its largest molecule, which the split halves (0.21 -> 0.12-0.18 ms) --, else 0.
                                      // Batches in flight side by side fill the GPU: there the split costs throughput (wavefront 1 of
                                      // a 17..20-atom molecule mostly waits: 175 instead of 214 M atoms/s), so engine.Pipeline sets 0.
    int opt_wave3 = 1;                // molecules of 33..48 / 49..64 atoms run on three / four wavefronts of the block-per-wavefront kernel
                                      // (0: the tiled kernels)
    bool wave2_attr = false, wave23_attr = false;
    int opt_wave_order = 0;           // order of a launch's wavefronts: 0 largest molecule first, 1 ends interleaved, 2 smallest first
    int opt_large_fused = 1;          // tiled path: one launch between two sweeps / pair passes (0: one kernel per stage)
    int opt_wave_prio = 18;           // fused kernel: molecules with >= this many atoms run at raised wave priority (0: off);
                                      // measured on the QM9-sized batch: 211 M atoms/s with 18 or 20, 206-208 M with 0 / 25 / 28
    int wave_lds = 20480;             // LDS bytes per wavefront of the wave-autonomous kernel (8 per CU)
    // large path workspace (epnn_large.hip.h)
    DevBuf l_a, l_P, l_R, l_zp, l_S0, l_corr, l_dl, l_tiles, l_csr_off, l_csr_ent, l_cnt, l_nm;
    DevBuf l_stasks, l_schunk, l_sfin, l_sfrac;
    DevBuf l_stasks2, l_sfrac2;                     // tile-workgroup sweep (k_lg_sweep2): tasks, correction-tile shares
    int l_nstasks2 = 0;
    int opt_large_sweep_old = 0;      // developer switch: every tiled molecule on the four-tile sweep kernel of rounds 1-4
    DevBuf l_Nn, l_Yb, l_qbuf, l_Pst, l_Rst;        // sweep operands (-R, b2 + W2^T R); EPN stack: charges, projections with q = 0
    DevBuf l_lmol, l_typrow, l_typtab, l_stype, l_typhash;     // first GNN step by atom types
    bool sweep_attr = false;                        // k_lg_sweep's dynamic LDS limit has been raised
    bool types_overflowed = false;                  // a molecule had more distinct feature rows than EPNN_TYPE_MAX: all-pairs sweep from now on
    int opt_large_dedupe = 1;         // tiled path, compact entry: first GNN step by atom types instead of the all-pairs sweep
    int opt_large_chunks = 0;         // developer switch: pieces the partner range of a tiled molecule's sweep is cut into (0: by size)
    int opt_large_merge = 1;          // compact entry: the pair-list launches are merged with the first projections / type sums (0: separate launches)
    // row-block partition of the all-pairs sweep over `part_world` processes (epnn_set_partition): this one runs the tile
    // groups [part_g0, part_g1) = atoms [part_row_lo, part_row_hi) and the callback completes S after every GNN step
    int part_rank = 0, part_world = 1;
    int part_row_lo = 0, part_row_hi = 0;
    epnn_exchange_fn part_exchange = nullptr;
    void *part_ctx = nullptr;
    std::vector<int> part_lo, part_hi;       // every process's row range (the plan is the same everywhere)
    int opt_part_collective = 0;             // developer switch: run the communicator's collectives (row exchange, gradient all-reduce, status guard) even at world size 1
    int opt_comm_guard = 1;                  // a status word is all-reduced (max) in front of every payload collective: see comm_guard below
    int opt_comm_inject_fail = 0;            // developer switch (tests): this rank reports a failure at its next status guard
    bool guard_pending = false;              // an entry point that will reach a collective has been entered and its status guard has not run yet
    DevBuf d_guard;
    // RCCL communicator (epnn_comm_init): gradient all-reduce of the train step, row exchange of a partitioned system
    ncclComm_t comm = nullptr;
    int comm_world = 1, comm_rank = 0;
    int l_natiles = 0, l_nstasks = 0, l_maxchunk = 0;
    // options / stats
    int opt_profile = 0, opt_force_path = 0;
    int opt_sync_spin_us = 2000;      // finish_forward / epnn_sync: poll the stream this long before sleeping on its completion (0: sleep at once)
    float timing[4] = {0, 0, 0, 0};
    int64_t stats[4] = {0, 0, 0, 0};
    // page-locked staging: the plan's index arrays with, behind them, the inputs of the host entry (one upload per forward;
    // reused once ev_ctl says the previous upload has run); the charges of the asynchronous host entry
    PinBuf pin_ctl, pin_out, pin_neff;
#ifdef EPNN_LG_CLOCKS
    DevBuf lg_clk;                    // development build: phase clocks of the tiled path's tail / EPN-step launches
#endif
    PinBuf pin_train, pin_tout;       // inputs of epnn_train_step_xyz / of a small dense call (one upload); loss terms + predictions of a
                                      // train step / charges of a small dense call
    DevBuf s_train;
    hipEvent_t ev_ctl = nullptr;
    bool ctl_uploading = false;
    struct HostCall {
        bool active = false;          // a begun forward has not been collected yet
        int A = 0;
    } hostcall;
    // deferred overflow handling for the asynchronous entry point
    struct Pending {
        bool active = false;
        std::function<int()> redo;    // re-enqueues the same forward after a capacity regrow
        int slot = 0;                 // its status slot / completion event
        const void *key[4] = {nullptr, nullptr, nullptr, nullptr};      // the device buffers of a device-resident compact forward
    } pending;
    // dense front-end workspace (epnn_dense.hip.h)
    DevBuf dn_den;
    DevBuf dn_xs, dn_hs, dn_qs, dn_nms, dn_flag, dn_neff, dn_xf, dn_hf, dn_qf, dn_nmf, dn_out, tr_realbuf;
    DevBuf sd_h, sd_e, sd_x, sd_q, sd_mask, sd_out;
    std::vector<int> dn_neff_host;
    void *train = nullptr;            // TrainState (epnn_train.hip.h)
};

// ---- fail-closed collectives.  RCCL has no timeout: a rank that leaves an entry point with an error BEFORE a collective its peers
// have already enqueued leaves them blocked for good.  Every payload collective of this library (the gradient all-reduce of a train
// step, the row exchange of a partitioned system) is therefore preceded by a 4-byte ncclAllReduce(max) of a status word on the same
// stream, read back before the payload is enqueued: if ANY rank reports a failure, every rank returns non-zero ("... aborted on every
// rank") and nobody enqueues the payload.  A rank that fails earlier -- argument checks, allocations, a launch error -- still joins
// that status collective from its entry point's exit path (epnn_handle::guard_pending says it owes one), so its peers are released.
// What this cannot cover is a rank whose GPU is gone (the guard itself then fails): rendezvous.launch_ranks stops the survivors.
// the model's update MLP needs the generic update stage (tiled kernels, one launch per stage)
// an update MLP that is neither [32, 32] nor embedded in it: the tiled path runs its generic update stage, the block-per-wavefront
// kernels are not built for it; the one-wavefront-per-molecule kernel has a 64-unit variant (upd_wide), without it everything is tiled
static inline bool upd_generic_stage(const epnn_handle *h) { return h->upd_generic && !h->upd_embed; }
static inline bool upd_tiled_only(const epnn_handle *h) { return h->upd_generic && !h->upd_embed && !h->upd_wide; }
static inline bool comm_collectives(const epnn_handle *h) { return h->comm && (h->comm_world > 1 || h->opt_part_collective); }
static inline int comm_guard(epnn_handle *h, int local_fail, const char *what) {
    h->guard_pending = false;
    if (!comm_collectives(h) || !h->opt_comm_guard) return local_fail ? 1 : 0;
    if (h->opt_comm_inject_fail) { local_fail = 1; h->opt_comm_inject_fail = 0; g_epnn_err = std::string(what) + ": injected failure (developer switch comm_inject_fail)"; }
    const std::string why = local_fail ? g_epnn_err : std::string();
    int *word = h->h_status_base + 8;                      // page-locked (epnn_create allocates 12 ints)
    *word = local_fail ? 1 : 0;
    bool ok = h->d_guard.ensure(sizeof(int)) == 0;
    ok = ok && hipMemcpyAsync(h->d_guard.p, word, sizeof(int), hipMemcpyHostToDevice, h->stream) == hipSuccess;
    ok = ok && ncclAllReduce(h->d_guard.p, h->d_guard.p, 1, ncclInt32, ncclMax, h->comm, h->stream) == ncclSuccess;
    ok = ok && hipMemcpyAsync(word, h->d_guard.p, sizeof(int), hipMemcpyDeviceToHost, h->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(h->stream) == hipSuccess;
    if (!ok) EPNN_FAIL("%s: the status collective itself failed on this rank%s%s", what, local_fail ? " after: " : "", why.c_str());
    if (local_fail) EPNN_FAIL("%s aborted on every rank; this rank failed: %s", what, why.c_str());
    if (*word != 0) EPNN_FAIL("%s aborted on every rank: another rank reported a failure before the collective", what);
    return 0;
}
// exit path of an entry point that owes its peers a status collective (it failed before reaching the payload collective)
static inline int comm_guard_exit(epnn_handle *h, int rc, const char *what) {
    if (rc && h->guard_pending) (void)comm_guard(h, 1, what);
    h->guard_pending = false;
    return rc;
}
