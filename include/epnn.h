/*
 * epnn.h -- C ABI of libepnn_hip.so, the MI355X (gfx950) implementation of the EPNN hot path.
 *
 * The reference (derekmetcalf/epnn) has no FFI: its "operator API" for this path is the Python/Keras layer
 * interface of charge_gn.py.  Each entry point below names the reference interface it stands in for; the
 * Python mirror of that interface (epnn_amd/charge_gn.py) binds these symbols with ctypes and nothing else.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; epnn_last_error() gives the message of the
 *     last failing call on the calling thread.
 *   - all tensors are float32, row-major, caller-owned.  "host" entry points take host pointers and copy;
 *     "_dev" entry points take pointers obtained from epnn_dev_alloc() on the same handle and run
 *     asynchronously on the handle's stream (epnn_sync() waits).
 *   - one handle per device; calls on one handle must be serialised by the caller.
 *   - flat ("ragged") molecule batches: B molecules, molecule b owns atoms [offsets[b], offsets[b+1]) of the
 *     flat atom arrays; N is the padded atom count the reference would have used (gen_padded_init_state pads
 *     every molecule to the directory maximum, charge_gn.py:340-364).  N enters the arithmetic: the reference
 *     sums messages over all N partners including the padded ones (charge_gn.py:70).
 *   - arithmetic: float32-grade throughout, like the reference's (TensorFlow float32).  The inference kernels run the Dense layers
 *     of the pair and update MLPs on the bf16 matrix pipe as six bf16 products of EXACT three-piece splits of both operands per product (what is
 *     left out is below 2^-24 of a product, the rounding of an f32 multiply-add); results agree with a
 *     float64 evaluation of the reference's algorithm to float32 rounding noise (1e-7 .. 1e-6 on the charges).  Every sum has a
 *     fixed order: results are bit-reproducible and independent of batch composition and of sharding.
 */
#ifndef EPNN_H
#define EPNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct epnn_handle epnn_handle;

/* Hyper-parameters fixed at make_model time (charge_gn.py:369-374, 413-418) and the featuriser constants of
 * get_init_edges (charge_gn.py:122: cutoff=3.0, eta=2.0, mu=linspace(0.1,cutoff,e_dim)); near_tol is the
 * 1e-5 of EPN_layer.call (charge_gn.py:90). */
typedef struct epnn_config {
    int32_t nx;       /* atom feature columns: 9 (infer.py table) or 10 (charge_gn.py table)          */
    int32_t h_dim;    /* channels of h, 1..48 (48 in the reference's scripts; must equal e_dim, charge_gn.py:377) */
    int32_t e_dim;    /* channels of e = Gaussians of get_init_edges, == h_dim                         */
    int32_t T;        /* message / electron passing steps, 1..8                                        */
    int32_t hidden;   /* 32: hidden width of every MLP and the message width (charge_gn.py:52,84,415)  */
    float cutoff;     /* 3.0                                                                           */
    float eta;        /* 2.0                                                                           */
    float near_tol;   /* 1e-5                                                                          */
} epnn_config;

enum { EPNN_W_MSG = 0, EPNN_W_UPD = 1, EPNN_W_PAS = 2 };

const char *epnn_last_error(void);
int epnn_version(void);
/* number of HIP devices visible (0 when there is none; never initialises a context beyond the count). */
int epnn_device_count(void);

/* make_model (charge_gn.py:369-391): creates the layer stack on `device`.  Weights start at zero.
 *
 * CONTRACT -- what of the reference's constructor arguments is free and what is fixed.  In the reference `layers` sizes the UPDATE
 * MLP only (make_model: MLP_layer(layers, out_dim=h_dim), charge_gn.py:371); the message and pass MLPs are ([32, 32], 32) and
 * ([32, 32], 1) by its own constants (:52, :84), and every width follows h_dim (:369-374).  Every checkpoint it ships and both of
 * its scripts use layers = [32, 32], h_dim = e_dim = 48 (:413-417, infer.py:47-50).
 *   free :  nx in 1..10 (atom feature columns; 9 and 10 are the reference's two tables), T in 1..8, cutoff, eta, near_tol,
 *           h_dim = e_dim in 1..48 (below 48 the model runs as a 48-channel one: zero channels of h and e, zero rows / columns of
 *           the kernels that touch them -- exact, the padding feeds nothing, stays zero through every step and has zero gradient;
 *           every weight, tensor and epnn_edges row at this interface has the MODEL's h_dim channels),
 *           the padded size N and the batch size of every call, every weight value; `layers` of the update MLP (1..7 hidden
 *           layers of 1..256 units each: epnn_set_update_layers) for every inference entry and the training step;
 *   fixed:  hidden == 32 (the reference's own constant for the message / pass MLPs), h_dim == e_dim (make_model gives e_inp h_dim
 *           channels, charge_gn.py:377) and at most 48 (the kernels' register / LDS layouts hold 48 channels).
 * epnn_create FAILS (returns non-zero, epnn_last_error says which field) for any other value.  `layers` == [32, 32] runs the
 * kernels DESIGN.md describes, and so does every `layers` of one or two hidden layers of at most 32 units: the library runs it as a
 * [32, 32] model on a zero-padded copy of its weights (units with zero weights and bias feed nothing; a missing second layer is the
 * identity on the first layer's non-negative outputs) -- exact, not an approximation.  One or two hidden layers of at most 64
 * units run molecules of up to 32 atoms on a 64-unit build of the fused kernel (the same embedding into [64, 64]; 181 M atoms/s on
 * the bench batch against 274 M for [32, 32]) and larger ones through the tiled kernels with one launch per stage and a generic
 * (f32 FMA) Dense stack as the update stage; any other `layers` (a width above 64, three or more hidden layers) runs every molecule
 * that way (28 M atoms/s on the bench batch: tools/bench_layers.py).  The training step of any `layers` but [32, 32] runs one launch per Dense layer ("train_fused" = 0's kernels)
 * on the model's own shapes -- the same results to float32 rounding, several times slower per small molecule. */
int epnn_create(const epnn_config *cfg, int device, epnn_handle **out);
int epnn_destroy(epnn_handle *h);
/* Leaves out `n` of the process's hardware queues: the HIP runtime deals a process's streams onto its hardware queues in the order
 * they are created (every handle owns one stream), and a pipeline of several handles runs faster with its lanes on every other
 * queue (engine.Pipeline calls this between two handles; no counterpart in the reference, which runs one model call at a time). */
int epnn_skip_hw_queues(int device, int n);

/* model.load_weights / layer.set_weights (infer.py:57): one Dense layer of one MLP.
 * which = EPNN_W_MSG (message_fns[t], charge_gn.py:52), EPNN_W_UPD (update_fn, t ignored, charge_gn.py:371),
 * EPNN_W_PAS (pass_fns[t], charge_gn.py:84); layer = 0..2 (EPNN_W_UPD: 0..n_hidden after epnn_set_update_layers); kernel is Keras
 * layout [in][out]. */
int epnn_set_weights(epnn_handle *h, int which, int t, int layer, const float *kernel, const float *bias);
/* model.trainable_variables / save_weights (charge_gn.py:462). */
int epnn_get_weights(epnn_handle *h, int which, int t, int layer, float *kernel, float *bias);
int epnn_weight_shape(epnn_handle *h, int which, int t, int layer, int32_t *n_in, int32_t *n_out);
/* make_model(layers, ...) / GNN_layer(message_fn, update_fn = MLP_layer(layers, out_dim = h_dim), T) (charge_gn.py:369-371): the
 * hidden widths of the update MLP, widths[n_hidden].  The default is {32, 32}.  Afterwards EPNN_W_UPD has n_hidden + 1 layers
 * (layer 0: h_dim + 32 -> widths[0]; the last: widths[n_hidden - 1] -> h_dim), all zero until epnn_set_weights fills them.
 * Fails on a handle that already holds training state (call it before epnn_train_init). */
int epnn_set_update_layers(epnn_handle *h, int n_hidden, const int32_t *widths);

/* get_init_edges (charge_gn.py:122-163): xyz[n][3] float32 -> e[n][n][e_dim] float32, host pointers. */
int epnn_edges(epnn_handle *h, int n, const float *xyz, float *e_out);
/* The same with the reference function's own parameters: num channels (charge_gn.py:122; mu = linspace(0.1, cutoff, num)),
 * cutoff and eta (the constants 3.0 and 2.0 of charge_gn.py:148-161), and, when c_out is not NULL, the cutoff weights
 * C[n][n] (float64) that get_init_edges returns tiled over the channels (charge_gn.py:163). */
int epnn_edges_ex(epnn_handle *h, int n, const float *xyz, int num, double cutoff, double eta, float *e_out, double *c_out);

/* Compact entry == gen_padded_init_state featurisation (charge_gn.py:292-366) + model([h,e,x,q,mask])
 * (charge_gn.py:369-391, infer.py:32-35) without materialising the dense (N,N,.) tensors:
 * xyz[A][3], x[A][nx] (Z + one-hot), Q[B] total charges -> q_out[A] predicted charges of the real atoms
 * (padded atoms are 0 in the reference output and are not stored).  Host pointers. */
int epnn_forward_xyz(epnn_handle *h, int B, int N, const int32_t *offsets, const float *xyz, const float *x,
                     const float *Q, float *q_out);
/* The same call in two halves (the loop of infer.py:62-76 with several batches in flight): _begin copies the host arrays
 * into page-locked staging owned by the handle (the caller may reuse them at once), queues uploads, kernels and the
 * download of the charges, and returns without waiting for the GPU; _end waits and writes q_out[A].  One forward per
 * handle between _begin and _end; use several handles to overlap batches (epnn_amd.engine.Pipeline.map). */
int epnn_forward_xyz_begin(epnn_handle *h, int B, int N, const int32_t *offsets, const float *xyz, const float *x,
                           const float *Q);
int epnn_forward_xyz_end(epnn_handle *h, float *q_out);
/* One large system on several GPUs (one process each, same inputs everywhere): the all-pairs sum of GNN_layer
 * (charge_gn.py:70), which is all of the cost of a system with thousands of atoms, is split by rows of atoms.  After every
 * GNN step the library calls `exchange(ctx, d_rows, row_len, n_rows, row_lo, row_hi)`: d_rows is a device array
 * [n_rows][row_len] float of which this process has filled rows row_lo..row_hi-1; the function must fill in the rows the
 * other processes own (an all-gather; epnn_memcpy_d2h / epnn_memcpy_h2d move rows) and return 0.  Molecules of at most 32
 * atoms are not partitioned.  Results are bit-identical to the unpartitioned run.
 * With exchange == NULL the handle's RCCL communicator (epnn_comm_init, same world size and rank) does it: every process
 * broadcasts its own rows in place, all of them grouped into one RCCL operation on the handle's stream -- no host
 * synchronisation inside the forward (one GPU per process; this is the multi-GPU form, the callback is the portable one). */
typedef int (*epnn_exchange_fn)(void *ctx, float *d_rows, int row_len, int n_rows, int row_lo, int row_hi);
int epnn_set_partition(epnn_handle *h, int rank, int world, epnn_exchange_fn exchange, void *ctx);
/* Same with device-resident xyz/x/Q/q_out (offsets stay on the host); asynchronous. */
int epnn_forward_xyz_dev(epnn_handle *h, int B, int N, const int32_t *offsets, const float *d_xyz,
                         const float *d_x, const float *d_Q, float *d_q_out);

/* Literal make_model call (charge_gn.py:376-389): h_inp/e_inp [B][N][N][h_dim], x_inp [B][N][N][nx],
 * q_inp/mask_inp [B][N][N][1] -> q_out [B][N][1].  Host pointers. */
int epnn_model_forward_dense(epnn_handle *h, int B, int N, const float *h_inp, const float *e_inp,
                             const float *x_inp, const float *q_inp, const float *mask_inp, float *q_out);
int epnn_model_forward_dense_dev(epnn_handle *h, int B, int N, const float *d_h_inp, const float *d_e_inp,
                                 const float *d_x_inp, const float *d_q_inp, const float *d_mask_inp,
                                 float *d_q_out);
/* GNN_layer.call (charge_gn.py:57-75): h[B][N][h_dim], e[B][N][N][e_dim], x[B][N][nx], q[B][N][1],
 * mask[B][N][N][1] -> h_out[B][N][h_dim].  Host pointers. */
int epnn_gnn_forward(epnn_handle *h, int B, int N, const float *hin, const float *e, const float *x,
                     const float *q, const float *mask, float *h_out);
/* EPN_layer.call (charge_gn.py:88-119): same inputs -> q_out[B][N][1].  Host pointers. */
int epnn_epn_forward(epnn_handle *h, int B, int N, const float *hin, const float *e, const float *x,
                     const float *q, const float *mask, float *q_out);

/* MLP_layer(nodes, out_dim, activation).call (charge_gn.py:31-45) for ANY `nodes`: n_layers Dense layers, dims[n_layers + 1] = n_in,
 * nodes..., out_dim (each 1..256), W[l] in Keras layout [dims[l]][dims[l + 1]], b[l][dims[l + 1]].  `activation` follows every layer
 * but the last (charge_gn.py:38-39: Dense(n, activation=activation) ..., Dense(out_dim, activation=None)): 0 = 'relu' (the
 * reference's default and only use), 1 = None / 'linear', 2 = 'tanh', 3 = 'sigmoid' (Keras' definitions); anything else fails.
 * x[rows][dims[0]] -> out[rows][dims[n_layers]].  Host pointers. */
int epnn_mlp_forward_layers(epnn_handle *h, int rows, int n_layers, const int32_t *dims, const float *const *W,
                            const float *const *b, const float *x, float *out, int activation);
/* MLP_layer.call (charge_gn.py:41-45) as a stand-alone operator: x[rows][n_in] -> relu 32 -> relu 32 -> out[rows][n_out];
 * kernels in Keras layout [in][out].  Host pointers. */
int epnn_mlp_forward(epnn_handle *h, int rows, int n_in, int n_out, const float *W1, const float *b1,
                     const float *W2, const float *b2, const float *W3, const float *b3, const float *x, float *out);

/* ---- training (charge_gn.py:393-402, 419): master weights, gradients and Adam moments live on the device as flat
 * vectors in model.trainable_variables order (update MLP, message MLPs t=0.., pass MLPs t=0..; kernel then bias).
 * epnn_train_init copies the current weights to the device and zeroes the Adam state (Keras-2 defaults are
 * lr 1e-3, beta1 0.9, beta2 0.999, eps 1e-7). */
int epnn_train_init(epnn_handle *h, float lr, float beta1, float beta2, float eps);
int epnn_param_count(epnn_handle *h, int64_t *out);
/* train_step on the literal make_model inputs (B,N,N,.), y and pred_out (B,N,1): loss = sum (y-p)^2, gradient of the
 * summed loss; apply != 0: all-reduce the gradient over the attached communicator (if any) and take one Adam step. */
int epnn_train_step_dense(epnn_handle *h, int B, int N, const float *h_inp, const float *e_inp, const float *x_inp,
                          const float *q_inp, const float *mask_inp, const float *y, float *pred_out, float *loss_out,
                          int apply);
/* same from a flat coordinate batch (y_flat, q_out_flat per real atom) */
int epnn_train_step_xyz(epnn_handle *h, int B, int N, const int32_t *offsets, const float *xyz, const float *x,
                        const float *Q, const float *y_flat, float *q_out_flat, float *loss_out, int apply);
int epnn_get_gradients(epnn_handle *h, float *out, int64_t count);
int epnn_set_gradients(epnn_handle *h, const float *in, int64_t count);
int epnn_train_apply(epnn_handle *h);
/* RCCL communicator (one rank per GPU): the gradient is summed with ONE ncclAllReduce of the flat vector; the same
 * communicator carries the row exchange of a partitioned large system (epnn_set_partition with exchange == NULL). */
int epnn_comm_unique_id(char *out128);
int epnn_comm_init(epnn_handle *h, const char *id128, int rank, int world);
/* ranks that joined the communicator (ncclCommCount) */
int epnn_comm_count(epnn_handle *h, int32_t *ranks_out);
/* all-reduce of n <= 1024 host doubles over the communicator, on the handle's stream, waited for (op 0 = sum, 1 = max): the
 * barrier and the MAX-over-ranks timing of a multi-process driver go through the same RCCL path as the product's collectives
 * (the reference has nothing distributed; bench.py --gpus N is the caller) */
int epnn_comm_allreduce(epnn_handle *h, double *inout, int32_t n, int32_t op);

/* (tests) the pair list of the last forward that built one outside the fused kernel: first / second atom and near weight
 * (is_near of charge_gn.py:90-94 as 1.0 / 0.0, times the mask for dense inputs) of up to `cap` pairs; *count_out = pairs listed */
int epnn_debug_pairs(epnn_handle *h, int32_t *pi, int32_t *pj, float *pwi, int64_t cap, int64_t *count_out);

/* Device memory and stream plumbing for callers that keep inputs resident (bench.py). */
int epnn_dev_alloc(epnn_handle *h, size_t bytes, void **out);
int epnn_dev_free(epnn_handle *h, void *p);
int epnn_memcpy_h2d(epnn_handle *h, void *dst, const void *src, size_t bytes);
int epnn_memcpy_d2h(epnn_handle *h, void *dst, const void *src, size_t bytes);
int epnn_sync(epnn_handle *h);

/* hipEvent timing on the handle's stream: begin/end bracket any number of calls; elapsed in ms.
 * epnn_last_timing: per-stage device times of the most recent forward when profiling is enabled with
 * epnn_set_option("profile", k): out[0]=front-end, out[1]=fused small-molecule kernel, out[2]=tiled
 * large-system kernels, out[3]=total. (infer.py:70-79 prints wall-clock; this is the device-side view.) */
int epnn_timer_begin(epnn_handle *h);
int epnn_timer_end(epnn_handle *h, float *elapsed_ms);
int epnn_last_timing(epnn_handle *h, float *out4);
/* same for the idx-th forward issued since "profile" was set (pool of that many event sets; no sync in between). */
int epnn_timing_at(epnn_handle *h, int idx, float *out4);
/* Options a caller may want to touch (every other name epnn_set_option accepts is a developer switch that selects between
 * implementations with identical results: include/epnn_dev.h).  Unknown names and out-of-range values fail.
 *   "profile"           0 = off (default); k > 0: keep the stage events of the last k forwards (epnn_timing_at)
 *   "force_path"        0 = by molecule size (default); 1 = fused small-molecule kernels only (fails above 32 atoms); 2 = tiled kernels only
 *   "pair_cap_per_atom" initial capacity of the near-pair list of the tiled / dense paths (it grows by itself; default 16)
 *   "wave2"             how batches of small molecules use the block-per-wavefront kernel: -1 (default) = molecules of 17..32 atoms
 *                       on two wavefronts and smaller ones two to a workgroup when the batch has at most 1024 molecules (halves the
 *                       latency of a lone batch); 0 = one wavefront per molecule throughout (set on handles whose launches overlap:
 *                       engine.Pipeline does); 17..32 = split from that many atoms
 *   "wave3"             1 (default) = molecules of 33..48 / 49..64 atoms run on three / four wavefronts of the fused kernel; 0 = on the
 *                       tiled kernels
 *   "sync_spin_us"      epnn_sync and every call that waits for a forward poll the stream this many microseconds (yielding the core
 *                       between polls) before they sleep on its completion; default 2000, 0 = sleep at once
 *   "large_dedupe"      1 (default) = the tiled path's first GNN step of the compact entry groups the atoms by feature row (h = 0 and one
 *                       q per molecule there: the all-pairs sum of charge_gn.py:70 takes (distinct rows)^2 pair evaluations instead of
 *                       n^2; a molecule with more than 64 distinct rows switches the handle back by itself); 0 = always the all-pairs sweep
 *   "train_fused"       1 (default) = train step with one workgroup per atom and pair MLP, Dense layers and weight gradients as f32
 *                       MFMA tiles, 2T + 2T + 1 launches; 0 = one launch per Dense layer on materialised rows (also taken above 96
 *                       atoms and for update layers other than [32, 32])
 *   "train_async"       1 (default) = a training step returns as soon as its forward pass is done (loss and predictions are on the host
 *                       then; backward and optimizer keep running, every call that reads weights or gradients waits for them); 0 = a
 *                       step returns when all of it is done
 *   "train_graph"       1 (default) = a train step that waits for its own end has its launch sequence captured once per (B, N, buffers)
 *                       and replayed as a hipGraph; 0 = kernel by kernel */
int epnn_set_option(epnn_handle *h, const char *name, int value);
/* The fused kernel's own front-end runs its G products in a 16-dimensional basis of the Gaussian edge features
 * (charge_gn.py:148-161: 48 overlapping bumps of one variable).  Returns max |e - B B^T e| over D in [0, cutoff], relative
 * to max e = 1 (5e-10 for cutoff 3, eta 2); the basis is used only when this is below 1e-8. */
double epnn_edge_basis_residual(epnn_handle *h);
/* counters of the most recent forward: out[0]=listed pairs (unordered, D < cutoff; the is_near test of charge_gn.py:90-94
 * enters as the pairs' weights), out[1]=molecules on the fused path,
 * out[2]=molecules on the tiled path, out[3]=pair-list regrows. */
int epnn_last_stats(epnn_handle *h, int64_t *out4);

#ifdef __cplusplus
}
#endif
#endif /* EPNN_H */
