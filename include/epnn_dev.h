/*
 * epnn_dev.h -- developer switches of libepnn_hip.so.  No symbols are declared here: these are further NAMES accepted by
 * epnn_set_option (include/epnn.h).  Each selects between two implementations of the same arithmetic -- the results are the same
 * bits unless a line says otherwise -- and exists for the measurements in HISTORY.md / profiles/ and for tests that compare the
 * implementations.  A product caller never needs them; defaults are what DESIGN.md describes.
 *
 *   "wave_front"        1 (default): batches of small molecules build their pair lists inside the fused kernel (16-dimensional edge
 *                       basis); 0: separate front-end kernels and 48-channel rows (results differ by float32 rounding)
 *   "wave_lds"          LDS bytes per wavefront of the fused kernels (default 20480; 16384..65536)
 *   "wave_prio"         fused kernel: molecules with at least this many atoms run at raised wave priority (default 18; 0 = off)
 *   "wave_order"        order of a launch's wavefronts: 0 largest molecule first (default), 1 largest / smallest interleaved, 2 smallest first
 *   "large_fused"       1 (default): the tiled path merges reduction, update MLP and next projections into one launch per GNN step;
 *                       0: one kernel per stage
 *   "large_merge"       1 (default): the compact entry launches the pair-list construction of tiled molecules merged with the work that
 *                       needs only the atoms (feature rows, atom types, first projections, the first step's type sums and correction
 *                       tiles); 0: every kernel its own launch
 *   "front_inline"      1 (default): compact entry, tiled systems of up to 8192 atoms: every workgroup of the pair list's fill pass works
 *                       out its rows' prefix sums itself (no single-workgroup scan launch between count and fill); 0: the scan launch
 *   "front_bits"        1 (default): the separate front-end's count pass leaves its D < cutoff decisions as a bit per candidate and the
 *                       fill pass walks the set bits (systems of up to 4096 atoms); 0: the fill pass measures every distance again
 *   "large_sweep_old"   1: every tiled molecule runs the four-tile sweep kernel (k_lg_sweep) that systems above 4096 atoms use
 *                       (results differ from the tile-workgroup kernel's by the order of the partner sum)
 *   "large_chunks"      number of pieces the partner range of a tiled molecule's all-pairs sweep is cut into (0, default: by size)
 *   "part_collective"   1: the communicator's collectives (row exchange of a partition, gradient all-reduce, their status guards) run
 *                       even at world size 1 (tests on one GPU)
 *   "comm_guard"        1 (default): every payload collective is preceded by a 4-byte all-reduce (max) of a status word that is read
 *                       back before the payload is enqueued -- a rank that failed makes every rank return non-zero instead of leaving
 *                       its peers blocked (RCCL has no timeout); 0: payload collectives only (one host synchronisation less per
 *                       collective, for measurements)
 *   "comm_inject_fail"  1: this rank reports a failure at its next status guard (tests)
 *   "forward_ahead"     1 (default): epnn_forward_xyz_dev called again with the batch and buffers of the call before enqueues the new
 *                       forward first and looks at the previous one's status after (two status slots; a forward that overflowed a
 *                       capacity is redone with its successor behind it); 0: every call waits for the one before it
 *   "dense_small"       1 (default): a make_model call on one or a few molecules (B N^2 <= 65536) builds per-atom features, flags,
 *                       effective atom counts and pair list in four launches; 0: the general sequence
 *   "dense_rowfused"    1 (default): such a call with B N <= 256 runs the row-fused forward kernels of the training step (the
 *                       reference's literal arithmetic; always when N <= 48, beyond that when the largest molecule fills more than
 *                       55 % of N); 0: always the fused / tiled kernels (results differ by float32 rounding)
 *   "train_skip_padded" 1 (default): epnn_train_step_xyz tells the row-fused kernels which atom slots are real; the workgroups of the
 *                       others return at once; 0: every slot is computed
 *   "train_inline"      1 (default): the inputs of an epnn_train_step_xyz step of up to ~69 atoms travel in the argument block of the
 *                       kernel that pads them; 0: always staged in page-locked memory and uploaded
 *   "train_split"       workgroups that share one atom's weight-gradient jobs in the backward launches (0, default: as many as fit the
 *                       XCD the atom's workgroups are placed on, at most 6; 1..8)
 */
#ifndef EPNN_DEV_H
#define EPNN_DEV_H
#include "epnn.h"
#endif /* EPNN_DEV_H */
