"""Molecule sharding over the GPUs of one node.

Inference over independent molecules needs no collective on the data path: the batch is partitioned by molecule,
balanced by the cost of the all-pairs GNN sweep (~n^2), every rank runs its shard through its own Engine and the
charges are concatenated back in the original order.  Per-molecule results do not depend on which other molecules
share the batch (every sum in the kernels has a fixed, molecule-local order), so the sharded result equals the
single-GPU result bit for bit.
"""
from __future__ import annotations

import numpy as np


def molecule_cost(ns):
    """Relative cost model: all-pairs message sweep (n^2 pair rows) + per-atom work + near-pair work (~n)."""
    ns = np.asarray(ns, dtype=np.float64)
    return ns * ns + 24.0 * ns


def partition_molecules(ns, world):
    """Longest-processing-time greedy partition; deterministic (ties by molecule index, then by rank).
    Returns a list of `world` sorted index arrays that together cover range(len(ns)) exactly once."""
    if world < 1:
        raise ValueError("world must be >= 1")
    cost = molecule_cost(ns)
    order = sorted(range(len(cost)), key=lambda i: (-cost[i], i))
    load = [0.0] * world
    parts = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += float(cost[i])
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


def take_molecules(offsets, xyz, x, Q, idx):
    """Sub-batch (offsets, xyz, x, Q) of the molecules `idx` of a flat batch."""
    offsets = np.asarray(offsets)
    idx = np.asarray(idx, dtype=np.int64)
    sizes = offsets[idx + 1] - offsets[idx]
    off = np.zeros(len(idx) + 1, dtype=np.int32)
    off[1:] = np.cumsum(sizes)
    if len(idx):
        rows = np.concatenate([np.arange(offsets[i], offsets[i + 1]) for i in idx])
    else:
        rows = np.zeros((0,), dtype=np.int64)
    return off, np.asarray(xyz)[rows], np.asarray(x)[rows], np.asarray(Q)[idx], rows


def forward_sharded(compute, offsets, xyz, x, Q, N, rank=0, world=1, dist=None):
    """Run `compute(offsets, xyz, x, Q, N) -> q` on this rank's shard and return the full charge vector (A,) on
    every rank.  `dist` is torch.distributed (initialised) when world > 1; the only communication is the final
    all_gather of the per-rank charge vectors."""
    offsets = np.asarray(offsets)
    A = int(offsets[-1])
    parts = partition_molecules(np.diff(offsets), world)
    off, xyz_s, x_s, Q_s, rows = take_molecules(offsets, xyz, x, Q, parts[rank])
    q_local = compute(off, xyz_s, x_s, Q_s, N) if len(parts[rank]) else np.zeros((0,), dtype=np.float32)
    out = np.zeros((A,), dtype=np.float32)
    if world == 1:
        out[rows] = q_local
        return out
    gathered = [None] * world
    dist.all_gather_object(gathered, (rows, np.asarray(q_local, dtype=np.float32)))
    for r_rows, r_q in gathered:
        out[r_rows] = r_q
    return out


# --------------------------------------------------------------------------------------------- one large system
def make_row_exchange(engine, dist, rank, world):
    """Exchange function for Engine.set_partition: all-gather of the rows of S every process owns, through
    torch.distributed on the host (gloo; the rows of one GNN step of a 100 000-atom system are 12.8 MB).  This is the
    portable form (ranks sharing a GPU, CPU-side tests); with one GPU per process leave the callback out and give the
    engine a communicator (Engine.comm_init): the rows then travel over RCCL on the engine's stream.  Every process calls it
    at the same points of the forward (after each GNN step).

    A failure on one rank must not leave the others waiting in the collective: every call first all-gathers a status word
    (local preparation done / failed); if any rank failed, ALL ranks raise, so the partitioned forward aborts on every
    process together (each then exits non-zero) instead of hanging."""
    import torch
    known = {}                                   # (n_rows, row_lo, row_hi) -> every process's row range (fixed per plan)

    def exchange(d_rows, row_len, n_rows, row_lo, row_hi):
        mine, err = None, None
        try:
            mine = engine.copy_rows_to_host(d_rows, row_len, row_lo, row_hi)
        except Exception as exc:                 # noqa: BLE001 -- reported to every rank below
            err = exc
        status = torch.tensor([0 if err is None else 1], dtype=torch.int32)
        every = [torch.zeros_like(status) for _ in range(world)]
        dist.all_gather(every, status)
        bad = [r for r, t in enumerate(every) if int(t.item()) != 0]
        if bad:
            raise RuntimeError(f"row exchange aborted on every rank: rank(s) {bad} failed" + (f" ({err})" if err else ""))
        key = (int(n_rows), int(row_lo), int(row_hi))
        if key not in known:
            ranges = [None] * world
            dist.all_gather_object(ranges, (int(row_lo), int(row_hi)))
            known.clear()
            known[key] = ranges
        ranges = known[key]
        width = max(hi - lo for lo, hi in ranges)
        send = torch.zeros((max(width, 1), row_len), dtype=torch.float32)
        send[:mine.shape[0]] = torch.from_numpy(mine)
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(recv, send)
        for r, (lo, hi) in enumerate(ranges):
            if r != rank and hi > lo:
                engine.copy_rows_to_device(d_rows, row_len, lo, recv[r][:hi - lo].numpy())

    return exchange


# --------------------------------------------------------------------------------------------- data-parallel training
def dp_step_molecules(order, world, step, per_rank=1):
    """Molecules of optimizer step `step` when `world` ranks each take `per_rank` molecules (train.py): rank r gets
    order[(step*world + r)*per_rank : (step*world + r + 1)*per_rank].  Returns one entry per rank: the molecule's index
    (per_rank == 1, the reference's batch of one per device) or the list of its `per_rank` indices."""
    if per_rank == 1:
        return [int(order[step * world + r]) for r in range(world)]
    return [[int(i) for i in order[(step * world + r) * per_rank:(step * world + r + 1) * per_rank]] for r in range(world)]


def allreduce_sum_host(vec, dist, failed=None):
    """Gradient all-reduce through the host (gloo) -- the portable counterpart of the RCCL all-reduce the library does
    on the device (epnn_comm_init / epnn_train_apply); used by CPU tests and as a fallback without RCCL.

    Fails closed like the device path (comm_guard in csrc/epnn_host.h): every rank first all-reduces (max) a status word --
    `failed` is this rank's exception, or a true value, if its step went wrong before the collective (then `vec` may be None)
    -- and if ANY rank reports a failure ALL ranks raise together; nobody is left waiting in the payload collective."""
    import torch
    status = torch.tensor([1 if failed else 0], dtype=torch.int32)
    dist.all_reduce(status, op=dist.ReduceOp.MAX)
    if int(status.item()) != 0:
        raise RuntimeError("gradient all-reduce aborted on every rank" +
                           (f"; this rank failed: {failed}" if failed else ": another rank reported a failure before the collective"))
    t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()
