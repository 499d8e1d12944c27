"""Host logic: molecule sharding (incl. a world_size-2 gloo run) and the synthetic workload generator. CPU only."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT


def test_partition_is_exact_cover_balanced_and_deterministic():
    from epnn_amd import shard, synth
    ns = np.diff(synth.qm9_like_batch(B=512, seed=3)[0])
    for world in (1, 2, 3, 8):
        parts = shard.partition_molecules(ns, world)
        allidx = np.sort(np.concatenate(parts))
        assert np.array_equal(allidx, np.arange(len(ns)))
        loads = [shard.molecule_cost(ns[p]).sum() for p in parts]
        assert max(loads) / (sum(loads) / world) < 1.02
        again = shard.partition_molecules(ns, world)
        assert all(np.array_equal(a, b) for a, b in zip(parts, again))


def test_data_parallel_step_molecules_cover_the_order_once():
    """shard.dp_step_molecules: the molecules of optimizer step s, one entry per rank -- one molecule each (the reference's batch
    of one per device) or `per_rank` of them (train.py --molecules-per-rank); over the steps every position of the order is
    taken exactly once, in order."""
    from epnn_amd import shard
    order = np.arange(100, 148)
    for world, per in ((1, 1), (2, 1), (8, 1), (2, 3), (4, 2)):
        seen = []
        for s in range(len(order) // (world * per)):
            parts = shard.dp_step_molecules(order, world, s, per)
            assert len(parts) == world
            for p in parts:
                seen += [p] if per == 1 else p
                assert per == 1 or len(p) == per
        assert seen == [int(v) for v in order[:len(seen)]] and len(seen) == (len(order) // (world * per)) * world * per


def test_take_molecules_round_trip():
    from epnn_amd import shard, synth
    offsets, xyz, x, Q, N = synth.qm9_like_batch(B=16, seed=1)
    idx = np.array([3, 7, 8])
    off, xyz_s, x_s, Q_s, rows = shard.take_molecules(offsets, xyz, x, Q, idx)
    assert off[-1] == sum(offsets[i + 1] - offsets[i] for i in idx)
    assert np.array_equal(xyz_s[off[1]:off[2]], xyz[offsets[7]:offsets[8]])
    assert np.array_equal(rows[:off[1]], np.arange(offsets[3], offsets[4]))


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch.distributed as dist
from epnn_amd import shard, synth, checkpoint
from oracle import epnn_oracle as orc      # test stand-in for the per-rank engine
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
w = checkpoint.load_epnn_weights(os.path.join(sys.argv[1], "models", "decay_model_weights"))
offsets, xyz, x, Q, N = synth.qm9_like_batch(B=12, seed=5)
def compute(off, xyz_s, x_s, Q_s, N):
    return np.concatenate([orc.forward_xyz(xyz_s[off[b]:off[b+1]], x_s[off[b]:off[b+1]], Q_s[b], w, N=N)[:off[b+1]-off[b]]
                           for b in range(len(off) - 1)]).astype(np.float32)
q = shard.forward_sharded(compute, offsets, xyz, x, Q, N, rank, world, dist)
if rank == 0:
    full = compute(offsets, xyz, x, Q, N)
    assert np.array_equal(q, full), np.abs(q - full).max()
    print("SHARD_OK", len(q))
dist.barrier()
dist.destroy_process_group()
'''


def test_forward_sharded_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script), ROOT],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "SHARD_OK" in out.stdout


def test_synthetic_qm9_like_batch_properties():
    from epnn_amd import synth
    offsets, xyz, x, Q, N = synth.qm9_like_batch(B=256, seed=0)
    ns = np.diff(offsets)
    assert N == 29 and ns.max() == 29 and ns.min() >= 7 and 16.5 < ns.mean() < 19.5
    assert x.shape == (offsets[-1], 9) and np.all(x[:, 1:].sum(1) == 1) and np.all(Q == 0)
    near = 0
    for b in range(64):
        p = xyz[offsets[b]:offsets[b + 1]].astype(np.float64)
        d = np.linalg.norm(p[:, None] - p[None], axis=-1)
        iu = np.triu_indices(len(p), 1)
        assert d[iu].min() >= 0.95 - 1e-6
        near += 2 * int((d[iu] < 3.0).sum())
    per_atom = near / offsets[64]
    assert 6.0 < per_atom < 13.0, per_atom


def test_algorithmic_flops_matches_survey_figure():
    """mean QM9 molecule n=18, nnz=145 ordered near pairs -> 10.8 Mflop per forward."""
    from epnn_amd import synth
    fl = synth.algorithmic_flops([18], 145 / 2)
    assert abs(fl / 1e6 - 10.8) < 0.1, fl


def test_roofline_of_the_two_matrix_pipes():
    """bench.py prices the forward against the bound of the two matrix pipes it runs on (synth.mixed_pipe_peak): the split of the
    algorithmic flops adds up to the total, the peak lies between the f32 MFMA peak and 2500 / 6, a system whose flops are all
    all-pairs Dense reaches the bf16 form's bound, and the bench batch's figures are the ones DESIGN.md section 5 quotes."""
    import numpy as np
    from epnn_amd import synth
    for ns, pairs in (([18], 72), ([2220], 12810), ([100000], 532253)):
        total = synth.algorithmic_flops(ns, pairs)
        for chains in (False, True):
            for edges in (False, True):
                f32, bf = synth.algorithmic_flops(ns, pairs, parts="pipes", chains_bf16=chains, edges_bf16=edges)
                assert f32 > 0 and bf > 0 and abs(f32 + bf - total) <= 1e-9 * total
                peak, share = synth.mixed_pipe_peak(ns, pairs, chains_bf16=chains, edges_bf16=edges)
                assert synth.FP32_MFMA_PEAK_TFLOPS < peak < synth.BF16X6_PEAK_TFLOPS and abs(share - bf / total) < 1e-12
    assert synth.mixed_pipe_peak([100000], 532253)[0] > 0.999 * synth.BF16X6_PEAK_TFLOPS
    off = synth.qm9_like_batch(1024, 0, 29)[0]
    peak, share = synth.mixed_pipe_peak(np.diff(off), 72439, chains_bf16=True)
    assert abs(peak - 413.8) < 0.5 and abs(share - 0.996) < 0.002
    peak, share = synth.mixed_pipe_peak(np.diff(off), 72439, chains_bf16=True, edges_bf16=False)     # (the edge products as f32 MFMAs)
    assert abs(peak - 277.2) < 0.5 and abs(share - 0.695) < 0.005


_DP_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
import torch.distributed as dist
from epnn_amd import shard
from oracle import epnn_oracle_train as ot      # test stand-in for the per-rank engine
from conftest import random_weights
from test_train_oracle import _tiny_batch
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
nx, T, N = 9, 1, 6
w = random_weights(nx, T, seed=2, scale=0.5)
h, e, x, q, mask, y = _tiny_batch(nx, N, [5, 4, 6, 3], seed=7)
order = np.array([2, 0, 3, 1])
theta = ot.flatten(w); opt = ot.Adam(theta.size)
for step in range(2):
    mine = shard.dp_step_molecules(order, world, step)[rank]
    sl = slice(mine, mine + 1)
    _, _, g = ot.loss_and_grads(h[sl], e[sl], x[sl], q[sl], mask[sl], y[sl], ot.unflatten(theta, w))
    gsum = shard.allreduce_sum_host(ot.flatten(g), dist)          # what ncclAllReduce(sum) does on the GPUs
    theta = opt.step(theta, gsum)
if rank == 0:
    # single process, the same global batches of two molecules with summed gradients
    theta1 = ot.flatten(w); opt1 = ot.Adam(theta1.size)
    for step in range(2):
        idx = shard.dp_step_molecules(order, world, step)
        _, _, g = ot.loss_and_grads(h[idx], e[idx], x[idx], q[idx], mask[idx], y[idx], ot.unflatten(theta1, w))
        theta1 = opt1.step(theta1, ot.flatten(g))
    assert np.abs(theta - theta1).max() < 1e-12, np.abs(theta - theta1).max()
    print("DP_OK")
dist.barrier(); dist.destroy_process_group()
'''


def test_data_parallel_gradient_sum_world2_gloo(tmp_path):
    """one molecule per rank, gradients summed by an all-reduce, identical Adam step on
    every rank == single-process step on the summed batch.  gloo on CPU stands in for RCCL."""
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29519", str(script), ROOT],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "DP_OK" in out.stdout


_EXCH_WORKER = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch.distributed as dist
from epnn_amd import shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)

class FakeEngine:                                  # host arrays stand in for the device rows
    def __init__(self, rows, fail=False):
        self.rows, self.fail = rows, fail
    def copy_rows_to_host(self, d_ptr, row_len, lo, hi):
        if self.fail:
            raise OSError("device copy failed")
        return self.rows[lo:hi].copy()
    def copy_rows_to_device(self, d_ptr, row_len, lo, rows):
        self.rows[lo:lo + len(rows)] = rows

n, L = 10, 4
full = np.arange(n * L, dtype=np.float32).reshape(n, L)
lo, hi = (0, 6) if rank == 0 else (6, 10)          # unequal ranges, like tile groups
rows = np.zeros_like(full); rows[lo:hi] = full[lo:hi]
ex = shard.make_row_exchange(FakeEngine(rows), dist, rank, world)
ex(0, L, n, lo, hi)
assert np.array_equal(rows, full)
ex(0, L, n, lo, hi)                                # second call: cached ranges
# a failure on ONE rank aborts the exchange on EVERY rank (nobody is left waiting in the collective)
bad = shard.make_row_exchange(FakeEngine(rows, fail=(rank == 1)), dist, rank, world)
t0 = time.time()
try:
    bad(0, L, n, lo, hi)
    raise SystemExit("exchange did not fail")
except RuntimeError as exc:
    assert "rank(s) [1] failed" in str(exc), str(exc)
assert time.time() - t0 < 30
dist.barrier()
if rank == 0:
    print("EXCHANGE_OK")
dist.destroy_process_group()
'''


def test_row_exchange_all_gathers_and_fails_together_world2_gloo(tmp_path):
    """shard.make_row_exchange (the host-staged exchange of a partitioned large system): unequal row ranges are gathered
    on every rank; a failure on one rank raises on all of them instead of leaving the others in the collective."""
    script = tmp_path / "ex_worker.py"
    script.write_text(_EXCH_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29521", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29521", str(script), ROOT],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "EXCHANGE_OK" in out.stdout


_GUARD_WORKER = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch.distributed as dist
from epnn_amd import shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)

class FakeEngine:                                  # the gradient of a train step, or the step's failure (EPNN_FAIL before the collective)
    def __init__(self, fail):
        self.fail = fail
    def get_gradients(self):
        if self.fail:
            raise OSError("hipMalloc failed: out of memory")
        return np.full(5, 1.0 + rank, np.float32)

def step(eng):
    g, err = None, None
    try:
        g = eng.get_gradients()
    except Exception as exc:                       # the step failed on THIS rank: its peers must not wait for it
        err = exc
    return shard.allreduce_sum_host(g, dist, failed=err)

assert np.array_equal(step(FakeEngine(False)), np.full(5, 3.0))
t0 = time.time()
try:
    step(FakeEngine(rank == 1))
    raise SystemExit("the all-reduce did not fail")
except RuntimeError as exc:
    assert "aborted on every rank" in str(exc), str(exc)
    assert ("this rank failed" in str(exc)) == (rank == 1), str(exc)
assert time.time() - t0 < 30
assert np.array_equal(step(FakeEngine(False)), np.full(5, 3.0))      # the group is still usable afterwards
dist.barrier()
if rank == 0:
    print("GUARD_OK")
dist.destroy_process_group()
'''


def test_gradient_allreduce_fails_together_world2_gloo(tmp_path):
    """The status guard in front of the gradient all-reduce (the host-staged counterpart of comm_guard in csrc/epnn_host.h): a rank
    whose step failed before the collective makes EVERY rank raise instead of leaving its peer blocked in the all-reduce."""
    script = tmp_path / "guard_worker.py"
    script.write_text(_GUARD_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script), ROOT],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "GUARD_OK" in out.stdout
