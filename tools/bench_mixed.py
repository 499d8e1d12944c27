#!/usr/bin/env python3
"""Timing of REAL data: the reference's 871-system validation batch of its `mixed` set (molecules of 3..38 atoms, N = 41):
molecules up to 32 atoms on the two-block fused kernel, 33..48 on three wavefronts of the block-per-wavefront kernel (`--opt wave3=0`: on the tiled
kernels, as before round 2).   python tools/bench_mixed.py [depth] [--opt=name:value ...]"""
import os, sys, tarfile, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_molecules, GOLDEN
from epnn_amd import checkpoint
from epnn_amd.engine import Engine
d = tempfile.mkdtemp(); tarfile.open(GOLDEN + "/mixed_val.tar.gz").extractall(d)
names = [str(n) for n in np.load(GOLDEN + "/val_names.npy", allow_pickle=True)]
mols, off, xyz, x, Q = load_molecules(d + "/mixed_val", names, 9)
ns = np.diff(off)
print("systems", len(ns), "atoms", int(off[-1]), "n > 32:", int((ns > 32).sum()), "with", int(ns[ns > 32].sum()), "atoms; histogram of n>32:", np.bincount(ns[ns > 32])[33:])
w = checkpoint.load_epnn_weights(ROOT + "/models/decay_model_weights")
eng = Engine(nx=9, T=5); eng.set_weights(w)
def timeit(sel, label):
    m = [mols[i] for i in sel]
    o = np.zeros(len(m) + 1, np.int32); o[1:] = np.cumsum([a[1].shape[0] for a in m])
    args = (o, np.concatenate([a[0] for a in m]), np.concatenate([a[1] for a in m]), np.array([a[2] for a in m], np.float32))
    dv = [eng.to_device(a) for a in args[1:]]; dq = eng.alloc(int(o[-1]) * 4)
    for _ in range(3): eng.forward_xyz_dev(o, dv[0], dv[1], dv[2], dq, 41)
    eng.sync(); t0 = time.perf_counter()
    for _ in range(20): eng.forward_xyz_dev(o, dv[0], dv[1], dv[2], dq, 41)
    eng.sync(); dt = (time.perf_counter() - t0) / 20
    print(f"{label}: {len(m)} systems, {int(o[-1])} atoms: {dt*1e3:.3f} ms per forward = {o[-1]/dt/1e6:.1f} M atoms/s; stats {eng.last_stats()}", flush=True)
only_pipe = len(sys.argv) > 1 and not sys.argv[1].startswith("--")
if not only_pipe: timeit(range(len(mols)), "all")
if not only_pipe: timeit([i for i in range(len(mols)) if ns[i] <= 32], "n <= 32 (fused kernel)")
if not only_pipe: timeit([i for i in range(len(mols)) if ns[i] > 32], "n > 32 (three wavefronts per molecule; tiled with wave3=0)")
eng.close()
from epnn_amd.engine import Pipeline
# pipelined: `depth` batches in flight
gold = np.load(GOLDEN + "/test_pred_charges.npy")
for depth in ((int(sys.argv[1]),) if only_pipe else (1, 4, 8, 14)):
    pipe = Pipeline(depth=depth, nx=9, T=5); pipe.set_weights(w)
    for o in [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--opt=")]:      # --opt=name:value on every lane
        pipe.set_option(o.split(":")[0], int(o.split(":")[1]))
    lanes = []
    for e in pipe.engines:
        dv = [e.to_device(a) for a in (xyz, x, Q)]; lanes.append((e, dv, e.alloc(int(off[-1]) * 4)))
    def step(k):
        e, dv, dq = lanes[k % depth]; e.forward_xyz_dev(off, dv[0], dv[1], dv[2], dq, 41)
    for k in range(60 * depth): step(k)
    pipe.sync(); t0 = time.perf_counter()
    nrun = 50 * depth
    for k in range(nrun): step(k)
    pipe.sync(); dt = (time.perf_counter() - t0) / nrun
    q = lanes[0][2].download((int(off[-1]),))
    err = max(float(np.abs(q[off[i]:off[i + 1]] - gold[i, :ns[i]]).max()) for i in range(len(ns)))
    print(f"pipelined depth {depth}: {dt*1e3:.4f} ms per batch of 871 systems = {off[-1]/dt/1e6:.1f} M atoms/s; max |dq| vs the stored TensorFlow outputs {err:.2e}", flush=True)
    pipe.close()
