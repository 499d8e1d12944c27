#!/usr/bin/env python3
"""Rate of the compact entry on the bench batch (1024 QM9-like molecules, N = 29) for a model whose update MLP has other `layers`
than the reference's [32, 32] (make_model(layers, ...), charge_gn.py:369-371): one or two hidden layers of at most 32 units run the
tuned kernels (zero-padded copy, exact), anything else the tiled kernels with the generic update stage.
    python tools/bench_layers.py [depth]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_weights
from epnn_amd import synth
from epnn_amd.engine import Pipeline
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
off, xyz, x, Q, N = synth.qm9_like_batch(1024, 0, 29)
A = int(off[-1])
rng = np.random.default_rng(0)
def upd(layers):
    dims = [80] + list(layers) + [48]
    return [((rng.normal(size=(i, o)) / np.sqrt(i)).astype(np.float32), (0.1 * rng.normal(size=(o,))).astype(np.float32)) for i, o in zip(dims[:-1], dims[1:])]
for layers in ([32, 32], [16], [24, 8], [64, 32], [48, 48], [8, 24, 40]):
    w = random_weights(9, 5, seed=3, scale=0.3)
    w["upd"] = upd(layers)
    pipe = Pipeline(depth=depth, nx=9, T=5); pipe.set_weights(w)
    lanes = [(e, [e.to_device(a) for a in (xyz, x, Q)], e.alloc(A * 4)) for e in pipe.engines]
    def step(k):
        e, dv, dq = lanes[k % depth]; e.forward_xyz_dev(off, dv[0], dv[1], dv[2], dq, N)
    for k in range(10 * depth): step(k)
    pipe.sync(); t0 = time.perf_counter()
    n = 40 * depth
    for k in range(n): step(k)
    pipe.sync(); dt = (time.perf_counter() - t0) / n
    st = lanes[0][0].last_stats()
    print(f"layers {str(layers):12s}: {dt*1e3:.3f} ms per batch = {A/dt/1e6:6.1f} M atoms/s ({depth} batches in flight; molecules on the fused / tiled kernels: {st[1]} / {st[2]})", flush=True)
    pipe.close()
